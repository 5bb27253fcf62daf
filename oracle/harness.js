"use strict"
/*
 * ORACLE HARNESS — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Restatements of the reference's test helpers, used to drive oracle/peritext_oracle.js:
 *   reference/test/generateDocs.ts:11-42      generateDocs
 *   reference/test/accumulatePatches.ts:9-80  accumulatePatches (rebuild spans from a Patch[] stream)
 *   reference/test/merge.ts:4-38              applyChanges (retry on causal gap), getMissingChanges
 *   reference/test/micromerge.ts:46-86        testConcurrentWrites (two replicas, cross-apply, 4 asserts)
 */
const O = require("./peritext_oracle")
const Micromerge = O.Micromerge

/** `count` replicas doc1..docN sharing one initial change (makeList + one insert of all chars) made on doc1. */
function generateDocs(text, count, opts) {
    if (text === undefined) text = "The Peritext editor"
    if (count === undefined) count = 2
    const docs = []
    for (let i = 0; i < count; i++) docs.push(new Micromerge("doc" + (i + 1), opts))
    const patches = docs.map(() => [])
    const first = docs[0].change([
        { path: [], action: "makeList", key: "text" },
        { path: ["text"], action: "insert", index: 0, values: text.split("") },
    ])
    patches[0] = first.patches
    for (let i = 1; i < count; i++) patches[i] = docs[i].applyChange(first.change)
    return { docs, patches, initialChange: first.change }
}

/** Naive per-character replay of a patch stream; must equal the batch getTextWithFormatting. */
function accumulatePatches(patches) {
    const cells = [] /* {character, marks} per visible char */
    for (const p of patches) {
        if (!(p.path.length === 1 && p.path[0] === "text")) {
            throw new Error("This implementation only supports a single path: 'text'")
        }
        if (p.action === "insert") {
            p.values.forEach((character, k) => {
                cells.splice(p.index + k, 0, { character, marks: Object.assign({}, p.marks) })
            })
        } else if (p.action === "delete") {
            cells.splice(p.index, p.count)
        } else if (p.action === "addMark") {
            for (let i = p.startIndex; i < p.endIndex; i++) {
                const m = cells[i].marks
                if (p.markType !== "comment") {
                    m[p.markType] = Object.assign({}, p.attrs || { active: true })
                } else if (m.comment === undefined) {
                    m.comment = [Object.assign({}, p.attrs)]
                } else if (!m.comment.find(c => c.id === p.attrs.id)) {
                    m.comment = O.sortedById(m.comment.concat([Object.assign({}, p.attrs)]))
                }
            }
        } else if (p.action === "removeMark") {
            for (let i = p.startIndex; i < p.endIndex; i++) delete cells[i].marks[p.markType]
        } else if (p.action !== "makeList") {
            throw new Error("unexpected patch action " + p.action)
        }
    }
    const spans = []
    for (const c of cells) O.pushRun(spans, [c.character], c.marks)
    return spans
}

/** Apply a bag of changes in any causally valid order: re-queue the ones that throw (merge.ts:4-22). */
function applyChanges(doc, changes, applied) {
    const queue = changes.slice()
    const patches = []
    let spins = 0
    while (queue.length > 0) {
        const c = queue.shift()
        try {
            for (const p of doc.applyChange(c)) patches.push(p)
            if (applied) applied.push(c)
        } catch (e) {
            queue.push(c)
        }
        if (spins++ > 10000) throw new Error("applyChanges did not converge")
    }
    return patches
}

/** Changes `source` has seen that `target` has not, by vector-clock difference (merge.ts:25-38). */
function getMissingChanges(source, target, queues) {
    const out = []
    for (const actor of Object.keys(source.clock)) {
        const have = target.clock[actor]
        const upto = source.clock[actor]
        if (have === undefined) for (const c of queues[actor].slice(0, upto)) out.push(c)
        if (have < upto) for (const c of queues[actor].slice(have, upto)) out.push(c)
    }
    return out
}

/**
 * The reference's two-replica scenario (test/micromerge.ts:46-86) as a function returning
 * everything a checker may want: both batch outputs, both patch-accumulated outputs and
 * the per-replica logs (changes in application order).
 */
function runConcurrentWrites(spec) {
    const withPath = ops => (ops || []).map(op => Object.assign({}, op, { path: ["text"] }))
    const g = generateDocs(spec.initialText === undefined ? "The Peritext editor" : spec.initialText)
    const doc1 = g.docs[0]
    const doc2 = g.docs[1]
    let p1 = g.patches[0]
    let p2 = g.patches[1]
    const log1 = [g.initialChange]
    const log2 = [g.initialChange]
    if (spec.preOps) {
        const r0 = doc1.change(withPath(spec.preOps))
        p1 = p1.concat(r0.patches)
        p2 = p2.concat(doc2.applyChange(r0.change))
        log1.push(r0.change)
        log2.push(r0.change)
    }
    const r1 = doc1.change(withPath(spec.inputOps1))
    p1 = p1.concat(r1.patches)
    const r2 = doc2.change(withPath(spec.inputOps2))
    p2 = p2.concat(r2.patches)
    p2 = p2.concat(doc2.applyChange(r1.change))
    p1 = p1.concat(doc1.applyChange(r2.change))
    log1.push(r1.change, r2.change)
    log2.push(r2.change, r1.change)
    return {
        batch1: doc1.getTextWithFormatting(["text"]),
        batch2: doc2.getTextWithFormatting(["text"]),
        patched1: accumulatePatches(p1),
        patched2: accumulatePatches(p2),
        logs: [log1, log2],
        docs: [doc1, doc2],
    }
}

module.exports = { generateDocs, accumulatePatches, applyChanges, getMissingChanges, runConcurrentWrites }
