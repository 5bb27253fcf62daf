#!/usr/bin/env node
"use strict"
/*
 * ORACLE PINNING TOOL — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Replays the reference's OWN mocha test file (reference/test/micromerge.ts, 46 `it` cases) against
 * oracle/peritext_oracle.js, in this container, without tsc/mocha:
 *   - the test source is read from where it lies under /root/reference (never copied into the repo),
 *     its handful of TypeScript annotations are erased in memory, and it is evaluated in a `vm`
 *     context whose `describe` / `it` / imports are shims bound to the oracle and oracle/harness.js;
 *   - every assertion in the file therefore runs against the oracle (batch spans, patch streams,
 *     exact Patch[] shapes, cursors).
 *
 * With --dump <file> it also records, for every replica created by every test case, the replica's
 * log (Change[] in application order) and final spans, plus the `expectedResult` literal of each
 * testConcurrentWrites case — these become tests/golden/kat_reference_tests.json, the golden
 * vectors the HIP path is checked against on the GPU box (where /root/reference does not exist).
 *
 * With --dump-scripts <file> it records, per test case, every Micromerge.change(ops) / applyChange(change) call in call order:
 * {replica, kind: "change", ops: InputOperation[], change: the Change it returned} | {replica, kind: "apply", change} —
 * tests/golden/kat_change_scripts.json, the known answers of the on-device change() (ptx_change): made with --impl ref the
 * Changes are the reference's own.
 *
 * Usage: node oracle/run_reference_tests.js [--ref /root/reference] [--impl oracle|ref] [--dump tests/golden/kat_reference_tests.json]
 *                                           [--dump-scripts tests/golden/kat_change_scripts.json]
 * Exit code 0 iff every case passed.
 */
const fs = require("fs")
const path = require("path")
const vm = require("vm")
const assert = require("assert")
const util = require("util")
const O = require("./peritext_oracle")
const H = require("./harness")

const argv = process.argv.slice(2)
function flag(name, dflt) {
    const i = argv.indexOf(name)
    return i >= 0 ? argv[i + 1] : dflt
}
const refRoot = flag("--ref", "/root/reference")
const dumpPath = flag("--dump", null)
const scriptsPath = flag("--dump-scripts", null)
/* --impl oracle (default) | ref : `ref` binds the shims to the type-erased reference in oracle/_ref instead */
const impl = flag("--impl", "oracle")
const Impl = impl === "ref" ? require("./_ref/micromerge").default : O.Micromerge
const testFile = path.join(refRoot, "test", "micromerge.ts")
if (!fs.existsSync(testFile)) {
    console.error("reference test file not found: " + testFile)
    process.exit(2)
}

/* ---- erase the TypeScript-only syntax of this one file (fails loudly if the file ever changes shape) ---- */
function eraseTypes(src) {
    const lines = src.split("\n")
    const out = []
    for (let i = 0; i < lines.length; i++) {
        const line = lines[i]
        if (/^import\s/.test(line)) continue
        if (/^export type\s/.test(line)) {
            /* skip to the end of the type declaration: balance (), {}, <> opened on these lines */
            let depth = 0
            let j = i
            for (;;) {
                for (const ch of lines[j]) {
                    if (ch === "{" || ch === "(") depth++
                    else if (ch === "}" || ch === ")") depth--
                }
                if (depth <= 0) break
                j++
            }
            i = j
            continue
        }
        out.push(line)
    }
    let js = out.join("\n")
    const subs = [
        [/^export const /gm, "const "],
        [/\(obj: any\): void =>/g, "(obj) =>"],
        [/\(start: number, end: number\): number\[\] =>/g, "(start, end) =>"],
        [/\(args: TraceSpec\): void =>/g, "(args) =>"],
        [/: InputOperation\[\] =/g, " ="],
        [/getRoot<RootDoc>\(\)/g, "getRoot()"],
        [/const testConcurrentWrites = \(args\) => \{/, "const testConcurrentWrites = (args) => { __capture(args);"],
    ]
    for (const [re, to] of subs) js = js.replace(re, to)
    return js
}

/* ---- shims ---- */
const results = []
const dump = []
const scripts = []
let current = null

function wrapDoc(doc) {
    /* record the replica's log: every change it generated or applied, in order */
    const log = []
    const change0 = doc.change.bind(doc)
    const apply0 = doc.applyChange.bind(doc)
    const me = current ? current.replicas.length : 0
    const events = current ? current.events : []
    doc.change = ops => {
        const r = change0(ops)
        log.push(JSON.parse(JSON.stringify(r.change)))
        events.push({ replica: me, kind: "change", ops: JSON.parse(JSON.stringify(ops)), change: JSON.parse(JSON.stringify(r.change)) })
        return r
    }
    doc.applyChange = c => {
        const r = apply0(c)
        log.push(JSON.parse(JSON.stringify(c)))
        events.push({ replica: me, kind: "apply", change: JSON.parse(JSON.stringify(c)) })
        return r
    }
    if (current) current.replicas.push({ doc, log })
    return doc
}

const sandbox = {
    assert,
    inspect: util.inspect,
    console,
    generateDocs: (text, count) => {
        const docs = []
        const n = count === undefined ? 2 : count
        for (let i = 0; i < n; i++) docs.push(wrapDoc(new Impl("doc" + (i + 1))))
        const patches = docs.map(() => [])
        const t = text === undefined ? "The Peritext editor" : text
        const first = docs[0].change([
            { path: [], action: "makeList", key: "text" },
            { path: ["text"], action: "insert", index: 0, values: t.split("") },
        ])
        patches[0] = first.patches
        for (let i = 1; i < n; i++) patches[i] = docs[i].applyChange(first.change)
        return { docs, patches, initialChange: first.change }
    },
    accumulatePatches: H.accumulatePatches,
    __capture: args => {
        if (current) current.spec = JSON.parse(JSON.stringify(args))
    },
    describe: null,
    it: null,
}
const prefix = []
function describe(title, body) {
    prefix.push(title)
    body()
    prefix.pop()
}
describe.only = describe
function it(title, body) {
    const full = prefix.concat([title]).join(" / ")
    current = { title: full, replicas: [], spec: null, events: [] }
    let ok = true
    let err = null
    try {
        body()
    } catch (e) {
        ok = false
        err = e
    }
    results.push({ title: full, ok, err })
    if (ok) {
        const entry = { title: full, replicas: [] }
        if (current.spec) entry.expected = current.spec.expectedResult
        for (const r of current.replicas) {
            let spans = null
            try {
                spans = r.doc.getTextWithFormatting(["text"])
            } catch (e) {
                spans = null
            }
            entry.replicas.push({ actor: r.doc.actorId, log: r.log, spans })
        }
        dump.push(entry)
        scripts.push({ title: full, actors: current.replicas.map(r => r.doc.actorId), events: current.events, spans: entry.replicas.map(r => r.spans) })
    }
    current = null
}
sandbox.describe = describe
sandbox.it = it

const js = eraseTypes(fs.readFileSync(testFile, "utf8"))
/* same realm as the oracle (assert.deepStrictEqual compares prototypes), names bound as parameters */
const names = Object.keys(sandbox)
const runner = vm.runInThisContext("(function (" + names.join(", ") + ") {" + js + "\n})", {
    filename: "reference/test/micromerge.ts (types erased in memory)",
})
runner.apply(null, names.map(n => sandbox[n]))

let failed = 0
for (const r of results) {
    if (!r.ok) {
        failed++
        console.log("FAIL  " + r.title + "\n      " + String(r.err && r.err.message).split("\n").slice(0, 12).join("\n      "))
    }
}
console.log(`reference test/micromerge.ts against ${impl === "ref" ? "oracle/_ref (erased reference)" : "the oracle"}: ${results.length - failed} passed, ${failed} failed, ${results.length} total`)
if (dumpPath && failed === 0) {
    const doc = {
        generated_by: "oracle/run_reference_tests.js",
        source: "reference/test/micromerge.ts (all `it` cases; `expected` = the reference's expectedResult literal)",
        n_cases: dump.length,
        cases: dump,
    }
    fs.writeFileSync(dumpPath, JSON.stringify(doc) + "\n")
    console.log("wrote " + dumpPath)
}
if (scriptsPath && failed === 0) {
    const doc = {
        generated_by: "oracle/run_reference_tests.js --dump-scripts" + (impl === "ref" ? " --impl ref" : ""),
        impl,
        source: "reference/test/micromerge.ts: every Micromerge.change(InputOperation[]) / applyChange call of every `it` case, in call order",
        n_cases: scripts.length,
        cases: scripts,
    }
    fs.writeFileSync(scriptsPath, JSON.stringify(doc) + "\n")
    console.log("wrote " + scriptsPath)
}
process.exit(failed === 0 ? 0 : 1)
