#!/usr/bin/env node
"use strict"
/*
 * ORACLE TOOLING — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Golden ProseMirror documents (SURVEY §8 f4): prosemirrorDocFromCRDT (reference/src/bridge.ts:394-414 with
 * getProsemirrorMarksForMarkMap :369-391) applied to span lists, as the JSON of Node.toJSON().
 *
 * What is PINNED by the reference here: everything the reference's own source decides — which mark types exist and in which order
 * marks are visited (ALL_MARKS, schema.ts:125), which attributes a mark type declares (markSpec[t].attrs, schema.ts:45-96: extra
 * properties of a MarkValue such as `active: true` are NOT attributes of the type and vanish), the rank order of the schema's mark
 * table (demoMarkSpec, :99-121), the empty-document short cut (bridge.ts:399-401).  These are read from oracle/_ref/schema.js, which
 * oracle/build_ref.js extracts from the reference's schema.ts — nothing about them is hard-coded below.
 *
 * What is RESTATED (prosemirror-model is not in this image; parity unpinned for these rules): Mark.toJSON = {type, attrs?} with attrs
 * only when the type declares some; Mark.setFrom sorts marks by type rank, stably; Node.toJSON of a text node = {type: "text",
 * marks?, text}; Fragment.fromArray joins ADJACENT text nodes with the same mark set (so two spans whose MarkMaps differ only by
 * `comment: []` vs no comment key become one text node); Node.toJSON omits empty content.
 *
 *   node oracle/gen_pm_golden.js --spans tests/golden/ptxgen_rich_700.json --out tests/golden/pm_docs.json
 */
const fs = require("fs")
const path = require("path")
const S = require(path.join(__dirname, "_ref", "schema.js"))

const argv = process.argv.slice(2)
const flag = (n, d) => (argv.indexOf(n) >= 0 ? argv[argv.indexOf(n) + 1] : d)

function marksOf(markMap) {
    /* bridge.ts:369-391 */
    const marks = []
    for (const t of S.ALL_MARKS) {
        const v = markMap[t]
        if (v === undefined) continue
        if (Array.isArray(v)) for (const one of v) marks.push(mark(t, one))
        else if (v) marks.push(mark(t, v))
    }
    /* Mark.setFrom: by rank of the type in the schema's mark table, stable */
    return marks.map((m, i) => [m, i]).sort((a, b) => S.markRankOrder.indexOf(a[0].type) - S.markRankOrder.indexOf(b[0].type) || a[1] - b[1]).map(x => x[0])
}
function mark(type, value) {
    /* schema.mark(type, attrs): only the attributes the type declares survive */
    const names = S.markSpec[type].attrs
    const m = { type }
    if (names.length) {
        m.attrs = {}
        for (const n of names) m.attrs[n] = value[n]
    }
    return m
}
function docOf(spans) {
    /* bridge.ts:394-414 */
    if (spans.length === 1 && spans[0].text === "") return { type: "doc", content: [{ type: "paragraph" }] }
    const nodes = []
    for (const s of spans) {
        if (s.text === "") throw new Error("Empty text nodes are not allowed") /* prosemirror-model: schema.text("") throws */
        const marks = marksOf(s.marks)
        const last = nodes[nodes.length - 1]
        if (last && JSON.stringify(last.marks || []) === JSON.stringify(marks)) last.text += s.text /* Fragment.fromArray joins */
        else {
            const n = { type: "text" }
            if (marks.length) n.marks = marks
            n.text = s.text
            nodes.push(n)
        }
    }
    const paragraph = { type: "paragraph" }
    if (nodes.length) paragraph.content = nodes
    return { type: "doc", content: [paragraph] }
}

const gen = JSON.parse(fs.readFileSync(flag("--spans"), "utf8"))
const cases = []
for (const d of gen.docs.slice(0, 2)) for (const e of d.expected) cases.push(e.spans)
/* corners */
cases.push([])
cases.push([{ text: "", marks: {} }])
cases.push([{ text: "ab", marks: { comment: [] } }, { text: "cd", marks: {} }, { text: "e", marks: { strong: { active: true } } }])
cases.push([{ text: "x", marks: { link: { url: "u" }, comment: [{ id: "a" }, { id: "b" }], em: { active: true }, strong: { active: true } } }])
fs.writeFileSync(flag("--out"), JSON.stringify({
    generated_by: "oracle/gen_pm_golden.js",
    schema: { ALL_MARKS: S.ALL_MARKS, markRankOrder: S.markRankOrder, attrs: Object.keys(S.markSpec).reduce((o, t) => Object.assign(o, { [t]: S.markSpec[t].attrs }), {}) },
    cases: cases.map(spans => ({ spans, doc: docOf(spans) })),
}))
console.log("wrote " + flag("--out") + ": " + cases.length + " cases")
