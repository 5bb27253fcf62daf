"use strict"
/*
 * The TypeScript-facing entry point end to end (GPU box): `Change[][][]` as JSON text — what reference/src/micromerge.ts:60-71 defines and test/fuzz.ts:16-20 exchanges —
 * through MergeEngine.applyChanges to FormatSpanWithText[] per replica.  Stages timed on their own: JSON.parse, encodeDocs (Change objects -> the typed arrays of the
 * C ABI), the whole applyChanges call (encode + upload + ptx_merge + download + span decode).
 *   node tools/js_host_bench.js <docs.json: [[Change[] per replica] per document]> [repeats]
 * Prints one JSON line.  Not part of `value`: bench.py's extras leg `js_host_end_to_end` runs it on documents the device generator made.
 */
const fs = require("fs")
const path = require("path")
const host = require(path.join(__dirname, "..", "peritext_amd", "node"))

const file = process.argv[2]
const repeats = parseInt(process.argv[3] || "3", 10)
const ms = t0 => Number(process.hrtime.bigint() - t0) / 1e6
const text = fs.readFileSync(file, "utf8")
let t0 = process.hrtime.bigint()
const docs = JSON.parse(text)
const parseMs = ms(t0)
let ops = 0
for (const d of docs) for (const log of d) for (const ch of log) ops += ch.ops.length
let encodeMs = Infinity
for (let r = 0; r < repeats; r++) {
    t0 = process.hrtime.bigint()
    host.encodeDocs(docs)
    encodeMs = Math.min(encodeMs, ms(t0))
}
const engine = new host.MergeEngine()
engine.applyChanges(docs.slice(0, 1)) /* (context, kernels and tables warm) */
let applyMs = Infinity, spans = 0
for (let r = 0; r < repeats; r++) {
    t0 = process.hrtime.bigint()
    const got = engine.applyChanges(docs)
    applyMs = Math.min(applyMs, ms(t0))
    spans = 0
    for (const d of got) for (const rep of d) spans += rep.length
}
engine.close()
console.log(JSON.stringify({ documents: docs.length, replica_logs: docs.reduce((a, d) => a + d.length, 0), ops, json_bytes: text.length, json_parse_ms: parseMs, encode_ms: encodeMs,
    apply_changes_ms: applyMs, spans_returned: spans, ops_per_s_from_change_objects: ops / applyMs * 1e3, ops_per_s_from_json_text: ops / (applyMs + parseMs) * 1e3,
    encode_ops_per_s: ops / encodeMs * 1e3, json_parse_ops_per_s: ops / parseMs * 1e3, host: "node " + process.version + ", one thread" }))
