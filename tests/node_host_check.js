"use strict"
/*
 * Test driver for the JS host (peritext_amd/node).  It does not import oracle/: expected values come from the
 * committed golden fixtures.
 *   node tests/node_host_check.js encode <fixture.json>   print sha256 of every encoded column (no GPU, no addon)
 *   node tests/node_host_check.js load                    the addon loads and dlopens libperitext_hip.so (no GPU)
 *   node tests/node_host_check.js run <fixture.json>...   GPU: applyChanges + the replica() surface vs the fixture's spans
 */
const fs = require("fs")
const path = require("path")
const crypto = require("crypto")
const assert = require("assert")
const host = require(path.join(__dirname, "..", "peritext_amd", "node"))

const cmd = process.argv[2]
const sha = ta => crypto.createHash("sha256").update(Buffer.from(ta.buffer, ta.byteOffset, ta.byteLength)).digest("hex")
const norm = spans => spans.map(s => ({ text: s.text, marks: JSON.parse(JSON.stringify(s.marks, Object.keys(s.marks).sort())) }))

if (cmd === "encode") {
    const gen = JSON.parse(fs.readFileSync(process.argv[3], "utf8"))
    const b = host.encodeDocs(gen.docs.map(d => d.logs))
    const out = { nLogs: b.nLogs, nOps: b.nOps, values: b.values, urls: b.urls, docComments: b.docComments }
    for (const k of ["logOff", "opId", "refA", "refB", "payload", "action", "markType", "sideA", "sideB", "logHdr", "chgOff", "chgActor", "chgSeq", "chgNops", "chgDeps"]) out[k] = sha(b[k])
    out.maxActors = b.maxActors
    console.log(JSON.stringify(out))
} else if (cmd === "load") {
    const addon = require(path.join(__dirname, "..", "peritext_amd", "node", "peritext_node.node"))
    const v = addon.open(path.join(__dirname, "..", "peritext_amd", "lib", "libperitext_hip.so"))
    assert.strictEqual(typeof addon.applyMaterialize, "function")
    console.log(JSON.stringify({ abi: v, kernel: addon.kernelName(), exports: Object.keys(addon).sort() }))
} else if (cmd === "run") {
    const engine = new host.MergeEngine()
    let logs = 0
    for (const f of process.argv.slice(3)) {
        const gen = JSON.parse(fs.readFileSync(f, "utf8"))
        const got = engine.applyChanges(gen.docs.map(d => d.logs))
        gen.docs.forEach((d, di) =>
            d.expected.forEach((e, ri) => {
                assert.deepStrictEqual(norm(got[di][ri]), norm(e.spans), f + " doc " + di + " replica " + ri)
                logs++
            })
        )
        /* the reference's per-replica surface, batched under the hood */
        const reps = gen.docs[0].logs.map(() => engine.replica(0))
        gen.docs[0].logs.forEach((log, ri) => log.forEach(ch => reps[ri].applyChange(ch)))
        reps.forEach((r, ri) => assert.deepStrictEqual(norm(r.getTextWithFormatting(["text"])), norm(gen.docs[0].expected[ri].spans)))
        engine.pending = []
    }
    /* a failed log throws RangeError like micromerge.ts:752 */
    const bad = [[[{ actor: "a", seq: 1, deps: {}, startOp: 1, ops: [
        { opId: "1@a", action: "makeList", obj: "_root", key: "text" },
        { opId: "2@a", action: "set", obj: "1@a", elemId: "9@zz", insert: true, value: "x" }] }]]]
    assert.throws(() => engine.applyChanges(bad), e => e instanceof RangeError && /List element not found/.test(e.message))
    /* causal admission: a dropped change -> RangeError "Expected sequence number" (micromerge.ts:503) */
    {
        const gen = JSON.parse(fs.readFileSync(process.argv[3], "utf8"))
        const log = gen.docs[0].logs[1].slice()
        log.splice(3, 1)
        assert.throws(() => engine.applyChanges([[log]]), e => e instanceof RangeError && /Expected sequence number|Missing dependency/.test(e.message))
    }
    engine.close()
    console.log(JSON.stringify({ ok: true, logs }))
} else if (cmd === "patches") {
    /* GPU: the Patch[] every applyChange returns (fixtures made by the reference itself, oracle/gen_patch_golden.js) */
    const engine = new host.MergeEngine()
    let logs = 0, patches = 0
    for (const f of process.argv.slice(3)) {
        const gen = JSON.parse(fs.readFileSync(f, "utf8"))
        const got = engine.applyChangesWithPatches(gen.docs.map(d => d.logs))
        gen.docs.forEach((d, di) =>
            d.expected.forEach((e, ri) => {
                assert.deepStrictEqual(norm(got.spans[di][ri]), norm(e.spans))
                assert.strictEqual(got.patches[di][ri].length, d.logs[ri].length, "one Patch[] per applied change")
                const flat = [].concat(...got.patches[di][ri])
                assert.deepStrictEqual(flat, e.patches, f + " doc " + di + " replica " + ri)
                logs++
                patches += flat.length
            })
        )
        /* per-replica handles: getPatches() = the returns of the queued applyChange calls */
        const reps = gen.docs[0].logs.map(() => engine.replica(0))
        gen.docs[0].logs.forEach((log, ri) => log.forEach(ch => assert.deepStrictEqual(reps[ri].applyChange(ch), [])))
        reps.forEach((r, ri) => assert.deepStrictEqual([].concat(...r.getPatches()), gen.docs[0].expected[ri].patches))
        engine.pending = []
    }
    engine.close()
    console.log(JSON.stringify({ ok: true, logs, patches }))
} else if (cmd === "pmdoc") {
    /* no GPU: ProseMirror doc JSON of every expected span list of a fixture */
    const gen = JSON.parse(fs.readFileSync(process.argv[3], "utf8"))
    console.log(JSON.stringify(gen.docs.map(d => d.expected.map(e => host.prosemirrorDocFromSpans(e.spans)))))
} else if (cmd === "decode") {
    /* no GPU: decodeChanges inverts encodeDocs */
    const gen = JSON.parse(fs.readFileSync(process.argv[3], "utf8"))
    const docs = gen.docs.map(d => d.logs)
    const b = host.encodeDocs(docs)
    let log = 0
    docs.forEach(logs => logs.forEach(l => assert.deepStrictEqual(host.decodeChanges(b, log++), l)))
    console.log(JSON.stringify({ ok: true, logs: log }))
} else if (cmd === "generate") {
    /* GPU: on-device change() reproduces the committed PTXGEN fixtures (config + seed in the file) and merges them to their spans */
    const engine = new host.MergeEngine()
    let logs = 0
    for (const f of process.argv.slice(3)) {
        const gen = JSON.parse(fs.readFileSync(f, "utf8"))
        const c = gen.cfg
        const got = engine.generate({ replicas: c.replicas, opsPerLog: c.opsPerLog, mix: c.mix, markTypes: c.markTypes, seed: gen.seed, nDocs: gen.docs.length, firstDoc: gen.docs[0].docIndex })
        gen.docs.forEach((d, di) =>
            d.logs.forEach((l, ri) => {
                assert.deepStrictEqual(got.docs[di][ri], l, f + " doc " + di + " replica " + ri)
                assert.deepStrictEqual(norm(got.spans[di][ri]), norm(d.expected[ri].spans))
                logs++
            })
        )
    }
    engine.close()
    console.log(JSON.stringify({ ok: true, logs }))
} else {
    console.error("usage: encode|load|run|patches|decode|generate")
    process.exit(2)
}
