#!/usr/bin/env node
"use strict"
/*
 * ORACLE CLI — TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Callers: tests/, __graft_entry__.smoke(),
 * bench.py's cpu_baseline leg.
 *
 *   node oracle/cli.js gen   --config mini|config2..5 [--docs D] [--seed S] [--first F] [--ops N] [--replicas R]
 *                            [--mix i,d,a,r] [--marks strong,em,...] [--initial-text T]
 *                            [--impl oracle|ref] --out FILE
 *       PTXGEN traces + expected output.  FILE = {config, seed, docs:[{docIndex, seed, actors,
 *       logs:[Change[] per replica], expected:[{spans, text} per replica]}]}
 *   node oracle/cli.js apply --in FILE [--impl oracle|ref] [--cursors] [--patches] [--timing] [--no-patches] [--list-key K] --out FILE
 *       FILE in  = {docs:[{logs:[Change[]...]}]} (e.g. a reference trace or a KAT);  every log is applied
 *       to a FRESH replica with applyChange (micromerge.ts:499) and flattened (peritext.ts:337).
 *       FILE out = {docs:[{expected:[{spans, text, error?}]}]}; --timing adds timing: {seconds, ops, logs} = the time spent in
 *       applyChange over every change + getTextWithFormatting of every log (the CPU-baseline leg of bench.py: whole logs)
 *   node oracle/cli.js change --in FILE [--impl oracle|ref] --out FILE
 *       FILE in  = {replicas:[{actor, log: Change[], calls: InputOperation[][]}]}: a replica `actor` is rebuilt from its log with
 *       applyChange (its own changes included; `seq` is then set to its clock entry — a replica is never rebuilt like that
 *       upstream), then every entry of `calls` is one doc.change(ops) (micromerge.ts:308).
 *       FILE out = {replicas:[{changes: Change[], error?}]}
 *   node oracle/cli.js time  --in FILE [--impl oracle|ref] [--budget-ms T] [--whole] [--spans-out FILE]
 *       CPU baseline: time applyChange over every change of every log + getTextWithFormatting, one log
 *       after another on this core until the budget is spent; prints one JSON line
 *       {impl, logs, ops, seconds, ops_per_s}.
 */
const fs = require("fs")
const path = require("path")
const O = require("./peritext_oracle")
const G = require("./ptxgen")

const argv = process.argv.slice(2)
const cmd = argv[0]
const flag = (n, d) => (argv.indexOf(n) >= 0 ? argv[argv.indexOf(n) + 1] : d)

function implClass(name) {
    if (name === "ref") return require(path.join(__dirname, "_ref", "micromerge.js")).default
    return O.Micromerge
}
function liveChange(change, impl) {
    /* the reference compares ROOT/HEAD by Symbol identity: translate the portable strings back */
    if (impl !== "ref") return change
    const R = require(path.join(__dirname, "_ref", "micromerge.js"))
    const ops = change.ops.map(op => {
        const o = Object.assign({}, op)
        if (o.obj === O.ROOT) o.obj = R.ROOT
        if (o.elemId === O.HEAD) o.elemId = R.HEAD
        return o
    })
    return Object.assign({}, change, { ops })
}

function applyLog(Impl, impl, log, patchSink) {
    /* --no-patches: the oracle's pure speed switch (opts.patches === false skips the Patch[] bookkeeping, not a single state
       transition); the reference has no such switch and ignores it */
    const doc = argv.indexOf("--no-patches") >= 0 && impl !== "ref" ? new Impl("oracle-reader", { patches: false }) : new Impl("oracle-reader")
    for (const c of log) {
        const patches = doc.applyChange(liveChange(O.normalizeChange(c), impl))
        /* the makeList patch is the raw op (incl. a Symbol obj in the reference): keep only its action */
        if (patchSink) for (const p of patches) patchSink.push(p.action === "makeList" ? { action: "makeList" } : p)
    }
    return doc
}
/* getRoot() (micromerge.ts:443-449) as plain JSON: the list objects (the text) only as a marker */
function rootOf(v) {
    if (Array.isArray(v)) return { $list: true }
    if (v && typeof v === "object") {
        const o = {}
        for (const k of Object.keys(v)) o[k] = rootOf(v[k])
        return o
    }
    return v
}

/* --list-key K: the list object under root key K instead of "text" (a document may hold several: micromerge.ts:589); "a.b": the list under key b of the map
 * under root key a (an OperationPath, micromerge.ts:178-196) */
const LIST_KEY = flag("--list-key", "text")
const LIST_PATH = LIST_KEY.split(".")
function expectedOf(doc) {
    let text = []
    try {
        let at = doc.root
        for (const k of LIST_PATH) at = at[k]
        text = (at || []).slice()
    } catch (e) {
        text = []
    }
    let spans
    try {
        spans = doc.getTextWithFormatting(LIST_PATH)
    } catch (e) {
        return { spans: null, text, error: String(e.message) }
    }
    return { spans, text }
}

if (cmd === "gen") {
    const name = flag("--config", "mini")
    const cfg = Object.assign({}, G.CONFIGS[name])
    if (!cfg.mix) throw new Error("unknown config " + name)
    if (flag("--ops", null)) cfg.opsPerLog = parseInt(flag("--ops"), 10)
    if (flag("--replicas", null)) cfg.replicas = parseInt(flag("--replicas"), 10)
    if (flag("--mix", null)) cfg.mix = flag("--mix").split(",").map(x => parseInt(x, 10)) /* percent: insert,delete,addMark,removeMark */
    if (flag("--marks", null) !== null) cfg.markTypes = flag("--marks") === "" ? [] : flag("--marks").split(",")
    if (flag("--initial-text", null) !== null) cfg.initialText = flag("--initial-text")
    const nDocs = parseInt(flag("--docs", "4"), 10)
    const first = parseInt(flag("--first", "0"), 10)
    const seed = parseInt(flag("--seed", "1"), 10)
    const impl = flag("--impl", "oracle")
    const Impl = implClass(impl)
    const out = { config: name, cfg, seed, impl, docs: [] }
    for (let d = first; d < first + nDocs; d++) {
        const g = G.generateDoc(id => new Impl(id, { patches: false }), cfg, seed, d)
        out.docs.push({
            docIndex: d,
            seed: g.seed,
            actors: g.actors,
            logs: g.logs,
            expected: g.replicas.map(expectedOf),
        })
    }
    fs.writeFileSync(flag("--out"), JSON.stringify(out))
} else if (cmd === "apply") {
    const impl = flag("--impl", "oracle")
    const Impl = implClass(impl)
    const input = JSON.parse(fs.readFileSync(flag("--in"), "utf8"))
    const out = { impl, docs: [] }
    const wantCursors = argv.indexOf("--cursors") >= 0
    const wantPatches = argv.indexOf("--patches") >= 0
    const wantRoots = argv.indexOf("--roots") >= 0
    const timing = { seconds: 0, ops: 0, logs: 0 }
    for (const d of input.docs) {
        const expected = []
        for (const log of d.logs) {
            try {
                const patches = wantPatches ? [] : null
                const s0 = process.hrtime.bigint()
                const doc = applyLog(Impl, impl, log, patches)
                const e = expectedOf(doc)
                timing.seconds += Number(process.hrtime.bigint() - s0) / 1e9
                timing.ops += log.reduce((a, c) => a + c.ops.length, 0) - 1 /* the makeList */
                timing.logs += 1
                if (wantPatches) e.patches = patches /* the concatenated returns of applyChange (micromerge.ts:499) */
                if (wantRoots) e.root = rootOf(doc.root)
                if (wantCursors) {
                    /* micromerge.ts:465-477: getCursor for every visible index, resolveCursor for every element ever inserted */
                    e.cursorAt = e.text.map((_, i) => doc.getCursor(["text"], i).elemId)
                    e.cursorResolve = {}
                    const objectId = e.text.length ? doc.getCursor(["text"], 0).objectId : null
                    for (const c of log)
                        for (const op of c.ops)
                            if (op.action === "set" && op.insert && objectId !== null) e.cursorResolve[op.opId] = doc.resolveCursor({ objectId, elemId: op.opId })
                }
                expected.push(e)
            } catch (e) {
                expected.push({ spans: null, text: [], error: (e instanceof RangeError ? "RangeError: " : "Error: ") + e.message })
            }
        }
        out.docs.push({ expected })
    }
    if (argv.indexOf("--timing") >= 0) out.timing = timing
    fs.writeFileSync(flag("--out"), JSON.stringify(out))
} else if (cmd === "change") {
    const impl = flag("--impl", "oracle")
    const Impl = implClass(impl)
    const input = JSON.parse(fs.readFileSync(flag("--in"), "utf8"))
    const out = { impl, replicas: [] }
    const R = impl === "ref" ? require(path.join(__dirname, "_ref", "micromerge.js")) : null
    for (const rep of input.replicas) {
        const changes = []
        try {
            const doc = new Impl(rep.actor)
            for (const c of rep.log) doc.applyChange(liveChange(O.normalizeChange(c), impl))
            doc.seq = doc.clock[rep.actor] || 0
            for (const ops of rep.calls) {
                const r = doc.change(ops)
                /* portable form: ROOT / HEAD as strings */
                const c = JSON.parse(JSON.stringify(r.change))
                r.change.ops.forEach((op, i) => {
                    if (op.obj === O.ROOT || (R && op.obj === R.ROOT)) c.ops[i].obj = O.ROOT
                    if (op.elemId === O.HEAD || (R && op.elemId === R.HEAD)) c.ops[i].elemId = O.HEAD
                })
                changes.push(c)
            }
            out.replicas.push({ changes })
        } catch (e) {
            out.replicas.push({ changes, error: (e instanceof RangeError ? "RangeError: " : "Error: ") + e.message })
        }
    }
    fs.writeFileSync(flag("--out"), JSON.stringify(out))
} else if (cmd === "time") {
    const impl = flag("--impl", "oracle")
    const Impl = implClass(impl)
    const budgetMs = parseFloat(flag("--budget-ms", "10000"))
    const whole = process.argv.includes("--whole") /* whole logs: the budget is only a deadline against a hang (a log it cuts is reported as cut) */
    const spansOut = flag("--spans-out", null)
    const outDocs = []
    const input = JSON.parse(fs.readFileSync(flag("--in"), "utf8"))
    let logs = 0
    let ops = 0
    let truncated = 0
    const t0 = process.hrtime.bigint()
    const spent = () => Number(process.hrtime.bigint() - t0) / 1e6
    let elapsed = 0
    outer: for (const d of input.docs) {
        for (const log of d.logs) {
            const live = log.map(c => liveChange(O.normalizeChange(c), impl))
            const s0 = process.hrtime.bigint()
            const doc = new Impl("oracle-reader")
            let done = 0
            let cut = false
            for (const c of live) {
                doc.applyChange(c)
                done += c.ops.length
                /* the per-op cost GROWS along a log (O(n) scans, bigger slot sets), so a log cut at the
                   deadline makes the CPU look faster than it is on whole logs: conservative for the CPU */
                if (spent() > budgetMs) {
                    cut = done < log.reduce((a, c2) => a + c2.ops.length, 0)
                    break
                }
            }
            doc.getTextWithFormatting(["text"])
            elapsed += Number(process.hrtime.bigint() - s0) / 1e6
            ops += done - 1 /* the makeList */
            if (cut) truncated++
            else logs++
            /* --spans-out: what the replica shows at the end of a WHOLE log (outside the timed section): bench.py compares it with the device's rows, so the
               logs this leg times through the reference's own code are parity checks against the reference as well */
            if (spansOut) outDocs.push({ expected: [cut ? null : expectedOf(doc)] })
            if (spent() > budgetMs) break outer
        }
    }
    if (spansOut) fs.writeFileSync(spansOut, JSON.stringify({ impl, docs: outDocs }))
    console.log(JSON.stringify({ impl, logs, truncated_logs: truncated, ops, seconds: elapsed / 1e3, ops_per_s: ops / (elapsed / 1e3) }))
} else {
    console.error("usage: cli.js gen|apply|change|time ... (see header)")
    process.exit(2)
}
