#!/usr/bin/env node
"use strict"
/*
 * Fixture generator — TEST INFRASTRUCTURE.  Reads reference/traces/*.json (op logs of 9 saved fuzz
 * failures; their recorded outputs are from older code and are NOT used, SURVEY.md §8c), restores the
 * ROOT/HEAD that JSON dropped, and emits for every trace several causally valid delivery orders of the
 * same change set (per-actor queues concatenated forward / reversed, with retry) plus the spans the
 * ERASED REFERENCE ITSELF (oracle/_ref) produces for them.  Output: tests/golden/reference_traces.json.
 * Usage: node oracle/export_traces.js   (needs /root/reference and oracle/_ref)
 */
const fs = require("fs")
const path = require("path")
const O = require("./peritext_oracle")
const Ref = require("./_ref/micromerge.js")
const dir = "/root/reference/traces"
const out = []
for (const name of fs.readdirSync(dir).filter(f => f.endsWith(".json")).sort()) {
    const queues = JSON.parse(fs.readFileSync(path.join(dir, name), "utf8")).queues
    const actors = Object.keys(queues)
    const orders = [actors, actors.slice().reverse()]
    const logs = []
    let spans = null
    for (const order of orders) {
        const pending = []
        for (const a of order) for (const c of queues[a]) pending.push(O.normalizeChange(c))
        const doc = new Ref.default("reader")
        const log = []
        let spins = 0
        while (pending.length) {
            const c = pending.shift()
            const live = Object.assign({}, c, { ops: c.ops.map(op => Object.assign({}, op, op.obj === O.ROOT ? { obj: Ref.ROOT } : {}, op.elemId === O.HEAD ? { elemId: Ref.HEAD } : {})) })
            try {
                doc.applyChange(live)
                log.push(c)
            } catch (e) {
                if (!(e instanceof RangeError)) throw e
                pending.push(c)
            }
            if (spins++ > 10000) throw new Error("no causal order for " + name)
        }
        const s = doc.getTextWithFormatting(["text"])
        if (spans && JSON.stringify(spans) !== JSON.stringify(s)) throw new Error("reference does not converge on " + name)
        spans = s
        logs.push(log)
    }
    out.push({ name, logs, spans })
}
fs.writeFileSync(path.join(__dirname, "..", "tests", "golden", "reference_traces.json"), JSON.stringify(out) + "\n")
console.log("wrote " + out.length + " traces")
