#!/usr/bin/env node
"use strict"
/*
 * DIFFERENTIAL FUZZ: oracle/peritext_oracle.js vs the reference itself (oracle/_ref, built by
 * oracle/build_ref.js from /root/reference).  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * For every seed, PTXGEN drives both implementations with identical random decisions and compares
 *   (1) every generated Change (so change()/changeMark/getListElementId agree, incl. lookAfterTombstones),
 *   (2) every Patch[] returned by change()/applyChange() (incremental path),
 *   (3) the final getTextWithFormatting of every replica (batch path).
 * This pins the oracle on behaviour no reference test covers: removeMark of all four mark types,
 * `comment: []`, concurrent comment add/remove, zero-width marks (SURVEY §8c "parity unpinned").
 *
 * Usage: node oracle/diff_fuzz.js [--config mini] [--docs 200] [--seed 1] [--ops N] [--zero-width]
 */
const path = require("path")
const assert = require("assert")
const O = require("./peritext_oracle")
const G = require("./ptxgen")

const argv = process.argv.slice(2)
const flag = (n, d) => (argv.indexOf(n) >= 0 ? argv[argv.indexOf(n) + 1] : d)
let Ref
try {
    Ref = require(path.join(__dirname, "_ref", "micromerge.js")).default
} catch (e) {
    console.error("oracle/_ref is not built (run: node oracle/build_ref.js): " + e.message)
    process.exit(2)
}
const cfg = Object.assign({}, G.CONFIGS[flag("--config", "mini")])
if (flag("--ops", null)) cfg.opsPerLog = parseInt(flag("--ops"), 10)
const nDocs = parseInt(flag("--docs", "200"), 10)
const seed = parseInt(flag("--seed", "1"), 10)

let mismatches = 0
let totalOps = 0
for (let d = 0; d < nDocs; d++) {
    const pa = []
    const pb = []
    const a = G.generateDoc(id => new O.Micromerge(id), cfg, seed, d, { onPatches: (i, p) => pa.push([i, JSON.parse(JSON.stringify(p))]) })
    const b = G.generateDoc(id => new Ref(id), cfg, seed, d, { onPatches: (i, p) => pb.push([i, G.portable({ ops: [] }) && JSON.parse(JSON.stringify(p))]) })
    try {
        assert.deepStrictEqual(a.logs, b.logs, "generated changes / application order differ")
        /* makeList patches carry the raw op incl. obj (a Symbol in the reference): compare without it */
        const strip = ps => ps.map(([i, list]) => [i, list.map(p => (p.action === "makeList" ? Object.assign({}, p, { obj: undefined }) : p))])
        assert.deepStrictEqual(strip(pa), strip(pb), "patch streams differ")
        for (let r = 0; r < cfg.replicas; r++) {
            assert.deepStrictEqual(
                a.replicas[r].getTextWithFormatting(["text"]),
                b.replicas[r].getTextWithFormatting(["text"]),
                "spans differ on replica " + r,
            )
        }
    } catch (e) {
        mismatches++
        console.log(`doc ${d} (seed ${a.seed}): ${e.message.split("\n")[0]}`)
        if (mismatches > 5) break
    }
    totalOps += a.logs[0].reduce((s, c) => s + c.ops.length, 0)
}
console.log(`diff_fuzz: ${nDocs} docs x ${cfg.replicas} replicas, ${totalOps} ops per replica set, config ${flag("--config", "mini")}: ${mismatches} mismatches`)
process.exit(mismatches === 0 ? 0 : 1)
