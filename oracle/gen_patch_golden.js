#!/usr/bin/env node
"use strict"
/*
 * ORACLE TOOLING — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Golden Patch[] streams (SURVEY §8 f1): PTXGEN documents, every replica log applied to a fresh replica with
 * applyChange (reference/src/micromerge.ts:499); the concatenated returns are the expected stream.
 *
 *   node oracle/gen_patch_golden.js --config mini --docs 6 --seed 11 [--ops N] [--impl oracle|ref] --out tests/golden/patches_mini.json
 *
 *   node oracle/gen_patch_golden.js --from tests/golden/ptxgen_config5_8192.json [--impl ref] --out tests/golden/patches_config5_8192.json
 *       the streams of an EXISTING fixture's logs; the logs are not repeated: FILE = {from, impl, docs:[{expected:[{patches}]}]}
 *
 * --impl ref uses the type-erased build of the reference itself (oracle/_ref, see build_ref.js): that is how the
 * committed fixtures were made.  FILE = {config, seed, impl, docs:[{logs, expected:[{spans, text, patches}]}]}
 */
const fs = require("fs")
const path = require("path")
const { execFileSync } = require("child_process")
const os = require("os")

const argv = process.argv.slice(2)
const flag = (n, d) => (argv.indexOf(n) >= 0 ? argv[argv.indexOf(n) + 1] : d)
const cli = path.join(__dirname, "cli.js")
const tmp = fs.mkdtempSync(path.join(os.tmpdir(), "ptxpatch-"))
const gen = path.join(tmp, "gen.json")
const out = path.join(tmp, "out.json")
const impl = flag("--impl", "ref")
if (flag("--from", null)) {
    execFileSync(process.execPath, [cli, "apply", "--in", flag("--from"), "--impl", impl, "--patches", "--out", out])
    const a = JSON.parse(fs.readFileSync(out, "utf8"))
    const docs = a.docs.map(d => ({ expected: d.expected.map(e => ({ patches: e.patches })) }))
    fs.writeFileSync(flag("--out"), JSON.stringify({ from: path.basename(flag("--from")), impl, docs }))
    process.exit(0)
}
const genArgs = [cli, "gen", "--config", flag("--config", "mini"), "--docs", flag("--docs", "4"), "--seed", flag("--seed", "1"), "--out", gen]
if (flag("--ops", null)) genArgs.push("--ops", flag("--ops"))
execFileSync(process.execPath, genArgs)
execFileSync(process.execPath, [cli, "apply", "--in", gen, "--impl", impl, "--patches", "--out", out])
const g = JSON.parse(fs.readFileSync(gen, "utf8"))
const a = JSON.parse(fs.readFileSync(out, "utf8"))
const docs = g.docs.map((d, i) => ({ docIndex: d.docIndex, logs: d.logs, expected: a.docs[i].expected }))
fs.writeFileSync(flag("--out"), JSON.stringify({ config: g.config, seed: g.seed, impl, docs }))
