"use strict"
/*
 * PTXGEN — seeded synthetic trace generator.  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Modelled on the reference's fuzzer (reference/test/fuzz.ts:23-199): R replicas start from
 * generateDocs("ABCDE", R) (fuzz.ts:157), then repeatedly a random replica makes one random edit
 * through its own `change()` and a random ordered pair of replicas syncs both ways
 * (fuzz.ts:181-199, merge.ts:4-38).  Deviations from fuzz.ts (it is unseeded, never emits removeMark
 * because of the bug at fuzz.ts:80, and never terminates) are the ones listed in SURVEY.md §8(d):
 *   - PRNG: mulberry32, seed = (0x5eed0000 + docIndex) ^ (configSeed * 0x9e3779b1); all draws are
 *     integer draws randInt(n) = floor(u32 * n / 2^32) so that the C++ port (peritext_amd/csrc/ptxgen.cc)
 *     makes bit-identical decisions;
 *   - insert: index uniform [0,len], 1-2 values from [0-9a-f];  delete: index uniform [1,len-1],
 *     count 1..min(3,len-index) (never index 0, fuzz.ts:127);  marks: start uniform [0,len-1],
 *     end = start+1+U[0,len-start-1] (fuzz.ts:34-35);  link urls "A.com".."Z.com" (fuzz.ts:28);
 *     comment ids "comment-<k>" (k = per-doc counter);  removeMark really emits removeMark and, for
 *     comments, only targets ids whose addMark the removing replica has already applied
 *     (SURVEY A.6-2: otherwise the REFERENCE ITSELF does not converge);
 *   - edits that are impossible in the current state degrade deterministically (delete/mark on a
 *     too-short doc -> insert; removeMark comment with no known id -> addMark comment);
 *   - sync delivers missing changes actor by actor in replica-index order, re-queueing a change
 *     whose dependencies are not met yet (same retry idea as merge.ts:11-17);
 *   - generation stops when the op log holds exactly `opsPerLog` internal ops (the single makeList
 *     excluded, the 5 initial inserts included), then a final full sync gives every replica the
 *     same op set in its own application order.
 *
 * The generator only touches the public surface (change / applyChange / root.text / clock /
 * getTextWithFormatting), so it can drive the oracle or the type-erased reference (oracle/_ref).
 */

function mulberry32(seed) {
    let a = seed >>> 0
    return function () {
        a = (a + 0x6d2b79f5) >>> 0
        let t = a
        t = Math.imul(t ^ (t >>> 15), t | 1)
        t ^= t + Math.imul(t ^ (t >>> 7), t | 61)
        return (t ^ (t >>> 14)) >>> 0
    }
}

/** Named workloads = BASELINE.json configs #2..#5 as concretised in SURVEY.md §8(d). */
const CONFIGS = {
    config2: { replicas: 1, opsPerLog: 256, mix: [70, 30, 0, 0], markTypes: [] },
    config3: { replicas: 1, opsPerLog: 1024, mix: [40, 20, 25, 15], markTypes: ["strong", "em"] },
    config4: { replicas: 3, opsPerLog: 4096, mix: [25, 25, 25, 25], markTypes: ["strong", "em", "link", "comment"] },
    config5: { replicas: 1, opsPerLog: 8192, mix: [20, 50, 20, 10], markTypes: ["link", "comment"] },
    /* insert-heavy, all mark types: documents GROW (hundreds of visible chars, many overlapping marks
       and comments) — stresses the mark sweep, which the tombstone-heavy BASELINE mixes barely touch */
    rich: { replicas: 3, opsPerLog: 1024, mix: [55, 10, 20, 15], markTypes: ["strong", "em", "link", "comment"] },
    /* the same at the headline's log length (bench.py extras: a batch-scale leg on documents that hold text) */
    rich4k: { replicas: 3, opsPerLog: 4096, mix: [55, 10, 20, 15], markTypes: ["strong", "em", "link", "comment"] },
    /* small all-features case used by unit tests and differential fuzzing */
    mini: { replicas: 3, opsPerLog: 96, mix: [25, 25, 25, 25], markTypes: ["strong", "em", "link", "comment"] },
}

const { normalizeChange } = require("./peritext_oracle")

/** JSON-safe copy of a change; the reference's ROOT/HEAD Symbols become "_root"/"_head" (SURVEY A.6-9). */
function portable(change) {
    return normalizeChange(JSON.parse(JSON.stringify(change)))
}

function docSeed(configSeed, docIndex) {
    return ((0x5eed0000 + docIndex) ^ Math.imul(configSeed >>> 0, 0x9e3779b1)) >>> 0
}

/**
 * Generate one document.
 * @param makeReplica  (actorId) -> replica object with the Micromerge surface
 * @param cfg          {replicas, opsPerLog, mix:[ins,del,add,rem] (percent), markTypes, initialText?}
 * @param configSeed   integer
 * @param docIndex     integer
 * @param hooks        optional {onPatches(replicaIndex, patches)} to observe every Patch[] produced
 * @returns {docIndex, seed, actors, logs: Change[][] (per replica, application order), replicas}
 */
function generateDoc(makeReplica, cfg, configSeed, docIndex, hooks) {
    const R = cfg.replicas
    const initialText = cfg.initialText === undefined ? "ABCDE" : cfg.initialText
    const seed = docSeed(configSeed, docIndex)
    const next = mulberry32(seed)
    const randInt = n => Math.floor((next() * n) / 4294967296)
    const onPatches = hooks && hooks.onPatches ? hooks.onPatches : () => {}

    const actors = []
    const docs = []
    for (let i = 0; i < R; i++) {
        actors.push("doc" + (i + 1))
        docs.push(makeReplica(actors[i]))
    }
    const queues = actors.map(() => []) /* every change made by actor i, in seq order (live objects) */
    const logs = actors.map(() => []) /* every change applied by replica i, in application order (portable copies) */
    const copies = new Map()
    const seen = actors.map(() => actors.map(() => 0)) /* seen[i][a] = #changes of actor a applied by replica i */
    const knownComments = actors.map(() => []) /* comment ids whose addMark replica i has applied */
    let commentCounter = 0

    const record = (i, change) => {
        if (!copies.has(change)) copies.set(change, portable(change))
        logs[i].push(copies.get(change))
        const a = actors.indexOf(change.actor)
        seen[i][a] = change.seq
        for (const op of change.ops) {
            if (op.action === "addMark" && op.markType === "comment") knownComments[i].push(op.attrs.id)
        }
    }

    /* generateDocs (generateDocs.ts:11-42): doc1 creates the list + initial text, everybody applies it */
    const first = docs[0].change([
        { path: [], action: "makeList", key: "text" },
        { path: ["text"], action: "insert", index: 0, values: initialText.split("") },
    ])
    onPatches(0, first.patches)
    queues[0].push(first.change)
    record(0, first.change)
    for (let i = 1; i < R; i++) {
        onPatches(i, docs[i].applyChange(first.change))
        record(i, first.change)
    }
    let opsSoFar = initialText.length

    /* deliver to `dst` everything `src` has applied and `dst` has not */
    const deliver = (src, dst) => {
        const pending = []
        for (let a = 0; a < R; a++) {
            for (let s = seen[dst][a]; s < seen[src][a]; s++) pending.push(queues[a][s])
        }
        let spins = 0
        while (pending.length > 0) {
            const c = pending.shift()
            let ok = true
            let patches
            try {
                patches = docs[dst].applyChange(c)
            } catch (e) {
                if (!(e instanceof RangeError)) throw e
                ok = false
            }
            if (ok) {
                onPatches(dst, patches)
                record(dst, c)
            } else {
                pending.push(c)
            }
            if (spins++ > 100000) throw new Error("ptxgen: sync did not converge")
        }
    }

    const HEX = "0123456789abcdef"
    while (opsSoFar < cfg.opsPerLog) {
        const k = randInt(R)
        const doc = docs[k]
        const len = doc.root.text.length
        const budget = cfg.opsPerLog - opsSoFar
        const x = randInt(100)
        let kind = x < cfg.mix[0] ? 0 : x < cfg.mix[0] + cfg.mix[1] ? 1 : x < cfg.mix[0] + cfg.mix[1] + cfg.mix[2] ? 2 : 3
        if (kind === 1 && len < 2) kind = 0
        if (kind >= 2 && (len < 1 || cfg.markTypes.length === 0)) kind = 0
        let input
        if (kind === 0) {
            const index = randInt(len + 1)
            let nvals = 1 + randInt(2)
            if (nvals > budget) nvals = budget
            const values = []
            for (let v = 0; v < nvals; v++) values.push(HEX[randInt(16)])
            input = { path: ["text"], action: "insert", index, values }
        } else if (kind === 1) {
            const index = 1 + randInt(len - 1)
            let count = 1 + randInt(Math.min(3, len - index))
            if (count > budget) count = budget
            input = { path: ["text"], action: "delete", index, count }
        } else {
            const startIndex = randInt(len)
            const endIndex = startIndex + 1 + randInt(len - startIndex)
            const markType = cfg.markTypes[randInt(cfg.markTypes.length)]
            let action = kind === 2 ? "addMark" : "removeMark"
            input = { path: ["text"], action, startIndex, endIndex, markType }
            if (markType === "link") {
                const url = String.fromCharCode(65 + randInt(26)) + ".com"
                if (action === "addMark") input.attrs = { url }
            } else if (markType === "comment") {
                if (action === "removeMark" && knownComments[k].length === 0) {
                    action = "addMark"
                    input.action = action
                }
                if (action === "addMark") input.attrs = { id: "comment-" + commentCounter++ }
                else input.attrs = { id: knownComments[k][randInt(knownComments[k].length)] }
            }
        }
        const made = doc.change([input])
        onPatches(k, made.patches)
        queues[k].push(made.change)
        record(k, made.change)
        opsSoFar += made.change.ops.length

        if (R > 1) {
            const left = randInt(R)
            let right = randInt(R - 1)
            if (right >= left) right++
            deliver(left, right)
            deliver(right, left)
        }
    }

    /* final full sync */
    for (let round = 0; round < R + 1; round++) {
        for (let i = 0; i < R; i++) for (let j = 0; j < R; j++) if (i !== j) deliver(i, j)
    }
    return { docIndex, seed, actors, logs, replicas: docs }
}

module.exports = { mulberry32, docSeed, generateDoc, portable, CONFIGS }
