"use strict"
/*
 * peritext_amd/node — the JavaScript/TypeScript host of the MI355X batch merge engine (types: index.d.ts).
 *
 * It keeps the reference's surface for the hot path — `Change` in (reference/src/micromerge.ts:60-71),
 * `FormatSpanWithText[]` out (src/peritext.ts:35-38) — but applies MANY replica op logs per call:
 *
 *     const { MergeEngine } = require("./peritext_amd/node")
 *     const engine = new MergeEngine()                       // one GPU context (N-API addon -> libperitext_hip.so)
 *     const spans = engine.applyChanges([[log1, log2], ...]) // docs -> replica logs -> Change[]  =>  spans per log
 *
 * and, for code written against the reference's per-replica calls, `engine.replica()` hands out objects with
 * `applyChange(change)` (micromerge.ts:499) and `getTextWithFormatting(["text"])` (:516) that batch under the hood.
 *
 * Encoding rules (same as peritext_amd/wire.py; SURVEY.md §8b): actor strings -> rank in UTF-16 code-unit order
 * per document so that integer order of (counter << 32 | rank) equals compareOpIds (micromerge.ts:812-827);
 * ROOT/HEAD -> 0 (JSON drops the reference's Symbols; "_root"/"_head" are accepted too); inserted values and
 * link urls -> ids in batch-wide string tables; comment ids -> doc-local dense ranks in code-unit order
 * (peritext.ts:318 keeps the arrays id-sorted).  Nothing here computes a merge: without the addon or without a
 * gfx950 device the constructor throws.
 */
const path = require("path")

const ACT = { MAKELIST: 0, INSERT: 1, DELETE: 2, ADDMARK: 3, REMOVEMARK: 4, NOP: 5, MAPSET: 6, MAPDEL: 7 }
const MAPV = { SCALAR: 0, MAP: 1, LIST: 2, DELETED: 3 } /* what a map row writes / ptx_root_entry.kind */
const MARK_NAMES = ["strong", "em", "comment", "link"] /* schema.ts:125 ALL_MARKS */
const SIDE_NAMES = ["before", "after", "startOfText", "endOfText"] /* peritext.ts:17-21 */
const ATTR = { STRONG: 0x10000000, EM: 0x20000000, LINK: 0x40000000, COMMENT: 0x80000000, ID_MASK: 0x0fffffff }
const STATUS_MESSAGES = {
    1: "List element not found" /* micromerge.ts:752 */,
    2: "Expected sequence number" /* :503 */,
    3: "Missing dependency" /* :507 */,
    4: "duplicate opId",
    5: "log exceeds on-chip capacity",
    6: "malformed op row",
    7: "List index out of bounds" /* :804 */,
}
const IN = { INSERT: 0, DELETE: 1, ADDMARK: 2, REMOVEMARK: 3, MAKELIST: 4, MAPSET: 5, MAPDEL: 6 }
const IN_OBJ_NEW = 0x80000000 /* ptx_input_ops.index of a map op: the object made by row k of this log's output */ /* ptx_input_ops.action */
const CHG_ACTOR_SHIFT = 20, CHG_NOPS = 0x000fffff, ENV_SATURATED = 65535
const envStride = maxActors => (1 + maxActors + 3) & ~3 /* PTX_ENV_STRIDE */

/** chgActor / chgSeq / chgNops / chgDeps -> the packed envelope the device reads: chgHdr = actor << 20 | nops, chgEnv rows of
 *  envStride(maxActors) u16 = seq, deps[...].  While every value is at most 65534 that is all; once some seq / dep is larger (seq and deps are
 *  plain numbers in the reference, micromerge.ts:499-511: one change per keystroke passes 65535) the values are split exactly into chgEnv
 *  (low halves) and chgEnvHi (high halves) — ptx_batch.chg_env_hi of include/peritext_hip.h. */
function packEnvelope(batch) {
    const n = batch.chgActor.length, es = envStride(batch.maxActors)
    batch.chgHdr = new Uint32Array(n)
    batch.chgEnv = new Uint16Array(n * es)
    delete batch.chgEnvHi
    let top = 0
    for (let c = 0; c < n; c++) top = Math.max(top, batch.chgSeq[c])
    for (let i = 0; i < batch.chgDeps.length; i++) top = Math.max(top, batch.chgDeps[i])
    const wide = top >= ENV_SATURATED
    if (wide) batch.chgEnvHi = new Uint16Array(n * es)
    for (let c = 0; c < n; c++) {
        if (batch.chgActor[c] > 4095 || batch.chgNops[c] > CHG_NOPS) throw new RangeError("a document has at most 4096 actors and a change at most " + CHG_NOPS + " ops")
        batch.chgHdr[c] = ((batch.chgActor[c] << CHG_ACTOR_SHIFT) | batch.chgNops[c]) >>> 0
        for (let k = 0; k <= batch.maxActors; k++) {
            const v = k === 0 ? batch.chgSeq[c] : batch.chgDeps[c * batch.maxActors + k - 1]
            batch.chgEnv[c * es + k] = v & 0xffff
            if (wide) batch.chgEnvHi[c * es + k] = v >>> 16
        }
    }
    return batch
}
/** the inverse, for batches that come back from the device (generate / change) */
function unpackEnvelope(batch) {
    const n = batch.chgHdr ? batch.chgHdr.length : 0, es = envStride(batch.maxActors)
    batch.chgActor = new Uint32Array(n)
    batch.chgNops = new Uint32Array(n)
    batch.chgSeq = new Uint32Array(n)
    batch.chgDeps = new Uint32Array(n * batch.maxActors)
    for (let c = 0; c < n; c++) {
        batch.chgActor[c] = batch.chgHdr[c] >>> CHG_ACTOR_SHIFT
        batch.chgNops[c] = batch.chgHdr[c] & CHG_NOPS
        const hi = batch.chgEnvHi && batch.chgEnvHi.length ? batch.chgEnvHi : null
        batch.chgSeq[c] = (batch.chgEnv[c * es] | (hi ? hi[c * es] << 16 : 0)) >>> 0
        for (let a = 0; a < batch.maxActors; a++) batch.chgDeps[c * batch.maxActors + a] = (batch.chgEnv[c * es + 1 + a] | (hi ? hi[c * es + 1 + a] << 16 : 0)) >>> 0
    }
    return batch
}
const ROOT = "_root"
const HEAD = "_head"
const ID_RE = /^([0-9]+)@([\s\S]*)$/

function splitOpId(id) {
    const m = ID_RE.exec(id)
    if (!m) throw new Error("Invalid operation ID: " + id)
    return [parseInt(m[1], 10), m[2]]
}

/** The list object a replica's root shows under `path` once it has applied `log`: every key of a map holds the write with the LARGEST opId
 *  (reference/src/micromerge.ts:572-602 keeps a write iff compareOpIds says its id is larger than the key's current one), so two replicas that made a list
 *  under one key concurrently both show the same one.  Its opId, or null when the path does not end at a list.  (wire.resolve_list_path is the Python twin.) */
function resolveListPath(log, path) {
    const win = new Map() /* obj + "\u0000" + key -> {ctr, actor, op} */
    for (const ch of log)
        for (const op of ch.ops) {
            if (op.key === undefined || op.elemId !== undefined) continue
            const a = op.action
            if (a !== "set" && a !== "del" && a !== "makeMap" && a !== "makeList") continue
            const obj = op.obj === undefined || op.obj === null || typeof op.obj === "symbol" ? ROOT : op.obj
            const k = obj + "\u0000" + op.key
            const id = splitOpId(op.opId)
            const cur = win.get(k)
            if (cur === undefined || id[0] > cur.ctr || (id[0] === cur.ctr && id[1] > cur.actor)) win.set(k, { ctr: id[0], actor: id[1], op })
        }
    let cur = ROOT
    for (let i = 0; i < path.length; i++) {
        const w = win.get(cur + "\u0000" + path[i])
        if (w === undefined || w.op.action !== (i === path.length - 1 ? "makeList" : "makeMap")) return null
        cur = w.op.opId
    }
    return cur
}

/** docs: Change[][][] (doc -> replica log -> changes in application order)  ->  SoA batch (include/peritext_hip.h).
 *  opts.extraActors / opts.extraComments: per document, actor names / comment ids that get a rank although no change uses them
 *  yet (a replica about to make its first change, comment ids a later InputOperation introduces: ranks are positions in the
 *  document's sorted id list, so they are reserved before the rows that use them exist).  opts.textObjs: per document the opId
 *  of the text list when the logs of this batch do not hold its makeList (a batch of newly arrived changes only).
 *  opts.listKeys (round 5; default ["text"]): the LIST objects of the root map to merge, by key.  The engine merges one list object per device log; a document that
 *  holds several (micromerge.ts:589: makeList under any key; :534-571: applyOp takes any list object) becomes one device log per (replica, key), in that order — each
 *  with the replica's whole Change envelope and, of the list ops, those of ITS list (the ops on the replica's other list objects are rows without effect there);
 *  batch.logList[l] names the key of device log l, batch.logReplica[l] its replica within the document. */
function encodeDocs(docs, opts) {
    const listKeys = (opts && opts.listKeys) || ["text"]
    const extraActors = (opts && opts.extraActors) || [], extraComments = (opts && opts.extraComments) || []
    const textObjs = (opts && opts.textObjs) || [] /* per document: opId of the text list when the logs do not hold its makeList (or one entry per log) */
    /* opts.seed: the tables of an earlier batch of the same documents — ids already given stay (the rows of Changes appended to a resident batch) */
    const seed = (opts && opts.seed) || {}
    const values = (seed.values || []).slice(), valueIx = new Map(values.map((v, i) => [v, i]))
    const urls = (seed.urls || []).slice(), urlIx = new Map(urls.map((v, i) => [v, i]))
    const keys = (seed.keys || ["text"]).slice(), keyIx = new Map(keys.map((v, i) => [v, i])) /* keys of the map objects (ref_b of the map rows); key 0 of every batch: the text list's */
    const mapValues = (seed.mapValues || []).slice(), mapValueIx = new Map(mapValues.map((v, i) => [v, i])) /* JSON text of the values the map rows set */
    const intern = (table, index, v) => {
        if (!index.has(v)) {
            index.set(v, table.length)
            table.push(v)
        }
        return index.get(v)
    }
    /* JSON with sorted keys: the same text wire.py makes (json.dumps(sort_keys=True) only differs in separators, which never reach the device) */
    /* The columns are written straight into typed arrays sized by a first pass over the Changes (an op of a replica is a row of each of its device logs);
     * the 64-bit ids as two 32-bit halves (low: actor rank, high: counter — no BigInt arithmetic per op), ids parsed by hand (no regular expression, no
     * array per id), an actor's rank found by comparing the id's tail with the few actor names of the document (round 5: 0.2 -> 1.5 M ops/s per core). */
    let totalRows = 0, totalChanges = 0
    for (const logs of docs)
        for (const log of logs) {
            totalChanges += log.length * listKeys.length
            for (const ch of log) totalRows += ch.ops.length * listKeys.length
        }
    const colOpId = new BigUint64Array(totalRows), colRefA = new BigUint64Array(totalRows), colRefB = new BigUint64Array(totalRows)
    const opId32 = new Uint32Array(colOpId.buffer), refA32 = new Uint32Array(colRefA.buffer), refB32 = new Uint32Array(colRefB.buffer)
    const colPayload = new Uint32Array(totalRows), colAction = new Uint8Array(totalRows), colMarkType = new Uint8Array(totalRows)
    const colSideA = new Uint8Array(totalRows), colSideB = new Uint8Array(totalRows)
    let nRow = 0
    const logOff = [0]
    const chgOff = [0], chgActor = new Uint32Array(totalChanges), chgSeq = new Uint32Array(totalChanges), chgNops = new Uint32Array(totalChanges)
    const depChg = [], depActor = [], depValue = [] /* (change, actor rank, value) of every dependency: the rows of chgDeps once the widest document is known */
    let nChg = 0
    let maxActors = 1
    const logDoc = [], docActors = [], docComments = [], logList = [], logReplica = []
    /* "<counter>@<actor>": the counter; idAt = where the actor starts.  The same ids the regular expression ^([0-9]+)@([\s\S]*)$ took */
    let idAt = 0
    const idCounter = id => {
        const at = id.indexOf("@")
        if (at <= 0) throw new Error("Invalid operation ID: " + id)
        let ctr = 0
        for (let i = 0; i < at; i++) {
            const c = id.charCodeAt(i) - 48
            if (c < 0 || c > 9) throw new Error("Invalid operation ID: " + id)
            ctr = ctr * 10 + c
        }
        idAt = at + 1
        return ctr
    }
    docs.forEach((logs, d) => {
        const actors = new Set(), comments = new Set()
        /* the actor of an id without a substring per id: most ids carry the actor of the id before them */
        let lastActor = null
        const noteActorOf = id => {
            if (typeof id !== "string") throw new Error("Invalid operation ID: " + id)
            idCounter(id)
            if (lastActor !== null && id.length - idAt === lastActor.length && id.endsWith(lastActor)) return
            lastActor = id.slice(idAt)
            actors.add(lastActor)
        }
        for (const log of logs)
            for (const ch of log) {
                actors.add(ch.actor)
                if (ch.deps) for (const a in ch.deps) actors.add(a)
                for (const op of ch.ops) {
                    noteActorOf(op.opId)
                    const r0 = op.elemId, r1 = op.start && op.start.elemId, r2 = op.end && op.end.elemId
                    if (typeof r0 === "string" && r0 !== HEAD && r0 !== ROOT) noteActorOf(r0)
                    if (typeof r1 === "string" && r1 !== HEAD && r1 !== ROOT) noteActorOf(r1)
                    if (typeof r2 === "string" && r2 !== HEAD && r2 !== ROOT) noteActorOf(r2)
                    if (op.markType === "comment") comments.add(op.attrs.id)
                    if (typeof op.obj === "string" && op.obj !== HEAD && op.obj !== ROOT) noteActorOf(op.obj)
                }
            }
        for (const a of extraActors[d] || []) actors.add(a)
        for (const c of extraComments[d] || []) comments.add(c)
        const actorList = Array.from(actors).sort() /* default sort = UTF-16 code-unit order = JS `<` (micromerge.ts:826) */
        /* comment ids: ranks in string order — or, for Changes appended to a resident batch (opts.commentOrder), the ids that batch knows keep theirs and the
         * new ones follow (the device only ever compares comment ids for equality; the decoders order them by their strings) */
        const known = (opts && opts.commentOrder && opts.commentOrder[d]) || []
        for (const c of known) comments.delete(c)
        const commentList = known.concat(Array.from(comments).sort())
        const arank = new Map(actorList.map((a, i) => [a, i]))
        const crank = new Map(commentList.map((c, i) => [c, i]))
        docActors.push(actorList)
        docComments.push(commentList)
        maxActors = Math.max(maxActors, actorList.length)
        /* id -> (counter, actor rank) written as the two halves of 64-bit word i of a column; undefined / HEAD / ROOT: 0 */
        let lastRank = -1
        lastActor = null
        const putId = (col32, i, s) => {
            if (s === undefined || s === null || s === HEAD || s === ROOT || typeof s === "symbol") return
            if (typeof s !== "string") throw new Error("Invalid operation ID: " + s)
            const ctr = idCounter(s)
            if (lastActor === null || s.length - idAt !== lastActor.length || !s.endsWith(lastActor)) {
                lastActor = s.slice(idAt)
                lastRank = arank.get(lastActor)
            }
            col32[2 * i] = lastRank
            col32[2 * i + 1] = ctr
        }
        logs.forEach((log, r) => listKeys.forEach(lkey => {
            const t0 = Array.isArray(textObjs[d]) ? textObjs[d][r] : textObjs[d]
            let textObj = t0 === undefined ? null : t0
            const firstRow = nRow
            /* the replica's list objects that are not this device log's: their ops are rows without effect here.  opts.otherLists[d][r]: the ones an EARLIER part of the
             * log created (a resident session encodes only the Changes that arrived since: ADVICE r5) */
            const seedLists = opts && opts.otherLists && opts.otherLists[d] && opts.otherLists[d][r]
            const otherLists = new Set(seedLists || [])
            const wantPath = String(lkey).split(".") /* "meta.notes": a list nested in map objects, by its path (micromerge.ts:178-196); one key: a list of the root map */
            const pathOf = new Map() /* map / list object -> the keys that lead to it from the root map, as the ops of this log made them */
            /* round 6 (ADVICE r5): WHICH list the path names is decided as the reference decides it — the last-writer-wins winner of every key on the way; null
             * (the path ends at no list) or a seeded textObj (Changes appended to a resident log): the first object made under the path, as before */
            const resolved = textObj === null ? resolveListPath(logs[r], wantPath) : null
            for (const ch of log) {
                /* the Change envelope (micromerge.ts:60-71): what applyChange's admission checks (:499-511) */
                chgActor[nChg] = arank.get(ch.actor)
                chgSeq[nChg] = ch.seq
                chgNops[nChg] = ch.ops.length
                if (ch.deps)
                    for (const a in ch.deps) {
                        depChg.push(nChg)
                        depActor.push(arank.get(a))
                        depValue.push(ch.deps[a])
                    }
                nChg++
                for (const op of ch.ops) {
                    const i = nRow++
                    putId(opId32, i, op.opId)
                    colAction[i] = ACT.NOP
                    const act = op.action
                    const onRoot = op.obj === undefined || op.obj === null || op.obj === ROOT || typeof op.obj === "symbol"
                    if ((act === "makeMap" || act === "makeList") && op.key !== undefined && (onRoot || pathOf.has(op.obj)) && !pathOf.has(op.opId))
                        pathOf.set(op.opId, (onRoot ? [] : pathOf.get(op.obj)).concat([op.key]))
                    if (act === "makeList" && onRoot && op.key === lkey && textObj === null && (resolved === null || op.opId === resolved)) {
                        colAction[i] = ACT.MAKELIST
                        refB32[2 * i] = intern(keys, keyIx, lkey) /* also a write of the root map's key */
                        textObj = op.opId
                    } else if (textObj !== null && op.obj === textObj) {
                        if (act === "set" && op.insert) {
                            const v = op.value
                            if (typeof v !== "string") throw new Error("Expected value inserted into text to be a string")
                            let vi = valueIx.get(v)
                            if (vi === undefined) {
                                vi = values.length
                                valueIx.set(v, vi)
                                values.push(v)
                            }
                            colAction[i] = ACT.INSERT
                            putId(refA32, i, op.elemId)
                            colPayload[i] = vi
                        } else if (act === "del" && op.elemId !== undefined) {
                            colAction[i] = ACT.DELETE
                            putId(refA32, i, op.elemId)
                        } else if (act === "addMark" || act === "removeMark") {
                            const mt = MARK_NAMES.indexOf(op.markType)
                            colAction[i] = act === "addMark" ? ACT.ADDMARK : ACT.REMOVEMARK
                            colMarkType[i] = mt
                            colSideA[i] = SIDE_NAMES.indexOf(op.start.type)
                            colSideB[i] = SIDE_NAMES.indexOf(op.end.type)
                            putId(refA32, i, op.start.elemId)
                            putId(refB32, i, op.end.elemId)
                            if (op.markType === "link" && act === "addMark") {
                                const u = op.attrs.url
                                if (!urlIx.has(u)) {
                                    urlIx.set(u, urls.length)
                                    urls.push(u)
                                }
                                colPayload[i] = urlIx.get(u)
                            } else if (op.markType === "comment") colPayload[i] = crank.get(op.attrs.id)
                        }
                    } else if (op.key !== undefined && op.elemId === undefined && (act === "set" || act === "del" || act === "makeMap" || act === "makeList")) {
                        /* an op on a MAP object (the root map or a nested one), micromerge.ts:572-602: last writer wins per (object, key) */
                        putId(refA32, i, op.obj)
                        refB32[2 * i] = intern(keys, keyIx, op.key)
                        if (act === "del") colAction[i] = ACT.MAPDEL
                        else {
                            colAction[i] = ACT.MAPSET
                            colMarkType[i] = act === "makeMap" ? MAPV.MAP : act === "makeList" ? MAPV.LIST : MAPV.SCALAR
                            if (act === "set") colPayload[i] = intern(mapValues, mapValueIx, JSON.stringify(op.value === undefined ? null : op.value))
                            if (act === "makeList") {
                                const p = pathOf.get(op.opId)
                                /* the nested list this device log merges: its makeList stays a write of its map's key */
                                if (textObj === null && wantPath.length > 1 && p && p.length === wantPath.length && p.every((k, j) => k === wantPath[j]) && (resolved === null || op.opId === resolved)) textObj = op.opId
                                else otherLists.add(op.opId)
                            }
                        }
                    } else if (otherLists.has(op.obj) && (act === "addMark" || act === "removeMark" || op.elemId !== undefined || op.insert)) {
                        /* an op on ANOTHER list object of this replica (merged by its own device log when its key is in listKeys): PTX_ACT_NOP here */
                    } else if (act === "addMark" || act === "removeMark" || op.elemId !== undefined || op.insert) {
                        /* a list op whose object no earlier makeList of this log created: the reference throws RangeError("Object does not exist")
                         * (micromerge.ts:538): rejected here.  (Ops on a list the log created under a key outside listKeys are rows without effect: that list's own checks, micromerge.ts:752, run only when the caller names its key) */
                        throw new RangeError("list op " + String(op.opId) + " on an object that no earlier makeList of this log created")
                    }
                }
            }
            logOff.push(logOff[logOff.length - 1] + (nRow - firstRow))
            chgOff.push(nChg)
            logDoc.push(d)
            logList.push(lkey)
            logReplica.push(r)
        }))
    })
    const nLogs = logOff.length - 1
    const batch = {
        nLogs,
        nOps: logOff[nLogs],
        logOff: BigUint64Array.from(logOff.map(BigInt)),
        opId: colOpId,
        refA: colRefA,
        refB: colRefB,
        payload: colPayload,
        action: colAction,
        markType: colMarkType,
        sideA: colSideA,
        sideB: colSideB,
        chgOff: BigUint64Array.from(chgOff.map(BigInt)),
        chgActor,
        chgSeq,
        chgNops,
        chgDeps: new Uint32Array(nChg * maxActors),
        maxActors,
        values, urls, logDoc, docActors, docComments, keys, mapValues,
    }
    if (listKeys.length !== 1 || listKeys[0] !== "text") {
        batch.logList = logList
        batch.logReplica = logReplica
    }
    for (let k = 0; k < depChg.length; k++) batch.chgDeps[depChg[k] * maxActors + depActor[k]] = depValue[k]
    packEnvelope(batch)
    batch.logHdr = census(batch)
    return batch
}

/**
 * InputOperation[] (micromerge.ts:133-148) of many replicas -> the columns of ptx_input_ops.  perLog[l] = the change() calls
 * of the replica behind log l (each an InputOperation[]), actors[l] = its actor id.  New strings / urls extend batch.values /
 * batch.urls; comment ids and actors must already have their rank in `batch` (encodeDocs opts).
 */
/** an InputOperation on a map object (micromerge.ts:109-131: makeMap / set / del, makeList of another key than the text's) */
function isMapInput(op) {
    if (op.action === "makeMap" || op.action === "set" || op.action === "del") return true
    return op.action === "makeList" && !((op.path || []).length === 0 && op.key === "text")
}
/**
 * metadata[CHILDREN] of every map object of the replica behind `log` (micromerge.ts:585-596): "obj:keyId" -> [child object id packed
 * counter << 12 | actor rank, kind].  A makeMap / makeList registers its child when it wins its key AT THE TIME it is applied; later
 * winners of the key that are no makeMap leave the entry alone — so this is a replay of the log's map rows in order.
 */
function mapChildrenOfLog(batch, log) {
    const last = new Map(), children = new Map()
    const pack = v => Number(((v >> 32n) << 12n) | (v & 0xfffn))
    for (let i = Number(batch.logOff[log]); i < Number(batch.logOff[log + 1]); i++) {
        const a = batch.action[i]
        if (a !== ACT.MAKELIST && a !== ACT.MAPSET && a !== ACT.MAPDEL) continue
        const key = (a === ACT.MAKELIST ? 0 : pack(batch.refA[i])) + ":" + Number(batch.refB[i] & 0xffffffffn)
        if (!last.has(key) || last.get(key) < batch.opId[i]) {
            last.set(key, batch.opId[i])
            const kind = a === ACT.MAKELIST ? MAPV.LIST : a === ACT.MAPSET ? batch.markType[i] : MAPV.DELETED
            if (kind === MAPV.MAP || kind === MAPV.LIST) children.set(key, [pack(batch.opId[i]), kind])
        }
    }
    return children
}

function encodeInputOps(batch, perLog, actors) {
    const valueIx = new Map(batch.values.map((v, i) => [v, i])), urlIx = new Map(batch.urls.map((u, i) => [u, i]))
    if (!batch.keys) batch.keys = []
    if (!batch.keys.length) batch.keys.push("text")
    if (!batch.mapValues) batch.mapValues = []
    const keyIx = new Map(batch.keys.map((k, i) => [k, i])), mapValueIx = new Map(batch.mapValues.map((v, i) => [v, i]))
    const intern = (table, ix, v) => {
        if (!ix.has(v)) {
            ix.set(v, table.length)
            table.push(v)
        }
        return ix.get(v)
    }
    const chgOff = [0], opOff = [0], action = [], markType = [], index = [], count = [], payload = [], values = [], actor = []
    perLog.forEach((calls, l) => {
        const d = batch.logDoc[l]
        const me = batch.docActors[d].indexOf(actors[l])
        if (me < 0) throw new Error("actor " + actors[l] + " has no rank in this batch (encodeDocs opts.extraActors)")
        actor.push(me)
        const crank = new Map(batch.docComments[d].map((c, i) => [c, i]))
        const children = calls.some(ops => ops.some(isMapInput)) ? mapChildrenOfLog(batch, l) : new Map()
        let made = 0 /* rows this log's calls have made so far */
        for (const ops of calls) {
            for (const op of ops) {
                let row
                if (isMapInput(op)) {
                    /* getObjectIdForPath (micromerge.ts:446-463): down the CHILDREN of the map objects, from the root */
                    let obj = 0
                    for (const elem of op.path || []) {
                        const child = children.get(obj + ":" + (keyIx.has(elem) ? keyIx.get(elem) : -1))
                        if (child === undefined) throw new Error("Child not found: " + elem + " in " + JSON.stringify(op.path))
                        if (child[1] !== MAPV.MAP) throw new RangeError("Object " + elem + " in path " + JSON.stringify(op.path) + " is a list")
                        obj = child[0]
                    }
                    const k = intern(batch.keys, keyIx, op.key)
                    if (op.action === "del") row = [IN.MAPDEL, 0, obj, k, 0]
                    else {
                        const kind = op.action === "makeMap" ? MAPV.MAP : op.action === "makeList" ? MAPV.LIST : MAPV.SCALAR
                        row = [IN.MAPSET, kind, obj, k, op.action === "set" ? intern(batch.mapValues, mapValueIx, JSON.stringify(op.value === undefined ? null : op.value)) : 0]
                        if (kind !== MAPV.SCALAR) children.set(obj + ":" + k, [(IN_OBJ_NEW | made) >>> 0, kind]) /* the newest op of the replica: it wins its key */
                    }
                    made += 1
                } else if (op.action === "makeList") {
                    if ((op.path || []).length !== 0 || op.key !== "text") throw new Error("only the text list of the root map is supported")
                    row = [IN.MAKELIST, 0, 0, 0, 0]
                    made += 1
                } else if (!Array.isArray(op.path) || op.path.length !== 1 || op.path[0] !== "text") {
                    throw new Error("Only the text list is supported: " + JSON.stringify(op.path))
                } else if (op.action === "insert") {
                    const first = values.length
                    for (const v of op.values) {
                        if (typeof v !== "string") throw new Error("Expected value inserted into text to be a string")
                        if (!valueIx.has(v)) {
                            valueIx.set(v, batch.values.length)
                            batch.values.push(v)
                        }
                        values.push(valueIx.get(v))
                    }
                    row = [IN.INSERT, 0, op.index, op.values.length, first]
                    made += op.values.length
                } else if (op.action === "delete") {
                    row = [IN.DELETE, 0, op.index, op.count, 0]
                    made += op.count
                }
                else if (op.action === "addMark" || op.action === "removeMark") {
                    const mt = MARK_NAMES.indexOf(op.markType)
                    if (mt < 0) throw new Error("unknown mark type " + op.markType)
                    let pl = 0
                    if (op.markType === "link" && op.action === "addMark") {
                        if (!urlIx.has(op.attrs.url)) {
                            urlIx.set(op.attrs.url, batch.urls.length)
                            batch.urls.push(op.attrs.url)
                        }
                        pl = urlIx.get(op.attrs.url)
                    } else if (op.markType === "comment") {
                        if (!crank.has(op.attrs.id)) throw new Error("comment id " + op.attrs.id + " has no rank in this batch (encodeDocs opts.extraComments)")
                        pl = crank.get(op.attrs.id)
                    }
                    row = [op.action === "addMark" ? IN.ADDMARK : IN.REMOVEMARK, mt, op.startIndex, op.endIndex, pl]
                    made += 1
                } else throw new Error("unsupported InputOperation action " + op.action)
                if (!(row[2] >= 0) || !(row[3] >= 0)) throw new RangeError("List index out of bounds: " + Math.min(row[2], row[3]))
                action.push(row[0]); markType.push(row[1]); index.push(row[2]); count.push(row[3]); payload.push(row[4])
            }
            opOff.push(action.length)
        }
        chgOff.push(opOff.length - 1)
    })
    return {
        chgOff: BigUint64Array.from(chgOff.map(BigInt)), opOff: BigUint64Array.from(opOff.map(BigInt)), action: Uint8Array.from(action), markType: Uint8Array.from(markType),
        index: Uint32Array.from(index), count: Uint32Array.from(count), payload: Uint32Array.from(payload), values: Uint32Array.from(values), actor: Uint32Array.from(actor),
        maxActors: Math.max(batch.maxActors, ...batch.docActors.map(a => a.length)),
    }
}

/** ptx_log_hdr rows (LOG_HDR_WORDS u32 per log: n_ins, n_del, n_mark[4], max_counter, max_actor, n_comment_ids, reserved) — what the
 *  encoder knows for free.  n_comment_ids = largest doc-local comment id the log uses + 1 (ids are ranks over the whole document). */
const LOG_HDR_WORDS = 10
function census(batch) {
    const hdr = new Uint32Array(batch.nLogs * LOG_HDR_WORDS)
    const action = batch.action, markType = batch.markType, payload = batch.payload
    const id32 = new Uint32Array(batch.opId.buffer, batch.opId.byteOffset, 2 * batch.opId.length) /* (low half: actor rank, high half: counter — no BigInt per row) */
    for (let l = 0; l < batch.nLogs; l++) {
        const b0 = Number(batch.logOff[l]), b1 = Number(batch.logOff[l + 1])
        const h = hdr.subarray(l * LOG_HDR_WORDS, (l + 1) * LOG_HDR_WORDS)
        let maxCtr = 0, maxAct = 0
        for (let i = b0; i < b1; i++) {
            const a = action[i]
            if (a === ACT.INSERT) h[0]++
            else if (a === ACT.DELETE) h[1]++
            else if ((a === ACT.ADDMARK || a === ACT.REMOVEMARK) && markType[i] < 4) {
                h[2 + markType[i]]++
                if (markType[i] === 2 && payload[i] + 1 > h[8]) h[8] = payload[i] + 1
            }
            const act = id32[2 * i], ctr = id32[2 * i + 1]
            if (ctr > maxCtr) maxCtr = ctr
            if (act > maxAct) maxAct = act
        }
        h[6] = maxCtr
        h[7] = maxAct
    }
    return hdr
}

/** FormatSpanWithText[] of one log (what getTextWithFormatting(["text"]) returns, peritext.ts:337-395). */
function decodeSpans(batch, res, log) {
    const r = res.logs.subarray(12 * log, 12 * log + 12)
    if (r[0] !== 0) throw new RangeError(STATUS_MESSAGES[r[0]] || "merge error " + r[0])
    /* the rows are compact (ABI 7): log l's at valueOff / spanOff / cintOff[l]; a result object without them (the mock addon of the tests) has them at the log's row offset */
    const b = Number(batch.logOff[log])
    const bv = res.valueOff ? Number(res.valueOff[log]) : b, bs = res.spanOff ? Number(res.spanOff[log]) : b, bc = res.cintOff ? Number(res.cintOff[log]) : b
    const nVisible = r[3], nSpans = r[4], nCints = r[5]
    const comments = batch.docComments[batch.logDoc[log]]
    const out = []
    for (let k = 0; k < nSpans; k++) {
        const start = res.spans[2 * (bs + k)], attr = res.spans[2 * (bs + k) + 1]
        const end = k + 1 < nSpans ? res.spans[2 * (bs + k + 1)] : nVisible
        const marks = {}
        if (attr & ATTR.STRONG) marks.strong = { active: true }
        if (attr & ATTR.EM) marks.em = { active: true }
        if ((attr & ATTR.COMMENT) !== 0) {
            const ids = []
            for (let c = 0; c < nCints; c++) {
                const id = res.cintervals[3 * (bc + c)], s = res.cintervals[3 * (bc + c) + 1], e = res.cintervals[3 * (bc + c) + 2]
                if (s <= start && start < e) ids.push(id)
            }
            marks.comment = ids.map(i => comments[i]).sort().map(id => ({ id })) /* by id string (= by rank, unless the table grew in arrival order) */
        }
        if (attr & ATTR.LINK) marks.link = { url: batch.urls[attr & ATTR.ID_MASK] }
        let text = ""
        for (let q = start; q < end; q++) text += batch.values[res.values[bv + q]]
        out.push({ text, marks })
    }
    return out
}

/**
 * Change[] of one log — the inverse of encodeDocs for the ops of the text list (micromerge.ts:60-71 Change, :150-212
 * Operation, peritext.ts:25-65 mark ops) in the JSON-portable form of the traces ("_root" / "_head").
 */
function decodeChanges(batch, log, textObjOfLog) {
    if (!batch.chgActor) unpackEnvelope(batch)
    const d = batch.logDoc[log]
    const actors = batch.docActors[d], comments = batch.docComments[d]
    const oid = v => String(v >> 32n) + "@" + actors[Number(v & 0xffffffffn)]
    const out = []
    let row = Number(batch.logOff[log])
    let textObj = textObjOfLog === undefined ? null : textObjOfLog /* opId of the text list when this batch holds only newly made changes */
    for (let c = Number(batch.chgOff[log]); c < Number(batch.chgOff[log + 1]); c++) {
        const nops = batch.chgNops[c]
        const deps = {}
        for (let a = 0; a < batch.maxActors; a++) {
            const v = batch.chgDeps[c * batch.maxActors + a]
            if (v) deps[actors[a]] = v
        }
        const ops = []
        for (let i = row; i < row + nops; i++) {
            const act = batch.action[i]
            const op = { opId: oid(batch.opId[i]) }
            if (act === ACT.MAKELIST) {
                Object.assign(op, { action: "makeList", obj: ROOT, key: "text" })
                textObj = op.opId
            } else if (act === ACT.INSERT) {
                Object.assign(op, { action: "set", obj: textObj, elemId: batch.refA[i] ? oid(batch.refA[i]) : HEAD, insert: true, value: batch.values[batch.payload[i]] })
            } else if (act === ACT.DELETE) {
                Object.assign(op, { action: "del", obj: textObj, elemId: oid(batch.refA[i]) })
            } else if (act === ACT.ADDMARK || act === ACT.REMOVEMARK) {
                const mt = MARK_NAMES[batch.markType[i]]
                const start = { type: SIDE_NAMES[batch.sideA[i]] }, end = { type: SIDE_NAMES[batch.sideB[i]] }
                if (batch.sideA[i] < 2) start.elemId = oid(batch.refA[i])
                if (batch.sideB[i] < 2) end.elemId = oid(batch.refB[i])
                Object.assign(op, { action: act === ACT.ADDMARK ? "addMark" : "removeMark", obj: textObj, start, end, markType: mt })
                if (mt === "link" && act === ACT.ADDMARK) op.attrs = { url: batch.urls[batch.payload[i]] }
                else if (mt === "comment") op.attrs = { id: comments[batch.payload[i]] }
            } else if (act === ACT.MAPSET || act === ACT.MAPDEL) { /* an op on a map object (the root map or a nested one) */
                Object.assign(op, { obj: batch.refA[i] ? oid(batch.refA[i]) : ROOT, key: batch.keys[Number(batch.refB[i])] })
                if (act === ACT.MAPDEL) op.action = "del"
                else {
                    const kind = batch.markType[i]
                    op.action = kind === MAPV.MAP ? "makeMap" : kind === MAPV.LIST ? "makeList" : "set"
                    if (kind === MAPV.SCALAR) op.value = JSON.parse(batch.mapValues[batch.payload[i]])
                }
            } else throw new Error("row " + i + " is not an op this engine models")
            ops.push(op)
        }
        out.push({ actor: actors[batch.chgActor[c]], seq: batch.chgSeq[c], deps, startOp: nops ? Number(batch.opId[row] >> 32n) : 0, ops })
        row += nops
    }
    return out
}

/**
 * prosemirrorDocFromCRDT (reference/src/bridge.ts:394-414 with getProsemirrorMarksForMarkMap :369-391) as the JSON that
 * prosemirror-model's Node.toJSON() gives for the document it builds: doc > paragraph > text nodes, one per span, marks
 * in ALL_MARKS order (= schema rank order, schema.ts:125,:146-149), attrs only where the mark spec has some (comment
 * {id}, link {url}; strong / em carry none, schema.ts:45-96), adjacent spans with equal ProseMirror marks joined.  What the
 * reference's own source decides is pinned by tests/golden/pm_docs.json (oracle/gen_pm_golden.js reads the reference's schema.ts);
 * prosemirror-model itself is not available in this image: its toJSON / joining rules are restated, PARITY UNPINNED for those.
 */
/**
 * getRoot() of the replica behind `log` (micromerge.ts:443-449) from ptx_root_map's entries: nested objects for the maps,
 * { $list: true } where a list object hangs (the text: its content is getTextWithFormatting's business), the set values elsewhere.
 */
function decodeRoot(batch, rm, log) {
    const b0 = Number(batch.logOff[log]), e0 = Number(rm.entryOff[log]), n = rm.logs[log * 4 + 1]
    if (rm.logs[log * 4] !== 0) throw new RangeError("Object does not exist (row " + rm.logs[log * 4 + 2] + " of the log)")
    const byObj = new Map()
    for (let k = e0; k < e0 + n; k++) {
        const obj = (BigInt(rm.entries[k * 6 + 1]) << 32n) | BigInt(rm.entries[k * 6])
        if (!byObj.has(obj)) byObj.set(obj, [])
        byObj.get(obj).push({ key: rm.entries[k * 6 + 2], row: rm.entries[k * 6 + 3], kind: rm.entries[k * 6 + 4], value: rm.entries[k * 6 + 5] })
    }
    const build = obj => {
        const out = {}
        for (const e of byObj.get(obj) || []) {
            if (e.kind === MAPV.DELETED) continue
            const k = (batch.keys && batch.keys.length ? batch.keys : ["text"])[e.key] /* batches made on the device hold the text list's makeList only: key id 0 */
            if (e.kind === MAPV.MAP) out[k] = build(batch.opId[b0 + e.row])
            else if (e.kind === MAPV.LIST) out[k] = { $list: true }
            else out[k] = JSON.parse(batch.mapValues[e.value])
        }
        return out
    }
    return build(0n)
}

function prosemirrorDocFromSpans(spans) {
    if (spans.length === 1 && spans[0].text === "") return { type: "doc", content: [{ type: "paragraph" }] } /* bridge.ts:399-401 */
    const text = []
    for (const s of spans) {
        if (s.text === "") throw new RangeError("Empty text nodes are not allowed") /* what prosemirror-model's schema.text("") throws */
        const marks = []
        for (const t of MARK_NAMES) {
            const v = s.marks[t]
            if (v === undefined) continue
            if (Array.isArray(v)) for (const one of v) marks.push({ type: t, attrs: { id: one.id } })
            else if (t === "link") marks.push({ type: t, attrs: { url: v.url } })
            else marks.push({ type: t })
        }
        const last = text[text.length - 1]
        if (last && JSON.stringify(last.marks || []) === JSON.stringify(marks)) {
            last.text += s.text /* Fragment.fromArray joins adjacent text nodes with the same marks */
            continue
        }
        const node = { type: "text" }
        if (marks.length) node.marks = marks
        node.text = s.text
        text.push(node)
    }
    const paragraph = { type: "paragraph" }
    if (text.length) paragraph.content = text
    return { type: "doc", content: [paragraph] }
}

const PATCH = { MAKELIST: 0, INSERT: 1, DELETE: 2, ADDMARK: 3, REMOVEMARK: 4, INSERT_COMMENT: 5 }

/**
 * Patch[][] of one log: entry c = what applyChange(change c) returned (micromerge.ts:499 -> Patch[]), in the
 * reference's shapes (insert :661-671, delete :696-703, add/removeMark peritext.ts:251-281; the makeList patch, the
 * raw op in the reference, is reduced to {action: "makeList"}).  `res` must come from applyMaterialize(batch, true).
 */
function decodePatches(batch, res, log) {
    const st = res.patchLogs[2 * log], n = res.patchLogs[2 * log + 1]
    if (st !== 0) throw new RangeError(STATUS_MESSAGES[st] || "merge error " + st)
    const b = Number(batch.logOff[log]), p0 = Number(res.patchOff[log])
    const comments = batch.docComments[batch.logDoc[log]]
    /* op row -> index of its change within the log */
    const c0 = Number(batch.chgOff[log]), c1 = Number(batch.chgOff[log + 1])
    const out = []
    const firstRow = []
    let row = 0
    for (let c = c0; c < c1; c++) {
        firstRow.push(row)
        row += batch.chgNops[c]
        out.push([])
    }
    let cur = 0
    for (let k = 0; k < n; k++) {
        const at = 4 * (p0 + k)
        const r = res.patches[at], kind = res.patches[at + 1], a = res.patches[at + 2], v = res.patches[at + 3]
        while (cur + 1 < firstRow.length && firstRow[cur + 1] <= r) cur++
        let patch
        if (kind === PATCH.MAKELIST) patch = { action: "makeList" }
        else if (kind === PATCH.INSERT) {
            const marks = {}
            if (v & ATTR.STRONG) marks.strong = { active: true }
            if (v & ATTR.EM) marks.em = { active: true }
            if ((v & ATTR.COMMENT) !== 0) {
                const ids = []
                while (k + 1 < n && res.patches[4 * (p0 + k + 1) + 1] === PATCH.INSERT_COMMENT) ids.push(res.patches[4 * (p0 + ++k) + 2])
                marks.comment = ids.map(i => comments[i]).sort().map(id => ({ id })) /* by id string (= by rank, unless the table grew in arrival order) */
            }
            if (v & ATTR.LINK) marks.link = { url: batch.urls[v & ATTR.ID_MASK] }
            patch = { path: ["text"], action: "insert", index: a, values: [batch.values[batch.payload[b + r]]], marks }
        } else if (kind === PATCH.DELETE) patch = { path: ["text"], action: "delete", index: a, count: v }
        else if (kind === PATCH.ADDMARK || kind === PATCH.REMOVEMARK) {
            const mt = batch.markType[b + r]
            patch = { action: kind === PATCH.ADDMARK ? "addMark" : "removeMark", markType: MARK_NAMES[mt], path: ["text"], startIndex: a, endIndex: v }
            if (kind === PATCH.ADDMARK && MARK_NAMES[mt] === "link") patch.attrs = { url: batch.urls[batch.payload[b + r] & ATTR.ID_MASK] }
            else if (kind === PATCH.ADDMARK && MARK_NAMES[mt] === "comment") patch.attrs = { id: comments[batch.payload[b + r]] }
        } else throw new Error("unknown patch kind " + kind)
        out[cur].push(patch)
    }
    return out
}

class MergeEngine {
    /** opts: {device?: number, libPath?: string, addonPath?: string, resident?: boolean} */
    constructor(opts) {
        const o = opts || {}
        this.addon = o.addon || require(o.addonPath || path.join(__dirname, "peritext_node.node")) /* (o.addon: the tests' stand-in that records what would be uploaded) */
        this.addon.open(o.libPath || path.join(__dirname, "..", "lib", "libperitext_hip.so"))
        this.ctx = this.addon.create(o.device || 0, 0) /* throws without a gfx950 device: there is no CPU fallback */
        this.pending = []
        /* resident replicas (default): the logs of a document's replica() handles stay in HBM between calls; a flush encodes and uploads only the Changes
         * that arrived since (ptx_batch_append), merges the resident logs and fetches the Patch[] records of the new rows only (ptx_replay_patches_from).
         * {resident: false}: every flush encodes, uploads and replays every handle's whole log in one launch (the round-2 behaviour). */
        this.resident = o.resident !== false
        this.sessions = new Map() /* docId -> resident state of the document's handles */
        this.stats = { residentUploads: 0, residentAppends: 0, rowsUploaded: 0, residentChanges: 0, residentChangeMs: 0, residentCursorCalls: 0 }
    }
    close() {
        if (this.ctx) {
            for (const st of this.sessions.values()) if (st.handle) this.addon.residentFree(this.ctx, st.handle)
            this.sessions.clear()
            this.addon.destroy(this.ctx)
        }
        this.ctx = null
    }
    /** Raw call: SoA batch -> result typed arrays (ptx_apply_materialize). */
    applyMaterialize(batch, wantPatches) {
        return this.addon.applyMaterialize(this.ctx, batch, !!wantPatches)
    }
    /** docs: Change[][][] -> {spans: FormatSpanWithText[][][], patches: Patch[][][][]} — patches[d][r][c] is what replica r of
     *  document d would have returned from applyChange(change c) (ptx_replay_patches). */
    applyChangesWithPatches(docs) {
        const batch = encodeDocs(docs)
        const res = this.applyMaterialize(batch, true)
        let log = 0, log2 = 0
        return {
            spans: docs.map(logs => logs.map(() => decodeSpans(batch, res, log++))),
            patches: docs.map(logs => logs.map(() => decodePatches(batch, res, log2++))),
        }
    }
    /** docs: Change[][][] -> getRoot() of every replica (ptx_root_map: the root map and the maps nested in it, last writer wins per
     *  key, micromerge.ts:572-602; list objects appear as { $list: true }).  A replica the reference would have thrown on
     *  ("Object does not exist") throws a RangeError here too. */
    roots(docs) {
        const batch = encodeDocs(docs)
        const rm = this.addon.rootMap(this.ctx, batch)
        let log = 0
        return docs.map(logs => logs.map(() => decodeRoot(batch, rm, log++)))
    }
    /**
     * On-device change(): whole edit histories made on the GPU (ptx_generate — Micromerge.change, micromerge.ts:308-441, under the
     * workload of the reference's fuzzer test/fuzz.ts:115-205 in its seeded form oracle/ptxgen.js) and merged there.
     * cfg: {replicas, opsPerLog, mix: [insert, delete, addMark, removeMark] percent, markTypes: MarkType[], seed, nDocs, firstDoc?, listCap?, initialText?}
     * Returns {docs: Change[][][] (doc -> replica -> changes in application order), spans: FormatSpanWithText[][][], kernelMs}.
     */
    generate(cfg) {
        const raw = this.addon.generate(this.ctx, Object.assign({}, cfg, { markTypes: (cfg.markTypes || []).map(m => MARK_NAMES.indexOf(m)) }))
        const batch = raw.batch
        const R = cfg.replicas, nDocs = cfg.nDocs
        /* the fixed string tables of a generated batch (include/peritext_hip.h ptx_generate) */
        batch.values = Array.from({ length: 128 }, (_, i) => String.fromCharCode(i))
        batch.urls = Array.from({ length: 26 }, (_, i) => String.fromCharCode(65 + i) + ".com")
        batch.logDoc = []
        batch.docActors = []
        batch.docComments = []
        for (let d = 0; d < nDocs; d++) {
            for (let r = 0; r < R; r++) batch.logDoc.push(d)
            batch.docActors.push(Array.from({ length: R }, (_, i) => "doc" + (i + 1)))
            batch.docComments.push(Array.from({ length: raw.nComments[d] }, (_, k) => "comment-" + k).sort()) /* UTF-16 order = the wire ranks */
        }
        const docs = [], spans = []
        for (let d = 0; d < nDocs; d++) {
            docs.push(Array.from({ length: R }, (_, r) => decodeChanges(batch, d * R + r)))
            spans.push(Array.from({ length: R }, (_, r) => decodeSpans(batch, raw.result, d * R + r)))
        }
        return { docs, spans, kernelMs: raw.kernelMs, batch }
    }
    /** docs: Change[][][]  ->  FormatSpanWithText[][][] (doc -> replica -> spans).  A failed log throws RangeError like the reference.
     *  opts.listKeys (round 5): the list objects of the root map to materialise, by key — then every replica's entry is an object {key: spans}
     *  (getTextWithFormatting([key]) of that replica, micromerge.ts:516) instead of the spans of "text" alone. */
    applyChanges(docs, opts) {
        const listKeys = opts && opts.listKeys
        const batch = encodeDocs(docs, listKeys ? { listKeys } : undefined)
        const res = this.applyMaterialize(batch)
        let log = 0
        if (!listKeys) return docs.map(logs => logs.map(() => decodeSpans(batch, res, log++)))
        return docs.map(logs => logs.map(() => {
            const out = {}
            for (const k of listKeys) out[k] = decodeSpans(batch, res, log++)
            return out
        }))
    }
    /** Per-document digests (2 x u64 as BigInt pairs) of every log: equal digests <=> deep-equal spans. */
    digests(docs) {
        const batch = encodeDocs(docs)
        const res = this.applyMaterialize(batch)
        const out = []
        for (let l = 0; l < batch.nLogs; l++) {
            const r = res.logs.subarray(12 * l, 12 * l + 12)
            out.push([(BigInt(r[9]) << 32n) | BigInt(r[8]), (BigInt(r[11]) << 32n) | BigInt(r[10])])
        }
        return out
    }
    /* ---- multi-GPU (one Node process per GPU): the digest all-gather over RCCL lives in the library (ptx_allgather_digests) ---- */
    /** Uint8Array(128) made on rank 0; the host's own channel carries it to the other ranks */
    commUniqueId() {
        return this.addon.commUniqueId(this.ctx)
    }
    /** collective over all ranks; returns an opaque communicator */
    commInit(id, rank, nRanks) {
        return this.addon.commInit(this.ctx, id, rank, nRanks)
    }
    commDestroy(comm) {
        this.addon.commDestroy(this.ctx, comm)
    }
    /**
     * This rank's documents (Change[][][], `replicas` logs each) are merged, the digests of ALL ranks gathered on the device and the
     * converged documents of the whole job counted there — the reference's `assert.deepStrictEqual(leftText, rightText)`
     * (test/fuzz.ts:277-278) for a sharded batch.  counts[r] = replica logs of rank r.
     * Returns {converged, total, digests: Array<[bigint, bigint]> of every rank (rank-major), statuses: number[] of this rank's logs}.
     */
    convergedDocs(docs, comm, counts, replicas) {
        const batch = encodeDocs(docs)
        const raw = this.addon.mergeAndGather(this.ctx, comm, batch, Uint32Array.from(counts), replicas)
        const digests = []
        for (let l = 0; 2 * l < raw.gathered.length; l++) digests.push([raw.gathered[2 * l], raw.gathered[2 * l + 1]])
        const statuses = []
        for (let l = 0; l < batch.nLogs; l++) statuses.push(raw.logs[12 * l])
        return { converged: raw.converged, total: digests.length / replicas, digests, statuses }
    }
    /**
     * Micromerge.change for many replicas in ONE call (ptx_change): docs = the replica logs applied so far (Change[][][]),
     * calls[d][r] = the change() calls of replica r of document d (each an InputOperation[]; [] = none), actors[d][r] = its actor id.
     * Returns {changes: Change[][][] (doc -> replica -> the Changes made, in call order), status: number[][]} — a replica whose
     * status is not 0 made no change (STATUS_MESSAGES: 7 = the reference's RangeError "List index out of bounds").
     */
    changeMany(docs, calls, actors, opts) {
        const batch = encodeDocs(docs, Object.assign({ extraActors: actors }, opts || {}))
        const flatCalls = [], flatActors = []
        docs.forEach((logs, d) => logs.forEach((_, r) => {
            flatCalls.push(calls[d][r] || [])
            flatActors.push(actors[d][r])
        }))
        const inputOps = encodeInputOps(batch, flatCalls, flatActors)
        const raw = this.addon.change(this.ctx, batch, inputOps)
        const made = Object.assign(raw.batch, { values: batch.values, urls: batch.urls, logDoc: batch.logDoc, docActors: batch.docActors, docComments: batch.docComments, keys: batch.keys, mapValues: batch.mapValues })
        let log = 0
        const changes = [], status = []
        docs.forEach((logs, d) => {
            changes.push([])
            status.push([])
            logs.forEach(changesOfLog => {
                let textObj = null
                for (const ch of changesOfLog) for (const op of ch.ops) if (op.action === "makeList" && textObj === null) textObj = op.opId
                changes[d].push(decodeChanges(made, log, textObj))
                status[d].push(raw.status[log])
                log++
            })
        })
        return { changes, status }
    }
    /**
     * A replica handle with the reference's per-replica calls; all handles of one engine are merged in ONE launch.
     * docId groups the replicas of a document (shared actor / comment ranks); actorId is needed for change().
     */
    replica(docId, actorId) {
        const self = this
        const rep = { changes: [], clock: {}, spans: null, patches: null, error: null, docId: docId === undefined ? this.pending.length : docId, actorId }
        this.pending.push(rep)
        const admit = change => {
            /* applyChange's causal admission (micromerge.ts:499-511), here and now: throws like the reference and leaves the
             * replica untouched, so a caller can retry out-of-order changes (reference/test/merge.ts:13-19) */
            const last = rep.clock[change.actor] || 0
            if (change.seq !== last + 1) throw new RangeError("Expected sequence number " + (last + 1) + ", got " + change.seq)
            for (const a of Object.keys(change.deps || {}))
                if (!rep.clock[a] || rep.clock[a] < change.deps[a]) throw new RangeError("Missing dependency: change " + change.deps[a] + " by actor " + a)
            /* a list op must name the document's text list (the encoder's rule, encodeDocs): checked HERE, before the replica changes, so that a bad
             * Change is refused once — like the reference's RangeError("Object does not exist") out of applyChange (micromerge.ts:538) — and the replica
             * stays readable (ADVICE r3: thrown later, by every re-encode of the queued Change, it made the replica unreadable for good) */
            let textObj = rep.textObj === undefined ? null : rep.textObj
            const lists = new Set(rep.otherLists || []) /* the replica's other list objects (round 5: their ops are accepted — rows without effect on the text list; the batch API, applyChanges(docs, {listKeys}), materialises them) */
            for (const op of change.ops || []) {
                const onRoot = op.obj === undefined || op.obj === null || op.obj === ROOT || typeof op.obj === "symbol"
                if (op.action === "makeList" && onRoot && op.key === "text" && textObj === null) textObj = op.opId
                else if (textObj !== null && op.obj === textObj) {
                    if (op.action === "set" && op.insert && typeof op.value !== "string") throw new Error("Expected value inserted into text to be a string")
                } else if (op.key !== undefined && op.elemId === undefined && ["set", "del", "makeMap", "makeList"].indexOf(op.action) >= 0) {
                    if (op.action === "makeList") lists.add(op.opId)
                } else if (lists.has(op.obj) && (op.action === "addMark" || op.action === "removeMark" || op.elemId !== undefined || op.insert)) continue
                else if (op.action === "addMark" || op.action === "removeMark" || op.elemId !== undefined || op.insert)
                    throw new RangeError("Object does not exist: list op " + String(op.opId) + " on an object that no earlier makeList of this replica created")
            }
            rep.textObj = textObj
            rep.otherLists = lists
            rep.clock[change.actor] = change.seq
            rep.changes.push(change)
            rep.spans = null
            rep.patches = null
        }
        return {
            applyChange(change) {
                admit(change)
                return [] /* the call is only queued: the patches it would have returned come from getPatches() */
            },
            /** Micromerge.change(ops) (micromerge.ts:308): the InputOperations are resolved against this replica's state on the device;
             *  returns {change, patches} like the reference (patches = what applying the change returned). */
            change(ops) {
                if (rep.actorId === undefined) throw new Error("engine.replica(docId, actorId): an actor id is needed to make changes")
                const onDevice = self.residentChange(rep, ops, admit) /* the resident write path: nothing of the document is encoded or uploaded */
                if (onDevice) {
                    const patches = this.getPatches()
                    return { change: onDevice, patches: patches[patches.length - 1] }
                }
                const mates = self.pending.filter(r => r.docId === rep.docId)
                const docs = [mates.map(r => r.changes)]
                const comments = []
                for (const op of ops) if (op.markType === "comment" && op.attrs && op.attrs.id !== undefined) comments.push(op.attrs.id)
                const r = self.changeMany(docs, [mates.map(m => (m === rep ? [ops] : []))], [mates.map(m => (m.actorId === undefined ? rep.actorId : m.actorId))], { extraComments: [comments] })
                const me = mates.indexOf(rep)
                const st = r.status[0][me]
                if (st !== 0) throw new RangeError(STATUS_MESSAGES[st] || "change error " + st)
                const change = r.changes[0][me][0]
                admit(change)
                const patches = this.getPatches()
                return { change, patches: patches[patches.length - 1] }
            },
            /** Micromerge.getCursor (micromerge.ts:465-473): {objectId, elemId} of the index-th visible character, resolved on the device */
            getCursor(p, index) {
                if (!Array.isArray(p) || p.length !== 1 || p[0] !== "text") throw new Error("Only the text list is supported: " + JSON.stringify(p))
                if (!(index >= 0)) throw new RangeError("List index out of bounds: " + index)
                const r = self.cursorQueries(rep, [[1, BigInt(index)]])
                if (r.status[0] !== 0) throw new RangeError((STATUS_MESSAGES[r.status[0]] || "error " + r.status[0]) + ": " + index)
                return { objectId: r.textObj, elemId: r.oid(r.out[0]) }
            },
            /** Micromerge.resolveCursor (:475-477): visible characters before the cursor's element (which may be deleted by now) */
            resolveCursor(cursor) {
                const r = self.cursorQueries(rep, [[0, cursor.elemId]])
                if (r.status[0] !== 0) throw new RangeError((STATUS_MESSAGES[r.status[0]] || "error " + r.status[0]) + ": " + cursor.elemId)
                return Number(r.out[0])
            },
            /** Patch[][]: entry c = what applyChange(c-th change) returns in the reference (one launch for all handles). */
            getPatches() {
                if (rep.patches === null && rep.error === null) self.flush(true)
                if (rep.error) {
                    const e = rep.error
                    rep.error = null /* reported once: the rejected change is no longer part of the log */
                    throw e
                }
                return rep.patches
            },
            /** Micromerge.getRoot() (micromerge.ts:443-449): the root map with the maps nested in it, resolved on the device (last
             *  writer wins per key); list objects appear as { $list: true } — their content is getTextWithFormatting's. */
            getRoot() {
                return self.roots([[rep.changes]])[0][0]
            },
            getTextWithFormatting(p) {
                if (!Array.isArray(p) || p.length !== 1 || p[0] !== "text") throw new Error("Only the text list is supported: " + JSON.stringify(p))
                if (rep.spans === null && rep.error === null) self.flush()
                if (rep.error) {
                    const e = rep.error
                    rep.error = null
                    throw e
                }
                return rep.spans
            },
        }
    }
    /** queries: [kind (0 resolve, 1 get), elemId string | BigInt index] of one replica -> device answers (ptx_resolve_cursors) */
    /** Micromerge.change on the RESIDENT logs (bridge.ts:535 makes one per keystroke): the replica's new Changes are in HBM already (or go up now, alone), the
     *  InputOperations are resolved there (ptx_change on the resident batch), the Change made is appended there (ptx_batch_append_device) and comes back as a
     *  few rows for the host's own bookkeeping.  null: not applicable (no resident sessions, ops on map objects, an actor or a comment id the session has no
     *  rank for yet) — the caller takes the path that encodes the document. */
    residentChange(rep, ops, admit) {
        if (!this.resident || ops.some(isMapInput)) return null
        const mates = this.pending.filter(r => r.docId === rep.docId)
        if (mates.some(r => r.error !== null)) return null
        const { batch, st } = this.residentApply(rep.docId, mates, false, true)
        if (st.actorList.indexOf(rep.actorId) < 0 || mates.some(m => m.actorId !== undefined && st.actorList.indexOf(m.actorId) < 0)) return null
        for (const op of ops) if (op.markType === "comment" && op.attrs && op.attrs.id !== undefined && st.commentList.indexOf(op.attrs.id) < 0) st.commentList.push(op.attrs.id) /* takes the next rank */
        const me = mates.indexOf(rep)
        const view = { values: st.tables.values, urls: st.tables.urls, keys: st.tables.keys, mapValues: st.tables.mapValues, logDoc: mates.map(() => 0), docActors: [st.actorList],
                       docComments: [st.commentList], maxActors: st.maxActors || st.actorList.length }
        const io = encodeInputOps(view, mates.map(m => (m === rep ? [ops] : [])), mates.map(m => (m.actorId === undefined ? rep.actorId : m.actorId)))
        io.maxActors = view.maxActors
        const t0 = process.hrtime.bigint()
        const raw = this.addon.change(this.ctx, st.handle, io)
        this.stats.residentChangeMs += Number(process.hrtime.bigint() - t0) / 1e6 /* merge of the resident logs + ptx_change + append on the device + the made rows back */
        st.handle = raw.handle
        this.stats.residentChanges++
        if (raw.status[me] !== 0) throw new RangeError(STATUS_MESSAGES[raw.status[me]] || "change error " + raw.status[me])
        const made = Object.assign(raw.batch, view)
        const change = decodeChanges(made, me, st.textObjs[me] === undefined ? null : st.textObjs[me])[0]
        admit(change)
        /* the host's view of the resident logs follows: the made rows of this replica, its Change is "seen" (it is in HBM, no upload pending) */
        const b = Number(made.logOff[me]), e = Number(made.logOff[me + 1])
        const grow = (old, add) => {
            const out = new old.constructor(old.length + add.length)
            out.set(old)
            out.set(add, old.length)
            return out
        }
        st.payload[me] = grow(st.payload[me], made.payload.subarray(b, e))
        st.markType[me] = grow(st.markType[me], made.markType.subarray(b, e))
        st.chgNops[me].push(e - b)
        st.rows[me] += e - b
        st.seen[me] = rep.changes.slice()
        if (st.textObjs[me] === undefined) for (const op of change.ops) if (op.action === "makeList" && op.key === "text") st.textObjs[me] = op.opId
        return change
    }
    cursorQueries(rep, queries) {
        const mates = this.pending.filter(r => r.docId === rep.docId)
        const useResident = this.resident && !mates.some(r => r.error !== null)
        const synced = useResident ? this.residentApply(rep.docId, mates, false, true) : null
        const batch = synced ? { docActors: [synced.st.actorList] } : encodeDocs([mates.map(r => r.changes)])
        const log = mates.indexOf(rep)
        const actors = batch.docActors[0]
        let textObj = null
        for (const ch of rep.changes) for (const op of ch.ops) if (op.action === "makeList" && textObj === null) textObj = op.opId
        const q = { log: new Uint32Array(queries.length).fill(log), kind: new Uint8Array(queries.length), arg: new BigUint64Array(queries.length) }
        const missing = []
        queries.forEach(([kind, arg], k) => {
            q.kind[k] = kind
            if (kind === 1) q.arg[k] = arg
            else {
                const [ctr, actor] = splitOpId(arg)
                const rank = actors.indexOf(actor)
                if (rank < 0) missing.push(k) /* an actor this document has never seen: no such element */
                q.arg[k] = rank < 0 ? 0n : (BigInt(ctr) << 32n) | BigInt(rank)
            }
        })
        if (synced) this.stats.residentCursorCalls++
        const r = this.addon.cursors(this.ctx, synced ? synced.st.handle : batch, q) /* the resident logs: nothing is encoded or uploaded */
        for (const k of missing) r.status[k] = 1
        return { out: r.out, status: r.status, textObj, oid: v => String(v >> 32n) + "@" + actors[Number(v & 0xffffffffn)] }
    }
    /**
     * One document's handles against their resident logs: what is in HBM is a prefix of every handle's Changes (checked by identity) and the Changes
     * since need no new actor / comment rank -> only they are encoded (against the tables of the resident batch) and appended; otherwise the document
     * is encoded and uploaded afresh.  Returns {batch: what decodeSpans / decodePatches read, res, firstChange: per handle, the first Change whose patches
     * `res` holds}.
     */
    residentApply(docId, group, wantPatches, syncOnly) {
        let st = this.sessions.get(docId)
        const textObjOf = changes => {
            for (const ch of changes) for (const op of ch.ops) if (op.action === "makeList" && op.key === "text" && (op.obj === undefined || op.obj === null || op.obj === ROOT || typeof op.obj === "symbol")) return op.opId
            return undefined
        }
        let ok = !!st && st.reps.length === group.length && group.every((r, i) => r === st.reps[i] && r.changes.length >= st.seen[i].length && st.seen[i].every((c, k) => c === r.changes[k]))
        let delta = null
        if (ok) {
            /* (the handles' admission remembers the list objects of the Changes already uploaded — rep.otherLists —: the delta's ops on them are rows without effect,
             * not "a list no makeList of this log created"; whatever else the delta cannot be encoded for on its own sends the document through the full path) */
            try {
                delta = encodeDocs([group.map((r, i) => r.changes.slice(st.seen[i].length))],
                                   { extraActors: [st.actorList], commentOrder: [st.commentList], textObjs: [st.textObjs], seed: st.tables, otherLists: [group.map(r => Array.from(r.otherLists || []))] })
            } catch (e) {
                if (!(e instanceof RangeError)) throw e
                delta = null
            }
            /* a new actor re-ranks the ids of the old rows (actor ranks follow the string order, compareOpIds): then everything is encoded again.  New comment
             * ids just take the next ranks. */
            ok = delta !== null && delta.docActors[0].length === st.actorList.length
            if (ok) st.commentList = delta.docComments[0]
        }
        if (!ok) {
            if (st && st.handle) this.addon.residentFree(this.ctx, st.handle)
            this.sessions.delete(docId)
            const full = encodeDocs([group.map(r => r.changes)])
            st = { reps: group.slice(), seen: group.map(() => []), patched: group.map(() => 0), actorList: full.docActors[0], commentList: full.docComments[0], handle: null,
                   rows: group.map(() => 0), chgNops: group.map(() => []), payload: group.map(() => new Uint32Array(0)), markType: group.map(() => new Uint8Array(0)) }
            delta = full
            st.handle = this.addon.residentUpload(this.ctx, full)
            this.sessions.set(docId, st)
            this.stats.residentUploads++
        } else if (delta.chgActor.length > 0) {
            st.handle = this.addon.residentAppend(this.ctx, st.handle, delta) /* (the old handle is released by the addon) */
            this.stats.residentAppends++
        }
        this.stats.rowsUploaded += delta.nOps
        st.tables = { values: delta.values, urls: delta.urls, keys: delta.keys, mapValues: delta.mapValues }
        /* the host's view of the resident logs: what the decoders read (rows of a log are contiguous, as in HBM) */
        const firstRow = new Uint32Array(group.length), firstChange = []
        group.forEach((r, i) => {
            const b = Number(delta.logOff[i]), e = Number(delta.logOff[i + 1]), c0 = Number(delta.chgOff[i]), c1 = Number(delta.chgOff[i + 1])
            const grow = (old, add) => {
                const out = new old.constructor(old.length + add.length)
                out.set(old)
                out.set(add, old.length)
                return out
            }
            st.payload[i] = grow(st.payload[i], delta.payload.subarray(b, e))
            st.markType[i] = grow(st.markType[i], delta.markType.subarray(b, e))
            for (let c = c0; c < c1; c++) st.chgNops[i].push(delta.chgNops[c])
            st.seen[i] = r.changes.slice()
            /* patches are wanted from the first Change that has none yet */
            let row = 0
            for (let c = 0; c < st.patched[i]; c++) row += st.chgNops[i][c]
            firstRow[i] = row
            firstChange.push(st.patched[i])
            st.rows[i] += e - b
        })
        st.textObjs = group.map(r => textObjOf(r.changes))
        const logOff = new BigUint64Array(group.length + 1), chgOff = new BigUint64Array(group.length + 1)
        group.forEach((r, i) => {
            logOff[i + 1] = logOff[i] + BigInt(st.rows[i])
            chgOff[i + 1] = chgOff[i] + BigInt(st.chgNops[i].length)
        })
        const cat = (parts, T) => {
            const out = new T(parts.reduce((n, p) => n + p.length, 0))
            let at = 0
            for (const p of parts) {
                out.set(p, at)
                at += p.length
            }
            return out
        }
        const batch = { nLogs: group.length, logOff, chgOff, chgNops: cat(st.chgNops, Uint32Array), payload: cat(st.payload, Uint32Array), markType: cat(st.markType, Uint8Array),
                        values: st.tables.values, urls: st.tables.urls, logDoc: group.map(() => 0), docActors: [st.actorList], docComments: [st.commentList] }
        st.maxActors = delta.maxActors !== undefined && delta.chgActor.length ? delta.maxActors : st.maxActors /* the row stride of the resident envelope */
        /* syncOnly: the resident logs are current (only the new Changes went up) — the caller works on them in HBM (change(), cursors) */
        const res = syncOnly ? null : this.addon.residentApply(this.ctx, st.handle, !!wantPatches, firstRow)
        return { batch, res, firstChange, st }
    }
    flush(wantPatches) {
        const byDoc = new Map()
        for (const r of this.pending) {
            if (!byDoc.has(r.docId)) byDoc.set(r.docId, [])
            byDoc.get(r.docId).push(r)
        }
        if (this.resident) {
            for (const [docId, g] of byDoc) {
                if (!g.some(r => r.error === null && (r.spans === null || (wantPatches && r.patches === null)))) continue /* nothing new for this document */
                const { batch, res, firstChange, st } = this.residentApply(docId, g, wantPatches)
                g.forEach((r, log) => {
                    try {
                        r.spans = decodeSpans(batch, res, log)
                        if (wantPatches) {
                            const tail = decodePatches(batch, res, log) /* entries before firstChange are empty: those Changes were patched by an earlier flush */
                            r.patches = (r.patchCache || []).slice(0, firstChange[log]).concat(tail.slice(firstChange[log]))
                            r.patchCache = r.patches
                            st.patched[log] = r.patches.length
                        }
                        r.error = null
                    } catch (e) {
                        this.dropFailedChange(r, res, log, e)
                        st.patched[log] = 0 /* the log changes under the resident copy: the next flush uploads the document again */
                        r.patchCache = null
                    }
                })
            }
            return
        }
        const groups = Array.from(byDoc.values())
        const batch = encodeDocs(groups.map(g => g.map(r => r.changes)))
        const res = this.applyMaterialize(batch, wantPatches)
        let log = 0
        for (const g of groups)
            for (const r of g) {
                try {
                    r.spans = decodeSpans(batch, res, log)
                    if (wantPatches) r.patches = decodePatches(batch, res, log)
                    r.error = null
                } catch (e) {
                    this.dropFailedChange(r, res, log, e)
                }
                log++
            }
    }
    /* an op of some change failed on the device (e.g. "List element not found", micromerge.ts:752): the reference would have thrown out of that
     * applyChange call.  The change is dropped from the log (ptx_log_result names the row) so that later changes are not held hostage; the error
     * surfaces once, at the next read. */
    dropFailedChange(r, res, log, e) {
        r.error = e
        r.spans = null
        r.patches = null
        const failRow = res.logs[12 * log + 7]
        if (failRow !== 0xffffffff) {
            let row = 0
            for (let c = 0; c < r.changes.length; c++) {
                const n = r.changes[c].ops.length
                if (failRow < row + n || (n === 0 && failRow === row)) {
                    const gone = r.changes.splice(c, 1)[0]
                    if (r.clock[gone.actor] === gone.seq) r.clock[gone.actor] = gone.seq - 1
                    break
                }
                row += n
            }
        }
    }
}

module.exports = { MergeEngine, encodeDocs, encodeInputOps, packEnvelope, unpackEnvelope, decodeSpans, decodePatches, decodeChanges, decodeRoot, MAPV, prosemirrorDocFromSpans, PATCH, census, ACT, IN, MARK_NAMES, SIDE_NAMES, ATTR, STATUS_MESSAGES, ROOT, HEAD }
