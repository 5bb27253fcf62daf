"use strict"
/*
 * ORACLE — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A CPU restatement (Node >= 12, zero npm dependencies) of the Peritext hot path:
 * applying an op log (insert / delete / addMark / removeMark) to a replica and
 * materialising the formatted spans.  It follows the *sequential* algorithm of the
 * reference so that it can serve as the definition the HIP path is checked against:
 *
 *   reference/src/micromerge.ts  :262-756  class Micromerge  (change / applyChange / applyOp /
 *                                          applyListInsert / applyListUpdate / findListElement)
 *   reference/src/micromerge.ts  :762-805  getListElementId (incl. lookAfterTombstones)
 *   reference/src/micromerge.ts  :812-827  compareOpIds
 *   reference/src/peritext.ts    :154-281  applyAddRemoveMark / calculateOpsForPosition / patches
 *   reference/src/peritext.ts    :294-326  opsToMarks
 *   reference/src/peritext.ts    :337-455  getTextWithFormatting / addCharactersToSpans
 *   reference/src/peritext.ts    :458-501  changeMark
 *   reference/src/schema.ts      :45-96    markSpec (inclusive / allowMultiple table)
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may execute this file.
 * Parity pinning: tests/test_oracle_kat.py runs every known-answer case of the reference's
 * test/micromerge.ts (re-expressed as data in tests/golden/kat_*.json) against this file, and
 * oracle/run_reference_tests.js replays the reference's own test file against it when
 * /root/reference is mounted.
 *
 * Differences from the reference that are deliberate and documented:
 *   - ROOT / HEAD are the strings "_root" / "_head" instead of Symbols, so that changes survive
 *     JSON.stringify (the reference's Symbols vanish in JSON: SURVEY A.6-9).  `normalizeChange`
 *     restores them when loading a reference trace.
 *   - actorId is mandatory (the reference defaults to uuid.v4(), micromerge.ts:283).
 *   - `opts.patches === false` skips patch bookkeeping (a pure speed switch used when generating
 *     large traces; document state is unaffected).
 */

const ROOT = "_root"
const HEAD = "_head"
const CONTENT_KEY = "text"

/* schema.ts:45-96 reduced to the two properties the CRDT reads. */
const MARK_SPEC = {
    strong: { inclusive: true, allowMultiple: false },
    em: { inclusive: true, allowMultiple: false },
    comment: { inclusive: false, allowMultiple: true },
    link: { inclusive: false, allowMultiple: false },
}
const ALL_MARKS = ["strong", "em", "comment", "link"] /* schema.ts:125 */

/* ---------- op ids (micromerge.ts:812-827) ---------- */

const ID_RE = /^([0-9]+)@(.*)$/
function splitOpId(id) {
    const m = ID_RE.exec(id)
    if (!m) throw new Error("Invalid operation ID: " + id)
    return [parseInt(m[1], 10), m[2]]
}

/** -1 / 0 / +1; counter first, then actor id by JS string order (UTF-16 code units). */
function compareOpIds(a, b) {
    if (a == b) return 0
    const pa = splitOpId(a)
    const pb = splitOpId(b)
    return pa[0] < pb[0] || (pa[0] === pb[0] && pa[1] < pb[1]) ? -1 : 1
}

/* ---------- tiny structural helpers (stand-ins for lodash isEqual / sortBy, peritext.ts:2) ---------- */

function deepEqual(x, y) {
    if (x === y) return true
    if (typeof x !== "object" || typeof y !== "object" || x === null || y === null) return false
    const ax = Array.isArray(x)
    if (ax !== Array.isArray(y)) return false
    if (ax) {
        if (x.length !== y.length) return false
        for (let i = 0; i < x.length; i++) if (!deepEqual(x[i], y[i])) return false
        return true
    }
    const kx = Object.keys(x)
    if (kx.length !== Object.keys(y).length) return false
    for (const k of kx) {
        if (!Object.prototype.hasOwnProperty.call(y, k) || !deepEqual(x[k], y[k])) return false
    }
    return true
}

function sortedById(list) {
    /* stable ascending by .id with JS string comparison (lodash sortBy semantics for strings) */
    return list
        .map((v, i) => [v, i])
        .sort((p, q) => (p[0].id < q[0].id ? -1 : p[0].id > q[0].id ? 1 : p[1] - q[1]))
        .map(p => p[0])
}

/* ---------- mark resolution (peritext.ts:294-326) ---------- */

/** `ops`: array of mark operations in the order they were added to the slot (= Set order). */
function opsToMarks(ops) {
    const marks = {}
    const winner = {}
    for (const op of ops) {
        const t = op.markType
        if (!MARK_SPEC[t].allowMultiple) {
            if (winner[t] === undefined || compareOpIds(op.opId, winner[t]) === 1) {
                winner[t] = op.opId
                if (op.action === "addMark") marks[t] = op.attrs || { active: true }
                else delete marks[t]
            }
        } else if (op.action === "addMark") {
            const cur = marks[t]
            if (!(cur && cur.find(c => c.id === op.attrs.id))) {
                marks[t] = sortedById((cur || []).concat([op.attrs]))
            }
        } else if (op.action === "removeMark") {
            marks[t] = (marks[t] || []).filter(c => c.id !== op.attrs.id)
        }
    }
    return marks
}

/* ---------- span building (peritext.ts:438-455) ---------- */

function pushRun(spans, chars, marks) {
    if (chars.length === 0) return
    const last = spans.length > 0 ? spans[spans.length - 1] : undefined
    if (last && deepEqual(last.marks, marks)) last.text = last.text.concat(chars.join(""))
    else spans.push({ text: chars.join(""), marks })
}

/* ---------- list helpers ---------- */

/** micromerge.ts:762-805.  `elems` = list metadata (document order incl. tombstones). */
function getListElementId(elems, index, options) {
    if (!Array.isArray(elems)) throw new Error("Expected array metadata for findListElement")
    let seen = -1
    for (let i = 0; i < elems.length; i++) {
        if (elems[i].deleted) continue
        seen++
        if (seen !== index) continue
        if (options && options.lookAfterTombstones) {
            /* anchor after the last directly-following tombstone that owns an `after` slot */
            let pick = i
            let lastMarked
            for (let j = i + 1; j < elems.length && elems[j].deleted; j++) {
                if (elems[j].markOpsAfter !== undefined) lastMarked = j
            }
            if (lastMarked) pick = lastMarked
            return elems[pick].elemId
        }
        return elems[i].elemId
    }
    throw new RangeError("List index out of bounds: " + index)
}

/* ---------- the replica ---------- */

class Micromerge {
    constructor(actorId, opts) {
        if (typeof actorId !== "string") throw new Error("oracle: actorId is mandatory")
        this.actorId = actorId
        this.seq = 0
        this.maxOp = 0
        this.clock = {}
        this.wantPatches = !(opts && opts.patches === false)
        /* objects: id -> JS value; metadata: id -> list metadata (array) or map metadata */
        this.objects = {}
        this.objects[ROOT] = {}
        this.metadata = {}
        this.metadata[ROOT] = { keys: {}, children: {} }
    }

    get root() {
        return this.objects[ROOT]
    }
    getRoot() {
        return this.objects[ROOT]
    }

    /* micromerge.ts:446-463 */
    getObjectIdForPath(path) {
        let id = ROOT
        for (const step of path) {
            const meta = this.metadata[id]
            if (meta === undefined) throw new RangeError("No object at path " + JSON.stringify(path))
            if (Array.isArray(meta)) {
                throw new RangeError("Object " + step + " in path " + JSON.stringify(path) + " is a list")
            }
            const child = meta.children[step]
            if (child === undefined) throw new Error("Child not found: " + step + " in " + String(id))
            id = child
        }
        return id
    }

    /* micromerge.ts:308-441 */
    change(inputOps) {
        const deps = Object.assign({}, this.clock)
        this.seq += 1
        this.clock[this.actorId] = this.seq
        const change = { actor: this.actorId, seq: this.seq, deps, startOp: this.maxOp + 1, ops: [] }
        const patches = []
        const emit = partial => {
            const r = this._newOp(change, partial)
            for (const p of r.patches) patches.push(p)
            return r.opId
        }
        for (const input of inputOps) {
            const objId = this.getObjectIdForPath(input.path)
            const obj = this.objects[objId]
            if (!obj) throw new Error("Object doesn't exist: " + String(objId))
            const meta = this.metadata[objId]
            if (!meta) throw new Error("Object ID not found: " + String(objId))
            const a = input.action
            if (Array.isArray(obj) && Array.isArray(meta)) {
                if (a === "insert") {
                    let ref =
                        input.index === 0 ? HEAD : getListElementId(meta, input.index - 1, { lookAfterTombstones: true })
                    for (const value of input.values) {
                        ref = emit({ action: "set", obj: objId, elemId: ref, insert: true, value })
                    }
                } else if (a === "delete") {
                    /* always the same visible index: each delete shifts the next char into it */
                    for (let k = 0; k < input.count; k++) {
                        emit({ action: "del", obj: objId, elemId: getListElementId(meta, input.index) })
                    }
                } else if (a === "addMark" || a === "removeMark") {
                    emit(changeMark(input, objId, meta, obj))
                } else if (a === "del") {
                    throw new Error("Use the remove action")
                } else {
                    throw new Error("Unimplemented")
                }
            } else if (a === "makeList" || a === "makeMap" || a === "del") {
                emit({ action: a, obj: objId, key: input.key })
            } else if (a === "set") {
                emit({ action: a, obj: objId, key: input.key, value: input.value })
            } else {
                throw new Error("Not a list: " + input.path)
            }
        }
        return { change, patches }
    }

    /* micromerge.ts:483-493 */
    _newOp(change, partial) {
        this.maxOp += 1
        const opId = this.maxOp + "@" + this.actorId
        const op = Object.assign({ opId }, partial)
        const patches = this._applyOp(op)
        change.ops.push(op)
        return { opId, patches }
    }

    /* micromerge.ts:465-477 */
    getCursor(path, index) {
        const objectId = this.getObjectIdForPath(path)
        return { objectId, elemId: getListElementId(this.metadata[objectId], index) }
    }
    resolveCursor(cursor) {
        return this._find(cursor.objectId, cursor.elemId).visible
    }

    /* micromerge.ts:499-514 — causal admission, then the ops in order.  State is NOT rolled back on throw. */
    applyChange(change) {
        const last = this.clock[change.actor] || 0
        if (change.seq !== last + 1) {
            throw new RangeError("Expected sequence number " + (last + 1) + ", got " + change.seq)
        }
        const deps = change.deps || {}
        for (const actor of Object.keys(deps)) {
            if (!this.clock[actor] || this.clock[actor] < deps[actor]) {
                throw new RangeError("Missing dependency: change " + deps[actor] + " by actor " + actor)
            }
        }
        this.clock[change.actor] = change.seq
        this.maxOp = Math.max(this.maxOp, change.startOp + change.ops.length - 1)
        const out = []
        for (const op of change.ops) {
            for (const p of this._applyOp(op)) out.push(p)
        }
        return out
    }

    /* micromerge.ts:516-529 */
    getTextWithFormatting(path) {
        const id = this.getObjectIdForPath(path)
        const text = this.objects[id]
        const meta = this.metadata[id]
        if (text === undefined || !Array.isArray(text)) throw new Error("Expected a list at object ID " + String(id))
        if (meta === undefined || !Array.isArray(meta)) {
            throw new Error("Expected list metadata for object ID " + String(id))
        }
        return getTextWithFormatting(text, meta)
    }

    /* micromerge.ts:534-608 */
    _applyOp(op) {
        const meta = this.metadata[op.obj]
        const obj = this.objects[op.obj]
        if (!meta || obj === undefined) throw new RangeError("Object does not exist: " + String(op.obj))
        if (op.action === "makeMap") {
            this.objects[op.opId] = {}
            this.metadata[op.opId] = { keys: {}, children: {} }
        } else if (op.action === "makeList") {
            this.objects[op.opId] = []
            this.metadata[op.opId] = []
        }
        if (Array.isArray(meta)) {
            if (!Array.isArray(obj)) throw new Error("Non-array object with array metadata: " + String(op.obj))
            if (op.action === "set") {
                if (op.elemId === undefined) throw new Error("Must specify elemId when calling set on an array")
                return this._listInsert(op)
            }
            if (op.action === "del") {
                if (op.elemId === undefined) throw new Error("Must specify elemId when calling del on an array")
                return this._listDelete(op)
            }
            if (op.action === "addMark" || op.action === "removeMark") {
                return applyAddRemoveMark(op, obj, meta, this.wantPatches)
            }
            throw new Error("Unimplemented")
        }
        if (op.action === "addMark" || op.action === "removeMark") {
            throw new Error("Can't call addMark or removeMark on a map")
        }
        if (op.key === undefined) throw new Error("Must specify key when calling set or del on a map")
        if (Array.isArray(obj)) throw new Error("Metadata is map but object is array: " + String(op.obj))
        /* last-writer-wins per key */
        const prev = meta.keys[op.key]
        if (prev === undefined || compareOpIds(prev, op.opId) === -1) {
            meta.keys[op.key] = op.opId
            if (op.action === "del") {
                delete obj[op.key]
            } else if (op.action === "makeList") {
                obj[op.key] = this.objects[op.opId]
                meta.children[op.key] = op.opId
                return [Object.assign({}, op, { path: [CONTENT_KEY] })]
            } else if (op.action === "makeMap") {
                obj[op.key] = this.objects[op.opId]
                meta.children[op.key] = op.opId
            } else if (op.action === "set") {
                obj[op.key] = op.value
            }
        }
        return []
    }

    /* micromerge.ts:614-672 — RGA insert */
    _listInsert(op) {
        const elems = this.metadata[op.obj]
        if (!Array.isArray(elems)) throw new Error("Not a list: " + String(op.obj))
        let at
        let visible
        if (op.elemId === HEAD) {
            at = 0
            visible = 0
        } else {
            const hit = this._find(op.obj, op.elemId)
            at = hit.index + 1
            visible = hit.visible + (elems[hit.index].deleted ? 0 : 1)
        }
        /* concurrent inserts at the same spot: the larger opId goes first */
        while (at < elems.length && compareOpIds(op.opId, elems[at].elemId) < 0) {
            if (!elems[at].deleted) visible++
            at++
        }
        elems.splice(at, 0, { elemId: op.opId, valueId: op.opId, deleted: false })
        const text = this.objects[op.obj]
        if (!Array.isArray(text)) throw new Error("Not a list: " + String(op.obj))
        if (typeof op.value !== "string") throw new Error("Expected value inserted into text to be a string")
        text.splice(visible, 0, op.value)
        if (!this.wantPatches) return []
        const marks = opsToMarks(closestOpsToLeft(elems, at))
        return [{ path: [CONTENT_KEY], action: "insert", index: visible, values: [op.value], marks }]
    }

    /* micromerge.ts:677-724 — tombstone; idempotent */
    _listDelete(op) {
        const hit = this._find(op.obj, op.elemId)
        const elems = this.metadata[op.obj]
        if (elems === undefined) throw new Error("Object not found: " + String(op.obj))
        if (!Array.isArray(elems)) throw new Error("Not a list: " + String(op.obj))
        const el = elems[hit.index]
        if (el.deleted) return []
        const text = this.objects[op.obj]
        if (!Array.isArray(text)) throw new Error("Not a list: " + String(op.obj))
        el.deleted = true
        text.splice(hit.visible, 1)
        return [{ path: [CONTENT_KEY], action: "delete", index: hit.visible, count: 1 }]
    }

    /* micromerge.ts:731-755 */
    _find(objectId, elemId) {
        const elems = this.metadata[objectId]
        if (!elems) throw new Error("Object ID not found: " + String(objectId))
        if (!Array.isArray(elems)) throw new Error("Expected array metadata for findListElement")
        let visible = 0
        for (let index = 0; index < elems.length; index++) {
            if (elems[index].elemId === elemId) return { index, visible }
            if (!elems[index].deleted) visible++
        }
        throw new RangeError("List element not found: " + String(elemId))
    }
}
Micromerge.contentKey = CONTENT_KEY

/* peritext.ts:405-436 specialised to side === "before" (the only caller, :328-330) */
function closestOpsToLeft(elems, index) {
    for (let i = index - 1; i >= 0; i--) {
        if (elems[i].markOpsAfter !== undefined) return elems[i].markOpsAfter
        if (elems[i].markOpsBefore !== undefined) return elems[i].markOpsBefore
    }
    return []
}

/* ---------- applying a mark op (peritext.ts:154-281) ---------- */

/**
 * Slot sets are arrays in insertion order (the reference uses Set, whose iteration order is
 * insertion order).  Walks the 2n boundary slots left to right.
 */
function applyAddRemoveMark(op, text, elems, wantPatches) {
    if (!Array.isArray(elems)) throw new Error("Expected list metadata for a list")
    if (!Array.isArray(text)) throw new Error("Expected list metadata for a list")
    const patches = []
    const visibleLength = text.length
    let visibleIndex = 0
    let carried = []
    let phase = 0 /* 0 BEFORE, 1 DURING, 2 AFTER */
    let open /* partial patch */

    const close = endIndex => {
        /* peritext.ts:269-281 */
        if (open === undefined) return
        if (endIndex > open.startIndex && open.startIndex < visibleLength) {
            patches.push(Object.assign({}, open, { endIndex: Math.min(endIndex, visibleLength) }))
        }
        open = undefined
    }

    for (let slot = 0; slot < 2 * elems.length && phase !== 2; slot++) {
        const el = elems[slot >> 1]
        const isAfter = (slot & 1) === 1
        const field = isAfter ? "markOpsAfter" : "markOpsBefore"
        const sideName = isAfter ? "after" : "before"
        if (el[field] !== undefined) carried = el[field]

        /* peritext.ts:225-249 */
        let next
        if (op.start.type === sideName && op.start.elemId === el.elemId) {
            phase = 1
            next = carried.concat([op])
        } else if (op.end.type === sideName && op.end.elemId === el.elemId) {
            phase = 2
            next = carried.filter(o => o !== op)
        } else if (phase === 1 && el[field] !== undefined) {
            next = carried.concat([op])
        }
        if (next !== undefined) el[field] = next

        if (isAfter && !el.deleted) visibleIndex += 1

        if (next !== undefined && wantPatches) {
            close(visibleIndex)
            if (phase === 1 && !deepEqual(opsToMarks(carried), opsToMarks(next))) {
                /* peritext.ts:251-267 */
                open = { action: op.action, markType: op.markType, path: [CONTENT_KEY], startIndex: visibleIndex }
                if (op.action === "addMark" && (op.markType === "link" || op.markType === "comment")) {
                    open.attrs = op.attrs
                }
            }
        }
    }
    close(visibleIndex)
    return patches
}

/* ---------- flattening (peritext.ts:337-395) ---------- */

function getTextWithFormatting(text, elems) {
    if (text === undefined || !Array.isArray(text)) throw new Error("Expected a list at object ID objectId")
    if (elems === undefined || !Array.isArray(elems)) throw new Error("Expected list metadata for object ID objectId")
    const spans = []
    let run = []
    let marks = {}
    let visible = 0
    for (let i = 0; i < elems.length; i++) {
        let fresh
        if (elems[i].markOpsBefore) fresh = opsToMarks(elems[i].markOpsBefore)
        else if (i > 0 && elems[i - 1].markOpsAfter) fresh = opsToMarks(elems[i - 1].markOpsAfter)
        if (fresh !== undefined) {
            pushRun(spans, run, marks)
            run = []
            marks = fresh
        }
        if (!elems[i].deleted) run.push(text[visible++])
    }
    pushRun(spans, run, marks)
    return spans
}

/* ---------- generation of mark ops (peritext.ts:458-501) ---------- */

function changeMark(input, objId, elems, text) {
    const endGrows = MARK_SPEC[input.markType].inclusive
    /* startGrows is hard-wired false in the reference (:466) */
    const start = { type: "before", elemId: getListElementId(elems, input.startIndex) }
    let end
    if (endGrows && input.endIndex >= text.length) end = { type: "endOfText" }
    else if (endGrows) end = { type: "before", elemId: getListElementId(elems, input.endIndex) }
    else end = { type: "after", elemId: getListElementId(elems, input.endIndex - 1) }
    const op = { action: input.action, obj: objId, start, end, markType: input.markType }
    if (input.attrs) op.attrs = input.attrs
    return op
}

/* ---------- JSON interchange ---------- */

/**
 * Restore what JSON.stringify of a reference Change loses (Symbols): a `makeList` without `obj`
 * targets ROOT, an inserting `set` without `elemId` anchors at HEAD (SURVEY A.6-9).  Returns a copy.
 */
function normalizeChange(change) {
    const ops = change.ops.map(op => {
        const o = Object.assign({}, op)
        if ((o.action === "makeList" || o.action === "makeMap") && o.obj === undefined) o.obj = ROOT
        if (o.action === "set" && o.insert && o.elemId === undefined) o.elemId = HEAD
        return o
    })
    return { actor: change.actor, seq: change.seq, deps: Object.assign({}, change.deps || {}), startOp: change.startOp, ops }
}

module.exports = {
    default: Micromerge,
    Micromerge,
    ROOT,
    HEAD,
    CONTENT_KEY,
    MARK_SPEC,
    ALL_MARKS,
    compareOpIds,
    splitOpId,
    getListElementId,
    applyAddRemoveMark,
    opsToMarks,
    getTextWithFormatting,
    changeMark,
    deepEqual,
    sortedById,
    pushRun,
    normalizeChange,
}
