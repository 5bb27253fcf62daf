/*
 * rootmap_core.h — the MAP objects of a replica (the root map and the maps nested in it): what Micromerge.getRoot() shows
 * (reference/src/micromerge.ts:443-449) after the replica has applied its log.
 *
 * applyOp on a map object (micromerge.ts:572-602) is last-writer-wins per key: an op takes effect iff its opId is greater
 * (compareOpIds, :812-827) than the last op that wrote the key; "del" removes the key, makeMap / makeList put the child object
 * they have just created (:541-547) there, "set" a value.  The order-independent closed form: per (object, key) the op with the
 * LARGEST opId of all ops of the log on that pair decides.  The op must find its object (:535-540 "Object does not exist"): the
 * root, or a map a makeMap row EARLIER in the log created; a key op on a list object is an error too (:581-583).
 *
 * These ops are rare (the text path never makes any but the list's own makeList), so the kernel is deliberately plain: one wave
 * per replica log collects the log's map rows in LDS (object, key, opId, row), and every collected op looks at every other one
 * (existence of its object, a later writer of its key).  Same two-platform discipline as merge_core.h.
 */
#pragma once
#include "merge_core.h"

struct PtxRootArgs {
    const uint64_t* log_off;
    const uint64_t* op_id;
    const uint64_t* ref_a;   /* map rows: the object (0 = root) */
    const uint64_t* ref_b;   /* map rows: key id */
    const uint32_t* payload;
    const uint8_t* action;
    const uint8_t* mark_type; /* PTX_ACT_MAPSET rows: PTX_MAPV_* */
    const uint64_t* entry_off; /* [n_logs + 1] capacity of the entry rows per log (>= its map rows) */
    ptx_root_entry* entries;
    ptx_root_log* rlogs;
    uint32_t n_logs;
    uint32_t lds_bytes;
};

struct PtxRootHdr {
    uint32_t m;    /* map rows collected */
    uint32_t ne;   /* entries written */
    uint32_t err;  /* first failing row, 0xFFFFFFFF = none */
    uint32_t pad;
};

struct PtxRootOp {
    uint64_t obj, id;
    uint32_t key, row;
    uint32_t kind, value; /* PTX_MAPV_* (PTX_MAPV_DELETED for a del) */
};

/* bytes of LDS for a log with m map rows */
PTX_HD uint64_t ptx_rootmap_lds_need(uint64_t m) { return ptx_a16(sizeof(PtxRootHdr)) + ptx_a16((m + 1) * sizeof(PtxRootOp)); }

template <uint32_t kThreads>
PTX_DEV void ptx_rootmap_log(const PtxRootArgs& A, uint32_t log, uint8_t* lds) {
    PtxRootHdr* H = (PtxRootHdr*)lds;
    PtxRootOp* ops = (PtxRootOp*)(lds + ptx_a16(sizeof(PtxRootHdr)));
    const uint64_t base = A.log_off[log];
    const uint32_t N = (uint32_t)(A.log_off[log + 1] - base);
    const uint64_t ebase = A.entry_off[log];
    const uint32_t ecap = (uint32_t)(A.entry_off[log + 1] - ebase);
    const uint32_t cap = A.lds_bytes > ptx_a16(sizeof(PtxRootHdr)) ? (uint32_t)((A.lds_bytes - ptx_a16(sizeof(PtxRootHdr))) / sizeof(PtxRootOp)) : 0u;
    PTX_LEADER {
        H->m = 0;
        H->ne = 0;
        H->err = 0xFFFFFFFFu;
    }
    PTX_SYNC_T();
    /* collect the map rows (any order) */
    PTX_FOR(i, N) {
        const uint32_t a = A.action[base + i];
        const bool is_map = a == PTX_ACT_MAPSET || a == PTX_ACT_MAPDEL || a == PTX_ACT_MAKELIST;
        const uint32_t at = ptx_append(&H->m, is_map);
        if (is_map && at < cap) {
            PtxRootOp o;
            o.obj = a == PTX_ACT_MAKELIST ? 0ull : A.ref_a[base + i]; /* the text list hangs off the root map */
            o.id = A.op_id[base + i];
            o.key = (uint32_t)A.ref_b[base + i];
            o.row = i;
            o.kind = a == PTX_ACT_MAPDEL ? (uint32_t)PTX_MAPV_DELETED : a == PTX_ACT_MAKELIST ? (uint32_t)PTX_MAPV_LIST : (uint32_t)A.mark_type[base + i];
            o.value = A.payload[base + i];
            ops[at] = o;
        }
    }
    PTX_SYNC_T();
    const uint32_t m = H->m;
    if (m > cap || m > ecap) {
        PTX_LEADER {
            ptx_root_log r;
            r.status = PTX_ERR_CAPACITY;
            r.n_entries = 0;
            r.first_bad_row = 0xFFFFFFFFu;
            r.reserved = 0;
            A.rlogs[log] = r;
        }
        return;
    }
    /* every op: its object must exist when it is applied; it decides its key iff no op on the same (object, key) has a larger opId */
    PTX_FOR(x, m) {
        const PtxRootOp me = ops[x];
        bool obj_ok = me.obj == 0ull, later = false;
        for (uint32_t y = 0; y < m; ++y) {
            const PtxRootOp o = ops[y];
            /* the op that created my object: a makeMap row applied before me (a makeList makes a list: key ops on it fail) */
            if (o.id == me.obj && o.kind == PTX_MAPV_MAP && o.row < me.row) obj_ok = true;
            if (y != x && o.obj == me.obj && o.key == me.key && o.id > me.id) later = true;
        }
        if (!obj_ok || me.kind > PTX_MAPV_DELETED) ptx_atomic_min(&H->err, me.row);
        else if (!later) {
            const uint32_t at = ptx_atomic_add(&H->ne, 1u);
            ptx_root_entry e;
            e.obj = me.obj;
            e.key = me.key;
            e.row = me.row;
            e.kind = me.kind;
            e.value = me.value;
            if (at < ecap) A.entries[ebase + at] = e;
        }
    }
    PTX_SYNC_T();
    PTX_LEADER {
        ptx_root_log r;
        r.status = H->err != 0xFFFFFFFFu ? (uint32_t)PTX_ERR_ELEM_NOT_FOUND : (uint32_t)PTX_OK;
        r.n_entries = r.status ? 0u : H->ne;
        r.first_bad_row = H->err;
        r.reserved = 0;
        A.rlogs[log] = r;
    }
}
