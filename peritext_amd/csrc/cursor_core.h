/*
 * cursor_core.h — cursor resolution on the device.
 *
 * What it replaces, per replica:
 *   Micromerge.getCursor(path, index)   reference/src/micromerge.ts:465-473 -> getListElementId(meta, index) (:762-805, without
 *                                       lookAfterTombstones): the elemId of the index-th VISIBLE element; RangeError "List index
 *                                       out of bounds" past the end
 *   Micromerge.resolveCursor(cursor)    :475-477 -> findListElement(objectId, elemId).visible (:731-755): the number of visible
 *                                       elements BEFORE the cursor's element (which may itself be a tombstone: a cursor survives
 *                                       the deletion of its character); RangeError "List element not found" for an unknown id
 *
 * The replica's element order comes from ptx_merge_kernel (`elem_rank`: document position + tombstone flag per insert row).  One
 * workgroup per replica log that has queries builds, in LDS, the id index of the inserts (bitmap + popcount prefix, as
 * merge_core.h), element -> row, document position -> row and the alive bitmap by document position with its popcount prefix;
 * every query is then O(1) (resolve: two lookups + one popcount) or O(log n) (get: a binary search over the prefix words).
 * Documents beyond that form's 16-bit row indices or the CU's LDS take the long-document form below (one pass over the rows per query).
 * Compiled two ways like merge_core.h (hipcc: the product; g++ -DPTX_EMU: CPU tests).
 */
#pragma once
#include "merge_core.h"

struct PtxCursorArgs {
    const uint64_t* log_off;
    const uint64_t* op_id;
    const uint8_t* action;
    const ptx_log_hdr* log_hdr;
    const ptx_log_result* res;
    const uint32_t* elem_rank;
    /* queries grouped by log: group g = log q_group_log[g], its queries are q_perm[q_group_off[g] .. q_group_off[g+1]) */
    const uint32_t* q_group_log;
    const uint64_t* q_group_off;
    const uint32_t* q_perm;
    const uint8_t* q_kind; /* PTX_CURSOR_RESOLVE / PTX_CURSOR_GET, indexed by query */
    const uint64_t* q_arg; /* resolve: the cursor's elemId (counter << 32 | actorRank); get: the visible index */
    uint64_t* out;         /* resolve: visible index; get: elemId */
    uint32_t* status;      /* PTX_OK / PTX_ERR_ELEM_NOT_FOUND / PTX_ERR_INDEX_OOB / the log's merge status / PTX_ERR_CAPACITY */
    uint32_t n_groups;
    uint32_t lds_bytes;
};

struct PtxCursorHdr {
    uint32_t V;
    uint32_t scan_tmp[36];
};

PTX_HD uint64_t ptx_cursor_lds_need(uint64_t n, uint64_t ks) {
    const uint64_t nw = (ks + 31) / 32, nwe = (n >> 5) + 2;
    return ptx_a16(sizeof(PtxCursorHdr)) + ptx_a16(8 * (nw + 1)) + 2 * ptx_a16(2 * (n + 1)) + ptx_a16(8 * nwe);
}
/* Documents beyond the indexed form's 16-bit row indices (32 766 elements / 65 534 rows) or one CU's LDS (round 5; the reference has no bound,
 * micromerge.ts:465-477): only the alive bitmap by document position lives in LDS (a quarter byte per element: ~600 000 elements in a CU's 160 KB); a query is
 * then one pass of the workgroup over the log's rows (resolve: the row that inserted the id; get: the row whose element sits at the wanted position) + one
 * popcount lookup.  O(rows) per query instead of O(1) — cursors are a handful per replica. */
PTX_HD uint64_t ptx_cursor_lds_need_long(uint64_t n) { return ptx_a16(sizeof(PtxCursorHdr)) + ptx_a16(8 * ((n >> 5) + 2)); }
PTX_HD bool ptx_cursor_indexed(uint64_t N, uint64_t n, uint64_t max_actor, uint64_t ks, uint64_t lds_bytes) {
    return n <= 32766u && N <= 65534u && max_actor <= 4095u && ks < (1ull << 31) && ptx_cursor_lds_need(n, ks) <= lds_bytes;
}

template <uint32_t kThreads>
PTX_DEV void ptx_cursor_group(const PtxCursorArgs& A, uint32_t g, uint8_t* lds) {
    PtxCursorHdr* H = (PtxCursorHdr*)lds;
    const uint32_t log = A.q_group_log[g];
    const uint64_t q0 = A.q_group_off[g], q1 = A.q_group_off[g + 1];
    const uint64_t base = A.log_off[log];
    const uint32_t N = (uint32_t)(A.log_off[log + 1] - base);
    const uint64_t* op_id = A.op_id + base;
    const uint32_t* erank = A.elem_rank + base;
    const uint32_t merge_status = A.res[log].status;
    const ptx_log_hdr hd = A.log_hdr[log];
    const uint32_t n = N ? hd.n_ins : 0u;
    PtxElemIndex ix;
    ix.max_ctr = N ? hd.max_counter : 0u;
    ix.max_actor = N ? hd.max_actor : 0u;
    ix.na1 = ix.max_actor + 1u;
    const uint64_t ks64 = ((uint64_t)ix.max_ctr + 1u) * ix.na1;
    if (merge_status == PTX_OK && !ptx_cursor_indexed(N, n, ix.max_actor, ks64, A.lds_bytes)) {
        /* ---- the long-document form: alive bitmap by document position + one pass over the rows per query ---- */
        const uint32_t nwl = (n >> 5) + 2;
        PtxBitWord* al = (PtxBitWord*)(lds + ptx_a16(sizeof(PtxCursorHdr)));
        if (ptx_cursor_lds_need_long(n) > A.lds_bytes) {
            PTX_FOR(k, (uint32_t)(q1 - q0)) {
                const uint32_t q = A.q_perm[q0 + k];
                A.status[q] = PTX_ERR_CAPACITY;
                A.out[q] = 0;
            }
            return;
        }
        PTX_FOR(w, nwl) {
            PtxBitWord z;
            z.bits = 0;
            z.pre = 0;
            al[w] = z;
        }
        PTX_SYNC();
        PTX_FOR(i, N) {
            if (A.action[base + i] == PTX_ACT_INSERT) {
                const uint32_t rk = erank[i], r = rk & PTX_RANK_MASK;
                if (r < n && !(rk & PTX_RANK_TOMBSTONE)) ptx_atomic_or(&al[r >> 5].bits, 1u << (r & 31));
            }
        }
        PTX_SYNC();
        PTX_FOR(w, nwl) al[w].pre = ptx_popc(al[w].bits);
        PTX_SYNC();
        const uint32_t Vl = ptx_scan_excl<uint32_t, 2, kThreads>(&al[0].pre, nwl, H->scan_tmp);
        for (uint64_t k = 0; k < q1 - q0; ++k) { /* (uniform: every thread works on the same query) */
            const uint32_t q = A.q_perm[q0 + k];
            const uint64_t arg = A.q_arg[q];
            const bool resolve = A.q_kind[q] == PTX_CURSOR_RESOLVE;
            uint32_t want_rank = 0xFFFFFFFFu;
            if (!resolve && arg < Vl) { /* the position of the arg-th visible element: the word whose prefix range holds the index, then the bit */
                const uint32_t want = (uint32_t)arg;
                uint32_t lo = 0, hi = nwl - 1u;
                while (lo < hi) {
                    const uint32_t mid = (lo + hi + 1u) >> 1;
                    if (al[mid].pre <= want) lo = mid;
                    else hi = mid - 1u;
                }
                uint32_t m = al[lo].bits;
                for (uint32_t s2 = al[lo].pre; s2 < want; ++s2) m &= m - 1u;
                want_rank = (lo << 5) + (uint32_t)__builtin_ctz(m);
            }
            PTX_LEADER {
                A.status[q] = resolve ? PTX_ERR_ELEM_NOT_FOUND : PTX_ERR_INDEX_OOB; /* micromerge.ts:752 / :804, unless a row answers below */
                A.out[q] = 0;
            }
            PTX_SYNC(); /* (the leader's defaults stand before a finder overwrites them) */
            if (resolve || want_rank != 0xFFFFFFFFu) {
                PTX_FOR(i, N) {
                    if (A.action[base + i] == PTX_ACT_INSERT) {
                        const uint32_t r = erank[i] & PTX_RANK_MASK;
                        if (resolve ? op_id[i] == arg : r == want_rank) { /* ids and positions are unique: at most one row answers */
                            A.out[q] = resolve ? (uint64_t)ptx_bitrank(al, r) : op_id[i];
                            A.status[q] = PTX_OK;
                        }
                    }
                }
            }
            PTX_SYNC();
        }
        return;
    }
    const uint32_t keyspace = (uint32_t)ks64;
    const uint32_t nw = (keyspace + 31) / 32, nwe = (n >> 5) + 2;
    PtxBump bp;
    bp.base = lds;
    bp.off = (uint32_t)ptx_a16(sizeof(PtxCursorHdr));
    bp.cap = A.lds_bytes;
    bp.high = bp.off;
    bp.overflow = false;
    ix.ib = ptx_alloc<PtxBitWord>(bp, nw + 1);
    uint16_t* row_of = ptx_alloc<uint16_t>(bp, n + 1); /* element (rank of its id among the inserts) -> row */
    uint16_t* row_at = ptx_alloc<uint16_t>(bp, n + 1); /* document position -> row */
    PtxBitWord* alive = ptx_alloc<PtxBitWord>(bp, nwe);
    uint32_t fail = merge_status; /* a replica the reference threw on has no cursors */
    if (fail == PTX_OK && (bp.overflow || ix.max_actor > 4095u || n > 32766u || N > 65534u)) fail = PTX_ERR_CAPACITY;
    if (fail != PTX_OK) {
        PTX_FOR(k, (uint32_t)(q1 - q0)) {
            const uint32_t q = A.q_perm[q0 + k];
            A.status[q] = fail;
            A.out[q] = 0;
        }
        return;
    }
    PTX_FOR(w, nw + 1) {
        PtxBitWord z;
        z.bits = 0;
        z.pre = 0;
        ix.ib[w] = z;
    }
    PTX_FOR(w, nwe) {
        PtxBitWord z;
        z.bits = 0;
        z.pre = 0;
        alive[w] = z;
    }
    PTX_SYNC();
    PTX_FOR(i, N) {
        if (A.action[base + i] == PTX_ACT_INSERT) {
            uint32_t key = 0;
            if (ptx_id_key(ix, op_id[i], key)) ptx_atomic_or(&ix.ib[key >> 5].bits, 1u << (key & 31));
            const uint32_t rk = erank[i];
            const uint32_t r = rk & PTX_RANK_MASK;
            if (r < n) {
                row_at[r] = (uint16_t)i;
                if (!(rk & PTX_RANK_TOMBSTONE)) ptx_atomic_or(&alive[r >> 5].bits, 1u << (r & 31));
            }
        }
    }
    PTX_SYNC();
    PTX_FOR(w, nw + 1) ix.ib[w].pre = ptx_popc(ix.ib[w].bits);
    PTX_FOR(w, nwe) alive[w].pre = ptx_popc(alive[w].bits);
    PTX_SYNC();
    ptx_scan_excl<uint32_t, 2, kThreads>(&ix.ib[0].pre, nw + 1, H->scan_tmp);
    const uint32_t V = ptx_scan_excl<uint32_t, 2, kThreads>(&alive[0].pre, nwe, H->scan_tmp);
    PTX_FOR(i, N) {
        if (A.action[base + i] == PTX_ACT_INSERT) {
            const int e = ptx_elem_lookup(ix, op_id[i]);
            if (e >= 0 && (uint32_t)e < n) row_of[e] = (uint16_t)i;
        }
    }
    PTX_SYNC();
    PTX_FOR(k, (uint32_t)(q1 - q0)) {
        const uint32_t q = A.q_perm[q0 + k];
        const uint64_t arg = A.q_arg[q];
        if (A.q_kind[q] == PTX_CURSOR_RESOLVE) {
            const int e = ptx_elem_lookup(ix, arg);
            if (e < 0) {
                A.status[q] = PTX_ERR_ELEM_NOT_FOUND; /* micromerge.ts:752 */
                A.out[q] = 0;
            } else {
                const uint32_t r = erank[row_of[e]] & PTX_RANK_MASK;
                A.status[q] = PTX_OK;
                A.out[q] = ptx_bitrank(alive, r); /* visible elements strictly before it */
            }
        } else {
            if (arg >= V) {
                A.status[q] = PTX_ERR_INDEX_OOB; /* micromerge.ts:804 */
                A.out[q] = 0;
            } else {
                /* the word whose prefix range holds the index, then the (index - prefix)-th set bit of it */
                const uint32_t want = (uint32_t)arg;
                uint32_t lo = 0, hi = nwe - 1u;
                while (lo < hi) {
                    const uint32_t mid = (lo + hi + 1u) >> 1;
                    if (alive[mid].pre <= want) lo = mid;
                    else hi = mid - 1u;
                }
                uint32_t m = alive[lo].bits;
                for (uint32_t s = alive[lo].pre; s < want; ++s) m &= m - 1u;
                const uint32_t r = (lo << 5) + (uint32_t)__builtin_ctz(m);
                A.status[q] = PTX_OK;
                A.out[q] = op_id[row_at[r]];
            }
        }
    }
}
