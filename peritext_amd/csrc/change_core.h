/*
 * change_core.h — `Micromerge.change(InputOperation[])` on the device for CALLER-SUPPLIED input operations.
 *
 * What it replaces, per replica: `doc.change(ops)` (reference/src/micromerge.ts:308-441) — index-based
 * InputOperations (:133-148: insert {index, values}, delete {index, count}, addMark / removeMark {startIndex, endIndex,
 * markType, attrs}, makeList {key: "text"}) resolved against the replica's CURRENT state into id-based Operations:
 *   insert  after the element at index-1, past the tombstones whose `after` slot is a defined one
 *           (getListElementId with lookAfterTombstones, :762-805), one op per value, each after the previous (:335-345)
 *   delete  `count` single deletes at the same visible index (:346-352)
 *   marks   start = before(elem[startIndex]); end = endOfText / before(elem[endIndex]) for the inclusive types
 *           (strong, em: schema.ts:45-96), after(elem[endIndex-1]) for link / comment  (changeMark, peritext.ts:458-501)
 * every op applied locally at once (makeNewOp :483-493: opId = ++maxOp @ actor), the Change = {actor, seq = clock+1,
 * deps = the clock before, startOp, ops}.  A bad index is the reference's RangeError "List index out of bounds" (:804).
 *
 * The replica's current state comes from its op log, already merged by ptx_merge_kernel: `elem_rank` gives every element's
 * document position and tombstone flag; which `after` slots are defined ones (markOpsAfter !== undefined) follows from the
 * mark ops of the log in closed form (applyAddRemoveMark's walk, peritext.ts:167-214: an op's end slot is written whenever its
 * element exists — unless it is the very slot the op starts on —, its start slot unless the end was met first); the clock is
 * the per-actor change count of the log's envelope; maxOp the largest counter of the log (Lamport, micromerge.ts:511).
 *
 * One 64-thread workgroup (one wave) per replica log, the element list in document order in LDS (one u32 per element:
 * dense element index | tombstone | after-defined), sequential over the InputOperations — the list primitives are the
 * 64-wide ballots of gen_core.h.  Compiled two ways like merge_core.h (hipcc: the product; g++ -DPTX_EMU: CPU tests).
 */
#pragma once
#include "gen_core.h"

#define PTX_CE_MASK 0x3FFFFFFFu /* the element inside a list word (bits 30 / 31: PTX_GK_DEAD / PTX_GK_AFTER): the ROW of the log that inserted it, or
                                   PTX_CE_NEW + j for the j-th element this call makes (round 6: rows up to 2^29 — the word was 16 + 4 bits before) */
#define PTX_CE_NEW 0x20000000u

struct PtxChangeArgs {
    /* the base batch (resident) and its merge result */
    const uint64_t* log_off;
    const uint64_t* op_id;
    const uint64_t* ref_a;
    const uint64_t* ref_b;
    const uint8_t* action;
    const uint8_t* mark_type;
    const uint8_t* side_a;
    const uint8_t* side_b;
    const ptx_log_hdr* log_hdr;
    const ptx_log_result* res;
    const uint32_t* elem_rank;
    const uint32_t* refs;      /* the merge's resolved references (PtxMergeArgs.out_refs): per mark row its boundary slots start | end << 16, 0xFFFF = none */
    const uint32_t* refs_hi;   /* (a log of more than 32 766 list elements) the slots' high halves, PtxMergeArgs.out_refs_hi */
    uint32_t* list_scratch;    /* optional: the element list of a log that does not fit one CU's LDS lives HERE (round 6) ... */
    const uint64_t* list_off;  /* ... [n_logs + 1], in words: the slice of log l (empty: the list fits the LDS) */
    const uint64_t* chg_off;   /* base envelope: the replica's clock = changes per actor */
    const uint32_t* chg_hdr;
    uint32_t max_actors;
    /* the InputOperations: log l makes changes [in_chg_off[l], in_chg_off[l+1]); change c holds input ops [in_op_off[c], in_op_off[c+1]) */
    const uint64_t* in_chg_off;
    const uint64_t* in_op_off;
    const uint8_t* in_action;
    const uint8_t* in_mark_type;
    const uint32_t* in_index;
    const uint32_t* in_count;
    const uint32_t* in_payload;
    const uint32_t* in_values;
    const uint32_t* actor;     /* [n_logs] actorRank of the replica behind log l */
    /* output, capacity layout: log l owns rows [out_off[l], out_off[l+1]) and envelope rows [in_chg_off[l], in_chg_off[l+1]) */
    const uint64_t* out_off;
    uint64_t* o_op_id;
    uint64_t* o_ref_a;
    uint64_t* o_ref_b;
    uint32_t* o_payload;
    uint8_t* o_action;
    uint8_t* o_mark_type;
    uint8_t* o_side_a;
    uint8_t* o_side_b;
    uint32_t* o_chg_hdr;
    uint16_t* o_chg_env;       /* rows of PTX_ENV_STRIDE(max_actors): the low halves of seq / deps */
    uint16_t* o_chg_env_hi;    /* the high halves (the wide column of include/peritext_hip.h), same shape */
    uint32_t* any_wide;        /* set to 1 when some high half is not zero (the host then keeps the wide column) */
    uint32_t* status;          /* [n_logs] PTX_OK / PTX_ERR_* */
    uint32_t* rows_made;       /* [n_logs] rows written (0 on error) */
    uint32_t* chgs_made;       /* [n_logs] */
    uint32_t n_logs;
    uint32_t lds_bytes;
};

struct PtxChangeHdr {
    uint32_t n;        /* list length incl. tombstones */
    uint32_t vis;      /* visible length */
    uint32_t max_op;
    uint32_t rows;     /* rows written */
    uint32_t has_list; /* the log holds the makeList of the text */
    uint32_t err;
    uint32_t scan_tmp[36];
};

/* LDS of one log: n elements now, `grow` inserts to come, id keyspace of ks bits, na actors */
PTX_HD uint64_t ptx_change_lds_need(uint64_t n, uint64_t grow, uint64_t ks, uint64_t na, bool list_in_hbm = false) {
    (void)ks; /* (no element index: the rows come resolved from the merge) */
    const uint64_t cap = n + grow + 64;
    return ptx_a16(sizeof(PtxChangeHdr)) + (list_in_hbm ? 0 : ptx_a16(4 * cap)) + ptx_a16(4 * (grow + 1)) + ptx_a16(4 * (na + 1));
}
PTX_HD uint64_t ptx_change_list_words(uint64_t n, uint64_t grow) { return (n + grow + 64 + 3) & ~3ull; } /* of list_scratch (16-byte blocks: ptx_list_shift_up) */

template <uint32_t kThreads>
PTX_DEV void ptx_change_log(const PtxChangeArgs& A, uint32_t log, uint8_t* lds) {
    PtxChangeHdr* H = (PtxChangeHdr*)lds;
    const uint64_t base = A.log_off[log];
    const uint32_t N = (uint32_t)(A.log_off[log + 1] - base);
    const uint64_t* op_id = A.op_id + base;
    const uint32_t* erank = A.elem_rank + base;
    const uint64_t ch0 = A.in_chg_off[log], ch1 = A.in_chg_off[log + 1];
    const uint64_t out0 = A.out_off[log];
    const uint32_t out_cap = (uint32_t)(A.out_off[log + 1] - out0);
    const uint32_t me = A.actor[log], na = A.max_actors;

#define PTX_CHANGE_FAIL(code_)                        \
    do {                                              \
        PTX_SYNC();                                   \
        PTX_LEADER {                                  \
            A.status[log] = (code_);                  \
            A.rows_made[log] = 0;                     \
            A.chgs_made[log] = 0;                     \
        }                                             \
        return;                                       \
    } while (0)

    if (ch1 == ch0) { /* nothing to do for this replica */
        PTX_LEADER {
            A.status[log] = PTX_OK;
            A.rows_made[log] = 0;
            A.chgs_made[log] = 0;
        }
        return;
    }
    const uint32_t merge_status = A.res[log].status;
    if (merge_status != PTX_OK) PTX_CHANGE_FAIL(merge_status); /* the replica itself is broken: nothing can be built on it */
    if (me >= na) PTX_CHANGE_FAIL(PTX_ERR_BAD_OP);

    const ptx_log_hdr hd = A.log_hdr[log];
    const uint32_t n0 = N ? hd.n_ins : 0u;
    uint32_t grow = 0; /* list elements this call adds */
    for (uint64_t q = A.in_op_off[ch0]; q < A.in_op_off[ch1]; ++q) grow += A.in_action[q] == PTX_IN_INSERT ? A.in_count[q] : 0u;
    const uint32_t max_ctr0 = N ? hd.max_counter : 0u;
    const uint32_t cap = n0 + grow + 64u;
    PtxBump bp;
    bp.base = lds;
    bp.off = (uint32_t)ptx_a16(sizeof(PtxChangeHdr));
    bp.cap = A.lds_bytes;
    bp.high = bp.off;
    bp.overflow = false;
    /* The list in document order, one word per element: the ROW that inserted it | PTX_GK_DEAD | PTX_GK_AFTER.  Everything it is built from comes resolved
     * from the merge — elem_rank (document position + tombstone of every insert row) and the boundary slots of every mark row (refs) — so no element index
     * is built here and a document of 32 766 elements fits one CU's LDS (round 4: with its own id bitmap, element -> row table and a second list-sized
     * buffer for opening a gap the kernel stopped at ~12 000). */
    /* Round 6: a list that does not fit the LDS (more than ~40 000 elements) lives in the log's slice of A.list_scratch — the same primitives over flat
     * addresses; the one wave's stores are complete and visible at every PTX_SYNC (a workgroup-scope barrier: the CU's own L1 is coherent for it). */
    const bool list_in_hbm = A.list_scratch && A.list_off[log + 1] - A.list_off[log] >= cap;
    uint32_t* L = list_in_hbm ? A.list_scratch + A.list_off[log] : ptx_alloc<uint32_t>(bp, cap);
    uint32_t* newctr = ptx_alloc<uint32_t>(bp, grow + 1); /* counter of the j-th element made here (its list word is PTX_CE_NEW + j) */
    uint32_t* clock = ptx_alloc<uint32_t>(bp, na + 1);
    if (bp.overflow || n0 + grow > 0x03FFFFFFu || N >= PTX_CE_NEW || grow >= PTX_CE_NEW || max_ctr0 + out_cap >= (1u << 19)) PTX_CHANGE_FAIL(PTX_ERR_CAPACITY);
    if (n0 > 32766u && !A.refs_hi) PTX_CHANGE_FAIL(PTX_ERR_CAPACITY); /* (a result without the slots' high halves) */

    /* ---- the replica's state from its merged log ---- */
    PTX_FOR(a, na + 1) clock[a] = 0;
    PTX_LEADER {
        H->n = n0;
        H->vis = N ? A.res[log].n_visible : 0u;
        H->max_op = max_ctr0;
        H->rows = 0;
        H->has_list = 0;
        H->err = 0;
    }
    PTX_SYNC();
    PTX_FOR(i, N) {
        const uint32_t a = A.action[base + i];
        if (a == PTX_ACT_INSERT) {
            const uint32_t rk = erank[i];
            if ((rk & PTX_RANK_MASK) < n0) L[rk & PTX_RANK_MASK] = i | ((rk & PTX_RANK_TOMBSTONE) ? PTX_GK_DEAD : 0u);
        } else if (a == PTX_ACT_MAKELIST) {
            H->has_list = 1;
        }
    }
    if (A.chg_off) {
        const uint64_t c0 = A.chg_off[log], c1 = A.chg_off[log + 1];
        PTX_FOR(c, (uint32_t)(c1 - c0)) {
            const uint32_t a = A.chg_hdr[c0 + c] >> PTX_CHG_ACTOR_SHIFT;
            if (a < na) ptx_atomic_add(&clock[a], 1u);
        }
    }
    PTX_SYNC();
    /* which `after` slots are defined ones: the walk of applyAddRemoveMark in closed form (positions never change once both
     * elements exist, so final ranks decide "the end is met before the start"); the slots as the merge resolved them (an element that was
     * not in the list when the op was applied: none) */
    PTX_FOR(i, N) {
        const uint32_t a = A.action[base + i];
        if ((a == PTX_ACT_ADDMARK || a == PTX_ACT_REMOVEMARK) && A.mark_type[base + i] < 4u) {
            const uint32_t v = A.refs[base + i];
            uint32_t slot_a = (v & 0xFFFFu) != 0xFFFFu ? v & 0xFFFFu : 0xFFFFFFFFu, slot_b = (v >> 16) != 0xFFFFu ? v >> 16 : 0xFFFFFFFFu;
            if (n0 > 32766u) { /* 32-bit slots: low halves | high halves, none = all ones in both */
                const uint32_t vh = A.refs_hi[base + i];
                slot_a = (v & 0xFFFFu) | (vh << 16);
                slot_b = (v >> 16) | (vh & 0xFFFF0000u);
            }
            const bool has_a = slot_a != 0xFFFFFFFFu, has_b = slot_b != 0xFFFFFFFFu;
            const bool end_first = has_b && (!has_a || slot_b < slot_a);
            const bool start_written = has_a && !end_first;
            const bool end_written = has_b && slot_b != slot_a;
            if (start_written && (slot_a & 1u)) ptx_atomic_or(&L[slot_a >> 1], PTX_GK_AFTER);
            if (end_written && (slot_b & 1u)) ptx_atomic_or(&L[slot_b >> 1], PTX_GK_AFTER);
        }
    }
    PTX_SYNC();

    /* id of the element behind a list word */
#define PTX_CE_ID(word_) (((word_) & PTX_CE_MASK) < PTX_CE_NEW ? op_id[(word_) & PTX_CE_MASK] : (((uint64_t)newctr[((word_) & PTX_CE_MASK) - PTX_CE_NEW] << 32) | me))
    /* one id-based op of the change under construction: row + local application bookkeeping */
#define PTX_CE_EMIT(act_, mt_, ra_, rb_, sa_, sb_, pay_)                        \
    do {                                                                        \
        const uint32_t k_ = H->rows, ctr_ = H->max_op + 1u;                     \
        PTX_SYNC();                                                             \
        PTX_LEADER {                                                            \
            if (k_ < out_cap) {                                                 \
                const uint64_t at_ = out0 + k_;                                 \
                A.o_op_id[at_] = ((uint64_t)ctr_ << 32) | me;                   \
                A.o_ref_a[at_] = (ra_);                                         \
                A.o_ref_b[at_] = (rb_);                                         \
                A.o_payload[at_] = (pay_);                                      \
                A.o_action[at_] = (uint8_t)(act_);                              \
                A.o_mark_type[at_] = (uint8_t)(mt_);                            \
                A.o_side_a[at_] = (uint8_t)(sa_);                               \
                A.o_side_b[at_] = (uint8_t)(sb_);                               \
            }                                                                   \
            H->rows = k_ + 1u;                                                  \
            H->max_op = ctr_;                                                   \
        }                                                                       \
        PTX_SYNC();                                                             \
        ++nops;                                                                 \
    } while (0)

    uint32_t made = 0, new_elems = 0;
    for (uint64_t c = ch0; c < ch1; ++c) {
        /* the Change header: deps = the clock before the change, seq = own clock + 1 (micromerge.ts:314-327) */
        const uint32_t seq = clock[me] + 1u;
        PTX_SYNC();
        const uint32_t es = PTX_ENV_STRIDE(na);
        PTX_FOR(b, es - 1u) { /* exact values: low halves here, high halves in the wide column */
            const uint32_t v = b < na ? clock[b] : 0u;
            A.o_chg_env[c * es + 1u + b] = (uint16_t)v;
            A.o_chg_env_hi[c * es + 1u + b] = (uint16_t)(v >> 16);
            if (v >= PTX_ENV_SATURATED) *A.any_wide = 1u; /* 65535 itself would read as "saturated" without the column */
        }
        PTX_SYNC();
        PTX_LEADER { clock[me] = seq; }
        PTX_SYNC();
        uint32_t nops = 0;
        for (uint64_t q = A.in_op_off[c]; q < A.in_op_off[c + 1]; ++q) {
            const uint32_t act = A.in_action[q], idx = A.in_index[q], cnt = A.in_count[q], pay = A.in_payload[q], mt = A.in_mark_type[q];
            if (act == PTX_IN_MAKELIST) {
                if (H->has_list) PTX_CHANGE_FAIL(PTX_ERR_BAD_OP); /* one text list per document */
                PTX_CE_EMIT(PTX_ACT_MAKELIST, 0, 0ull, (uint64_t)cnt, 0, 0, 0u); /* ref_b: the key's id ("text" is key 0 of every batch) */
                PTX_LEADER { H->has_list = 1; }
                PTX_SYNC();
                continue;
            }
            if (act == PTX_IN_MAPSET || act == PTX_IN_MAPDEL) {
                /* an op on a map object (micromerge.ts:400-425): the host has resolved the path; nothing of the list state is involved */
                uint64_t obj = 0;
                if (idx & PTX_IN_OBJ_NEW) {
                    const uint32_t k = idx & ~PTX_IN_OBJ_NEW;
                    if (k >= H->rows) PTX_CHANGE_FAIL(PTX_ERR_BAD_OP);
                    obj = A.o_op_id[out0 + k]; /* a map made earlier in this very call */
                } else if (idx) obj = ((uint64_t)(idx >> 12) << 32) | (uint64_t)(idx & 4095u);
                if (act == PTX_IN_MAPSET && mt > PTX_MAPV_LIST) PTX_CHANGE_FAIL(PTX_ERR_BAD_OP);
                PTX_CE_EMIT(act == PTX_IN_MAPSET ? PTX_ACT_MAPSET : PTX_ACT_MAPDEL, act == PTX_IN_MAPSET ? mt : 0u, obj, (uint64_t)cnt, 0, 0, act == PTX_IN_MAPSET ? pay : 0u);
                continue;
            }
            if (!H->has_list || act > PTX_IN_MAKELIST) PTX_CHANGE_FAIL(PTX_ERR_BAD_OP); /* "Child not found: text" / unknown action */
            const uint32_t n = H->n, vis = H->vis;
            if (act == PTX_IN_INSERT) {
                uint64_t ref = 0;
                uint32_t at = 0;
                if (idx != 0u) {
                    const uint32_t p = ptx_gen_select(L, n, idx - 1u);
                    if (p == 0xFFFFFFFFu) PTX_CHANGE_FAIL(PTX_ERR_INDEX_OOB); /* micromerge.ts:804 */
                    const uint32_t p2 = ptx_gen_after_tombstones(L, n, p);
                    ref = PTX_CE_ID(L[p2]);
                    at = p2 + 1u;
                }
                for (uint32_t v = 0; v < cnt; ++v) {
                    /* the new element has the largest id of the replica: nothing to skip (micromerge.ts:630), it lands right after its reference */
                    const uint32_t nn = H->n;
                    ptx_list_shift_up<uint32_t>(L, at, nn); /* L[at + 1 .. nn] = L[at .. nn - 1], whole 16-byte blocks from the top down */
                    PTX_SYNC();
                    PTX_LEADER {
                        L[at] = PTX_CE_NEW + new_elems;
                        newctr[new_elems] = H->max_op + 1u;
                        H->n = nn + 1u;
                        H->vis += 1u;
                    }
                    PTX_SYNC();
                    PTX_CE_EMIT(PTX_ACT_INSERT, 0, ref, 0ull, 0, 0, A.in_values[pay + v]);
                    ref = ((uint64_t)H->max_op << 32) | me;
                    ++new_elems;
                    ++at;
                }
            } else if (act == PTX_IN_DELETE) {
                for (uint32_t v = 0; v < cnt; ++v) { /* always the same visible index (micromerge.ts:346-352) */
                    const uint32_t p = ptx_gen_select(L, H->n, idx);
                    if (p == 0xFFFFFFFFu) PTX_CHANGE_FAIL(PTX_ERR_INDEX_OOB);
                    const uint64_t target = PTX_CE_ID(L[p]);
                    PTX_LEADER {
                        L[p] |= PTX_GK_DEAD;
                        H->vis -= 1u;
                    }
                    PTX_SYNC();
                    PTX_CE_EMIT(PTX_ACT_DELETE, 0, target, 0ull, 0, 0, 0u);
                }
            } else { /* changeMark (peritext.ts:458-501): idx = startIndex, cnt = endIndex */
                if (mt > PTX_MARK_LINK) PTX_CHANGE_FAIL(PTX_ERR_BAD_OP);
                const uint32_t ps = ptx_gen_select(L, n, idx);
                if (ps == 0xFFFFFFFFu) PTX_CHANGE_FAIL(PTX_ERR_INDEX_OOB);
                const uint64_t ra = PTX_CE_ID(L[ps]);
                const bool inclusive = mt == PTX_MARK_STRONG || mt == PTX_MARK_EM; /* schema.ts:45-96 */
                uint64_t rb = 0;
                uint32_t sb = PTX_SIDE_END_OF_TEXT;
                if (inclusive) {
                    if (cnt < vis) {
                        const uint32_t pe = ptx_gen_select(L, n, cnt);
                        if (pe == 0xFFFFFFFFu) PTX_CHANGE_FAIL(PTX_ERR_INDEX_OOB);
                        rb = PTX_CE_ID(L[pe]);
                        sb = PTX_SIDE_BEFORE;
                    }
                } else {
                    const uint32_t pe = cnt == 0u ? 0xFFFFFFFFu : ptx_gen_select(L, n, cnt - 1u); /* index -1 is out of bounds */
                    if (pe == 0xFFFFFFFFu) PTX_CHANGE_FAIL(PTX_ERR_INDEX_OOB);
                    rb = PTX_CE_ID(L[pe]);
                    sb = PTX_SIDE_AFTER;
                    PTX_LEADER { L[pe] |= PTX_GK_AFTER; } /* its end slot is a defined one from now on (peritext.ts:240) */
                    PTX_SYNC();
                }
                PTX_CE_EMIT(act == PTX_IN_ADDMARK ? PTX_ACT_ADDMARK : PTX_ACT_REMOVEMARK, mt, ra, rb, PTX_SIDE_BEFORE, sb,
                            (mt == PTX_MARK_COMMENT || (mt == PTX_MARK_LINK && act == PTX_IN_ADDMARK)) ? pay : 0u);
            }
        }
        PTX_LEADER {
            A.o_chg_hdr[c] = (me << PTX_CHG_ACTOR_SHIFT) | nops;
            A.o_chg_env[c * es] = (uint16_t)seq;
            A.o_chg_env_hi[c * es] = (uint16_t)(seq >> 16);
            if (seq >= PTX_ENV_SATURATED) *A.any_wide = 1u; /* 65535 itself would read as "saturated" without the column */
        }
        ++made;
    }
#undef PTX_CE_EMIT
#undef PTX_CE_ID
    PTX_SYNC();
    if (H->rows != out_cap) PTX_CHANGE_FAIL(PTX_ERR_INVALID_ARG); /* the host sized the rows from the same InputOperations */
    PTX_LEADER {
        A.status[log] = PTX_OK;
        A.rows_made[log] = H->rows;
        A.chgs_made[log] = made;
    }
#undef PTX_CHANGE_FAIL
}
