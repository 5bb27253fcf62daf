/*
 * replay_core.h — the incremental Patch[] stream of one replica log.
 *
 * What it replaces: the patches that `applyChange` (reference/src/micromerge.ts:499 -> applyOp :534) RETURNS, op by
 * op, in the replica's application order:
 *   insert  {action:"insert", index, values:[v], marks}      micromerge.ts:661-671, marks = getActiveMarksAtIndex
 *                                                            (peritext.ts:328) = the closest defined slot to the left
 *   delete  {action:"delete", index, count:1}                micromerge.ts:696-703 (nothing for a tombstone)
 *   add/removeMark  one patch per run of boundary slots whose effective marks change
 *                                                            peritext.ts:154-220, :251-281
 *   makeList  the op itself                                  micromerge.ts:575
 *
 * How: ptx_merge_kernel has already produced the FINAL document position (rank incl. tombstones) of every element
 * (`elem_rank`).  Because the RGA order of any two elements never changes once both exist, the state of the
 * replica at application time t is the final order restricted to the elements inserted before t.  The log is therefore
 * replayed in time order over arrays indexed by final rank / final boundary slot (slot = 2*rank + side):
 *   present   bit per rank  {bits, running popcount prefix}: inserted and not deleted -> visible index = popcount below
 *   defined   bit per slot : the reference's `markOpsBefore/After !== undefined` (peritext.ts:167-214): set at the start
 *             and end slot of every applied mark op; patches break at every defined slot inside the op's range
 *   win[3]    per defined slot: row of the max-opId covering op per non-multi mark type (opsToMarks, :304-313)
 *   anyc      bit per slot : some comment op covers (the `comment: []` state)
 *   comment ops: [start, end) slots + per-id chains in application order (the LAST-applied covering op of an id
 *             decides its presence, :314-321)
 * One 64-thread workgroup (one wave) per log: the replay is sequential in t, every step is a handful of wave-wide
 * passes over bitmap words / the defined slots of the range.  As ONE wave the phases need no s_barrier and no wait for the patches
 * just stored to HBM (nothing this kernel writes there is read back): PTX_SYNC_T is a compiler fence then.
 *
 * Compiled two ways like merge_core.h (hipcc: the product kernel; g++ -DPTX_EMU: CPU test tooling only).
 */
#pragma once
#include "merge_core.h"

#define PTX_SLOT_NONE 0xFFFFu /* start never matches / end never reached */
#define PTX_RCHUNK 64u
enum { PTX_RK_SKIP = 0, PTX_RK_MAKELIST = 1, PTX_RK_INSERT = 2, PTX_RK_DELETE = 3, PTX_RK_MARK = 4 };

struct PtxReplayArgs {
    const uint64_t* log_off;
    const uint64_t* op_id;
    const uint64_t* ref_a;
    const uint64_t* ref_b;
    const uint32_t* payload;
    const uint8_t* action;
    const uint8_t* mark_type;
    const uint8_t* side_a;
    const uint8_t* side_b;
    const ptx_log_hdr* log_hdr;
    const ptx_log_result* res;  /* of ptx_merge on the same batch */
    const uint32_t* elem_rank;  /* of ptx_merge on the same batch */
    const uint32_t* refs;       /* of the same ptx_merge (PtxMergeArgs.out_refs): target row of every delete, boundary slots of every mark op */
    const uint64_t* patch_off;  /* [n_logs + 1] capacity offsets into `patches` */
    ptx_patch* patches;
    ptx_patch_log* plogs;
    uint32_t n_logs;
    uint32_t lds_bytes;
    const uint32_t* first_row; /* optional [n_logs]: only the records of the rows from here on are produced (the rows before are replayed for their state alone) */
    uint32_t seg_lds;      /* (win_scratch) entries of a mark op's slot list kept in the LDS; the rest of the list lives behind the winner arrays */
    uint16_t* win_scratch; /* optional: the per-slot winner arrays of every log live HERE (16 bytes per row of the batch + 128 per log; the list of a mark op's defined slots too) instead of in LDS */
};

struct PtxReplayHdr {
    uint32_t npatch;   /* patches produced so far (may run past the capacity: only the count is kept then) */
    uint32_t ncom;     /* comment ops registered */
    uint32_t tmp;      /* per-step scratch: max / counter */
    uint32_t nvis;     /* visible length */
    uint32_t scan_tmp[36];
};

/* gwin: the per-slot winner arrays live in global memory and only the first seg_lds entries of a mark op's slot list in the LDS (PtxReplayArgs.seg_lds) */
PTX_HD uint64_t ptx_replay_lds_need(uint64_t n, uint64_t K, uint64_t Kc, uint64_t ks, uint64_t Kid, bool gwin = false, uint64_t seg_lds = 0) {
    const uint64_t nw = (ks + 31) / 32, nwe = (n >> 5) + 2, nws = ((2 * n + 2) >> 5) + 2;
    const uint64_t segcap = (2 * n + 2 < 2 * K + 2 ? 2 * n + 2 : 2 * K + 2) + 1;
    (void)nw;
    return ptx_a16(sizeof(PtxReplayHdr)) + ptx_a16(8 * nwe) + 2 * ptx_a16(4 * nws) +
           ptx_a16(4 * (nws + 1)) + (gwin ? ptx_a16(2 * (segcap < seg_lds ? segcap : seg_lds)) : 3 * ptx_a16(2 * (2 * n + 2)) + ptx_a16(2 * segcap)) + ptx_a16(8 * ((segcap >> 5) + 2)) + 3 * ptx_a16(4 * nws) +
           ptx_a16(8 * PTX_RCHUNK) + ptx_a16(4 * PTX_RCHUNK) + 3 * ptx_a16(2 * PTX_RCHUNK) + ptx_a16(PTX_RCHUNK) +
           4 * ptx_a16(2 * (Kc + 1)) + ptx_a16(2 * (Kid + 1)) + ptx_a16(Kc + 1);
}
PTX_HD uint64_t ptx_replay_lds_need_hdr(const ptx_log_hdr& h, bool gwin = false, uint64_t seg_lds = 0) {
    const uint64_t K = (uint64_t)h.n_mark[0] + h.n_mark[1] + h.n_mark[2] + h.n_mark[3];
    const uint64_t ks = ((uint64_t)h.max_counter + 1) * ((uint64_t)(h.max_actor > 4095u ? 4095u : h.max_actor) + 1);
    return ptx_replay_lds_need(h.n_ins, K, h.n_mark[PTX_MARK_COMMENT], ks, h.n_mark[PTX_MARK_COMMENT] ? h.n_comment_ids : 0u, gwin, seg_lds);
}
/* bytes of win_scratch a batch takes, and where the arrays of a log start (in u16 units): three winner arrays and the slot list of a mark op's range, each of at
 * most 2 N + 3 entries, 16-byte aligned */
PTX_HD uint64_t ptx_replay_win_bytes(uint64_t n_ops, uint64_t n_logs) { return 16 * n_ops + 128 * n_logs + 64; }
PTX_HD uint64_t ptx_replay_win_at(uint64_t base_row, uint64_t log) { return 8 * base_row + 64 * log; }

/* one patch record; rows past the capacity are counted, not written */
PTX_DEV void ptx_patch_put(const PtxReplayArgs& A, uint64_t pbase, uint32_t pcap, uint32_t idx, uint32_t row, uint32_t kind, uint32_t a, uint32_t b) {
    if (idx < pcap) {
        ptx_patch p;
        p.row = row;
        p.kind = kind;
        p.a = a;
        p.b = b;
        A.patches[pbase + idx] = p;
    }
}

/* kGWin: the winner arrays live in global memory (A.win_scratch), read and written past the L1 (device-scope relaxed atomics) with the wave's outstanding
 * stores waited for wherever one lane reads what another has written */
template <uint32_t kThreads, bool kGWin = false>
PTX_DEV void ptx_replay_log(const PtxReplayArgs& A, uint32_t log, uint8_t* lds) {
    PtxReplayHdr* H = (PtxReplayHdr*)lds;
    const uint64_t base = A.log_off[log];
    const uint32_t N = (uint32_t)(A.log_off[log + 1] - base);
    const uint64_t pbase = A.patch_off[log];
    const uint32_t pcap_all = (uint32_t)(A.patch_off[log + 1] - pbase);
    const uint32_t first = A.first_row ? A.first_row[log] : 0u; /* the stream starts with this row's records */
    const uint64_t* op_id = A.op_id + base;
    const uint32_t* payload = A.payload + base;
    const uint8_t* action = A.action + base;
    const uint8_t* mark_type = A.mark_type + base;
    const uint32_t* erank = A.elem_rank + base;
    const uint32_t* refs = A.refs + base;

    const uint32_t merge_status = A.res[log].status;
    if (merge_status != PTX_OK || N == 0) { /* the reference threw somewhere in this log: no stream (the status says why) */
        PTX_LEADER {
            ptx_patch_log pl;
            pl.status = merge_status;
            pl.n_patches = 0;
            A.plogs[log] = pl;
        }
        return;
    }
    const ptx_log_hdr hd = A.log_hdr[log];
    const uint32_t n = hd.n_ins, Kc = hd.n_mark[PTX_MARK_COMMENT];
    const uint32_t Kid = Kc ? hd.n_comment_ids : 0u; /* id space of the document's comments as this log has seen it */
    const uint32_t K = hd.n_mark[0] + hd.n_mark[1] + hd.n_mark[2] + hd.n_mark[3];
    const uint32_t nwe = (n >> 5) + 2, nws = ((2 * n + 2) >> 5) + 2;
    const uint32_t segcap = (2 * n + 2 < 2 * K + 2 ? 2 * n + 2 : 2 * K + 2) + 1;
    /* LWW winners of strong / em per slot: where every op id of the log has a dense key (counter * (max actor + 1) + actor, the merge kernel's) below
     * 65 535, the slot holds the winner's key + 1 — compareOpIds is then a compare of two LDS values — instead of its row (whose op id would have to
     * be fetched from HBM for every slot of every mark op: a dependent round trip in a sequential replay).  Links keep the row: their url is read through it. */
    const uint32_t na1 = hd.max_actor + 1u;
    const bool key_mode = (uint64_t)(hd.max_counter + 1ull) * na1 <= 65535ull;

    PtxBump bp;
    bp.base = lds;
    bp.off = (uint32_t)ptx_a16(sizeof(PtxReplayHdr));
    bp.cap = A.lds_bytes;
    bp.high = bp.off;
    bp.overflow = false;
    PtxBitWord* present = ptx_alloc<PtxBitWord>(bp, nwe);
    uint32_t* defined = ptx_alloc<uint32_t>(bp, nws);
    uint32_t* anyc = ptx_alloc<uint32_t>(bp, nws);
    uint32_t* wcnt = ptx_alloc<uint32_t>(bp, nws + 1); /* defined slots of the range per word -> prefix */
    uint16_t* win[3];
    if (kGWin) {
        const uint32_t stride = (2u * n + 2u + 7u) & ~7u;
        uint16_t* g = A.win_scratch + ptx_replay_win_at(base, log);
        win[0] = g;
        win[1] = g + stride;
        win[2] = g + 2u * stride;
    } else {
        win[0] = ptx_alloc<uint16_t>(bp, 2 * n + 2); /* strong */
        win[1] = ptx_alloc<uint16_t>(bp, 2 * n + 2); /* em */
        win[2] = ptx_alloc<uint16_t>(bp, 2 * n + 2); /* link */
    }
#define PTX_WIN_LD(p_) (kGWin ? ptx_coherent_load16(p_) : *(p_))
#define PTX_WIN_ST(p_, v_) do { if (kGWin) ptx_coherent_store16((p_), (uint16_t)(v_)); else *(p_) = (uint16_t)(v_); } while (0)
#define PTX_WIN_FENCE() do { if (kGWin) ptx_global_stores_done(); } while (0)
    uint32_t* won[3]; /* per slot: the winner of the type is an addMark (saves re-reading its action from HBM) */
    won[0] = ptx_alloc<uint32_t>(bp, nws);
    won[1] = ptx_alloc<uint32_t>(bp, nws);
    won[2] = ptx_alloc<uint32_t>(bp, nws);
    /* the next PTX_RCHUNK rows, resolved in parallel (element lookups, boundary slots) before they are replayed in order */
    uint64_t* c_id = ptx_alloc<uint64_t>(bp, PTX_RCHUNK);
    uint32_t* c_pay = ptx_alloc<uint32_t>(bp, PTX_RCHUNK);
    uint16_t* c_a = ptx_alloc<uint16_t>(bp, PTX_RCHUNK);   /* insert / delete: final rank; mark: start slot */
    uint16_t* c_b = ptx_alloc<uint16_t>(bp, PTX_RCHUNK);   /* mark: end slot */
    uint16_t* c_key = ptx_alloc<uint16_t>(bp, PTX_RCHUNK); /* (key_mode) the op id's dense key + 1 */
    uint8_t* c_kind = ptx_alloc<uint8_t>(bp, PTX_RCHUNK);  /* PTX_RK_* | mark type << 4 | addMark << 6 */
    /* defined slots of the op's range, ascending.  (global winners) the first A.seg_lds of them stay in the LDS — most ranges end there, and the list is
     * read right after it is filled: a global one costs the op two more round trips — the rest goes to global memory behind the winner arrays */
    const uint32_t segl = kGWin ? (segcap < A.seg_lds ? segcap : A.seg_lds) : segcap;
    uint16_t* seg_l = ptx_alloc<uint16_t>(bp, segl);
    uint16_t* seg = kGWin ? A.win_scratch + ptx_replay_win_at(base, log) + 3u * ((2u * n + 2u + 7u) & ~7u) : seg_l;
#define PTX_SEG_LD(j_) ((j_) < segl ? seg_l[j_] : ptx_coherent_load16(&seg[j_]))
    PtxBitWord* cf = ptx_alloc<PtxBitWord>(bp, (segcap >> 5) + 2); /* bit j: slot seg[j] opens a patch; prefix = its place */
    uint16_t* ca = ptx_alloc<uint16_t>(bp, Kc + 1);       /* comment op: first covered slot */
    uint16_t* cb = ptx_alloc<uint16_t>(bp, Kc + 1);       /*             first slot not covered (PTX_SLOT_NONE = to the end) */
    uint16_t* ccid = ptx_alloc<uint16_t>(bp, Kc + 1);
    uint16_t* cprev = ptx_alloc<uint16_t>(bp, Kc + 1);    /* chain of the ops with the same id, latest first from ctail[id] */
    uint16_t* ctail = ptx_alloc<uint16_t>(bp, Kid + 1);   /* per id: last registered op */
    uint8_t* cadd = ptx_alloc<uint8_t>(bp, Kc + 1);
    if (bp.overflow || n > 32766u || N > 65534u || Kid > 65535u) {
        PTX_LEADER {
            ptx_patch_log pl;
            pl.status = PTX_ERR_CAPACITY;
            pl.n_patches = 0;
            A.plogs[log] = pl;
        }
        return;
    }

    /* ---- set-up (the rows come resolved from the merge: no element index here) ---- */
    PTX_FOR(w, nwe) {
        PtxBitWord z;
        z.bits = 0;
        z.pre = 0;
        present[w] = z;
    }
    PTX_FOR(w, nws) {
        defined[w] = 0;
        anyc[w] = 0;
        won[0][w] = 0;
        won[1][w] = 0;
        won[2][w] = 0;
    }
    PTX_FOR(c, Kid + 1) ctail[c] = PTX_SLOT_NONE;
    PTX_LEADER {
        H->npatch = 0;
        H->ncom = 0;
        H->tmp = 0;
        H->nvis = 0;
    }
    PTX_SYNC_T();

    /* last defined slot strictly below `lim` -> out_ = slot + 1 (0 = none), the same in every lane; every thread calls it.  The workgroup is ONE
     * wave: every lane keeps the best of its own words and a wave-wide maximum (register shuffles) makes it common — no LDS atomic, no read back */
#define PTX_LAST_DEFINED_BELOW(lim_, out_) const uint32_t out_ = ptx_last_set_below(defined, (lim_));

    /* make slot s_ a defined one: its state is that of the closest defined slot to the left (peritext.ts:176).  The leader reads the three winners before it
     * stores any (one round trip when they are global). */
#define PTX_COPY_SLOT_STATE(s_, l1_, v0_, v1_, v2_)                                               \
    do {                                                                                        \
        PTX_WIN_ST(&win[0][s_], v0_);                                                           \
        PTX_WIN_ST(&win[1][s_], v1_);                                                           \
        PTX_WIN_ST(&win[2][s_], v2_);                                                           \
        if ((l1_) && ptx_bittest(anyc, (l1_)-1u)) anyc[(s_) >> 5] |= 1u << ((s_)&31u);          \
        for (int ty_ = 0; ty_ < 3; ++ty_)                                                       \
            if ((l1_) && ptx_bittest(won[ty_], (l1_)-1u)) won[ty_][(s_) >> 5] |= 1u << ((s_)&31u); \
        defined[(s_) >> 5] |= 1u << ((s_)&31u);                                                 \
    } while (0)
#define PTX_DEFINE_SLOT(s_)                                                                     \
    do {                                                                                        \
        if (!ptx_bittest(defined, (s_))) {                                                      \
            PTX_LAST_DEFINED_BELOW(s_, l1_)                                                     \
            PTX_WIN_FENCE(); /* (global winners) the stores of the ops before have landed: the wait stands at the reader, where it is usually over */ \
            PTX_LEADER {                                                                        \
                const uint16_t v0_ = l1_ ? PTX_WIN_LD(&win[0][l1_ - 1u]) : (uint16_t)0;         \
                const uint16_t v1_ = l1_ ? PTX_WIN_LD(&win[1][l1_ - 1u]) : (uint16_t)0;         \
                const uint16_t v2_ = l1_ ? PTX_WIN_LD(&win[2][l1_ - 1u]) : (uint16_t)0;         \
                PTX_COPY_SLOT_STATE(s_, l1_, v0_, v1_, v2_);                                    \
            }                                                                                   \
            PTX_SYNC_T();                                                                       \
        }                                                                                       \
    } while (0)
    /* ---- the replay: PTX_RCHUNK rows are resolved in parallel, then applied one at a time ---- */
#pragma nounroll
    for (uint32_t t0 = 0; t0 < N; t0 += PTX_RCHUNK) {
    const uint32_t chunk_n = N - t0 < PTX_RCHUNK ? N - t0 : PTX_RCHUNK;
    PTX_FOR(i, chunk_n) {
        const uint32_t tt = t0 + i, a_ = action[tt];
        uint32_t kind = PTX_RK_SKIP, va = PTX_SLOT_NONE, vb = PTX_SLOT_NONE;
        if (a_ == PTX_ACT_MAKELIST) {
            kind = PTX_RK_MAKELIST;
        } else if (a_ == PTX_ACT_INSERT) {
            kind = PTX_RK_INSERT;
            va = erank[tt] & PTX_RANK_MASK; /* final rank of the element */
        } else if (a_ == PTX_ACT_DELETE) {
            const uint32_t r = refs[tt]; /* the row that inserted the target (always one, in a log the merge accepted) */
            if (r < N) {
                kind = PTX_RK_DELETE;
                va = erank[r] & PTX_RANK_MASK;
            }
        } else if ((a_ == PTX_ACT_ADDMARK || a_ == PTX_ACT_REMOVEMARK) && mark_type[tt] < 4u) {
            /* boundary slots as the walk of peritext.ts:167-214 meets them: resolved by merge_core.h P5a */
            const uint32_t v = refs[tt];
            va = v & 0xFFFFu;
            vb = v >> 16;
            kind = PTX_RK_MARK | ((uint32_t)mark_type[tt] << 4) | (a_ == PTX_ACT_ADDMARK ? 64u : 0u);
        }
        c_kind[i] = (uint8_t)kind;
        c_a[i] = (uint16_t)va;
        c_b[i] = (uint16_t)vb;
        c_pay[i] = payload[tt];
        c_id[i] = op_id[tt];
        c_key[i] = (uint16_t)(key_mode ? (uint32_t)(op_id[tt] >> 32) * na1 + (uint32_t)op_id[tt] + 1u : 0u);
    }
    PTX_SYNC_T();
#pragma nounroll
    for (uint32_t ci = 0; ci < chunk_n; ++ci) {
        const uint32_t t = t0 + ci;
        const uint32_t pcap = t >= first ? pcap_all : 0u; /* the rows before `first` count records (npatch) but write none ... */
        if (t == first && first != 0u) {                  /* ... and the count starts again at the first row asked for */
            PTX_LEADER { H->npatch = 0; }
            PTX_SYNC_T();
        }
        const uint32_t kind = c_kind[ci] & 15u;
        if (kind == PTX_RK_MAKELIST) {
            PTX_LEADER {
                ptx_patch_put(A, pbase, pcap, H->npatch, t, PTX_PATCH_MAKELIST, 0u, 0u);
                H->npatch += 1;
            }
            PTX_SYNC_T();
        } else if (kind == PTX_RK_INSERT) {
            const uint32_t r = c_a[ci];
            PTX_LAST_DEFINED_BELOW(2u * r, l1) /* slot + 1 */
            const uint32_t p0 = H->npatch;
            PTX_WIN_FENCE();
            PTX_SYNC_T();
            PTX_LEADER {
                uint32_t attr = 0;
                if (l1) {
                    const uint32_t l = l1 - 1u;
                    if (ptx_bittest(won[0], l)) attr |= PTX_ATTR_STRONG;
                    if (ptx_bittest(won[1], l)) attr |= PTX_ATTR_EM;
                    if (ptx_bittest(won[2], l)) attr |= PTX_ATTR_LINK | (payload[PTX_WIN_LD(&win[2][l]) - 1u] & PTX_ATTR_ID_MASK);
                    if (ptx_bittest(anyc, l)) attr |= PTX_ATTR_COMMENT;
                }
                ptx_patch_put(A, pbase, pcap, p0, t, PTX_PATCH_INSERT, ptx_bitrank(present, r), attr);
                H->tmp = 0; /* comment ids of this patch */
            }
            PTX_SYNC_T();
            if (l1 && ptx_bittest(anyc, l1 - 1u)) {
                const uint32_t l = l1 - 1u, nc = H->ncom;
                PTX_FOR(kc, nc) {
                    if (cadd[kc] && ca[kc] <= l && l < cb[kc]) {
                        bool last = true; /* no later-applied covering op of the same id: the chain of the id, latest first, down to this op */
                        for (uint32_t y = ctail[ccid[kc]]; y != kc && y != PTX_SLOT_NONE; y = cprev[y])
                            if (ca[y] <= l && l < cb[y]) {
                                last = false;
                                break;
                            }
                        if (last) ptx_patch_put(A, pbase, pcap, p0 + 1u + ptx_atomic_add(&H->tmp, 1u), t, PTX_PATCH_INSERT_COMMENT, ccid[kc], 0u);
                    }
                }
                PTX_SYNC_T();
            }
            /* the element is visible from now on */
            PTX_FOR(w, nwe) {
                if (w == (r >> 5)) present[w].bits |= 1u << (r & 31);
                else if (w > (r >> 5)) present[w].pre += 1;
            }
            PTX_LEADER {
                H->npatch = p0 + 1u + H->tmp;
                H->nvis += 1;
            }
            PTX_SYNC_T();
        } else if (kind == PTX_RK_DELETE) {
            const uint32_t r = c_a[ci];
            const bool was = (present[r >> 5].bits >> (r & 31)) & 1u;
            PTX_SYNC_T();
            if (was) {
                PTX_LEADER {
                    ptx_patch_put(A, pbase, pcap, H->npatch, t, PTX_PATCH_DELETE, ptx_bitrank(present, r), 1u);
                    H->npatch += 1;
                    H->nvis -= 1;
                }
                PTX_SYNC_T();
                PTX_FOR(w, nwe) {
                    if (w == (r >> 5)) present[w].bits &= ~(1u << (r & 31));
                    else if (w > (r >> 5)) present[w].pre -= 1;
                }
                PTX_SYNC_T();
            }
        } else if (kind == PTX_RK_MARK) {
            const uint32_t ty = (c_kind[ci] >> 4) & 3u, act = (c_kind[ci] & 64u) ? (uint32_t)PTX_ACT_ADDMARK : (uint32_t)PTX_ACT_REMOVEMARK;
            uint32_t slot_a = c_a[ci], slot_b = c_b[ci];
            if (slot_a != PTX_SLOT_NONE && slot_b == slot_a) slot_b = PTX_SLOT_NONE; /* the start test fires first (A.6-3) */
            if (slot_a == PTX_SLOT_NONE || slot_b < slot_a) {
                /* the end is met while the op has not started: its slot becomes a defined one (a copy of the state to
                 * its left), the op covers nothing and the walk stops (peritext.ts:240-243) */
                if (slot_b != PTX_SLOT_NONE) PTX_DEFINE_SLOT(slot_b);
                continue;
            }
            PTX_DEFINE_SLOT(slot_a);
            if (slot_b != PTX_SLOT_NONE) PTX_DEFINE_SLOT(slot_b); /* inherits the state BEFORE this op from inside the range (reading both ends' sources in one
                                                                   * round trip was measured: 4 % slower) */
            /* the defined slots of [slot_a, lim), ascending */
            const uint32_t lim = slot_b != PTX_SLOT_NONE ? slot_b : 2u * n;
            const uint32_t wlo = slot_a >> 5, whi = (lim + 31u) >> 5;
#define PTX_RANGE_BITS(w_, m_)                                                       \
    uint32_t m_ = defined[w_];                                                       \
    if ((w_) == wlo) m_ &= ~((1u << (slot_a & 31u)) - 1u);                           \
    if (((w_) << 5) + 32u > lim) m_ &= (lim & 31u) ? (1u << (lim & 31u)) - 1u : 0u;
            PTX_FOR(wi, whi - wlo + 1u) {
                const uint32_t w = wlo + wi;
                uint32_t c = 0;
                if (w < whi) {
                    PTX_RANGE_BITS(w, m)
                    c = ptx_popc(m);
                }
                wcnt[wi] = c;
            }
            PTX_SYNC_T();
            const uint32_t S = ptx_scan_excl<uint32_t, 1, kThreads>(wcnt, whi - wlo + 1u, H->scan_tmp);
            PTX_FOR(wi, whi - wlo) {
                const uint32_t w = wlo + wi;
                PTX_RANGE_BITS(w, m)
                uint32_t o = wcnt[wi];
                while (m) {
                    const uint32_t b = (uint32_t)__builtin_ctz(m);
                    m &= m - 1u;
                    if (o < segl) seg_l[o] = (uint16_t)((w << 5) + b);
                    else if (o < segcap) PTX_WIN_ST(&seg[o], (w << 5) + b);
                    ++o;
                }
            }
#undef PTX_RANGE_BITS
            /* (global winners) the slot list's tail is read by other lanes than the ones that filled it, and the per-slot loop below loads winners that an
             * EARLIER op's other lanes stored: one wait for the wave's outstanding stores orders both (without it the second relied on same-wave
             * store -> load ordering through the L1 alone) */
            PTX_WIN_FENCE();
            PTX_SYNC_T();
            const uint32_t nvis = H->nvis, nc = H->ncom;
            const uint32_t cfw = (S >> 5) + 1u; /* words of the patch-opening bitmap (+1 for the total) */
            PTX_FOR(w, cfw + 1u) {
                PtxBitWord z;
                z.bits = 0;
                z.pre = 0;
                cf[w] = z;
            }
            /* a changed slot opens a patch that the next defined slot (or the end of the range / text) closes;
             * zero-width ones are dropped (peritext.ts:269-281) */
            const uint32_t v_end = slot_b != PTX_SLOT_NONE ? ptx_bitrank(present, (slot_b + 1u) >> 1) : nvis;
#define PTX_VIS_AT(s_) ptx_bitrank(present, ((uint32_t)(s_) + 1u) >> 1) /* visible index at a boundary slot */
            PTX_SYNC_T();
            const uint32_t my_id = c_pay[ci];
            const uint64_t my_op = c_id[ci];
            const uint32_t my_key1 = c_key[ci];
            const bool by_key = key_mode && ty != PTX_MARK_LINK;
            /* per defined slot: did the effective marks change (peritext.ts:208), new state, visible index */
            PTX_FOR(j, S) {
                const uint32_t s = PTX_SEG_LD(j);
                bool changed = false;
                if (ty != PTX_MARK_COMMENT) {
                    uint16_t* wt = win[ty == PTX_MARK_STRONG ? 0 : ty == PTX_MARK_EM ? 1 : 2];
                    uint32_t* wo = won[ty == PTX_MARK_STRONG ? 0 : ty == PTX_MARK_EM ? 1 : 2];
                    const uint32_t w = PTX_WIN_LD(&wt[s]);
                    const bool old_on = ptx_bittest(wo, s);
                    /* compareOpIds: counter, then actor (ranks keep the string order); the dense keys keep that order */
                    const bool wins = !w || (by_key ? my_key1 > w : my_op > op_id[w - 1u]);
                    if (wins) {
                        const bool new_on = act == PTX_ACT_ADDMARK;
                        changed = new_on != old_on;
                        if (new_on && old_on && ty == PTX_MARK_LINK) changed = (my_id & PTX_ATTR_ID_MASK) != (payload[w - 1u] & PTX_ATTR_ID_MASK);
                        PTX_WIN_ST(&wt[s], by_key ? my_key1 : t + 1u);
                        if (new_on != old_on) {
                            if (new_on) ptx_atomic_or(&wo[s >> 5], 1u << (s & 31));
                            else ptx_atomic_and(&wo[s >> 5], ~(1u << (s & 31)));
                        }
                    }
                } else {
                    /* the last-applied covering op with this id (this op is not registered yet) */
                    int state = -1; /* -1 none, 0 removed, 1 present */
                    for (uint32_t y = my_id < Kid ? ctail[my_id] : PTX_SLOT_NONE; y != PTX_SLOT_NONE; y = cprev[y])
                        if (ca[y] <= s && s < cb[y]) {
                            state = cadd[y] ? 1 : 0;
                            break;
                        }
                    const bool any = ptx_bittest(anyc, s);
                    changed = act == PTX_ACT_ADDMARK ? state != 1 : (state == 1 || !any); /* remove on no comment key: undefined -> [] */
                }
                if (changed) {
                    const uint32_t ve = j + 1u < S ? PTX_VIS_AT(PTX_SEG_LD(j + 1u)) : v_end;
                    if (ve > PTX_VIS_AT(s)) ptx_atomic_or(&cf[j >> 5].bits, 1u << (j & 31));
                }
            }
            PTX_SYNC_T(); /* (global winners: the ops that read what was stored here wait for it themselves) */
            if (ty == PTX_MARK_COMMENT) {
                PTX_FOR(j, S) {
                    const uint32_t s = PTX_SEG_LD(j);
                    ptx_atomic_or(&anyc[s >> 5], 1u << (s & 31));
                }
            }
            PTX_FOR(w, cfw) cf[w].pre = ptx_popc(cf[w].bits);
            PTX_SYNC_T();
            const uint32_t P = ptx_scan_excl<uint32_t, 2, kThreads>(&cf[0].pre, cfw + 1u, H->scan_tmp);
            const uint32_t p0 = H->npatch;
            PTX_FOR(j, S) {
                if ((cf[j >> 5].bits >> (j & 31)) & 1u) {
                    const uint32_t ve = j + 1u < S ? PTX_VIS_AT(PTX_SEG_LD(j + 1u)) : v_end;
                    ptx_patch_put(A, pbase, pcap, p0 + ptx_bitrank(cf, j), t, act == PTX_ACT_ADDMARK ? PTX_PATCH_ADDMARK : PTX_PATCH_REMOVEMARK, PTX_VIS_AT(PTX_SEG_LD(j)), ve);
                }
            }
#undef PTX_VIS_AT
            PTX_SYNC_T();
            PTX_LEADER {
                H->npatch = p0 + P;
                if (ty == PTX_MARK_COMMENT && my_id < Kid && nc < Kc) {
                    ca[nc] = (uint16_t)slot_a;
                    cb[nc] = (uint16_t)slot_b;
                    ccid[nc] = (uint16_t)my_id;
                    cadd[nc] = act == PTX_ACT_ADDMARK ? 1 : 0;
                    cprev[nc] = ctail[my_id];
                    ctail[my_id] = (uint16_t)nc;
                    H->ncom = nc + 1u;
                }
            }
            PTX_SYNC_T();
        }
    }
    PTX_SYNC_T(); /* the chunk buffers are rewritten next */
    }
#undef PTX_LAST_DEFINED_BELOW
#undef PTX_DEFINE_SLOT
#undef PTX_COPY_SLOT_STATE
#undef PTX_WIN_LD
#undef PTX_WIN_ST
#undef PTX_WIN_FENCE
#undef PTX_SEG_LD
    PTX_SYNC_T();
    PTX_LEADER {
        ptx_patch_log pl;
        const uint32_t produced = first < N ? H->npatch : 0u; /* (first >= N: nothing was asked for) */
        pl.status = produced > pcap_all ? (uint32_t)PTX_ERR_CAPACITY : (uint32_t)PTX_OK;
        pl.n_patches = produced;
        A.plogs[log] = pl;
    }
}
