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
 * replayed in time order over BITMAPS indexed by final rank / final boundary slot (slot = 2*rank + side); a mark op is a
 * handful of passes with one 32-slot word per lane (round 4; before, one defined slot per lane out of a compacted list,
 * with per-slot winner arrays):
 *   present   bit per rank  {bits, running popcount prefix}: inserted and not deleted -> visible index = popcount below
 *   defined   bit per slot : the reference's `markOpsBefore/After !== undefined` (peritext.ts:167-214): set at the start
 *             and end slot of every applied mark op; patches break at every defined slot inside the op's range
 *   on[3]     bit per defined slot and LWW type (strong, em, link): the max-opId op that covers the slot is an addMark
 *             (opsToMarks, :304-313).  A slot defined later copies the closest defined slot to its left (:176) — which is
 *             the same as saying that an op covers the slots of its interval [start, end) for good.  So "does this op win
 *             at slot s" (compareOpIds against the slot's winner) needs no per-slot winner: it LOSES exactly at the slots
 *             covered by an earlier-applied op of its type with a larger opId.  Those are rare (an op that arrives after
 *             a concurrent one with a larger id): the op whose id exceeds every applied id of its type (kept in a
 *             register) wins everywhere; the others scan the table of applied ops of the type (row, start, end) once
 *             and OR the intervals of the larger ones into a mask.
 *   url       per slot, links only: the url of the winner where the link is on (an addMark of a link over a linked
 *             slot changes it iff the urls differ, :208; the marks of an inserted char name it)
 *   anyc      bit per slot : some comment op covers (the `comment: []` state)
 *   comment ops: [start, end) slots + per-id chains in application order (the LAST-applied covering op of an id
 *             decides its presence, :314-321): per word, the chain of the op's id is walked latest first over masks
 * A changed slot opens a patch that the next defined slot closes; zero-width ones are dropped (:269-281): per word, the
 * visible chars are spread to their "before" slots and one reversed addition hands every one of them to the defined slot
 * that governs it, so only the records that will be written are counted, ranked (one scan) and written.
 * One 64-thread workgroup (one wave) per log: the replay is sequential in t.  As ONE wave the phases need no s_barrier and
 * no wait for the patches just stored to HBM (nothing this kernel writes there is read back): PTX_SYNC_T is a compiler fence then.
 *
 * Compiled two ways like merge_core.h (hipcc: the product kernel; g++ -DPTX_EMU: CPU test tooling only).
 */
#pragma once
#include "merge_core.h"

#define PTX_SLOT_NONE 0xFFFFu /* start never matches / end never reached (the 16-bit form: logs of up to 32 766 list elements; the wide build: 0xFFFFFFFF) */
#define PTX_CHAIN_NONE 0xFFFFu /* end of a chain of comment ops (indices of comment ops stay 16 bits wide in both builds) */
#define PTX_RCHUNK 32u /* rows resolved together (their LDS buffers: 17 bytes per row) */
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
    const uint32_t* refs_hi;    /* (the wide build) PtxMergeArgs.out_refs_hi: the high halves of the boundary slots of a log of more than 32 766 list elements */
    const uint64_t* patch_off;  /* [n_logs + 1] capacity offsets into `patches` */
    ptx_patch* patches;
    ptx_patch_log* plogs;
    uint32_t n_logs;
    uint32_t lds_bytes;
    const uint32_t* first_row; /* optional [n_logs]: only the records of the rows from here on are produced (the rows before are replayed for their state alone) */
    uint16_t* win_scratch; /* optional: the per-slot link urls, the tables of applied mark ops and the comment ops' id tables of every log live HERE instead of in LDS ... */
    const uint64_t* win_off; /* ... [n_logs + 1], in u16 units: the slice of log l (ptx_replay_win_units of its header) */
    /* optional: overflow extents.  A log whose records outgrow its capacity [patch_off[l], patch_off[l + 1]) takes an extent of `patches` from the arena
     * [arena_base, arena_base + arena_cap) (an atomic bump of *arena_next) and goes on writing there — sized from its own record rate so far; should that run
     * out too, a second, much larger one.  ext_off[3 l ..] = {first extent, second extent (~0: none), records the first holds}.  The host packs the parts. */
    unsigned long long* arena_next;
    uint64_t arena_base, arena_cap;
    uint64_t* ext_off; /* [3 * n_logs] */
};

/* the mark state of 32 slots, one 16-byte LDS access: "some comment op covers" and, per LWW type (strong, em, link), "the winner is an addMark" */
struct PtxMarkBits {
    uint32_t ac, on[3];
};

/* Round 6: the WIDE build (kWide) for logs of more than 32 766 list elements or 65 534 rows (merged by the HBM-staged kernel): ranks and boundary slots are 32 bits
 * wide (a chunk row is 24 bytes, an entry of the tables of applied LWW ops two 8-byte words, the comment ops' intervals 32-bit pairs); everything else is the same
 * text.  What bounds such a log is the LDS its bitmaps take — 8 bytes per 32 elements + 28 per 32 slots: about 70 000 elements. */
template <bool kWide> struct PtxSlotT { typedef uint16_t type; };
template <> struct PtxSlotT<true> { typedef uint32_t type; };
/* one of the next PTX_RCHUNK rows, resolved (one 16-byte LDS access when its turn comes; 24 bytes in the wide build) */
template <bool kWide> struct PtxChunkRowT {
    uint64_t id;   /* the op id (mark ops: compareOpIds) */
    uint32_t pay;  /* payload: url / comment id */
    typename PtxSlotT<kWide>::type a, b; /* insert / delete: final rank; mark: start slot, end slot */
};

struct PtxReplayHdr {
    uint32_t tmp;      /* per-step scratch: counter */
    uint32_t ext_ok;   /* the extent asked for was granted */
    uint32_t ext_cap[2]; /* the log's overflow extents: records each holds (0: none) ... */
    uint32_t ext_lo[2], ext_hi[2]; /* ... and where they start in `patches` */
    uint32_t scan_tmp[4]; /* (ptx_scan_excl's scratch: a one-wave workgroup takes the total from a lane, no trip through the LDS — the kernel is only built for 64 threads) */
};

/* gscratch: the per-slot link urls and the op tables live in global memory (PtxReplayArgs.win_scratch) */
PTX_HD uint64_t ptx_replay_lds_need(uint64_t n, uint64_t K, uint64_t Kc, uint64_t ks, uint64_t Kid, bool gscratch = false, bool wide = false) {
    const uint64_t nwe = (n >> 5) + 2, nws = ((2 * n + 2) >> 5) + 2, Kl = K - Kc, sw = wide ? 4 : 2;
    (void)ks;
    return ptx_a16(sizeof(PtxReplayHdr)) + ptx_a16(8 * nwe) + ptx_a16(4 * nws) + ptx_a16(16 * nws) + 2 * ptx_a16(4 * (nws + 1)) +
           (gscratch ? 0 : ptx_a16(4 * (2 * n + 2)) + ptx_a16((wide ? 16 : 8) * (Kl + 1)) + ptx_a16(2 * (Kc + 1)) + ptx_a16(2 * (Kid + 1))) +
           ptx_a16((wide ? 24 : 16) * PTX_RCHUNK) + ptx_a16(PTX_RCHUNK) +
           2 * ptx_a16(sw * (Kc + 1)) + ptx_a16(2 * (Kc + 1)) + ptx_a16(4 * ((Kc >> 5) + 1));
}
PTX_HD uint64_t ptx_replay_lds_need_hdr(const ptx_log_hdr& h, bool gscratch = false, bool wide = false) {
    const uint64_t K = (uint64_t)h.n_mark[0] + h.n_mark[1] + h.n_mark[2] + h.n_mark[3];
    return ptx_replay_lds_need(h.n_ins, K, h.n_mark[PTX_MARK_COMMENT], 0, h.n_mark[PTX_MARK_COMMENT] ? h.n_comment_ids : 0u, gscratch, wide);
}
/* u16 units of win_scratch a log takes: per-slot urls (4 bytes x (2 n + 2)), the table of the applied LWW mark ops (8 bytes each), the comment ops' ids and the
 * last op per comment id; every array 16-byte aligned */
PTX_HD uint64_t ptx_replay_win_units(uint64_t n, uint64_t K, uint64_t Kc, uint64_t Kid, bool wide = false) {
    const uint64_t Kl = K - Kc;
    return ((2 * (2 * n + 2) + 7) & ~7ull) + (wide ? 8 : 4) * ((Kl + 1 + 7) & ~7ull) + ((Kc + 1 + 7) & ~7ull) + ((Kid + 1 + 7) & ~7ull);
}
PTX_HD uint64_t ptx_replay_win_units_hdr(const ptx_log_hdr& h, bool wide = false) {
    const uint64_t K = (uint64_t)h.n_mark[0] + h.n_mark[1] + h.n_mark[2] + h.n_mark[3];
    return ptx_replay_win_units(h.n_ins, K, h.n_mark[PTX_MARK_COMMENT], h.n_mark[PTX_MARK_COMMENT] ? h.n_comment_ids : 0u, wide);
}
/* does the log need the wide build?  (ranks and slots 2 rank + side beyond 16 bits, or row numbers beyond them) */
PTX_HD bool ptx_replay_wants_wide(uint64_t N, const ptx_log_hdr& h) { return h.n_ins > 32766u || N > 65534u; }

/* where a log's records go: its own capacity first, then its overflow extent */
struct PtxPatchDst {
    ptx_patch* out;            /* the log's own capacity ... */
    uint32_t pcap;             /* ... in records */
    const PtxReplayHdr* H;     /* the overflow extents (the rare path reads them from the LDS) */
    ptx_patch* patches;
};
/* one patch record; rows past the capacity (and the extents) are counted, not written.  `open`: the row is one of those asked for (first_row) */
PTX_DEV void ptx_patch_put(const PtxPatchDst& d, bool open, uint32_t idx, uint32_t row, uint32_t kind, uint32_t a, uint32_t b) {
    if (!open) return;
#if defined(PTX_REPLAY_EXP) && (PTX_REPLAY_EXP & 2) /* timing experiment only: what the record stores cost */
    if (idx != 0xFFFFFFFFu) return;
#endif
    ptx_patch p;
    p.row = row;
    p.kind = kind;
    p.a = a;
    p.b = b;
    if (idx < d.pcap) {
        d.out[idx] = p;
    } else {
        uint32_t x = idx - d.pcap;
        for (int k = 0; k < 2; ++k) {
            const uint32_t c = d.H->ext_cap[k];
            if (x < c) {
                d.patches[(((uint64_t)d.H->ext_hi[k] << 32) | d.H->ext_lo[k]) + x] = p;
                break;
            }
            x -= c;
        }
    }
}

PTX_DEV uint32_t ptx_bits_from(uint32_t b) { return b >= 32u ? 0u : ~0u << b; }        /* bits [b, 32) */
PTX_DEV uint32_t ptx_bits_below(uint32_t b) { return b >= 32u ? ~0u : (1u << b) - 1u; } /* bits [0, b) */
/* the slots of [a, b) that fall into word w */
PTX_DEV uint32_t ptx_span_mask(uint32_t a, uint32_t b, uint32_t w) {
    const uint32_t lo = w << 5;
    if (b <= lo || a >= lo + 32u || a >= b) return 0u;
    return ptx_bits_from(a > lo ? a - lo : 0u) & ptx_bits_below(b - lo);
}
/* the same for a word w that the non-empty interval reaches: a >> 5 <= w < (b + 31) >> 5 (no range checks left) */
PTX_DEV uint32_t ptx_span_mask_in(uint32_t a, uint32_t b, uint32_t w) {
    const uint32_t lo = w << 5;
    const uint32_t from = a > lo ? a - lo : 0u, to = b - lo < 32u ? b - lo : 32u; /* 0 .. 31, 1 .. 32 */
    return (~0u << from) & (~0u >> (32u - to));
}
/* bit i of the low half -> bit 2 i */
PTX_DEV uint32_t ptx_spread16(uint32_t x) {
    x &= 0xFFFFu;
    x = (x | (x << 8)) & 0x00FF00FFu;
    x = (x | (x << 4)) & 0x0F0F0F0Fu;
    x = (x | (x << 2)) & 0x33333333u;
    x = (x | (x << 1)) & 0x55555555u;
    return x;
}

/* kGWin: the per-slot urls and the op tables live in global memory (A.win_scratch), read and written past the L1 (workgroup-scope relaxed atomics) with the
 * wave's outstanding stores waited for wherever one lane reads what another has written */
template <uint32_t kThreads, bool kGWin = false, bool kWide = false>
PTX_DEV void ptx_replay_log(const PtxReplayArgs& A, uint32_t log, uint8_t* lds) {
    typedef typename PtxSlotT<kWide>::type slot_t;
    typedef PtxChunkRowT<kWide> PtxChunkRow;
    const uint32_t SLOT_NONE = kWide ? 0xFFFFFFFFu : (uint32_t)PTX_SLOT_NONE;
    PtxReplayHdr* H = (PtxReplayHdr*)lds;
    const uint64_t base = A.log_off[log];
    const uint32_t N = (uint32_t)(A.log_off[log + 1] - base);
    PtxPatchDst dst;
    dst.out = A.patches + A.patch_off[log];
    dst.pcap = (uint32_t)(A.patch_off[log + 1] - A.patch_off[log]);
    dst.H = H;
    dst.patches = A.patches;
    uint32_t room = dst.pcap;                   /* records the log can hold: its capacity + its extents */
    uint32_t ext_left = A.arena_next ? 2u : 0u; /* extents it may still ask for */
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
            if (A.ext_off) A.ext_off[3 * (uint64_t)log] = A.ext_off[3 * (uint64_t)log + 1] = ~0ull;
        }
        return;
    }
    const ptx_log_hdr hd = A.log_hdr[log];
    const uint32_t n = hd.n_ins, Kc = hd.n_mark[PTX_MARK_COMMENT];
    const uint32_t Kid = Kc ? hd.n_comment_ids : 0u; /* id space of the document's comments as this log has seen it */
    const uint32_t K = hd.n_mark[0] + hd.n_mark[1] + hd.n_mark[2] + hd.n_mark[3];
    const uint32_t nwe = (n >> 5) + 2, nws = ((2 * n + 2) >> 5) + 2;
    /* tables of the applied LWW mark ops per type (0 strong, 1 em, 2 link), in application order */
    const uint32_t Kl = K - Kc;
    const uint32_t toff[3] = {0u, hd.n_mark[PTX_MARK_STRONG], hd.n_mark[PTX_MARK_STRONG] + hd.n_mark[PTX_MARK_EM]};

    PtxBump bp;
    bp.base = lds;
    bp.off = (uint32_t)ptx_a16(sizeof(PtxReplayHdr));
    bp.cap = A.lds_bytes;
    bp.high = bp.off;
    bp.overflow = false;
    PtxBitWord* present = ptx_alloc<PtxBitWord>(bp, nwe);
    uint32_t* defined = ptx_alloc<uint32_t>(bp, nws);
    PtxMarkBits* mb = ptx_alloc<PtxMarkBits>(bp, nws); /* per defined slot: covered by a comment op; the winner of the LWW type is an addMark */
    /* per word of a mark op's range: the changed slots -> the slots that open a record; their count -> its prefix */
    uint32_t* cw = ptx_alloc<uint32_t>(bp, nws + 1);
    uint32_t* cnt = ptx_alloc<uint32_t>(bp, nws + 1);
    uint32_t* lurl;
    uint64_t* tab;   /* applied LWW mark op: row | start slot << 16 | end of its interval << 32 (the wide build: two words, row | start slot << 32 and the end) */
    uint16_t* ccid;  /* comment op: its id */
    uint16_t* ctail; /* per comment id: the last registered op (the chain of the ops with one id starts here, latest first) */
    if (kGWin) {
        uint16_t* g = A.win_scratch + A.win_off[log];
        const uint32_t ucols = (2u * (2u * n + 2u) + 7u) & ~7u, tcols = (Kl + 1u + 7u) & ~7u;
        lurl = (uint32_t*)g;
        tab = (uint64_t*)(g + ucols);
        ccid = g + ucols + (kWide ? 8u : 4u) * tcols;
        ctail = ccid + ((Kc + 1u + 7u) & ~7u);
    } else {
        lurl = ptx_alloc<uint32_t>(bp, 2 * n + 2);
        tab = ptx_alloc<uint64_t>(bp, (kWide ? 2u : 1u) * (Kl + 1));
        ccid = ptx_alloc<uint16_t>(bp, Kc + 1);
        ctail = ptx_alloc<uint16_t>(bp, Kid + 1);
    }
#define PTX_G_LD16(p_) (kGWin ? ptx_coherent_load16(p_) : *(p_))
#define PTX_G_ST16(p_, v_) do { if (kGWin) ptx_coherent_store16((p_), (uint16_t)(v_)); else *(p_) = (uint16_t)(v_); } while (0)
#define PTX_G_LD32(p_) (kGWin ? ptx_coherent_load32(p_) : *(p_))
#define PTX_G_LD64(p_) (kGWin ? ptx_coherent_load64(p_) : *(p_))
#define PTX_G_ST64(p_, v_) do { if (kGWin) ptx_coherent_store64((p_), (uint64_t)(v_)); else *(p_) = (uint64_t)(v_); } while (0)
#define PTX_G_ST32(p_, v_) do { if (kGWin) ptx_coherent_store32((p_), (uint32_t)(v_)); else *(p_) = (uint32_t)(v_); } while (0)
#if defined(PTX_REPLAY_EXP) && (PTX_REPLAY_EXP & 1) /* timing experiment only (wrong results possible): what the waits for the wave's outstanding stores cost */
#define PTX_G_FENCE() do { } while (0)
#else
#define PTX_G_FENCE() do { if (kGWin) ptx_global_stores_done(); } while (0)
#endif
    /* the next PTX_RCHUNK rows, resolved in parallel (element lookups, boundary slots) before they are replayed in order */
    PtxChunkRow* c_row = ptx_alloc<PtxChunkRow>(bp, PTX_RCHUNK);
    uint8_t* c_kind = ptx_alloc<uint8_t>(bp, PTX_RCHUNK);  /* PTX_RK_* | mark type << 4 | addMark << 6 */
    slot_t* ca = ptx_alloc<slot_t>(bp, Kc + 1);           /* comment op: first covered slot */
    slot_t* cb = ptx_alloc<slot_t>(bp, Kc + 1);           /*             first slot not covered (SLOT_NONE = to the end) */
    uint16_t* cprev = ptx_alloc<uint16_t>(bp, Kc + 1);    /* chain of the ops with the same id, latest first from ctail[id] */
    uint32_t* cadd = ptx_alloc<uint32_t>(bp, (Kc >> 5) + 1); /* bit per comment op: it is an addMark */
    if (bp.overflow || (!kWide && (n > 32766u || N > 65534u)) || n > 0x03FFFFFFu || Kc > 65534u || Kid > 65535u) {
        PTX_LEADER {
            ptx_patch_log pl;
            pl.status = PTX_ERR_CAPACITY;
            pl.n_patches = 0;
            A.plogs[log] = pl;
            if (A.ext_off) A.ext_off[3 * (uint64_t)log] = A.ext_off[3 * (uint64_t)log + 1] = ~0ull;
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
        PtxMarkBits z;
        z.ac = z.on[0] = z.on[1] = z.on[2] = 0;
        mb[w] = z;
    }
    PTX_FOR(c, Kid + 1) PTX_G_ST16(&ctail[c], PTX_CHAIN_NONE);
    PTX_FOR(c, (Kc >> 5) + 1) cadd[c] = 0;
    PTX_LEADER {
        H->tmp = 0;
        H->ext_cap[0] = H->ext_cap[1] = 0;
    }
    PTX_SYNC_T();
    /* the wave's own counters: the same value in every lane */
    uint32_t npatch = 0, ncom = 0, nvis = 0;
    uint32_t ntab[3] = {0u, 0u, 0u};
    uint64_t maxop[3] = {0ull, 0ull, 0ull}; /* largest opId applied so far per LWW type */

#ifndef PTX_REPLAY_EXP
#define PTX_REPLAY_EXP 0 /* timing experiments only (wrong results): bits switch parts of the step off */
#endif
#define PTX_VIS_AT(s_) ptx_bitrank(present, ((uint32_t)(s_) + 1u) >> 1) /* visible index at a boundary slot */
    /* Room for the records, looked after once per chunk of rows (not per record): a log that may outgrow its room within the chunk — three times its record
     * rate so far, at least eight per row — asks for an extent that holds that rate for all the rows still to come; should that run out too, one sixteen times
     * as generous.  (A chunk that beats even that loses records: PTX_ERR_CAPACITY with the exact count, and the host launches again with exact capacities.) */
#define PTX_RESERVE(t_)                                                                                        \
    do {                                                                                                       \
        uint32_t rate_ = 3u * (npatch / ((t_) > first ? (t_) - first + 1u : 1u) + 1u);                         \
        rate_ = rate_ < 8u ? 8u : rate_;                                                                       \
        if (ext_left && (t_) + PTX_RCHUNK > first && npatch + rate_ * PTX_RCHUNK > room) {                     \
            rate_ *= ext_left == 2u ? 1u : 16u;                                                                \
            const uint64_t want64_ = (uint64_t)rate_ * (N - (t_)) + 1024u;                                     \
            const uint32_t want_ = want64_ < 0x7FFFFFFFull - room ? (uint32_t)want64_ : 0x7FFFFFFFu - room;    \
            PTX_LEADER {                                                                                       \
                const unsigned long long at_ = ptx_atomic_add64(A.arena_next, (unsigned long long)want_);      \
                const bool ok_ = at_ + want_ <= A.arena_cap;                                                   \
                H->ext_ok = ok_ ? 1u : 0u;                                                                     \
                if (!ok_) (void)ptx_atomic_add64(A.arena_next, 0ull - (unsigned long long)want_); /* hand it back: a smaller request of another log may still fit */ \
                if (ok_) {                                                                                     \
                    const uint32_t k_ = 2u - ext_left;                                                         \
                    H->ext_cap[k_] = want_;                                                                    \
                    H->ext_lo[k_] = (uint32_t)(A.arena_base + at_);                                            \
                    H->ext_hi[k_] = (uint32_t)((A.arena_base + at_) >> 32);                                    \
                }                                                                                              \
            }                                                                                                  \
            PTX_SYNC_T();                                                                                      \
            if (PTX_U32(H->ext_ok)) {                                                                          \
                room += want_;                                                                                 \
                ext_left -= 1u;                                                                                \
            } else {                                                                                           \
                ext_left = 0u; /* the arena is exhausted */                                                    \
            }                                                                                                  \
            PTX_SYNC_T();                                                                                      \
        }                                                                                                      \
    } while (0)
    /* make slot s_ a defined one: its state is that of the closest defined slot to the left (peritext.ts:176) */
#define PTX_DEFINE_SLOT(s_)                                                                     \
    do {                                                                                        \
        if (!((PTX_U32(defined[(s_) >> 5]) >> ((s_)&31u)) & 1u)) {                              \
            const uint32_t l1_ = (PTX_REPLAY_EXP & 8) ? 0u : PTX_U32(ptx_last_set_below(defined, (s_))); /* slot + 1, the same in every lane */ \
            PTX_LEADER {                                                                        \
                const uint32_t bit_ = 1u << ((s_)&31u), ws_ = (s_) >> 5;                        \
                if (l1_) { /* (one 16-byte read of the neighbour's word, one of the slot's own, one write) */ \
                    const uint32_t l_ = l1_ - 1u, lb_ = l_ & 31u;                               \
                    const PtxMarkBits src_ = mb[l_ >> 5];                                       \
                    PtxMarkBits own_ = (l_ >> 5) == ws_ ? src_ : mb[ws_];                       \
                    if ((src_.ac >> lb_) & 1u) own_.ac |= bit_;                                 \
                    if ((src_.on[0] >> lb_) & 1u) own_.on[0] |= bit_;                           \
                    if ((src_.on[1] >> lb_) & 1u) own_.on[1] |= bit_;                           \
                    if ((src_.on[2] >> lb_) & 1u) own_.on[2] |= bit_;                           \
                    mb[ws_] = own_;                                                             \
                    if ((src_.on[2] >> lb_) & 1u) {                                             \
                        PTX_G_FENCE(); /* (global urls) the stores of the ops before have landed */ \
                        PTX_G_ST32(&lurl[s_], PTX_G_LD32(&lurl[l_]));                           \
                    }                                                                           \
                }                                                                               \
                defined[ws_] |= bit_;                                                           \
            }                                                                                   \
            PTX_SYNC_T();                                                                       \
        }                                                                                       \
    } while (0)
    /* ---- the replay: PTX_RCHUNK rows are resolved in parallel, then applied one at a time ---- */
#pragma nounroll
    for (uint32_t t0 = 0; t0 < N; t0 += PTX_RCHUNK) {
    const uint32_t chunk_n = N - t0 < PTX_RCHUNK ? N - t0 : PTX_RCHUNK;
    PTX_RESERVE(t0);
    PTX_FOR(i, chunk_n) {
        const uint32_t tt = t0 + i, a_ = action[tt];
        uint32_t kind = PTX_RK_SKIP, va = SLOT_NONE, vb = SLOT_NONE;
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
            if (kWide) { /* a log of more than 32 766 elements carries the high halves beside (none: all ones in both); a shorter one of the same batch has none */
                if (n > 32766u) {
                    const uint32_t vh = A.refs_hi[base + tt];
                    va |= vh << 16;
                    vb |= vh & 0xFFFF0000u;
                } else {
                    va = va == 0xFFFFu ? SLOT_NONE : va;
                    vb = vb == 0xFFFFu ? SLOT_NONE : vb;
                }
            }
            kind = PTX_RK_MARK | ((uint32_t)mark_type[tt] << 4) | (a_ == PTX_ACT_ADDMARK ? 64u : 0u);
        }
        c_kind[i] = (uint8_t)kind;
        PtxChunkRow cr;
        cr.id = op_id[tt];
        cr.pay = payload[tt];
        cr.a = (slot_t)va;
        cr.b = (slot_t)vb;
        c_row[i] = cr;
    }
    PTX_SYNC_T();
#pragma nounroll
    for (uint32_t ci = 0; ci < chunk_n; ++ci) {
        const uint32_t t = t0 + ci;
        const bool open = t >= first;  /* the rows before `first` count records (npatch) but write none ... */
        if (t == first) npatch = 0u;   /* ... and the count starts again at the first row asked for */
        const uint32_t kindb = PTX_U32(c_kind[ci]); /* (one LDS address: the same in every lane) */
        uint32_t kind = kindb & 15u;
        if ((PTX_REPLAY_EXP & 256) && kind == PTX_RK_MARK) kind = PTX_RK_SKIP;
        if ((PTX_REPLAY_EXP & 512) && (kind == PTX_RK_INSERT || kind == PTX_RK_DELETE)) kind = PTX_RK_SKIP;
        if (kind == PTX_RK_MAKELIST) {
            PTX_LEADER { ptx_patch_put(dst, open, npatch, t, PTX_PATCH_MAKELIST, 0u, 0u); }
            npatch += 1u;
        } else if (kind == PTX_RK_INSERT) {
            const uint32_t r = PTX_U32(c_row[ci].a);
            const uint32_t l1 = (PTX_REPLAY_EXP & 32) ? 0u : PTX_U32(ptx_last_set_below(defined, 2u * r)); /* slot + 1 */
            const uint32_t p0 = npatch;
            uint32_t attr = 0;
            bool coms = false;
            if (l1) { /* the marks of the closest defined slot to the left (every lane computes them: LDS broadcasts, one url load) */
                const uint32_t l = l1 - 1u;
                const uint32_t lw = l >> 5, lb = l & 31u;
                const PtxMarkBits st = mb[lw]; /* (one 16-byte read) */
                if ((PTX_U32(st.on[0]) >> lb) & 1u) attr |= PTX_ATTR_STRONG;
                if ((PTX_U32(st.on[1]) >> lb) & 1u) attr |= PTX_ATTR_EM;
                if ((PTX_U32(st.on[2]) >> lb) & 1u) {
                    PTX_G_FENCE();
                    attr |= PTX_ATTR_LINK | (PTX_U32(PTX_G_LD32(&lurl[l])) & PTX_ATTR_ID_MASK);
                }
                coms = (PTX_U32(st.ac) >> lb) & 1u;
                if (coms) attr |= PTX_ATTR_COMMENT;
            }
            PTX_LEADER { ptx_patch_put(dst, open, p0, t, PTX_PATCH_INSERT, ptx_bitrank(present, r), attr); }
            uint32_t extra = 0;
            if (coms) {
                const uint32_t l = l1 - 1u;
                PTX_G_FENCE();
                PTX_FOR(kc, ncom) {
                    if (ptx_bittest(cadd, kc) && ca[kc] <= l && l < cb[kc]) {
                        bool last = true; /* no later-applied covering op of the same id: the chain of the id, latest first, down to this op */
                        const uint32_t id = PTX_G_LD16(&ccid[kc]);
                        for (uint32_t y = PTX_G_LD16(&ctail[id]); y != kc && y != PTX_CHAIN_NONE; y = cprev[y])
                            if (ca[y] <= l && l < cb[y]) {
                                last = false;
                                break;
                            }
                        if (last) ptx_patch_put(dst, open, p0 + 1u + ptx_atomic_add(&H->tmp, 1u), t, PTX_PATCH_INSERT_COMMENT, id, 0u);
                    }
                }
                PTX_SYNC_T();
                extra = PTX_U32(H->tmp);
                PTX_SYNC_T();
                PTX_LEADER { H->tmp = 0; }
            }
            /* the element is visible from now on */
            if (!(PTX_REPLAY_EXP & 64)) PTX_FOR(w, nwe) {
                if (w == (r >> 5)) present[w].bits |= 1u << (r & 31);
                else if (w > (r >> 5)) present[w].pre += 1;
            }
            npatch = p0 + 1u + extra;
            nvis += 1u;
            PTX_SYNC_T();
        } else if (kind == PTX_RK_DELETE) {
            const uint32_t r = PTX_U32(c_row[ci].a);
            const bool was = (PTX_U32(present[r >> 5].bits) >> (r & 31)) & 1u;
            if (was) {
                    PTX_LEADER { ptx_patch_put(dst, open, npatch, t, PTX_PATCH_DELETE, ptx_bitrank(present, r), 1u); }
                npatch += 1u;
                nvis -= 1u;
                PTX_SYNC_T();
                if (!(PTX_REPLAY_EXP & 64)) PTX_FOR(w, nwe) {
                    if (w == (r >> 5)) present[w].bits &= ~(1u << (r & 31));
                    else if (w > (r >> 5)) present[w].pre -= 1;
                }
                PTX_SYNC_T();
            }
        } else if (kind == PTX_RK_MARK) {
            const uint32_t ty = (kindb >> 4) & 3u;
            const bool add = (kindb & 64u) != 0u;
            const PtxChunkRow cr = c_row[ci];
            uint32_t slot_a = PTX_U32(cr.a), slot_b = PTX_U32(cr.b);
            if (slot_a != SLOT_NONE && slot_b == slot_a) slot_b = SLOT_NONE; /* the start test fires first (A.6-3) */
            if (slot_a == SLOT_NONE || slot_b < slot_a) {
                /* the end is met while the op has not started: its slot becomes a defined one (a copy of the state to
                 * its left), the op covers nothing and the walk stops (peritext.ts:240-243) */
                if (slot_b != SLOT_NONE) PTX_DEFINE_SLOT(slot_b);
                continue;
            }
            PTX_DEFINE_SLOT(slot_a);
            if (slot_b != SLOT_NONE) PTX_DEFINE_SLOT(slot_b); /* inherits the state BEFORE this op from inside the range */
            /* the words of [slot_a, lim) */
            const uint32_t lim = slot_b != SLOT_NONE ? slot_b : 2u * n;
            const uint32_t wlo = slot_a >> 5, whi = (lim + 31u) >> 5, nw = whi > wlo && lim > slot_a && !(PTX_REPLAY_EXP & 128) ? whi - wlo : 0u;
            const uint32_t my_id = PTX_U32(cr.pay);
            const uint64_t my_op = ((uint64_t)PTX_U32((uint32_t)(cr.id >> 32)) << 32) | PTX_U32((uint32_t)cr.id);
            /* the defined slots of the range in word w_ */
#define PTX_RANGE_MASK(w_) (ptx_span_mask_in(slot_a, lim, (w_))) /* wlo <= w_ < whi */
            /* first defined slot of the range in the words after w_, else the end of the range */
#define PTX_NEXT_AFTER_WORD(w_, out_)                                  \
    uint32_t out_ = lim;                                               \
    for (uint32_t v_ = (w_) + 1u; v_ < whi; ++v_) {                    \
        const uint32_t mm_ = defined[v_] & PTX_RANGE_MASK(v_);         \
        if (mm_) {                                                     \
            out_ = (v_ << 5) + (uint32_t)__builtin_ctz(mm_);           \
            break;                                                     \
        }                                                              \
    }
            /* word wi_ = w_: of the changed slots ch_ (defined slots m_ of the range) keep those whose patch — up to the next defined slot, or the end of the
             * range — holds a visible char (peritext.ts:269-281): a char of rank r sits between slots 2 r and 2 r + 1, so it belongs to the patch of the last
             * defined slot at or below 2 r.  Chars spread to their even slots; a reversed addition carries every one of them down to the defined slot that
             * governs it; what follows the word belongs to its highest defined slot. */
#define PTX_FINISH_WORD(wi_, w_, m_, ch_)                                                                               \
    do {                                                                                                                \
        uint32_t R_ = 0;                                                                                                \
        if ((ch_) && !(PTX_REPLAY_EXP & 16)) {                                                                          \
            const uint32_t pw_ = present[(w_) >> 1].bits;                                                               \
            const uint32_t P2_ = ptx_spread16(((w_)&1u) ? pw_ >> 16 : pw_) & PTX_RANGE_MASK(w_);                        \
            uint32_t G_ = P2_ & (m_);                                                                                   \
            const uint32_t Q_ = P2_ & ~(m_);                                                                            \
            if (Q_) {                                                                                                   \
                const uint32_t Dr_ = ptx_brev(m_);                                                                      \
                G_ |= ptx_brev((~Dr_ + ptx_brev(Q_)) & Dr_);                                                            \
            }                                                                                                           \
            const uint32_t top_ = 31u - (uint32_t)__builtin_clz(m_);                                                    \
            if ((((ch_) & ~G_) >> top_) & 1u) {                                                                         \
                PTX_NEXT_AFTER_WORD(w_, nx_)                                                                            \
                if (nx_ > (((w_) + 1u) << 5) && PTX_VIS_AT(nx_) > ptx_bitrank(present, ((w_) + 1u) << 4)) G_ |= 1u << top_; \
            }                                                                                                           \
            R_ = (ch_) & G_;                                                                                            \
        }                                                                                                               \
        cw[wi_] = R_;                                                                                                   \
        cnt[wi_] = ptx_popc(R_);                                                                                        \
    } while (0)
            uint32_t y0 = PTX_CHAIN_NONE; /* (comments) the last registered op of this op's id */
            if (ty != PTX_MARK_COMMENT) {
                const uint32_t li = ty == PTX_MARK_STRONG ? 0u : ty == PTX_MARK_EM ? 1u : 2u;
                /* compareOpIds (counter, then actor: the op ids keep that order): this op loses at the slots an applied op of its type with a larger id covers */
                const bool fast = my_op > maxop[li];
                if (!fast) {
                    PTX_FOR(wi, nw) cw[wi] = 0u;
                    PTX_G_FENCE();
                    PTX_SYNC_T();
                    PTX_FOR(e, ntab[li]) {
                        const uint64_t ent = PTX_G_LD64(&tab[(kWide ? 2u : 1u) * (toff[li] + e)]);
                        if (op_id[kWide ? (uint32_t)ent : (uint32_t)ent & 0xFFFFu] > my_op) {
                            const uint32_t ya = kWide ? (uint32_t)(ent >> 32) : (uint32_t)(ent >> 16) & 0xFFFFu;
                            const uint32_t yl = kWide ? (uint32_t)PTX_G_LD64(&tab[2u * (toff[li] + e) + 1u]) : (uint32_t)(ent >> 32);
                            const uint32_t v0 = (ya >> 5) > wlo ? ya >> 5 : wlo, v1 = ((yl + 31u) >> 5) < whi ? (yl + 31u) >> 5 : whi;
                            for (uint32_t v = v0; v < v1; ++v) ptx_atomic_or(&cw[v - wlo], ptx_span_mask(ya, yl, v));
                        }
                    }
                    PTX_SYNC_T();
                }
                const bool per_slot = li == 2u && add; /* the url of every slot the op wins is stored; where the link was on, it decides "changed" */
                PTX_FOR(wi, nw) {
                    const uint32_t w = wlo + wi;
                    const uint32_t m = defined[w] & PTX_RANGE_MASK(w);
                    const uint32_t upd = fast ? m : m & ~cw[wi];
                    const uint32_t old = mb[w].on[li];
                    if (upd) mb[w].on[li] = add ? old | upd : old & ~upd;
                    if (per_slot) {
                        cw[wi] = upd & ~old; /* (the two together: the slots the op wins) */
                        cnt[wi] = upd & old;
                    } else {
                        const uint32_t ch = upd & (add ? ~old : old);
                        PTX_FINISH_WORD(wi, w, m, ch);
                    }
                }
                if (per_slot) {
                    PTX_G_FENCE();
                    PTX_SYNC_T();
                    const uint32_t my_url = my_id & PTX_ATTR_ID_MASK;
                    PTX_FOR(j, nw << 5) {
                        const uint32_t wi = j >> 5, bit = j & 31u;
                        const bool both = (cnt[wi] >> bit) & 1u; /* the link was on: changed iff the urls differ */
                        if (both || ((cw[wi] >> bit) & 1u)) { /* (a lane only ever sets ITS bit of cw, and only where `both`: the test reads what the pass before wrote) */
                            const uint32_t s = ((wlo + wi) << 5) + bit;
                            if (both && (PTX_G_LD32(&lurl[s]) & PTX_ATTR_ID_MASK) != my_url) ptx_atomic_or(&cw[wi], 1u << bit);
                            PTX_G_ST32(&lurl[s], my_id);
                        }
                    }
                    PTX_SYNC_T();
                    PTX_FOR(wi, nw) {
                        const uint32_t w = wlo + wi;
                        const uint32_t m = defined[w] & PTX_RANGE_MASK(w);
                        const uint32_t ch = cw[wi];
                        PTX_FINISH_WORD(wi, w, m, ch);
                    }
                }
                /* the op joins the table of its type */
                PTX_LEADER {
                    if (kWide) {
                        PTX_G_ST64(&tab[2u * (toff[li] + ntab[li])], (uint64_t)t | ((uint64_t)slot_a << 32));
                        PTX_G_ST64(&tab[2u * (toff[li] + ntab[li]) + 1u], (uint64_t)lim);
                    } else {
                        PTX_G_ST64(&tab[toff[li] + ntab[li]], (uint64_t)t | ((uint64_t)slot_a << 16) | ((uint64_t)lim << 32));
                    }
                }
                ntab[li] += 1u;
                if (fast) maxop[li] = my_op;
            } else {
                /* comments: the last-applied covering op with this id decides (this op is not registered yet): per word, the id's chain latest first */
                PTX_G_FENCE();
                y0 = my_id < Kid ? PTX_U32(PTX_G_LD16(&ctail[my_id])) : (uint32_t)PTX_CHAIN_NONE;
                PTX_FOR(wi, nw) {
                    const uint32_t w = wlo + wi;
                    const uint32_t m = defined[w] & PTX_RANGE_MASK(w);
                    uint32_t und = m, onm = 0;
                    for (uint32_t y = y0; y != PTX_CHAIN_NONE && und; y = cprev[y]) {
                        const uint32_t c = ptx_span_mask(ca[y], cb[y], w) & und;
                        if (ptx_bittest(cadd, y)) onm |= c;
                        und &= ~c;
                    }
                    const uint32_t any = mb[w].ac;
                    const uint32_t ch = add ? m & ~onm : m & (onm | ~any); /* remove on no comment key: undefined -> [] */
                    if (m) mb[w].ac = any | m;
                    PTX_FINISH_WORD(wi, w, m, ch);
                }
            }
            PTX_SYNC_T();
            const uint32_t P = ptx_scan_excl<uint32_t, 1, kThreads>(cnt, nw, H->scan_tmp);
            const uint32_t p0 = npatch;
            if (P) {
                PTX_FOR(wi, nw) {
                    uint32_t R = cw[wi];
                    if (R) {
                        const uint32_t w = wlo + wi;
                        const uint32_t m = defined[w] & PTX_RANGE_MASK(w);
                        uint32_t o = p0 + cnt[wi];
                        while (R) {
                            const uint32_t b = (uint32_t)__builtin_ctz(R);
                            R &= R - 1u;
                            const uint32_t above = m & ptx_bits_from(b + 1u);
                            uint32_t nxt;
                            if (above) {
                                nxt = (w << 5) + (uint32_t)__builtin_ctz(above);
                            } else {
                                PTX_NEXT_AFTER_WORD(w, nx)
                                nxt = nx;
                            }
                            ptx_patch_put(dst, open, o++, t, add ? PTX_PATCH_ADDMARK : PTX_PATCH_REMOVEMARK, PTX_VIS_AT((w << 5) + b), PTX_VIS_AT(nxt));
                        }
                    }
                }
            }
            npatch = p0 + P;
            if (ty == PTX_MARK_COMMENT && my_id < Kid && ncom < Kc) {
                PTX_LEADER {
                    ca[ncom] = (slot_t)slot_a;
                    cb[ncom] = (slot_t)slot_b;
                    PTX_G_ST16(&ccid[ncom], my_id);
                    if (add) cadd[ncom >> 5] |= 1u << (ncom & 31u);
                    cprev[ncom] = (uint16_t)y0;
                    PTX_G_ST16(&ctail[my_id], ncom);
                }
                ncom += 1u;
            }
            PTX_SYNC_T();
#undef PTX_FINISH_WORD
#undef PTX_NEXT_AFTER_WORD
#undef PTX_RANGE_MASK
        }
    }
    PTX_SYNC_T(); /* the chunk buffers are rewritten next */
    }
#undef PTX_DEFINE_SLOT
#undef PTX_RESERVE
#undef PTX_VIS_AT
#undef PTX_G_LD16
#undef PTX_G_ST16
#undef PTX_G_LD32
#undef PTX_G_LD64
#undef PTX_G_ST64
#undef PTX_G_ST32
#undef PTX_G_FENCE
    PTX_LEADER {
        ptx_patch_log pl;
        const uint32_t produced = first < N ? npatch : 0u; /* (first >= N: nothing was asked for) */
        pl.status = produced > room ? (uint32_t)PTX_ERR_CAPACITY : (uint32_t)PTX_OK;
        pl.n_patches = produced;
        A.plogs[log] = pl;
        if (A.ext_off) {
            A.ext_off[3 * (uint64_t)log] = H->ext_cap[0] ? ((uint64_t)H->ext_hi[0] << 32) | H->ext_lo[0] : ~0ull;
            A.ext_off[3 * (uint64_t)log + 1] = H->ext_cap[1] ? ((uint64_t)H->ext_hi[1] << 32) | H->ext_lo[1] : ~0ull;
            A.ext_off[3 * (uint64_t)log + 2] = H->ext_cap[0];
        }
    }
}
