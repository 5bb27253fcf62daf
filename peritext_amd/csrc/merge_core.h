/*
 * merge_core.h — the per-replica-log merge algorithm of the MI355X engine.
 *
 * One workgroup applies ONE replica op log and materialises its formatted document, entirely in
 * LDS: the op columns are read from HBM once (re-reads of a column are L2 hits), the outputs are
 * written once.
 *
 * It replaces, for one log, the reference's sequential
 *     for change of log: doc.applyChange(change)        reference/src/micromerge.ts:499-514
 *         applyListInsert  (:614-672)   applyListUpdate (:677-724)   findListElement (:731-755)
 *         applyAddRemoveMark            reference/src/peritext.ts:154-249
 *     doc.getTextWithFormatting(["text"])               reference/src/peritext.ts:337-395, opsToMarks :294-326
 * with the order-independent closed form of SURVEY.md Appendix A.3/A.5/A.7:
 *   P0  (batches with the Change envelope) applyChange's causal admission: seq == clock[actor] + 1 and
 *       deps <= clock for every change, the vector clock carried along the log as a wave prefix sum
 *   P1  ONE pass over the rows (sized by the per-log header, which it verifies): row lists per class in
 *       row order, and the element index: a bitmap over the insert ids (counter * actors + actor) +
 *       popcount prefix gives every list element a dense index e = its rank in compareOpIds order
 *       (micromerge.ts:812-827) and turns every elemId reference into one 8-byte LDS read
 *   P3  RGA causal tree: element order = pre-order DFS, children by DESCENDING opId (equivalent to
 *       the skip loop at micromerge.ts:630-635).  Children are bucketed per parent (counting sort)
 *       and ranked inside the bucket; the Euler tour of the tree is ranked work-efficiently (splitter
 *       walks + in-place pointer jumping over the splitters) -> document position of every element
 *   P4  tombstones: clear "alive" bits by document position, popcount prefix -> visible index
 *       (the `visible` counters of micromerge.ts:747-750)
 *   P5  marks: boundary slots 2*rank+side -> visible interval; per visible char the max-opId covering
 *       op per non-multi mark type (LWW, peritext.ts:304-313) through a range-chmax tree; comments
 *       (allowMultiple, :314-321) as per-id presence intervals decided by the LAST-APPLIED covering op
 *   P6  spans = maximal runs of visible chars with equal marks (peritext.ts:438-455) + 128-bit digest
 *
 * The code is written as phases of `PTX_FOR` (a parallel loop over the workgroup) separated by
 * `PTX_SYNC()`; no iteration reads what another iteration of the same phase writes except through
 * commutative atomics (or, in the pointer-jumping rounds, through single-word reads of a value whose
 * every intermediate state is valid).  That discipline lets the SAME source be compiled two ways:
 *   - by hipcc for gfx950 as the body of the kernel in peritext_hip.hip (the product), and
 *   - by g++ against the test-suite's own platform header as a single-threaded emulation used ONLY by the CPU test-suite
 *     (tests/emu) to check the kernel's logic where no GPU exists.  The emulation is not linked
 *     into libperitext_hip.so and is never a fallback for the product path.
 */
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <type_traits>
#include "../../include/peritext_hip.h"

#ifndef PTX_U64
#define PTX_U64 1
#endif
#ifndef PTX_U128
#define PTX_U128 2
#endif
#ifndef PTX_U
#define PTX_U (kThreads == 64u ? PTX_U64 : kThreads == 128u ? PTX_U128 : 2) /* rows in flight per thread in the batched loops (per build, like PTX_UV below) */
#endif

#define PTX_IN(i0, u) ((i0) + (uint32_t)(u) * _T < _n)
#define PTX_JSTEPS(n) PTX_JSTEPS_U(n, PTX_U)
#define PTX_J_OF(st, u) PTX_J_OF_U(st, u, PTX_U)
#ifndef PTX_U1_128
#define PTX_U1_128 4
#endif
#ifndef PTX_U1
#define PTX_U1 (kThreads == 128u ? PTX_U1_128 : 4) /* consecutive rows per thread and step in the row pass P1: 3 or 4.  Round 6: four (the ids as two 16-byte loads, the class bytes of the four rows one dword each) is six steps instead of eight for a 4 097-row log of three waves — a step of this pass is a trip to HBM however many rows it carries: -1 % on BASELINE config #4, -1.5 % on #3, same box (rounds 1-2 measured 4 slower: the scalar registers it spilled then are gone) */
#endif
#define PTX_MAX_THREADS 1024u
#define PTX_BYTE_PAD 4u /* bytes the library allocates past the end of the action / mark_type columns (the row pass reads them a dword at a time) */
#define PTX_UA 1 /* changes per thread in flight in the (rare) many-actor admission passes */
#define PTX_INA(i0, u) PTX_IN(i0, u)
#define PTX_IXA(i0, u) PTX_IX(i0, u)
#ifndef PTX_AC
#define PTX_AC 4u /* consecutive changes per lane and step in the admission pass (16-byte loads: one of headers and two of envelope rows per four changes) */
#endif
#ifndef PTX_UM
#define PTX_UM 1u /* mark ops per thread and step in P5a: five gathers per op, so one op in flight + one in work */
#endif
#ifndef PTX_UB64
#define PTX_UB64 1u
#endif
#ifndef PTX_UB128
#define PTX_UB128 2u
#endif
#ifndef PTX_UB
#define PTX_UB (kThreads == 64u ? PTX_UB64 : kThreads == 128u ? PTX_UB128 : 2u)
#endif
/* (PTX_UB: mark ops per thread and step in the LWW pass P5b — one opId gather each, issued together) */
#define PTX_NCLK 32 /* phase stamps of the diagnostic build (slot PTX_CLK_EXACT_WALKS counts the logs whose admission was walked twice) */
#define PTX_CLK_EXACT_WALKS 15
#ifndef PTX_KO_P5A_RA
#define PTX_KO_P5A_RA 0 /* knock-out (WRONG results, timing experiments only): the mark ops do not read ref_a a second time */
#endif
#ifndef PTX_UV64
#define PTX_UV64 1
#endif
#ifndef PTX_UV128
#define PTX_UV128 4
#endif
#ifndef PTX_UV
/* items per thread and step in the loops that are chains of dependent LDS reads per item (tree order, list ranking, unpark): the chains of a step run side by side.
 * Eight where a log is three waves or more; the one- and two-wave builds (logs of a few hundred rows, some eighty to three hundred elements) take fewer — an item
 * slot past the end of a short list still executes its instructions (round 6, same box: 4 instead of 8 is -8.2 % on BASELINE config #2, -6.5 % on #3, -0.2 % on #4) */
#define PTX_UV (kThreads == 64u ? PTX_UV64 : kThreads == 128u ? PTX_UV128 : 8)
#endif

/* the machine: gfx950.  (The CPU test-suite compiles these sources against a header of its own that plays the workgroup with one host
 * thread: its driver names that header in PTX_PLATFORM_HEADER before it includes this file; nothing in csrc/ knows where it lives.) */
#ifndef PTX_PLATFORM_HEADER
#define PTX_PLATFORM_HEADER "ptx_platform_gfx950.h"
#endif
#include PTX_PLATFORM_HEADER


/* kernel arguments: device pointers (host pointers in the test-suite's emulation) */
struct PtxMergeArgs {
    const uint64_t* log_off;
    const uint64_t* op_id;
    const uint64_t* ref_a;
    const uint64_t* ref_b;
    const uint32_t* payload;
    const uint8_t* action;
    const uint8_t* mark_type;
    const uint8_t* side_a;
    const uint8_t* side_b;
    /* causal envelope (optional: chg_off == nullptr skips causal admission) */
    const uint64_t* chg_off;
    const uint32_t* chg_hdr;  /* actor << 20 | nops */
    const uint16_t* chg_env;  /* rows of PTX_ENV_STRIDE(max_actors) u16: seq, deps[...] */
    const uint16_t* chg_env_hi; /* optional: the high halves of the same values (only the HBM-staged kernel reads them: the host routes every log that needs them there) */
    const ptx_log_hdr* log_hdr; /* [n_logs] per-log census (always present: the host computes it when the caller did not) */
    ptx_log_result* res;
    uint32_t* out_values;
    ptx_span* out_spans;
    ptx_cinterval* out_cints;
    uint32_t* out_rank;
    uint32_t* out_refs;  /* optional, with out_rank: per delete row the row of its target's insert, per mark row its boundary slots (start | end << 16,
                            0xFFFF = none) as the walk of peritext.ts:167-214 meets them — what the patch-stream replay (replay_core.h) resolves rows with */
    unsigned long long* clocks; /* optional [PTX_NCLK]: per-phase cycle totals (profiling builds of the host) */
    uint32_t n_logs;
    uint32_t lds_bytes;
    uint32_t max_actors;
    uint32_t stop_after; /* diagnostic: leave after the phase with this stamp index (0 = run everything) */
    uint32_t div_magic;  /* floor(2^32 / threads per workgroup) + 1, see PTX_DIV_T */
    const uint32_t* log_index; /* optional: workgroup i handles log log_index[i] (launches over a subset of the logs) */
    /* the HBM-staged path for logs beyond one CU's LDS (biglog_core.h): workgroup i works in big_scratch[big_off[i] .. big_off[i + 1]) */
    uint8_t* big_scratch;
    const uint64_t* big_off;
    uint32_t* grid_bar; /* a large log merged by the workgroups of ONE cooperative launch: the two words (arrivals, generation) of its grid barrier, zeroed by the host */
    uint32_t* out_refs_hi; /* optional, beside out_refs: the HIGH halves of the boundary slots of the mark rows (start >> 16 | end >> 16 << 16; 0xFFFF beside a low half of
                              0xFFFF = none) — written by the HBM-staged kernel alone, for logs of more than 32 766 list elements, whose slots 2 rank + side pass 16 bits
                              (round 6: the replay and change() on such logs) */
};

#define PTX_END 0xFFFFu
#ifndef PTX_S
#define PTX_S 16u /* every PTX_S-th node of the Euler tour is a splitter of the list ranking (measured: 16 is 1 % faster than 8 and needs 0.6 KB less, 4 is 7 % slower) */
#endif
#define PTX_TILE_4 128u  /* visible chars up to which the four LWW trees are resident at once */
#ifndef PTX_QUERY9
#define PTX_QUERY9 1 /* the short-document form reads a tree's nine nodes at once (not in the three-wave lean build: config #4 +0.2 %, its documents show a dozen characters) */
#endif
#define PTX_TILE_1 512u  /* tile of the visible axis for longer documents (one tree, reused per mark type) */
#define PTX_SMALL_BUCKET 8u  /* child buckets up to this size: one lane per member */
#define PTX_HUGE_BUCKET 256u /* beyond this size: bitmap ranking, one bucket at a time */

/* ---- digest: 128-bit multiset hash of the canonical output (restated in peritext_amd/canon.py) ----
 * Per item (tag, a, b, c) four 32-bit words, mixed by four add-rotate-xor quarter rounds (the ChaCha quarter round over the four words, the words' roles rotated
 * from round to round), summed modulo 2^64 as two 64-bit halves over the items of a log.  Round 6: ONLY full-rate 32-bit instructions — the digest of rounds
 * 1-5 (seven 64-bit multiplies per item, each four quarter-rate v_mul_lo/hi_u32) was 7 % of the time of a 256-op log and 3 % of a 1K-op one, whose builds the
 * counters show bound by vector-instruction issue (knock-out timing, `-DPTX_KO_DIGEST=1`).  The values of two builds with different digest functions are not
 * comparable: a digest is only ever compared with another digest of the same library (replicas of a document, ranks of a job). */
#ifndef PTX_KO_DIGEST
#define PTX_KO_DIGEST 0 /* knock-out (WRONG digests, timing experiments only): what the digest arithmetic costs */
#endif
PTX_DEV uint32_t ptx_rotl32(uint32_t x, uint32_t r) { return (x << r) | (x >> (32u - r)); } /* (v_alignbit_b32) */
#define PTX_DIGEST_QR(a_, b_, c_, d_) \
    a_ += b_; d_ ^= a_; d_ = ptx_rotl32(d_, 16u); c_ += d_; b_ ^= c_; b_ = ptx_rotl32(b_, 12u); a_ += b_; d_ ^= a_; d_ = ptx_rotl32(d_, 8u); c_ += d_; b_ ^= c_; b_ = ptx_rotl32(b_, 7u);
PTX_DEV void ptx_digest_item(uint64_t& h1, uint64_t& h2, uint32_t tag, uint32_t a, uint32_t b, uint32_t c) {
    if (PTX_KO_DIGEST) {
        h1 += tag + a;
        h2 += b + c;
        return;
    }
    uint32_t x0 = a ^ 0x9E3779B9u, x1 = b ^ 0x85EBCA6Bu, x2 = c ^ 0xC2B2AE35u, x3 = (tag | (tag << 16)) ^ 0x165667B1u;
    PTX_DIGEST_QR(x0, x1, x2, x3)
    PTX_DIGEST_QR(x1, x2, x3, x0)
    PTX_DIGEST_QR(x2, x3, x0, x1)
    PTX_DIGEST_QR(x3, x0, x1, x2)
    h1 += (uint64_t)x0 | ((uint64_t)x1 << 32);
    h2 += (uint64_t)x2 | ((uint64_t)x3 << 32);
}

/* ---- LDS header ---- */
struct PtxHdr {
    uint32_t err;          /* min over ((row*2+level) << 4 | code) of every detected op-level error; ~0 = none */
    uint32_t adm;          /* the same for the change-level (causal admission) errors of P0 */
    uint32_t max_ctr, max_actor;
    uint32_t cur_big, cur_med, cur_huge;
    uint32_t V, S, I;
    uint32_t n_ins, n_applied;
    uint32_t cur[8]; /* list cursors per row class (0 insert, 1 delete, 2..5 mark type 0..3, 6/7 unlisted rows) */
    unsigned long long h1, h2;
    uint32_t scan_tmp[36];
    unsigned long long clk[PTX_NCLK + 1];
};

/* LDS bytes of the header: the phase stamps are the diagnostic kernel's alone (its launch adds PTX_HDR_DIAG_EXTRA to the window) */
#define PTX_HDR_BYTES ((uint32_t)((offsetof(PtxHdr, clk) + 15u) & ~15u))
#define PTX_HDR_BYTES_DIAG ((uint32_t)((sizeof(PtxHdr) + 15u) & ~15u))
#define PTX_HDR_DIAG_EXTRA (PTX_HDR_BYTES_DIAG - PTX_HDR_BYTES)
#define PTX_NO_ERR 0xFFFFFFFFu
/* the reference throws at the FIRST failing op in application order: keep the minimum position.
 * level 0 = change-level check (seq / deps, micromerge.ts:501-509), 1 = op-level (:752) */
PTX_DEV void ptx_raise(PtxHdr* H, uint32_t row, uint32_t level, uint32_t code) {
    ptx_atomic_min(&H->err, ((row * 2u + level) << 4) | code);
}

/* a thread's share of the digest goes to the header as two 8-byte LDS atomics, issued by the threads that have one (in a sparse document a few dozen per
 * log: the wave-wide butterfly cost every wave ~70 vector instructions per flush, three flushes per log).  ptx_digest_flush_dense is the butterfly: for
 * the passes in which most lanes of a wave carry a share (same-address atomics are served one lane at a time). */
PTX_DEV void ptx_digest_flush(PtxHdr* H, uint64_t h1, uint64_t h2) {
    if ((h1 | h2) != 0) {
        ptx_atomic_add64(&H->h1, (unsigned long long)h1);
        ptx_atomic_add64(&H->h2, (unsigned long long)h2);
    }
}
PTX_DEV void ptx_digest_flush_dense(PtxHdr* H, uint64_t h1, uint64_t h2) {
    ptx_reduce_add64(&H->h1, (unsigned long long)h1);
    ptx_reduce_add64(&H->h2, (unsigned long long)h2);
}
PTX_DEV void ptx_digest_flush_dense_dpp(PtxHdr* H, uint64_t h1, uint64_t h2) { /* (the one- and two-wave builds) */
    ptx_reduce_add64_dpp(&H->h1, (unsigned long long)h1);
    ptx_reduce_add64_dpp(&H->h2, (unsigned long long)h2);
}

/* ---- bit-rank: one 8-byte LDS word per 32 positions = {bits, exclusive popcount prefix} ---- */
struct PtxBitWord {
    uint32_t bits;
    uint32_t pre;
};
PTX_DEV uint32_t ptx_bitrank(const PtxBitWord* b, uint32_t pos) { /* # set bits strictly below pos */
    const PtxBitWord w = b[pos >> 5];
    return w.pre + ptx_popc(w.bits & ((1u << (pos & 31)) - 1u));
}
/* rank of `pos` if its bit is set, else -1 */
PTX_DEV int ptx_bitrank_if_set(const PtxBitWord* b, uint32_t pos) {
    const PtxBitWord w = b[pos >> 5];
    const uint32_t s = pos & 31;
    if (!((w.bits >> s) & 1u)) return -1;
    return (int)(w.pre + ptx_popc(w.bits & ((1u << s) - 1u)));
}
PTX_DEV bool ptx_bittest(const uint32_t* bits, uint32_t pos) { return (bits[pos >> 5] >> (pos & 31)) & 1u; }
/* pre = exclusive popcount prefix over m bit words, by ONE wave (64 words per step: a DPP prefix sum, no barrier inside); every thread calls it, it ends
 * with the barrier that publishes the words and returns the total.  tmp: one LDS word of the CALL SITE's own (a thread that is slow to read the total must
 * not find the next prefix's there; H->scan_tmp[16 ..], the block-wide scan uses the words below) */
template <uint32_t kThreads>
PTX_DEV uint32_t ptx_bitwords_prefix(PtxBitWord* b, uint32_t m, uint32_t* tmp) {
    PTX_ONE_WAVE {
        uint32_t run = 0;
#pragma nounroll
        for (uint32_t w0 = 0; w0 < m; w0 += PTX_WS) {
            const uint32_t w = w0 + PTX_LANE_ID;
            const uint32_t c = w < m ? ptx_popc(b[w].bits) : 0u;
            const uint32_t incl = ptx_wave_incl_scan(c);
            if (w < m) b[w].pre = run + incl - c;
            run += ptx_wave_last(incl);
        }
        if (PTX_LANE_ID == 0u) *tmp = run;
    }
    PTX_SYNC_LDS();
    return *tmp;
}
/* the same for a plain array of counts: in place, exclusive; returns the total */
template <uint32_t kThreads, class T>
PTX_DEV uint32_t ptx_counts_prefix(T* a, uint32_t m, uint32_t* tmp) {
    PTX_ONE_WAVE {
        uint32_t run = 0;
#pragma nounroll
        for (uint32_t w0 = 0; w0 < m; w0 += PTX_WS) {
            const uint32_t w = w0 + PTX_LANE_ID;
            const uint32_t c = w < m ? (uint32_t)a[w] : 0u;
            const uint32_t incl = ptx_wave_incl_scan(c);
            if (w < m) a[w] = (T)(run + incl - c);
            run += ptx_wave_last(incl);
        }
        if (PTX_LANE_ID == 0u) *tmp = run;
    }
    PTX_SYNC_LDS();
    return *tmp;
}

/* ---- elemId -> element index ---- */
struct PtxElemIndex {
    PtxBitWord* ib; /* bitmap over the keys of the INSERT ops */
    uint32_t na1, max_ctr, max_actor; /* key = counter * na1 + actor, na1 = max_actor + 1: a dense id keyspace */
};
PTX_DEV bool ptx_id_key(const PtxElemIndex& ix, uint64_t id, uint32_t& key) {
    const uint32_t ctr = (uint32_t)(id >> 32), actor = (uint32_t)id;
    if (ctr == 0 || ctr > ix.max_ctr || actor > ix.max_actor) return false;
    key = ctr * ix.na1 + actor;
    return true;
}
/* dense index (rank in compareOpIds order among the inserts) of the list element with this id, or -1 */
PTX_DEV int ptx_elem_lookup(const PtxElemIndex& ix, uint64_t id) {
    uint32_t key;
    if (!ptx_id_key(ix, id, key)) return -1;
    return ptx_bitrank_if_set(ix.ib, key);
}

/* The same for the logs this file's kernel takes (max_counter < 2^19 and max_actor < 4096 were checked against the header: every factor fits 24 bits): the key
 * by ONE full-rate v_mad_u32_u24 — the compiler's own choice for `ctr * na1 + actor` is the quarter-rate 64-bit multiply-add —, an id outside the header's bounds
 * looks up key 0 (a counter of 0: no insert of an accepted log has it), and no branch: one 8-byte LDS read, a select at the end. */
PTX_DEV int ptx_elem_lookup24(const PtxElemIndex& ix, uint64_t id) {
    const uint32_t ctr = (uint32_t)(id >> 32), actor = (uint32_t)id;
    const bool ok = (ctr - 1u < ix.max_ctr) & (actor <= ix.max_actor);
    const uint32_t key0 = ptx_mul24(ctr, ix.na1) + actor; /* (not the inline-asm form: the compiler branches around an asm it cannot speculate) */
    const uint32_t key = ok ? key0 : 0u;
    const PtxBitWord w = ix.ib[key >> 5];
    const uint32_t s = key & 31u;
    const uint32_t r = w.pre + ptx_popc(w.bits & ((1u << s) - 1u));
    return ((w.bits >> s) & 1u) ? (int)r : -1;
}

/* ---- LDS bump allocator ---- */
struct PtxBump {
    uint8_t* base;
    uint32_t off, cap, high;
    bool overflow;
};
template <class T>
PTX_DEV T* ptx_alloc(PtxBump& b, uint32_t count) {
    const uint32_t bytes = (uint32_t)(((uint64_t)count * sizeof(T) + 15u) & ~15ull);
    T* p = (T*)(b.base + b.off);
    if ((uint64_t)b.off + bytes > b.cap) {
        b.overflow = true;
        return (T*)b.base; /* never dereferenced: callers bail out on overflow */
    }
    b.off += bytes;
    if (b.off > b.high) b.high = b.off;
    PTX_LDS_ALLOCATED(p, (uint64_t)count * sizeof(T), bytes);
    return p;
}

/* allocate from the recycled region `bd` while it has room, else from the top of the bump `bp` */
template <class T>
PTX_DEV T* ptx_alloc2(PtxBump& bd, PtxBump& bp, uint32_t count) {
    const uint32_t bytes = (uint32_t)(((uint64_t)count * sizeof(T) + 15u) & ~15ull);
    if ((uint64_t)bd.off + bytes <= bd.cap) {
        T* p = (T*)(bd.base + bd.off);
        bd.off += bytes;
        PTX_LDS_ALLOCATED(p, (uint64_t)count * sizeof(T), bytes);
        return p;
    }
    return ptx_alloc<T>(bp, count);
}

/* from the recycled region only, and only if it has room (nullptr otherwise): for scratch the caller can do without */
template <class T>
PTX_DEV T* ptx_try_alloc(PtxBump& bd, uint32_t count) {
    const uint32_t bytes = (uint32_t)(((uint64_t)count * sizeof(T) + 15u) & ~15ull);
    if ((uint64_t)bd.off + bytes > bd.cap) return nullptr;
    T* p = (T*)(bd.base + bd.off);
    bd.off += bytes;
    PTX_LDS_ALLOCATED(p, (uint64_t)count * sizeof(T), bytes);
    return p;
}

PTX_DEV uint32_t ptx_ceil_log2(uint32_t x) { /* smallest k with (1<<k) >= x, x>=1 */
    return x <= 1u ? 0u : 32u - (uint32_t)__builtin_clz(x - 1u);
}

/* ---- LDS working set of ptx_merge_log (mirrors its ptx_alloc calls; used by the host to size the
 *      launch and by the tests to check the bound).  N rows, n inserts, D deletes, K mark ops of
 *      which Kc comment ops over Kid comment ids, id keyspace of ks bits. ---- */
PTX_HD uint64_t ptx_a16(uint64_t x) { return (x + 15) & ~15ull; }
/* bytes that do not fit the recycled region when arrays of the given sizes are placed first-fit in order */
PTX_HD uint64_t ptx_overflow4(uint64_t free_bytes, uint64_t s0, uint64_t s1, uint64_t s2, uint64_t s3) {
    uint64_t over = 0;
    const uint64_t sz[4] = {ptx_a16(s0), ptx_a16(s1), ptx_a16(s2), ptx_a16(s3)};
    for (int i = 0; i < 4; ++i) {
        if (sz[i] <= free_bytes) free_bytes -= sz[i];
        else over += sz[i];
    }
    return over;
}
PTX_HD uint64_t ptx_overflow3(uint64_t free_bytes, uint64_t s0, uint64_t s1, uint64_t s2) { return ptx_overflow4(free_bytes, s0, s1, s2, 0); }
PTX_HD uint64_t ptx_lds_need(uint64_t N, uint64_t n, uint64_t D, uint64_t K, uint64_t Kc, uint64_t ks, uint64_t Kid) {
    (void)N;
    (void)D; /* the row list of the deletes lives in HBM (the park), like the mark ops' between P1 and P5 */
    const uint64_t nw = (ks + 31) / 32, nwe = n / 32 + 1;
    const uint64_t elem = ptx_a16(8 * (nw + 1)) + 2 * ptx_a16(2 * (n + 1)) + ptx_a16(4 * (nwe + 1)); /* recycled after P5a */
    const uint64_t persist = PTX_HDR_BYTES + ptx_a16(4 * (K / 32 + 1)) + elem;
    /* P1 .. P3: insert rows / sorted children + child counters (later the ranking words), keys / bucket members / successors, the parents with many children,
     * the parents with two or more (later the bitmap of a huge bucket) */
    const uint64_t aux_words = n / 4 + 1 > 2 * (nwe + 1) ? n / 4 + 1 : 2 * (nwe + 1);
    const uint64_t p3 = ptx_a16(4 * (n + 3)) + ptx_a16(2 * (n + 2)) + ptx_a16(2 * (n / (PTX_SMALL_BUCKET + 1) + 2)) + ptx_a16(4 * aux_words);
    /* the tail phases take the recycled element region first (the list of the live mark ops only what their worst case leaves of it) */
    const uint64_t rem = elem;
    const uint64_t comments = Kc ? ptx_overflow3(rem, 4 * (Kid + 1), 4 * (Kid + 1), 8 * (Kc + 1)) : 0; /* (the list of interval rows only takes what is left of the recycled region) */
    uint64_t T4 = PTX_TILE_4; /* the short-document tile is the visible length rounded up to a power of two: at most that of the inserts */
    if (n < T4) {
        T4 = 1;
        while (T4 < n) T4 <<= 1;
    }
    const uint64_t T1 = PTX_TILE_1;
    const uint64_t trees4 = ptx_overflow3(rem, K ? 4 * 4 * 2 * T4 : 0, 4 * (T4 + 1), 8 * (T4 / 32 + 2)); /* no mark ops: no trees */
    const uint64_t trees1 = n > PTX_TILE_4 ? ptx_overflow3(rem, K ? 4 * 2 * T1 : 0, 4 * (T1 + 1), 8 * (T1 / 32 + 2)) : 0;
    uint64_t tail = comments > trees4 ? comments : trees4;
    if (trees1 > tail) tail = trees1;

    /* P5: mark rows / interval starts, alive bits, comment breaks, interval ends, comment ids + the tail phases' overflow */
    const uint64_t p5 = ptx_a16(2 * (K + 1)) + ptx_a16(8 * (nwe + 1)) + ptx_a16(4 * (nwe + 1)) + ptx_a16(2 * (K + 1)) + ptx_a16(2 * (Kc + 1)) + tail;
    return persist + (p3 > p5 ? p3 : p5);
}
/* the same from a log header */
/* P0 scratch on top of the header: per-actor table starts + the (actor, seq) -> change table */
PTX_HD uint64_t ptx_lds_need_admission(uint64_t n_changes, uint64_t max_actors) {
    if (max_actors <= 3) return PTX_HDR_BYTES + 2 * ptx_a16(4 * (1024 / 64 + 2)) + ptx_a16(4 * 12 * (1024 / 64 + 1)); /* per-wave clock totals and check records */
    const uint64_t table = PTX_HDR_BYTES + ptx_a16(4 * (max_actors + 2)) + ptx_a16(4 * (n_changes + 1));
    const uint64_t walk = PTX_HDR_BYTES + ptx_a16(4 * 28 * (1024 / 64 + 1)); /* (four to fifteen actors: the per-wave records of the one-pass check, before the table) */
    return max_actors <= 15 && walk > table ? walk : table;
}
PTX_HD uint64_t ptx_lds_need_hdr(uint64_t N, const ptx_log_hdr& h) {
    const uint64_t K = (uint64_t)h.n_mark[0] + h.n_mark[1] + h.n_mark[2] + h.n_mark[3];
    const uint64_t ks = ((uint64_t)h.max_counter + 1) * ((uint64_t)(h.max_actor > 4095u ? 4095u : h.max_actor) + 1);
    return ptx_lds_need(N, h.n_ins, h.n_del, K, h.n_mark[PTX_MARK_COMMENT], ks, h.n_comment_ids);
}

/* census of one log, sequential (host side: the emulation driver; the device has ptx_census_kernel) */
static inline void ptx_census_rows(const uint64_t* op_id, const uint8_t* action, const uint8_t* mark_type, const uint32_t* payload, uint64_t n_rows, ptx_log_hdr* out) {
    ptx_log_hdr h;
    h.n_ins = h.n_del = h.max_counter = h.max_actor = h.n_comment_ids = h.reserved = 0;
    h.n_mark[0] = h.n_mark[1] = h.n_mark[2] = h.n_mark[3] = 0;
    for (uint64_t i = 0; i < n_rows; ++i) {
        const uint32_t ctr = (uint32_t)(op_id[i] >> 32), act = (uint32_t)op_id[i];
        if (ctr > h.max_counter) h.max_counter = ctr;
        if (act > h.max_actor) h.max_actor = act;
        if (action[i] == PTX_ACT_INSERT) h.n_ins++;
        else if (action[i] == PTX_ACT_DELETE) h.n_del++;
        else if ((action[i] == PTX_ACT_ADDMARK || action[i] == PTX_ACT_REMOVEMARK) && mark_type[i] < 4) {
            h.n_mark[mark_type[i]]++;
            if (mark_type[i] == PTX_MARK_COMMENT && payload[i] >= h.n_comment_ids) h.n_comment_ids = payload[i] == 0xFFFFFFFFu ? payload[i] : payload[i] + 1u;
        }
    }
    *out = h;
}

/* a log's header by the scalar unit, word by word (a 40-byte struct copy through the constant address space came out as vector loads) */
PTX_DEV ptx_log_hdr ptx_load_log_hdr(const ptx_log_hdr* p) {
    const uint32_t* w = (const uint32_t*)p;
    ptx_log_hdr h;
    h.n_ins = PTX_CONST_LOAD(w + 0);
    h.n_del = PTX_CONST_LOAD(w + 1);
    h.n_mark[0] = PTX_CONST_LOAD(w + 2);
    h.n_mark[1] = PTX_CONST_LOAD(w + 3);
    h.n_mark[2] = PTX_CONST_LOAD(w + 4);
    h.n_mark[3] = PTX_CONST_LOAD(w + 5);
    h.max_counter = PTX_CONST_LOAD(w + 6);
    h.max_actor = PTX_CONST_LOAD(w + 7);
    h.n_comment_ids = PTX_CONST_LOAD(w + 8);
    h.reserved = 0;
    return h;
}

template <bool kDiag>
PTX_DEV void ptx_write_result(const PtxMergeArgs& A, uint32_t log, PtxHdr* H, uint32_t status, uint32_t lds_high) {
    PTX_LEADER {
        ptx_log_result r;
        r.status = status;
        r.n_ops = status ? 0 : H->n_applied;
        r.n_elems = status ? 0 : H->n_ins;
        r.n_visible = status ? 0 : H->V;
        r.n_spans = status ? 0 : H->S;
        r.n_cintervals = status ? 0 : H->I;
        r.reserved[0] = lds_high; /* LDS bytes this log needed (diagnostic; tests bound it by ptx_lds_need) */
        /* the row at which a sequential replay would have thrown (first op of the change for seq / deps failures) */
        const uint32_t ew = H->err < H->adm ? H->err : H->adm;
        r.reserved[1] = status && ew != PTX_NO_ERR ? ew >> 5 : 0xFFFFFFFFu;
        r.digest[0] = status ? 0 : (uint64_t)H->h1;
        r.digest[1] = status ? 0 : (uint64_t)H->h2;
        A.res[log] = r;
        if (kDiag) ptx_flush_clocks(A.clocks, H->clk, PTX_NCLK);
    }
}

/* range-chmax on an implicit segment tree with P leaves (tree[1] root, leaves at P..2P-1) */
PTX_DEV void ptx_tree_chmax(uint32_t* tree, uint32_t P, uint32_t lo, uint32_t hi, uint32_t val) {
    uint32_t l = lo + P, r = hi + P;
    while (l < r) {
        if (l & 1u) ptx_atomic_max(&tree[l++], val);
        if (r & 1u) ptx_atomic_max(&tree[--r], val);
        l >>= 1;
        r >>= 1;
    }
}
PTX_DEV uint32_t ptx_tree_query(const uint32_t* tree, uint32_t P, uint32_t q) {
    uint32_t w = 0;
    for (uint32_t p = q + P; p >= 1; p >>= 1) w = tree[p] > w ? tree[p] : w;
    return w;
}
/* the same for a tree of up to 256 leaves (the short-document form: 2 PTX_TILE_4), its nine nodes read AT ONCE — the loop above is a chain of LDS round trips, one per
 * level and mark type for every visible character (round 6, last session: a sixth of a 1K-op log's time was its tree passes) */
PTX_DEV uint32_t ptx_tree_query9(const uint32_t* tree, uint32_t P, uint32_t q) {
    uint32_t v[9];
#pragma unroll
    for (uint32_t i = 0; i < 9u; ++i) {
        const uint32_t p = (q + P) >> i;
        v[i] = tree[p ? p : 1u]; /* (beyond the root of a smaller tree: the root again) */
    }
    uint32_t w = 0;
#pragma unroll
    for (uint32_t i = 0; i < 9u; ++i) w = v[i] > w ? v[i] : w;
    return w;
}
/* the same for the tiles of a long document (ten or eleven nodes from the leaf to the root), all read at once — as a loop they are ten LDS round
 * trips one after the other */
PTX_DEV uint32_t ptx_tree_query_tile(const uint32_t* tree, uint32_t P, uint32_t q) { /* P = 512 or 1 024 leaves */
    static_assert(PTX_TILE_1 == 512u, "ten levels, eleven for the double tile");
    uint32_t v[11];
#pragma unroll
    for (uint32_t i = 0; i < 11u; ++i) {
        const uint32_t p = (q + P) >> i;
        v[i] = tree[p ? p : 1u]; /* (level 11 of a 512-leaf tree: the root again) */
    }
    uint32_t w = 0;
#pragma unroll
    for (uint32_t i = 0; i < 11u; ++i) w = v[i] > w ? v[i] : w;
    return w;
}

struct PtxCEntry {
    uint16_t lo, hi, t, add;
};

/*
 * Sweep the presence function of ONE comment id over the visible axis.
 * ent[0..m): the (visible interval, application index, add/remove) of every op with this id.
 * presence(p) = action of the covering op with the largest application index (peritext.ts:315-320
 * iterates the slot's ops in application order, so the last one decides).
 * Calls emit(start,end) for every maximal present interval in ascending order; returns their count.
 */
/* The same for an id with at most eight ops (nearly every id: a comment is added once and removed or re-added a few times), out of REGISTERS: the entries are read
 * once (eight 8-byte LDS reads in flight), every step of the sweep is then a few dozen vector instructions.  The loops of the general form below read an entry from
 * LDS at every turn — 8 m^2 dependent round trips for the two sweeps of an id, a lane per id, the wave as slow as its busiest id: 110 k of the 430 k cycles of a
 * 4 096-op log that keeps its text (round 6, `rich4k`). */
template <class F>
PTX_DEV uint32_t ptx_comment_sweep8(const PtxCEntry* ent, uint32_t m, F emit) {
    uint32_t lo_[8], hi_[8], ta_[8]; /* t << 1 | add (an entry past m: an empty interval that covers nothing) */
#pragma unroll
    for (uint32_t j = 0; j < 8u; ++j) {
        const PtxCEntry e = ent[j < m ? j : 0u];
        lo_[j] = j < m ? (uint32_t)e.lo : 0xFFFFFFFFu;
        hi_[j] = j < m ? (uint32_t)e.hi : 0xFFFFFFFFu;
        ta_[j] = ((uint32_t)e.t << 1) | (e.add ? 1u : 0u);
    }
    uint32_t count = 0, start = 0;
    int64_t cur = -1;
    bool present = false;
    for (;;) {
        uint32_t p = 0xFFFFFFFFu;
#pragma unroll
        for (uint32_t j = 0; j < 8u; ++j) {
            if ((int64_t)lo_[j] > cur && lo_[j] < p) p = lo_[j];
            if ((int64_t)hi_[j] > cur && hi_[j] < p) p = hi_[j];
        }
        if (p == 0xFFFFFFFFu) break;
        int best = -1;
#pragma unroll
        for (uint32_t j = 0; j < 8u; ++j)
            if (lo_[j] <= p && p < hi_[j] && (int)ta_[j] > best) best = (int)ta_[j]; /* (t decides: the low bit rides along) */
        const bool now = best >= 0 && (best & 1);
        if (now != present) {
            if (now) start = p;
            else {
                emit(start, p);
                ++count;
            }
            present = now;
        }
        cur = (int64_t)p;
    }
    return count;
}
#ifndef PTX_SWEEP8
#define PTX_SWEEP8 1
#endif
template <bool kRegs = true, class E, class F>
PTX_DEV uint32_t ptx_comment_sweep(const E* ent, uint32_t m, F emit) {
    if constexpr (PTX_SWEEP8 && kRegs && std::is_same<E, PtxCEntry>::value) { /* (kRegs: the builds for any launch shape, which take the documents that keep their text; the lean builds of the three usual shapes — logs that show a few dozen characters, an op or two per comment id — keep the plain form: the unrolled one cost BASELINE config #4 1.2 %) */
        if (m <= 8u) return ptx_comment_sweep8(ent, m, emit);
    }
    uint32_t count = 0;
    int64_t cur = -1;
    bool present = false;
    uint32_t start = 0;
    for (;;) {
        uint32_t p = 0xFFFFFFFFu;
        for (uint32_t j = 0; j < m; ++j) {
            if ((int64_t)ent[j].lo > cur && ent[j].lo < p) p = ent[j].lo;
            if ((int64_t)ent[j].hi > cur && ent[j].hi < p) p = ent[j].hi;
        }
        if (p == 0xFFFFFFFFu) break;
        int best = -1;
        uint32_t best_add = 0;
        for (uint32_t j = 0; j < m; ++j) {
            if (ent[j].lo <= p && p < ent[j].hi && (int)ent[j].t > best) {
                best = (int)ent[j].t;
                best_add = ent[j].add;
            }
        }
        const bool now = best >= 0 && best_add != 0;
        if (now != present) {
            if (now) start = p;
            else {
                emit(start, p);
                ++count;
            }
            present = now;
        }
        cur = (int64_t)p;
    }
    return count;
}

/* ---- P0, documents of up to three actors: one step of a wave's admission walk (PTX_AC consecutive changes per lane).
 *      h = Change headers, e0 = seq | deps[0] << 16, e1 = deps[1] | deps[2] << 16 of the lane's changes; kTail: only the first
 *      `nvalid` of them exist.  See the call site for the scheme. ---- */
struct PtxAdmWave { /* the same in every lane of the wave */
    uint32_t bx, by;   /* changes per actor before this step, relative to the segment: actor 0 << 16 | -, actor 1 | actor 2 << 16 */
    uint32_t gx, gy;   /* G per actor, same packing */
    uint32_t known;    /* bit b: G of actor b has been learned */
};
template <bool kTail>
PTX_DEV void ptx_adm_step(PtxAdmWave& S, const uint32_t* h, const uint32_t* e0, const uint32_t* e1, uint32_t nvalid,
                          uint32_t& mx0, uint32_t& mx1, uint32_t& bad, uint32_t& amax, uint32_t& hsum) {
    uint32_t ohx[PTX_AC], ohy[PTX_AC], sel[PTX_AC], s[PTX_AC], tx = 0, ty = 0;
#pragma unroll
    for (uint32_t u = 0; u < PTX_AC; ++u) {
        const bool in = !kTail || u < nvalid;
        const uint32_t hu = in ? h[u] : 0u;
        amax = hu > amax ? hu : amax; /* the actor sits in the top bits */
        hsum += hu;
        const uint32_t a = hu >> PTX_CHG_ACTOR_SHIFT;
        /* one-hot of the actor in the clock packing: actor 0 -> bit 16, 1 -> bit 32, 2 -> bit 48, a change of no actor (or of an
         * actor the document does not have: the log fails through amax) counts nowhere that matters */
        const uint64_t oh = in ? ptx_shl64(0x10000ull, a << 4) : 0ull;
        ohx[u] = (uint32_t)oh;
        ohy[u] = (uint32_t)(oh >> 32);
        tx += ohx[u];
        ty += ohy[u];
        /* byte selector of the own actor's clock in {cy, cx}: bytes 2,3 / 4,5 / 6,7 */
        sel[u] = in ? 0x0c0c0302u + a * 0x0202u : 0x0c0c0c0cu;
    }
    const uint32_t ix = ptx_wave_incl_scan(tx), iy = ptx_wave_incl_scan(ty);
    uint32_t cx = S.bx + ix - tx, cy = S.by + iy - ty; /* relative clock before this lane's first change */
#pragma unroll
    for (uint32_t u = 0; u < PTX_AC; ++u) {
        const bool in = !kTail || u < nvalid;
        const uint32_t e0u = in ? e0[u] : 0u, e1u = in ? e1[u] : 0u;
        s[u] = ptx_pk_subsat_u16(e0u, ptx_perm(cy, cx, sel[u]));   /* low half: seq (-) clock[actor] */
        mx0 = ptx_pk_max_u16(mx0, ptx_pk_subsat_u16(e0u, cx));      /* high half: deps[0] (-) clock[0] */
        mx1 = ptx_pk_max_u16(mx1, ptx_pk_subsat_u16(e1u, cy));      /* deps[1] (-) clock[1] | deps[2] (-) clock[2] */
        cx += ohx[u];
        cy += ohy[u];
    }
    if (S.known != 7u) { /* wave-uniform; normally only in the first step of a segment */
        for (uint32_t b = 0; b < 3u; ++b) {
            if ((S.known >> b) & 1u) continue;
            for (uint32_t u = 0; u < PTX_AC; ++u) {
                const bool mine = (!kTail || u < nvalid) && (h[u] >> PTX_CHG_ACTOR_SHIFT) == b;
                uint32_t v = 0;
                if (ptx_wave_pick(mine, s[u] & 0xFFFFu, v)) {
                    if (b == 0u) S.gx = v << 16;
                    else if (b == 1u) S.gy = (S.gy & 0xFFFF0000u) | v;
                    else S.gy = (S.gy & 0xFFFFu) | (v << 16);
                    S.known |= 1u << b;
                    break;
                }
            }
        }
    }
#pragma unroll
    for (uint32_t u = 0; u < PTX_AC; ++u) bad |= (s[u] ^ ptx_perm(S.gy, S.gx, sel[u])) & 0xFFFFu;
    S.bx += ptx_wave_last(ix);
    S.by += ptx_wave_last(iy);
}

/* ---- P0, documents of four to fifteen actors (envelope rows of kW dwords: seq, deps[0 .. 2 kW - 1); kW = 4 up to seven actors, 6 up to eleven, 8 up to
 *      fifteen): the same one-pass walk with the relative vector clock in kW words, packed like the row it is compared with — half k of the row holds
 *      deps[k - 1], half k of the clock words the changes of actor k - 1 so far (half 0, beside seq, stays 0).  h = Change headers, e[u][j] = word j of the row of
 *      the lane's change u.  Round 5: such documents took the (actor, seq) -> change table for every log — three passes of one change per thread and step,
 *      1.85 x the per-log cost of a three-actor document at five actors (profiles/r05_d_*); the table is now what names the error of a log that fails this check. ---- */
template <uint32_t kW>
struct PtxAdmWaveN { /* the same in every lane of the wave */
    uint32_t b[kW]; /* changes per actor before this step, relative to the segment */
    uint32_t g[kW]; /* G per actor (seq - relative clock of its changes), same packing */
    uint32_t known; /* bit a: G of actor a has been learned */
};
/* the 16-bit half k of kW packed words, through a byte permute of the word PAIR that holds it (k >> 2) */
template <uint32_t kW>
PTX_DEV uint32_t ptx_adm_half(const uint32_t (&w)[kW], uint32_t k, uint32_t sel) {
    if constexpr (kW == 4) return ((k >> 2) & 1u) ? ptx_perm(w[3], w[2], sel) : ptx_perm(w[1], w[0], sel);
    uint32_t v = ptx_perm(w[1], w[0], sel);
#pragma unroll
    for (uint32_t p = 1; p < kW / 2u; ++p) v = (k >> 2) == p ? ptx_perm(w[2u * p + 1u], w[2u * p], sel) : v;
    return v;
}
template <uint32_t kW, uint32_t kAC, bool kTail>
PTX_DEV void ptx_adm_step_n(PtxAdmWaveN<kW>& S, uint32_t na, const uint32_t* h, const uint32_t (*e)[kW], uint32_t nvalid, uint32_t (&mx)[kW], uint32_t& bad, uint32_t& amax,
                            uint32_t& rows) {
    uint32_t t[kW];
#pragma unroll
    for (uint32_t j = 0; j < kW; ++j) t[j] = 0u;
#pragma unroll
    for (uint32_t u = 0; u < kAC; ++u) {
        const bool in = !kTail || u < nvalid;
        const uint32_t hu = in ? h[u] : 0u;
        amax = hu > amax ? hu : amax; /* the actor sits in the top bits */
        rows += hu & PTX_CHG_NOPS;
        const uint32_t k = (hu >> PTX_CHG_ACTOR_SHIFT) + 1u; /* the actor's half of the packing (an actor the document does not have: the log fails through amax) */
        const uint32_t one = in ? 1u << (16u * (k & 1u)) : 0u;
#pragma unroll
        for (uint32_t j = 0; j < kW; ++j) t[j] += (k >> 1) == j ? one : 0u;
    }
    uint32_t c[kW], incl[kW];
#pragma unroll
    for (uint32_t j = 0; j < kW; ++j) {
        incl[j] = ptx_wave_incl_scan(t[j]);
        c[j] = S.b[j] + incl[j] - t[j]; /* relative clock before this lane's first change */
    }
    /* (nothing per change is kept between the two loops below: a lane holds its changes' headers and rows as it is) */
#define PTX_ADMN_K(u_, in_) ((((in_) ? h[u_] : 0u) >> PTX_CHG_ACTOR_SHIFT) + 1u)
#define PTX_ADMN_SEL(k_, in_) ((in_) ? 0x0c0c0100u + ((k_) & 3u) * 0x0202u : 0x0c0c0c0cu) /* the two bytes of half k & 3 of a word pair */
    const uint32_t all = (1u << na) - 1u;
    if ((S.known & all) != all) { /* wave-uniform; normally only in the first step of a segment: G of an actor = seq (-) clock of one of its changes */
        uint32_t c2[kW];
#pragma unroll
        for (uint32_t j = 0; j < kW; ++j) c2[j] = c[j];
        for (uint32_t u = 0; u < kAC; ++u) {
            const bool in = !kTail || u < nvalid;
            const uint32_t k = PTX_ADMN_K(u, in);
            const uint32_t su = ptx_pk_subsat_u16(in ? e[u][0] : 0u, ptx_adm_half<kW>(c2, k, PTX_ADMN_SEL(k, in))) & 0xFFFFu;
            for (uint32_t b = 0; b < na; ++b) {
                uint32_t v = 0;
                if (!((S.known >> b) & 1u) && ptx_wave_pick(in && k == b + 1u, su, v)) {
                    const uint32_t sh = 16u * ((b + 1u) & 1u);
                    S.g[(b + 1u) >> 1] = (S.g[(b + 1u) >> 1] & ~(0xFFFFu << sh)) | (v << sh); /* (an indexed register array: the compiler keeps these few words in scratch for this rare block — measured against the unrolled forms, which cost the whole kernel 16 VGPRs) */
                    S.known |= 1u << b;
                }
            }
            const uint32_t one = in ? 1u << (16u * (k & 1u)) : 0u;
#pragma unroll
            for (uint32_t j = 0; j < kW; ++j) c2[j] += (k >> 1) == j ? one : 0u;
        }
    }
#pragma unroll
    for (uint32_t u = 0; u < kAC; ++u) {
        const bool in = !kTail || u < nvalid;
        const uint32_t k = PTX_ADMN_K(u, in), sel = PTX_ADMN_SEL(k, in);
        const uint32_t su = ptx_pk_subsat_u16(in ? e[u][0] : 0u, ptx_adm_half<kW>(c, k, sel)); /* low half: seq (-) clock[actor] */
        bad |= (su ^ ptx_adm_half<kW>(S.g, k, sel)) & 0xFFFFu;                              /* every seq (-) clock of an actor is its G */
#pragma unroll
        for (uint32_t j = 0; j < kW; ++j) mx[j] = ptx_pk_max_u16(mx[j], ptx_pk_subsat_u16(in ? e[u][j] : 0u, c[j])); /* deps (-) clock, per half (half 0: ignored) */
        const uint32_t one = in ? 1u << (16u * (k & 1u)) : 0u;
#pragma unroll
        for (uint32_t j = 0; j < kW; ++j) c[j] += (k >> 1) == j ? one : 0u;
    }
#undef PTX_ADMN_K
#undef PTX_ADMN_SEL
#pragma unroll
    for (uint32_t j = 0; j < kW; ++j) S.b[j] += ptx_wave_last(incl[j]);
}
#define PTX_ADMN_WREC 28u /* words a wave leaves for the validation: b[kW], g[kW], mx[kW] (kW <= 8), known, bad, amax */
#define PTX_ADMN_HALF(w_, k_) (((w_)[(k_) >> 1] >> (16u * ((k_) & 1u))) & 0xFFFFu)

/* the walk of one log: true = every change is admitted (the same answer in every thread; ends with the barrier after which wrec may be reused) */
template <uint32_t kW, uint32_t kAC, uint32_t kThreads, class HdrT>
PTX_DEV bool ptx_adm_walk_n(const PtxMergeArgs& A, HdrT* H, uint32_t* wrec, const uint32_t* c_hdr, const uint16_t* c_env, uint32_t C, uint32_t N, uint32_t na) {
    (void)A;
    PTX_LEADER { H->cur[7] = 0; }
    PTX_SYNC_LDS();
    const uint32_t nwv_ = PTX_NWAVES;
    const uint32_t step = PTX_WS * kAC; /* changes per wave and step: kAC consecutive changes per lane (four of 16-byte rows; two of the longer ones: their words are what a lane holds in registers) */
    const uint32_t seg = ((C + nwv_ - 1u) / nwv_ + step - 1u) / step * step; /* changes per wave, whole steps */
    PTX_FOR_WAVE(w, lane) {
        const uint32_t lo = w * seg < C ? w * seg : C, hi = lo + seg < C ? lo + seg : C;
        PtxAdmWaveN<kW> S;
        uint32_t mx[kW], bad = 0, amax = 0, rows = 0;
#pragma unroll
        for (uint32_t j = 0; j < kW; ++j) S.b[j] = S.g[j] = mx[j] = 0u;
        S.known = 0u;
#pragma nounroll
        for (uint32_t cb = lo; cb < hi; cb += step) {
            uint32_t h[kAC], e[kAC][kW];
            const uint32_t cl0 = cb + lane * kAC;
            const uint32_t cl = cl0 < hi ? cl0 : (hi ? hi - 1u : 0u);
            PTX_ADM_HDRSN(h, cl, kAC)
            PTX_ADM_ROWSN(e, cl, kW, kAC)
            if (cb + step <= hi) ptx_adm_step_n<kW, kAC, false>(S, na, h, e, kAC, mx, bad, amax, rows);
            else ptx_adm_step_n<kW, kAC, true>(S, na, h, e, cl0 < hi ? hi - cl0 : 0u, mx, bad, amax, rows); /* the last, partial step: lanes past `hi` play changes of no actor */
        }
#pragma unroll
        for (uint32_t j = 0; j < kW; ++j) mx[j] = ptx_wave_pk_max_u16(mx[j]);
        bad = ptx_wave_max(bad);
        amax = ptx_wave_max(amax);
        ptx_reduce_add32(&H->cur[7], rows);
        if (lane == 0u) {
            uint32_t* r = wrec + w * PTX_ADMN_WREC;
#pragma unroll
            for (uint32_t j = 0; j < kW; ++j) {
                r[j] = S.b[j];
                r[8u + j] = S.g[j];
                r[16u + j] = mx[j];
            }
            r[24] = S.known;
            r[25] = bad;
            r[26] = amax;
        }
    }
    PTX_SYNC_LDS();
    bool admitted = H->cur[7] == N; /* the same answer in every thread; the changes must tile the rows of the log exactly */
    {
        uint32_t B[2u * kW - 1u];
#pragma unroll
        for (uint32_t b = 0; b < 2u * kW - 1u; ++b) B[b] = 0u;
        for (uint32_t w = 0; w < nwv_; ++w) {
            const uint32_t* r = wrec + w * PTX_ADMN_WREC;
            if ((r[26] >> PTX_CHG_ACTOR_SHIFT) >= na || r[25] != 0u) admitted = false; /* an actor beyond the document's; two changes of an actor disagree on seq - clock */
#pragma unroll
            for (uint32_t b = 0; b < 2u * kW - 1u; ++b) {
                const uint32_t k = b + 1u;
                if (b < na && ((r[24] >> b) & 1u) && PTX_ADMN_HALF(r + 8, k) != B[b] + 1u) admitted = false; /* some seq != clock + 1 */
                if (b < na && PTX_ADMN_HALF(r + 16, k) > B[b]) admitted = false;                              /* some dep > clock */
                B[b] += PTX_ADMN_HALF(r, k);
            }
        }
    }
    PTX_SYNC_LDS(); /* (wrec has been read by everyone) */
    return admitted;
}

/* ---- the mark ops of a log in ROW order although their list is grouped by mark type (four runs, each in row order): block b takes
 *      the b-th slice of every run, so that the ops of a block come from one stretch of the log and their gathers of ref_a / ref_b /
 *      sides / payload share cache lines (visiting run after run fetched every line of those columns once per run).  A block holds at
 *      most PTX_JB_CAP ops; lane t of block b gets op k (false: none). ---- */
struct PtxMarkBlocks {
    uint32_t B;               /* blocks */
    uint32_t off[4], sz[4];   /* first list slot and length of the run of mark type g */
    uint32_t c[4];            /* ops of run g per block */
};
PTX_DEV PtxMarkBlocks ptx_mark_blocks(uint32_t moff1, uint32_t moff2, uint32_t moff3, uint32_t K, uint32_t cap) {
    PtxMarkBlocks M;
    M.off[0] = 0;
    M.off[1] = moff1;
    M.off[2] = moff2;
    M.off[3] = moff3;
    M.sz[0] = moff1;
    M.sz[1] = moff2 - moff1;
    M.sz[2] = moff3 - moff2;
    M.sz[3] = K - moff3;
    M.B = K ? (K + cap - 5u) / (cap - 4u) : 0u; /* ceil(K / (cap - 4)): then the four slices of a block, each rounded up, fit cap */
    for (int g = 0; g < 4; ++g) M.c[g] = M.B ? (M.sz[g] + M.B - 1u) / M.B : 0u;
    return M;
}
/* lane t of EVERY block works on the same run: run g owns the lanes [c[0] + .. + c[g-1], .. + c[g]) (the counts per block are rounded up, so the four
 * slices of a block need not be packed: a lane beyond the end of its run idles in the last block).  What a thread needs per step is then one multiply-add
 * and one compare; the lane's share is loop-invariant (threadIdx.x on the GPU) and computed once. */
struct PtxMarkLane {
    uint32_t kbase, c, lim; /* the lane's mark op of block b: k = kbase + b * c, it exists iff b * c < lim */
};
PTX_DEV PtxMarkLane ptx_mark_lane(const PtxMarkBlocks& M, uint32_t t) {
    PtxMarkLane L;
    L.kbase = L.c = L.lim = 0;
    uint32_t o = 0;
    bool found = false;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        if (!found && t < o + M.c[g]) {
            const uint32_t j = t - o;
            L.kbase = M.off[g] + j;
            L.c = M.c[g];
            L.lim = j < M.sz[g] ? M.sz[g] - j : 0u;
            found = true;
        }
        o += M.c[g];
    }
    return L;
}
PTX_DEV bool ptx_mark_of(const PtxMarkBlocks& M, uint32_t b, uint32_t t, uint32_t& k) {
    const PtxMarkLane L = ptx_mark_lane(M, t);
    const uint32_t s = ptx_mul24(b, L.c);
    k = s < L.lim ? L.kbase + s : 0u;
    return s < L.lim;
}
/* The lane's share where it does not change from step to step (the GPU: lane = threadIdx.x & 63), worked out ONCE per log and packed into two registers that the
 * compiler is told to keep (left to itself it works the four-way choice out again at each of the three uses of a step — a dozen vector and as many scalar
 * instructions each, round 5's instruction counts): kbase | end << 16 (both at most K < 65 536; op k = kbase + b * c exists iff it is below `end` of its run) and c. */
struct PtxMarkLaneKept {
    uint32_t kb_end, c;
};
PTX_DEV PtxMarkLaneKept ptx_mark_lane_kept(const PtxMarkBlocks& M, uint32_t t) {
    const PtxMarkLane L = ptx_mark_lane(M, t);
    PtxMarkLaneKept P;
    P.kb_end = L.kbase | ((L.kbase + L.lim) << 16);
    P.c = L.c;
    PTX_KEEP_VGPR(P.kb_end);
    PTX_KEEP_VGPR(P.c);
    return P;
}
PTX_DEV bool ptx_mark_of_kept(const PtxMarkLaneKept& P, uint32_t b, uint32_t& k) {
    const uint32_t kk = ptx_mad24(b, P.c, P.kb_end & 0xFFFFu);
    const bool has = kk < (P.kb_end >> 16);
    k = has ? kk : 0u;
    return has;
}

/* Uniform early exit on a per-log error.  The error word is sampled between two barriers so that a
 * later phase's error write can never be seen by a thread that is still at this check point. */
#define PTX_BAIL_IF_ERROR()                                        \
    do {                                                           \
        PTX_SYNC_LDS();                                                \
        uint32_t _st = H->err;                                     \
        if (_st != PTX_NO_ERR && H->adm < _st) _st = H->adm; /* an earlier change failed admission first */ \
        PTX_SYNC_LDS();                                                \
        if (_st != PTX_NO_ERR) {                                   \
            lds_high = bp.high;                                    \
            return _st & 15u;                                      \
        }                                                          \
    } while (0)

#define PTX_BAIL_CAPACITY()                                        \
    do {                                                           \
        if (bp.overflow) {                                         \
            lds_high = bp.high;                                    \
            return PTX_ERR_CAPACITY;                               \
        }                                                          \
    } while (0)


/* ================================================================================================ */
/* Applies log `log`; returns its status (PTX_OK or a per-log PTX_ERR_*) and the LDS high-water mark.  The caller
 * writes the result row (ptx_write_result) — ONE copy of that code instead of one per early exit. */
template <int kManyActors, uint32_t kThreads, bool kDiag, bool kLean>
PTX_DEV uint32_t ptx_merge_log_body(const PtxMergeArgs& A, uint32_t log, uint8_t* lds, uint32_t& lds_high) {
    /* What the phases take from the kernel arguments and the log's header is derived AGAIN at the head of every phase (PTX_REMAT below): scalar loads and scalar
     * arithmetic, which the vector units do not see.  Carried from the top of the kernel these ~70 values outlive every loop, the 96 scalar registers that seven
     * waves per SIMD allow cannot hold them, and each trip through a VGPR lane is a v_readlane — 6 % of the kernel's vector instructions in round 5, on the unit
     * that bounds it ("What bounds the kernel"). */
    uint64_t base = A.log_off[log];
    const uint64_t N64 = A.log_off[log + 1] - base;
    /* The builds of one and two waves per log (a few hundred rows: a dozen dependent trips to memory in all) read the log's Change offsets and its header HERE, with
     * its row offsets — three scalar loads that depend on nothing but the kernel arguments, one wait — and do not read the row offsets a second time after the
     * admission: read where they are used they were four more round trips in a row before the row pass could start (round 6). */
    constexpr bool kShort = kThreads == 64u || kThreads == 128u;
    uint64_t chg0_early = 0, chg1_early = 0;
    ptx_log_hdr hd_early = ptx_log_hdr();
    if (kShort) {
        if (A.chg_off) {
            chg0_early = A.chg_off[log];
            chg1_early = A.chg_off[log + 1];
        }
        hd_early = A.log_hdr[log];
    }
    PtxHdr* H = (PtxHdr*)lds;
    PTX_LEADER {
        H->err = PTX_NO_ERR;
        H->adm = PTX_NO_ERR;
        H->max_ctr = H->max_actor = 0;
        H->cur_big = H->cur_med = H->cur_huge = 0;
        H->n_ins = H->n_applied = 0;
        H->V = H->S = H->I = 0;
        H->h1 = H->h2 = 0;
        if (kDiag)
            for (int k = 0; k <= PTX_NCLK; ++k) H->clk[k] = 0;
    }
    PtxBump bp;
    bp.base = lds;
    bp.off = (kDiag ? PTX_HDR_BYTES_DIAG : PTX_HDR_BYTES);
    bp.cap = A.lds_bytes;
    bp.high = bp.off;
    bp.overflow = false;
    PTX_SYNC_LDS();
    PTX_STAMP(0);
    if (N64 > 65534u) {
        lds_high = bp.high;
        return PTX_ERR_CAPACITY;
    }
    uint32_t N = (uint32_t)N64;
    const uint64_t* op_id = A.op_id + base;
    const uint64_t* ref_a = A.ref_a + base;
    const uint64_t* ref_b = A.ref_b + base;
    const uint32_t* payload = A.payload + base;
    const uint8_t* action = A.action + base;
    const uint8_t* mark_type = A.mark_type + base;
    const uint8_t* side_a = A.side_a + base;
    const uint8_t* side_b = A.side_b + base;
    uint32_t* out_values = A.out_values + base;
    ptx_span* out_spans = A.out_spans + base;
    ptx_cinterval* out_cints = A.out_cints + base;
    /* the row-side values again, from the kernel arguments as they stand in the kernarg segment */
#define PTX_REMAT_ROWS()                                                    \
    do {                                                                    \
        const PtxMergeArgs& F_ = PTX_FRESH_ARGS(A);                         \
        PTX_REMAT_ROWS_FROM(F_);                                            \
    } while (0)
#define PTX_REMAT_ROWS_FROM(F_)                                             \
    do {                                                                    \
        base = PTX_CONST_LOAD(&F_.log_off[log]);                            \
        N = (uint32_t)(PTX_CONST_LOAD(&F_.log_off[log + 1]) - base);        \
        op_id = F_.op_id + base;                                            \
        ref_a = F_.ref_a + base;                                            \
        ref_b = F_.ref_b + base;                                            \
        payload = F_.payload + base;                                        \
        action = F_.action + base;                                          \
        mark_type = F_.mark_type + base;                                    \
        side_a = F_.side_a + base;                                          \
        side_b = F_.side_b + base;                                          \
        out_values = F_.out_values + base;                                  \
        out_spans = F_.out_spans + base;                                    \
        out_cints = F_.out_cints + base;                                    \
    } while (0)
    if (N == 0) { /* a log with no rows: an empty document */
        PTX_LEADER {
            uint64_t g1 = 0, g2 = 0;
            ptx_digest_item(g1, g2, 4u, 0u, 0u, 0u);
            ptx_digest_item(g1, g2, 4u, 1u, 0u, 0u);
            H->h1 += g1;
            H->h2 += g2;
        }
        PTX_SYNC_LDS();
        lds_high = bp.high;
        return PTX_OK;
    }

    /* ---- the loads of P1's first step (declared here: with the usual three-actor admission they go out as soon as its walk has let go of its registers, and
     *      are in flight during the validation of the walk and the clearing of the bitmaps: most of one trip to HBM, of the handful a 256-op log is) ---- */
    bool p1_loaded = false;
    /* Round 6 (last session): a log of k full steps of the row pass PLUS ONE ROW whose first row is the document's makeList — the 257 rows of a 256-op log
     * (one wave: 256 rows per step), the 1 025 of a 1 024-op one (two waves: 512) — would spend a whole step, i.e. a trip to HBM and the pass's few hundred
     * vector instructions per wave, on its last row.  A makeList is listed nowhere: all the row pass does with it is its bit in the duplicate-id bitmap and its id
     * in the bounds check.  The leader does that by hand and the pass runs over the rows FROM THE SECOND ON (every load one row further: the columns are read
     * with 8- / 1-byte aligned loads anyway, a log's first row stands wherever the batch puts it).  Not in the three-wave lean build (4 097 rows are five steps
     * and a third either way). */
    constexpr bool kHeadRow = kThreads != 192u;
    uint32_t p1_head = 0u;
    if (kHeadRow && N > 1u && ((N - 1u) % PTX_U1) == 0u) {
        const uint32_t g1_ = (N - 1u) / PTX_U1;
        if (PTX_WHOLE_STEPS(g1_) && PTX_U32(PTX_CONST_LOAD(&action[0])) == PTX_ACT_MAKELIST) p1_head = 1u;
    }
    const uint32_t p1_N = N - p1_head;
    const uint64_t* const p1_op_id = op_id + p1_head;
    const uint8_t* const p1_action = action + p1_head;
    const uint8_t* const p1_mark_type = mark_type + p1_head;
    const uint32_t p1_groups = (p1_N + PTX_U1 - 1u) / PTX_U1, p1_full = p1_N / PTX_U1, p1_steps = PTX_STEPS(p1_groups);
    static_assert(PTX_U1 == 3 || PTX_U1 == 4, "the class bytes of a thread's rows come from ONE dword (with three rows its fourth byte belongs to the next thread's first row)");
    uint64_t id[PTX_U1], id_n[PTX_U1];
    uint32_t a4, mt4, a4_n, mt4_n; /* action / mark type of the thread's PTX_U1 rows, one byte each */
    /* this thread's PTX_U1 consecutive rows of a step.  A wave whose rows all exist reads them from one address; the wave that
     * holds the end of the log clamps every row index (effects of the rows past the end are masked).  The two byte columns are
     * read with ONE (unaligned) 4-byte load each: the library pads its copies of them by PTX_BYTE_PAD bytes. */
#define PTX_P1_LOAD(g_, id_, a_, mt_)                                       \
{                                                                       \
    const uint32_t r0_ = (g_) * PTX_U1;                                 \
    if (PTX_WAVE_FIRST(g_) + PTX_WS <= p1_full) { /* one address, 16 + 8 bytes */ \
        PTX_P1_IDS(id_, p1_op_id + r0_)                                 \
    } else {                                                            \
        _Pragma("unroll") for (int u = 0; u < PTX_U1; ++u) {            \
            const uint32_t r_ = r0_ + (uint32_t)u;                      \
            id_[u] = p1_op_id[r_ < p1_N ? r_ : p1_N - 1u];              \
        }                                                               \
    }                                                                   \
    PTX_P1_BYTES(p1_action, r0_, a_, p1_N)                              \
    PTX_P1_BYTES(p1_mark_type, r0_, mt_, p1_N)                          \
}
    /* ---- P0: causal admission (micromerge.ts:499-511), when the batch carries the Change envelope ----
     * Sequential rule: change c of actor a is admitted iff seq == clock[a] + 1 and clock[b] >= deps[b] for all b,
     * where clock[b] counts the changes of b applied before c.  Envelope per change: chg_hdr = actor << 20 | nops and one
     * chg_env row of u16 {seq, deps[0 .. max_actors)}.  This kernel takes logs of at most 65533 changes whose values fit 16 bits
     * (the census sends the others to biglog_core.h): a value saturated at 65535 can never be admitted here — the same error as
     * the true one. */
    if (A.chg_off) {
        const uint64_t c0 = kShort ? chg0_early : A.chg_off[log];
        const uint64_t C64 = (kShort ? chg1_early : A.chg_off[log + 1]) - c0;
        const uint32_t na = A.max_actors;
        if (C64 > 65533u || na == 0u || na > 4096u) {
            lds_high = bp.high;
            return C64 > 65533u ? PTX_ERR_CAPACITY : PTX_ERR_BAD_OP;
        }
        const uint32_t C = (uint32_t)C64;
        const uint32_t estride = PTX_ENV_STRIDE(na);
        const uint32_t* c_hdr = A.chg_hdr + c0;
        const uint16_t* c_env = A.chg_env + c0 * estride;
        /* first row of change c, only needed to place an error: the reference throws at the first failing change */
#define PTX_CHANGE_ROW(c_, row_)                                              \
    do {                                                                      \
        uint32_t r_ = 0;                                                      \
        for (uint32_t q_ = 0; q_ < (c_); ++q_) r_ += c_hdr[q_] & PTX_CHG_NOPS; \
        (row_) = r_ < 65535u ? r_ : 65535u;                                   \
    } while (0)
        if (na <= 3u) {
            /* Up to three actors (the usual case; envelope rows of 8 bytes): the vector clock itself is carried along the log.  Every WAVE owns a
             * contiguous segment of the changes and walks it 64 * PTX_AC changes at a time, PTX_AC consecutive changes per
             * lane, read with 16-byte loads (4 headers / 2 envelope rows each); inside a step the clock before each change
             * is a DPP prefix sum of one-hot counts, 16 bits per actor (actors 0,1 in one word, 2,3 in the other); the
             * clock before a wave's segment is the sum of the earlier waves' totals.  seq == clock[actor] + 1 and
             * deps[b] <= clock[b] (micromerge.ts:501-509) are then plain compares. */
            uint32_t* wt01 = ptx_alloc<uint32_t>(bp, PTX_MAX_THREADS / 64 + 2);
            uint32_t* wt23 = ptx_alloc<uint32_t>(bp, PTX_MAX_THREADS / 64 + 2);
            PTX_BAIL_CAPACITY();
            PTX_LEADER { H->cur[7] = 0; }
            PTX_SYNC_LDS();
            const uint32_t nwv_ = PTX_NWAVES;
            const uint32_t step = PTX_WS * PTX_AC; /* changes per wave and step */
            const uint32_t seg = ((C + nwv_ - 1u) / nwv_ + step - 1u) / step * step; /* changes per wave, whole steps */
            /* FAST CHECK, one pass: every wave walks its segment with clocks RELATIVE to the segment's start (the loads of the next
             * step in flight).  For every change of actor a,  seq (-) relative clock[a]  must be one and the same number G[a] (then
             * G[a] - 1 is the clock before the segment), and per actor b the maximum of  deps[b] (-) relative clock[b]  must not
             * exceed the clock before the segment ((-) saturates at 0).  The clocks before the segments are only known once every
             * wave has counted its own: the few per-wave numbers are validated after the pass.  A log that fails (rare) is walked
             * again by the exact two-pass code below, which names the first failing change. */
            uint32_t* wrec = ptx_alloc<uint32_t>(bp, (PTX_MAX_THREADS / 64 + 1) * 12u);
            PTX_BAIL_CAPACITY();
            PTX_FOR_WAVE(w, lane) {
                const uint32_t lo = w * seg < C ? w * seg : C, hi = lo + seg < C ? lo + seg : C;
                /* Relative clocks of the wave's segment, packed like the envelope row they are compared with (u16 seq, deps[0..3)):
                 * cx = clock[0] << 16 (beside deps[0]; the half beside seq stays 0), cy = clock[1] | clock[2] << 16 (beside deps[1], deps[2]).
                 * Per change a handful of packed 16-bit operations: the own actor's clock through a byte permute, seq (-) clock and
                 * deps (-) clock as saturating packed subtractions, running packed maxima.  "All seq - clock of an actor are equal"
                 * is checked against the value G of the FIRST change of that actor the wave meets (learned once per wave, scalar
                 * code); G itself is validated against the true clock before the segment after the pass. */
                PtxAdmWave S;
                S.bx = S.by = S.gx = S.gy = 0u;
                S.known = (7u << na) & 7u; /* (an actor the document does not have has no G to learn: with one or two actors the search for it ran in EVERY step, round 6) */
                uint32_t mx0 = 0, mx1 = 0, bad = 0, amax = 0, hsum = 0;
                uint32_t h[PTX_AC], h_n[PTX_AC], e0[PTX_AC], e1[PTX_AC], e0_n[PTX_AC], e1_n[PTX_AC];
#define PTX_ADM_LOAD(cb_, h_, e0_, e1_)                                     \
    {                                                                       \
        const uint32_t cl0_ = (cb_) + lane * PTX_AC;                        \
        const uint32_t cl_ = cl0_ < hi ? cl0_ : (hi ? hi - 1u : 0u);        \
        PTX_ADM_HDRS(h_, cl_)                                               \
        PTX_ADM_ENVS32(e0_, e1_, cl_)                                       \
    }
                /* two register sets in turn (no copies from "next" to "current"): set A holds the even steps, set B the odd ones */
#define PTX_ADM_STEP(cb_, h_, e0_, e1_)                                                                           \
    if ((cb_) + step <= hi) {                                                                                     \
        ptx_adm_step<false>(S, h_, e0_, e1_, PTX_AC, mx0, mx1, bad, amax, hsum);                                  \
    } else { /* the last, partial step of the segment: lanes past `hi` play changes of no actor */                \
        const uint32_t cl = (cb_) + lane * PTX_AC;                                                                \
        ptx_adm_step<true>(S, h_, e0_, e1_, cl < hi ? hi - cl : 0u, mx0, mx1, bad, amax, hsum);                   \
    }
                PTX_ADM_LOAD(lo, h, e0, e1)
#pragma nounroll
                for (uint32_t cb = lo; cb < hi; cb += 2u * step) {
                    PTX_ADM_LOAD(cb + step, h_n, e0_n, e1_n)
                    PTX_ADM_STEP(cb, h, e0, e1)
                    if (cb + step < hi) { /* wave-uniform */
                        PTX_ADM_LOAD(cb + 2u * step, h, e0, e1)
                        PTX_ADM_STEP(cb + step, h_n, e0_n, e1_n)
                    }
                }
#undef PTX_ADM_STEP
#undef PTX_ADM_LOAD
                PTX_P1_LOAD(PTX_G_OF(0u, p1_steps), id, a4, mt4)
                p1_loaded = true;
                constexpr bool kDpp = kThreads == 64u || kThreads == 128u; /* (one- and two-wave logs: the reductions without LDS round trips; round 6: -3.7 % / -3.0 % on configs #2 / #3) */
                mx0 = kDpp ? ptx_wave_pk_max_u16_dpp(mx0) : ptx_wave_pk_max_u16(mx0);
                mx1 = kDpp ? ptx_wave_pk_max_u16_dpp(mx1) : ptx_wave_pk_max_u16(mx1);
                bad = kDpp ? ptx_wave_max_dpp(bad) : ptx_wave_max(bad);
                amax = kDpp ? ptx_wave_max_dpp(amax) : ptx_wave_max(amax);
                /* rows of the segment = sum of the headers' low 20 bits = sum of the headers - (actors << 20), modulo 2^32 */
                hsum -= lane == 0u ? ((S.by & 0xFFFFu) + 2u * (S.by >> 16)) << PTX_CHG_ACTOR_SHIFT : 0u;
                if (kDpp) ptx_reduce_add32_dpp(&H->cur[7], hsum);
                else ptx_reduce_add32(&H->cur[7], hsum);
                if (lane == 0u) {
                    uint32_t* r = wrec + w * 12u;
                    r[0] = S.bx;
                    r[1] = S.by;
                    r[2] = S.gx;
                    r[3] = S.gy;
                    r[4] = S.known;
                    r[5] = bad;
                    r[6] = mx0;
                    r[7] = mx1;
                    r[8] = amax;
                }
            }
            PTX_SYNC_LDS();
            bool admitted = H->cur[7] == N; /* the same answer in every thread: wrec is complete.  The changes must tile the rows of the log exactly */
            {
                uint32_t B[3] = {0u, 0u, 0u};
                for (uint32_t w = 0; w < nwv_; ++w) {
                    const uint32_t* r = wrec + w * 12u;
                    const uint32_t G[3] = {r[2] >> 16, r[3] & 0xFFFFu, r[3] >> 16}, M[3] = {r[6] >> 16, r[7] & 0xFFFFu, r[7] >> 16};
                    if ((r[8] >> PTX_CHG_ACTOR_SHIFT) >= na || r[5] != 0u) admitted = false; /* an actor beyond the document's; two changes of an actor disagree on seq - clock */
#pragma unroll
                    for (uint32_t b = 0; b < 3; ++b) {
                        if (b < na && ((r[4] >> b) & 1u) && G[b] != B[b] + 1u) admitted = false; /* some seq != clock + 1 */
                        if (b < na && M[b] > B[b]) admitted = false;                     /* some dep > clock */
                    }
                    B[0] += r[0] >> 16;
                    B[1] += r[1] & 0xFFFFu;
                    B[2] += r[1] >> 16;
                }
            }
            if (!admitted) {
                /* EXACT walk of a failing log: which change fails first, and how (the reference throws there) */
                PTX_NOTE_EXACT_WALK(); /* test / diagnostic hook: valid logs must never come here */
                PTX_SYNC_LDS();
                PTX_LEADER { H->cur[7] = 0; }
                PTX_SYNC_LDS();
            /* pass A: per-wave totals of changes per actor, rows per log, actor range */
                PTX_FOR_WAVE(w, lane) {
                    const uint32_t lo = w * seg < C ? w * seg : C, hi = lo + seg < C ? lo + seg : C;
                    uint32_t t01 = 0, t23 = 0, rows = 0, badc = 0xFFFFFFFFu;
#pragma nounroll
                    for (uint32_t cb = lo; cb < hi; cb += step) {
                        uint32_t h[PTX_AC];
                        const uint32_t cl = cb + lane * PTX_AC;
                        PTX_ADM_HDRS(h, cl < hi ? cl : hi - 1u)
#pragma unroll
                        for (uint32_t u = 0; u < PTX_AC; ++u) {
                            const bool in = cl + u < hi;
                            const uint32_t a = h[u] >> PTX_CHG_ACTOR_SHIFT;
                            rows += in ? h[u] & PTX_CHG_NOPS : 0u;
                            if (in && a >= na) badc = cl + u < badc ? cl + u : badc;
                            t01 += in && a < 2u ? 1u << (16u * a) : 0u;
                            t23 += in && (a & ~1u) == 2u ? 1u << (16u * (a & 1u)) : 0u;
                        }
                    }
                    t01 = ptx_wave_total(t01);
                    t23 = ptx_wave_total(t23);
                    if (lane == 0u) {
                        wt01[w] = t01;
                        wt23[w] = t23;
                    }
                    ptx_reduce_add32(&H->cur[7], rows);
                    if (badc != 0xFFFFFFFFu) {
                        uint32_t row;
                        PTX_CHANGE_ROW(badc, row);
                        ptx_atomic_min(&H->adm, ((row * 2u) << 4) | PTX_ERR_BAD_OP);
                    }
                }
                PTX_SYNC_LDS();
                if (H->adm != PTX_NO_ERR || H->cur[7] != N) { /* malformed envelope; the changes must tile the rows of the log exactly */
                    lds_high = bp.high;
                    return PTX_ERR_BAD_OP;
                }
                /* pass B: the checks; the loads of the NEXT step are in flight while this step is checked */
                PTX_FOR_WAVE(w, lane) {
                    const uint32_t lo = w * seg < C ? w * seg : C, hi = lo + seg < C ? lo + seg : C;
                    uint32_t b01 = 0, b23 = 0; /* the clock before this wave's segment */
                    for (uint32_t q = 0; q < w; ++q) {
                        b01 += wt01[q];
                        b23 += wt23[q];
                    }
                    uint32_t h[PTX_AC], h_n[PTX_AC];
                    uint16_t e[PTX_AC][4], e_n[PTX_AC][4];
#define PTX_ADM_LOAD(cb_, h_, e_)                                           \
    {                                                                       \
        const uint32_t cl0_ = (cb_) + lane * PTX_AC;                        \
        const uint32_t cl_ = cl0_ < hi ? cl0_ : (hi ? hi - 1u : 0u);        \
        PTX_ADM_HDRS(h_, cl_)                                               \
        PTX_ADM_ENVS(e_, cl_)                                               \
    }
                    PTX_ADM_LOAD(lo, h, e)
#pragma nounroll
                    for (uint32_t cb = lo; cb < hi; cb += step) {
                        PTX_ADM_LOAD(cb + step, h_n, e_n)
                        const uint32_t cl = cb + lane * PTX_AC;
                        uint32_t o01[PTX_AC], o23[PTX_AC], t01 = 0, t23 = 0;
#pragma unroll
                        for (uint32_t u = 0; u < PTX_AC; ++u) {
                            const bool in = cl + u < hi;
                            const uint32_t a = h[u] >> PTX_CHG_ACTOR_SHIFT;
                            o01[u] = in && a < 2u ? 1u << (16u * a) : 0u;
                            o23[u] = in && (a & ~1u) == 2u ? 1u << (16u * (a & 1u)) : 0u;
                            t01 += o01[u];
                            t23 += o23[u];
                        }
                        const uint32_t i01 = ptx_wave_incl_scan(t01), i23 = ptx_wave_incl_scan(t23);
                        uint32_t w01 = b01 + i01 - t01, w23 = b23 + i23 - t23; /* the clock before this lane's first change */
#pragma unroll
                        for (uint32_t u = 0; u < PTX_AC; ++u) {
                            const uint32_t c = cl + u;
                            const uint32_t a = h[u] >> PTX_CHG_ACTOR_SHIFT;
                            const uint32_t clk[3] = {w01 & 0xFFFFu, w01 >> 16, w23 & 0xFFFFu};
                            const uint32_t mine = ((a < 2u ? w01 : w23) >> (16u * (a & 1u))) & 0xFFFFu;
                            const bool bad_seq = (uint32_t)e[u][0] != mine + 1u;
                            bool bad_dep = false;
#pragma unroll
                            for (uint32_t b = 0; b < 3; ++b) bad_dep = bad_dep || (b < na && (uint32_t)e[u][1u + b] > clk[b]);
                            if (c < hi && (bad_seq || bad_dep)) {
                                uint32_t row;
                                PTX_CHANGE_ROW(c, row);
                                ptx_atomic_min(&H->adm, ((row * 2u) << 4) | (bad_seq ? PTX_ERR_SEQ_GAP : PTX_ERR_MISSING_DEP));
                            }
                            w01 += o01[u];
                            w23 += o23[u];
                        }
                        b01 += ptx_wave_last(i01);
                        b23 += ptx_wave_last(i23);
#pragma unroll
                        for (uint32_t u = 0; u < PTX_AC; ++u) {
                            h[u] = h_n[u];
#pragma unroll
                            for (uint32_t b = 0; b < 4; ++b) e[u][b] = e_n[u][b];
                        }
                    }
#undef PTX_ADM_LOAD
                }
            }
            PTX_SYNC_LDS();
            bp.off = (kDiag ? PTX_HDR_BYTES_DIAG : PTX_HDR_BYTES);
        } else if constexpr (!kManyActors) {
            /* this build of the kernel carries only the <= 3-actor admission; the host launches the other one */
            lds_high = bp.high;
            return PTX_ERR_CAPACITY;
        } else {
        bool fast_admitted = false;
        if (kManyActors == 2 ? na >= 8u && na <= 15u : na <= 7u) {
            /* Four to fifteen actors (envelope rows of 16 / 24 / 32 bytes): the one-pass check of the three-actor path with the clock in as many words as a row has
             * (ptx_adm_step_n).  A log that passes is admitted; one that fails (rare) goes through the table below, which names the first failing change.
             * Two builds (kManyActors 1: up to seven actors, the walk over 16-byte rows; 2: eight to fifteen — its wider words would cost the first build
             * two waves per SIMD); documents of more than fifteen actors take the table of the first. */
            uint32_t* wrec = ptx_alloc<uint32_t>(bp, (PTX_MAX_THREADS / 64 + 1) * PTX_ADMN_WREC);
            PTX_BAIL_CAPACITY();
            if constexpr (kManyActors == 2) {
                fast_admitted = na <= 11u ? ptx_adm_walk_n<6, 2, kThreads>(A, H, wrec, c_hdr, c_env, C, N, na) : ptx_adm_walk_n<8, 2, kThreads>(A, H, wrec, c_hdr, c_env, C, N, na);
            } else {
                fast_admitted = ptx_adm_walk_n<4, 4, kThreads>(A, H, wrec, c_hdr, c_env, C, N, na);
            }
            bp.off = (kDiag ? PTX_HDR_BYTES_DIAG : PTX_HDR_BYTES);
            if (!fast_admitted) PTX_NOTE_EXACT_WALK();
        }
        if (!fast_admitted) {
        /* More than fifteen actors, or a log that failed the check above (rare): tbl[first[a] + seq - 1] = index of the change (a, seq) makes "the seqs of an
         * actor are 1, 2, ... in log order" and "dependency (b, d) sits earlier in the log" one LDS read each.  Kept
         * deliberately plain (one change per thread and step, serial prefix by the leader). */
        uint32_t* first = ptx_alloc<uint32_t>(bp, na + 2); /* changes per actor -> first table slot of the actor */
        uint32_t* tbl = ptx_alloc<uint32_t>(bp, C + 1);     /* (actor, seq) -> index of the EARLIEST change that claims it (atomic minimum: which change a duplicate
                                                               fails at must not depend on the order the threads run in; the reference throws at the later one) */
        PTX_BAIL_CAPACITY();
        PTX_FOR(a, na + 2) first[a] = 0;
        PTX_FOR(c, C + 1) tbl[c] = 0xFFFFFFFFu;
        PTX_LEADER { H->cur[7] = 0; }
        PTX_SYNC_LDS();
        PTX_FOR(c, C) {
            const uint32_t a = c_hdr[c] >> PTX_CHG_ACTOR_SHIFT;
            ptx_atomic_add(&H->cur[7], c_hdr[c] & PTX_CHG_NOPS);
            if (a >= na) {
                uint32_t row;
                PTX_CHANGE_ROW(c, row);
                ptx_atomic_min(&H->adm, ((row * 2u) << 4) | PTX_ERR_BAD_OP);
            } else ptx_atomic_add(&first[a], 1u);
        }
        PTX_SYNC_LDS();
        if (H->adm != PTX_NO_ERR || H->cur[7] != N) { /* malformed envelope; the changes must tile the rows of the log exactly */
            lds_high = bp.high;
            return PTX_ERR_BAD_OP;
        }
        PTX_LEADER {
            uint32_t run = 0;
            for (uint32_t a = 0; a < na + 2u; ++a) {
                const uint32_t v = first[a];
                first[a] = run;
                run += v;
            }
        }
        PTX_SYNC_LDS();
        PTX_FOR(c, C) {
            const uint32_t a = c_hdr[c] >> PTX_CHG_ACTOR_SHIFT, sq = c_env[(uint64_t)c * estride];
            const uint32_t f = first[a], cnt_a = first[a + 1] - f;
            if (sq - 1u < cnt_a) ptx_atomic_min(&tbl[f + sq - 1u], c); /* a later claimant of the slot is caught below */
        }
        PTX_SYNC_LDS();
        PTX_FOR(c, C) {
            const uint32_t a = c_hdr[c] >> PTX_CHG_ACTOR_SHIFT, sq = c_env[(uint64_t)c * estride];
            const uint32_t f = first[a], cnt_a = first[a + 1] - f;
            /* seq == clock[a] + 1  (micromerge.ts:501-504) */
            const bool bad_seq = !(sq - 1u < cnt_a) || tbl[f + sq - 1u] != c || (sq > 1u && tbl[f + sq - 2u] >= c);
            bool bad_dep = false;
            /* clock[b] >= deps[b] for every actor  (micromerge.ts:505-509) */
            for (uint32_t b = 0; b < na && !bad_seq; ++b) {
                const uint32_t d = c_env[(uint64_t)c * estride + 1u + b];
                if (d != 0u) {
                    const uint32_t fb = first[b], cnt_b = first[b + 1] - fb;
                    if (!(d <= cnt_b) || tbl[fb + d - 1u] >= c) bad_dep = true;
                }
            }
            if (bad_seq || bad_dep) {
                uint32_t row;
                PTX_CHANGE_ROW(c, row);
                ptx_atomic_min(&H->adm, ((row * 2u) << 4) | (bad_seq ? PTX_ERR_SEQ_GAP : PTX_ERR_MISSING_DEP));
            }
        }
        /* a failed admission stays pending in H->adm: an op-level error of an EARLIER row (found by the phases
         * below, which still run) wins over it, exactly as in a sequential replay */
        PTX_SYNC_LDS();
        bp.off = (kDiag ? PTX_HDR_BYTES_DIAG : PTX_HDR_BYTES);
        } /* the table */
        } /* na > 3 */
#undef PTX_CHANGE_ROW
    }


    PTX_STAMP(1); /* (diagnostic builds: end of the admission phase) */
    /* ---- the log header (census) sizes everything; the row pass below verifies it ---- */
    ptx_log_hdr hd;
    if (kShort) hd = hd_early;
    else {
        PTX_REMAT_ROWS();
        hd = ptx_load_log_hdr(&PTX_FRESH_ARGS(A).log_hdr[log]);
    }
    uint32_t n, D, moff1, moff2, moff3, Kc, Kid, K; /* list elements (inserts); deletes; mark ops, listed grouped by type: type t owns [moff_t, moff_{t+1}) */
    PtxElemIndex ix;
    uint32_t kbits, keyspace, nw, nwe;
    /* comment ids are doc-local ranks over ALL replicas of the document: a log that has seen only some of the comments
     * still carries the document's ranks, so the per-id tables are sized by the id space, not by the log's comment ops */
#define PTX_HDR_SCALARS()                                                   \
    do {                                                                    \
        n = hd.n_ins;                                                       \
        D = hd.n_del;                                                       \
        moff1 = hd.n_mark[0];                                               \
        moff2 = moff1 + hd.n_mark[1];                                       \
        moff3 = moff2 + hd.n_mark[2];                                       \
        Kc = hd.n_mark[PTX_MARK_COMMENT];                                   \
        Kid = Kc ? hd.n_comment_ids : 0u;                                   \
        K = moff3 + hd.n_mark[3];                                           \
        ix.max_ctr = hd.max_counter;                                        \
        ix.max_actor = hd.max_actor;                                        \
        ix.na1 = ix.max_actor + 1u;                                         \
        kbits = ptx_ceil_log2(K + 1);                                       \
        keyspace = (ix.max_ctr + 1u) * ix.na1;                              \
        nw = (keyspace + 31) / 32;                                          \
        nwe = (n >> 5) + 1; /* words of an element-indexed bitmap (bit positions 0..n) */ \
    } while (0)
    PTX_HDR_SCALARS();
#define PTX_TYPE_OF(k) (((k) >= moff1 ? 1u : 0u) + ((k) >= moff2 ? 1u : 0u) + ((k) >= moff3 ? 1u : 0u))
    if ((uint64_t)n + D + K > N) {
        lds_high = bp.high;
        return PTX_ERR_BAD_OP;
    }
    if (ix.max_actor > 4095u || ix.max_ctr >= (1u << 19) || n > 32766u || Kid > 65535u) { /* keyspace far below 2^31 bits; 2n+1 tour nodes in 16 bits */
        lds_high = bp.high;
        return PTX_ERR_CAPACITY;
    }
    if (((uint64_t)(keyspace + 1u) << kbits) > 0xFFFFFFFFull) { /* (key+1) << kbits | mark index in one u32 */
        lds_high = bp.high;
        return PTX_ERR_CAPACITY;
    }
    /* The row lists of the deletes and of the mark ops never live in LDS during P1 .. P4: the row pass writes them straight to their PARK in HBM — the log's own
     * span rows (8 bytes per row of the log = 2 N entries of 4 bytes, written by nobody before P6): entry = row | id key << 16, the deletes at [0, D), the mark ops
     * at the TOP, [mp0, 2 N) with mp0 = 2 N - K, type t from mp0 + moff_t.  P3a reads the deletes back, P5a the marks (both coalesced, each two steps ahead of the gathers that go through them); rows and keys
     * of the marks that still cover a visible character are read from the park by P5c / P5b — also AFTER the first span rows are out (long documents go tile by
     * tile): a log has at most n spans and N > n + D + K rows, so span row s (entries 2 s, 2 s + 1 < 2 n) never reaches entry mp0 > 2 n. */
    uint32_t* park = (uint32_t*)out_spans;
    uint32_t park_top = 2u * N - 1u; /* last 4-byte entry of the park (a lying header's stores are kept inside it) */
    uint32_t mp0 = 2u * N - K;       /* (K <= N was checked above) */
    /* element-side state: dead once the mark intervals are known (P5a), then reused as scratch of the tail phases.  The id bitmap comes first: its LDS address is a
     * compile-time constant, so the word address of a look-up is its instruction's offset field (one vector instruction less per id look-up and per row of P1) */
    const uint32_t elem_lds = (kDiag ? PTX_HDR_BYTES_DIAG : PTX_HDR_BYTES);
    bp.off = elem_lds;
    ix.ib = ptx_alloc<PtxBitWord>(bp, nw + 1);
    uint16_t* row_of = ptx_alloc<uint16_t>(bp, n + 1);         /* element -> op row */
    uint16_t* par = ptx_alloc<uint16_t>(bp, n + 1);            /* element -> parent element (n = HEAD); later: document position */
    uint32_t* delbits = ptx_alloc<uint32_t>(bp, nwe + 1);      /* element -> tombstone */
    uint32_t elem_end = bp.off;
    uint32_t* maddbits = ptx_alloc<uint32_t>(bp, (K >> 5) + 1); /* mark op k is an addMark (the tail phases need no look back at `action`); lives to the end */
    PTX_BAIL_CAPACITY();
    uint32_t mark_lds = bp.off; /* everything above this mark is phase scratch */
    /* scratch of P1..P3, three n-sized arrays:
     *   RW  one block of n + 3 words = `ilist` (rows of the inserts in row order; P3a leaves the deletes' target elements there; P3c: `srt`) followed by `cntw`
     *       (children per parent -> bucket ends); once the successor list stands both are dead and the block is `R`, the {next, weight} words of the list ranking
     *   L   `klist` (id keys of the inserts; P3a leaves the deletes' rows there), then `seg` (bucket members in arrival order), then `nx` (successors)
     *   aux the list of the parents with many children and the bitmap that ranks a huge bucket */
    uint32_t* RW = ptx_alloc<uint32_t>(bp, n + 3);
    uint16_t* ilist = (uint16_t*)RW;
    uint32_t* cntw = RW + (n + 2) / 2; /* behind the n + 1 entries of ilist */
    uint16_t* L = ptx_alloc<uint16_t>(bp, n + 2);
    /* when every id key fits 16 bits (the usual case): key of the insert in slot s, written beside ilist[s] by P1, so that P3a
     * needs no second look at the op_id column (the marks' keys go to the park beside their rows, for P5b) */
    /* kLean: the build for batches whose every log has 16-bit id keys and whose caller wants neither elem_rank nor the resolved references (the host checks both:
     * census + PTX_FLAG_NO_ELEM_RANK) — the code of the wide-key paths and of the two optional outputs is not in it (fewer scalar registers spilled) */
    const bool small_keys = kLean || keyspace <= 65536u;
    if (kLean && keyspace > 65536u) { /* (never launched for such a log) */
        lds_high = bp.high;
        return PTX_ERR_CAPACITY;
    }
    uint32_t* const out_rank = kLean ? nullptr : A.out_rank;
    uint32_t* const out_refs = kLean ? nullptr : A.out_refs;
    uint16_t* klist = L;
    uint16_t* bigp = ptx_alloc<uint16_t>(bp, n / (PTX_SMALL_BUCKET + 1u) + 2); /* parents with more than PTX_SMALL_BUCKET children */
    /* parents with two or more children (at most half of the elements' parents); once they are sorted the same words hold the bitmap that ranks a huge bucket */
    const uint32_t aux_words = n / 4u + 1u > 2u * (nwe + 1u) ? n / 4u + 1u : 2u * (nwe + 1u);
    uint32_t* auxw = ptx_alloc<uint32_t>(bp, aux_words);
    uint16_t* plist = (uint16_t*)auxw;
    PtxBitWord* hb = (PtxBitWord*)auxw;
    PTX_BAIL_CAPACITY();
    uint32_t d_fused = D < n + 1u ? D : n + 1u; /* deletes that ride along with the inserts in P3a */
    /* the arrays of P4 .. P6 that stand where the tree scratch stood (allocated after P3) */
    PtxBitWord* alive = nullptr;
    uint32_t* brkbits = nullptr;
    uint16_t *mrk_lo = nullptr, *mrk_hi = nullptr, *cid = nullptr;
    uint16_t* rnk = par; /* (document positions: the parents' array, from the end of P3 on) */
    /* PTX_REMAT(): every scalar above again — the rows' side from the kernel arguments, the header from HBM (scalar loads), the LDS addresses by the
     * allocator's own arithmetic (ptx_a16 sizes in its order).  Called at the head of a phase, after the barrier that ends the one before. */
#ifndef PTX_REMAT_MASK
#define PTX_REMAT_MASK 0x82u /* which of the numbered call sites below derive the scalars again (bit k: site k).  Two are enough for a build without a single spilled scalar register — the head of P3 and its end —; all thirteen cost +1 % (their scalar loads and arithmetic), round 6 */
#endif
#ifndef PTX_REMAT_MASK64
#define PTX_REMAT_MASK64 PTX_REMAT_MASK /* the one-wave build's own choice (it is held to 80 scalar registers: eight waves per SIMD) */
#endif
#ifndef PTX_REMAT_MASK128
#define PTX_REMAT_MASK128 PTX_REMAT_MASK
#endif
#define PTX_REMAT_AT(k_, call_)                                                             \
    do {                                                                                    \
        if (((kThreads == 64u ? PTX_REMAT_MASK64 : kThreads == 128u ? PTX_REMAT_MASK128 : PTX_REMAT_MASK) >> (k_)) & 1u) { call_; } \
    } while (0)
#define PTX_LDS_AT(T_, off_) ((T_*)(lds + (off_)))
#define PTX_REMAT()                                                                          \
    do {                                                                                     \
        const PtxMergeArgs& F_ = PTX_FRESH_ARGS(A);                                          \
        PTX_REMAT_ROWS_FROM(F_);                                                             \
        hd = ptx_load_log_hdr(&F_.log_hdr[log]);                                             \
        PTX_HDR_SCALARS();                                                                   \
        park = (uint32_t*)out_spans;                                                         \
        park_top = 2u * N - 1u;                                                              \
        mp0 = 2u * N - K;                                                                    \
        d_fused = D < n + 1u ? D : n + 1u;                                                   \
        uint32_t o_ = elem_lds;                                                              \
        ix.ib = PTX_LDS_AT(PtxBitWord, o_);                                                  \
        o_ += (uint32_t)ptx_a16(8u * (nw + 1u));                                             \
        row_of = PTX_LDS_AT(uint16_t, o_);                                                   \
        o_ += (uint32_t)ptx_a16(2u * (n + 1u));                                              \
        par = PTX_LDS_AT(uint16_t, o_);                                                      \
        rnk = par;                                                                           \
        o_ += (uint32_t)ptx_a16(2u * (n + 1u));                                              \
        delbits = PTX_LDS_AT(uint32_t, o_);                                                  \
        o_ += (uint32_t)ptx_a16(4u * (nwe + 1u));                                            \
        elem_end = o_;                                                                       \
        maddbits = PTX_LDS_AT(uint32_t, o_);                                                 \
        o_ += (uint32_t)ptx_a16(4u * ((K >> 5) + 1u));                                       \
        mark_lds = o_; /* the tree scratch ... */                                            \
        RW = PTX_LDS_AT(uint32_t, o_);                                                       \
        ilist = (uint16_t*)RW;                                                               \
        cntw = RW + (n + 2) / 2;                                                             \
        L = PTX_LDS_AT(uint16_t, o_ + (uint32_t)ptx_a16(4u * (n + 3u)));                     \
        klist = L;                                                                           \
        bigp = PTX_LDS_AT(uint16_t, o_ + (uint32_t)ptx_a16(4u * (n + 3u)) + (uint32_t)ptx_a16(2u * (n + 2u))); \
        auxw = PTX_LDS_AT(uint32_t, o_ + (uint32_t)ptx_a16(4u * (n + 3u)) + (uint32_t)ptx_a16(2u * (n + 2u)) + (uint32_t)ptx_a16(2u * (n / (PTX_SMALL_BUCKET + 1u) + 2u))); \
        plist = (uint16_t*)auxw;                                                             \
        hb = (PtxBitWord*)auxw;                                                              \
        /* ... and what P4 .. P6 keep in its place */                                        \
        alive = PTX_LDS_AT(PtxBitWord, o_);                                                  \
        o_ += (uint32_t)ptx_a16(8u * (nwe + 1u));                                            \
        brkbits = PTX_LDS_AT(uint32_t, o_);                                                  \
        o_ += (uint32_t)ptx_a16(4u * (nwe + 1u));                                            \
        mrk_lo = PTX_LDS_AT(uint16_t, o_);                                                   \
        o_ += (uint32_t)ptx_a16(2u * (K + 1u));                                              \
        mrk_hi = PTX_LDS_AT(uint16_t, o_);                                                   \
        o_ += (uint32_t)ptx_a16(2u * (K + 1u));                                              \
        cid = PTX_LDS_AT(uint16_t, o_);                                                      \
    } while (0)

    /* P3a's loads, declared here because its first step is issued as soon as P1 has completed the lists (its latency then hides
     * behind the duplicate check and the prefix scan of the id bitmap) */
    uint32_t p3_i[PTX_U], p3_di[PTX_U], p3_dq[PTX_U];
    uint64_t p3_id[PTX_U], p3_ra[PTX_U], p3_dra[PTX_U];
    /* the park entries of this thread's deletes of a step (coalesced; issued TWO steps ahead: the gathers through them are a second trip) */
#define PTX_P3A_DQ(st_, dq_)                                                \
    _Pragma("unroll") for (int u = 0; u < PTX_U; ++u) {                     \
        const uint32_t j_ = PTX_J_OF(st_, u);                               \
        dq_[u] = ptx_coherent_load32(&park[j_ < d_fused ? PTX_JX(j_, D) : 0u]); \
    }
    /* rows of this thread's inserts and deletes of a step (list reads, then the column gathers) */
#define PTX_P3A_LOAD(st_, i_, id_, ra_, di_, dra_, dq_)                     \
    _Pragma("unroll") for (int u = 0; u < PTX_U; ++u) {                     \
        const uint32_t j_ = PTX_J_OF(st_, u);                               \
        const uint32_t s_ = j_ < n ? PTX_JX(j_, n) : 0u;                    \
        const uint32_t r_ = ilist[s_];                                      \
        i_[u] = r_ < N ? r_ : N - 1u;                                       \
        if (small_keys) id_[u] = klist[s_];                                 \
        const uint32_t dr_ = dq_[u] & 0xFFFFu;                              \
        di_[u] = dr_ < N ? dr_ : N - 1u;                                    \
    }                                                                       \
    _Pragma("unroll") for (int u = 0; u < PTX_U; ++u) {                     \
        if (!small_keys) id_[u] = op_id[i_[u]];                             \
        ra_[u] = ref_a[i_[u]];                                              \
        dra_[u] = ref_a[di_[u]];                                            \
    }
    /* ---- P1: ONE pass over the rows: id bitmaps, row lists per class ---- */
    {
        uint32_t err4 = 0, ctr_hi = 0, act_hi = 0; /* malformed class bytes; max counter - 1 and max actor met */
        if (!p1_loaded) { PTX_P1_LOAD(PTX_G_OF(0u, p1_steps), id, a4, mt4) } /* (no admission phase ahead: the first rows go out here, while the bitmaps are cleared) */
        /* during this pass ib[w] = {ids of the inserts, ids of ALL ops (duplicate detection)}: one 8-byte LDS atomic per row */
        PTX_FOR(w, nw + 1) {
            PtxBitWord z;
            z.bits = 0;
            z.pre = 0;
            ix.ib[w] = z;
        }
        PTX_FOR(w, nwe + 1) delbits[w] = 0;
        PTX_FOR(w, (K >> 5) + 1) maddbits[w] = 0;
        if (out_rank) { /* the insert rows are overwritten in P5a, by other threads: these stores must have completed by then (the one full barrier) */
            PTX_FOR(i, N) out_rank[base + i] = 0xFFFFFFFFu;
            PTX_SYNC_FULL();
        }
        /* The insert list's cursor is ABSOLUTE: an index of 16-bit words from the start of the log's LDS window; the cursors of the deletes and of the four
         * mark types are entries of the park (deletes from 0, mark type t from D + moff_t).  A row's slot is one number whatever its class; rows that are
         * listed nowhere (makeList, NOP, map ops, malformed) store nothing. */
        uint16_t* const lds16 = (uint16_t*)lds;
        const uint32_t i_at = (uint32_t)(ilist - lds16);
        const uint32_t k_delta = (uint32_t)(klist - ilist);   /* the key of an insert sits this far behind its list entry */
        const uint32_t top16 = A.lds_bytes / 2u - 1u;          /* last 16-bit word of the window */
        PTX_LEADER {
            /* class 0 insert -> ilist (LDS), 1 delete -> park[0 ..), 2..5 mark type 0..3 -> its range of park[mp0 ..) */
            H->cur[0] = i_at;
            H->cur[1] = 0u;
            H->cur[2] = mp0;
            H->cur[3] = mp0 + moff1;
            H->cur[4] = mp0 + moff2;
            H->cur[5] = mp0 + moff3;
            H->cur[6] = H->cur[7] = 0; /* cur[7]: some row is malformed */
            H->n_ins = n;
            H->n_applied = n + D + K;
        }
        PTX_SYNC_LDS();
        if (kHeadRow && p1_head) { /* the makeList the pass leaves out: its id's bit in the duplicate bitmap, its id in the bounds check (the leader's registers) */
            PTX_LEADER {
                const uint64_t id0 = op_id[0];
                const uint32_t ctr = (uint32_t)(id0 >> 32), act = (uint32_t)id0;
                ctr_hi = ctr - 1u;
                act_hi = act;
                const uint32_t key = ptx_min(ptx_mad24_su(ctr, ix.na1, act), keyspace - 1u);
                ptx_atomic_or64((unsigned long long*)&ix.ib[key >> 5], (unsigned long long)(1u << (key & 31u)) << 32);
            }
        }
        /* Branch-free row loop, PTX_U1 consecutive rows per thread and step.  What the header promised is NOT re-checked per
         * row: a list that overflows (more rows of a class than the header says) overwrites scratch of this log only — every
         * store is kept inside the log's window — and the census check after the pass rejects the log; a row with a malformed
         * action / mark type / op id only raises a flag here, and the (rare) pass below names the first such row. */
        /* the work on one thread's rows; kMasked: only the first `nv` of them exist */
        auto p1_rows = [&](auto masked, uint32_t g, uint32_t nv, const uint64_t (&id)[PTX_U1], uint32_t a4, uint32_t mt4) {
            constexpr bool kMasked = decltype(masked)::value;
            const uint32_t r0 = g * PTX_U1 + (kHeadRow ? p1_head : 0u); /* the thread's first row, as the log numbers it */
            constexpr uint32_t kAllRows = PTX_U1 == 4 ? 0xFFFFFFFFu : 0x00FFFFFFu;
            const uint32_t live = kMasked ? (nv >= 4u ? 0xFFFFFFFFu : (1u << (8u * nv)) - 1u) : kAllRows; /* bytes of a4 / mt4 that are rows of this thread */
            /* class of the rows, four bytes at a time (byte permutes as table look-ups):
             * action 0 makeList -> 6, 1 insert -> 0, 2 delete -> 1, 3 / 4 add / removeMark -> 2 + mark type, 5 nop / 6, 7 map ops -> 6 */
            const uint32_t k4 = ptx_perm(0x06060680u, 0x80010006u, a4 & 0x07070707u); /* 0x80: a mark op; 6: listed nowhere (makeList, NOP, map ops) */
            const uint32_t mk = (k4 >> 7) & 0x01010101u;     /* 1 in the bytes of the mark ops */
            /* 0xFF there: three rows' bytes fit the 24-bit multiply; with four, byte by byte (0x80 - 0x01 = 0x7F, or-ed with the 0x80: no borrow crosses a byte) */
            const uint32_t mmask = PTX_U1 == 3 ? ptx_mul24(mk, 255u) : ((k4 & 0x80808080u) | ((k4 & 0x80808080u) - mk));
            uint32_t c4 = ((k4 & ~mmask) | (((mt4 & 0x03030303u) + 0x02020202u) & mmask)) & 0x07070707u;
            /* unknown action (beyond the table), mark type beyond 3: flagged; the row runs on under some class */
            err4 |= ((a4 & 0xF8F8F8F8u) | (mt4 & 0xFCFCFCFCu & mmask)) & live;
            if (kMasked) c4 = (c4 & live) | (0x07070707u & ~live);
            const uint32_t add4 = a4 & mk & live; /* bit 0 tells PTX_ACT_ADDMARK (3) from PTX_ACT_REMOVEMARK (4) */
            uint32_t slot[PTX_U1];
            ptx_wave_slots4<PTX_U1>(H->cur, 0u, c4, slot);
#pragma unroll
            for (int u = 0; u < PTX_U1; ++u) {
                const uint32_t i = r0 + (uint32_t)u;
                const bool in = !kMasked || (uint32_t)u < nv;
                const uint32_t c = (c4 >> (8u * (uint32_t)u)) & 255u;
                const uint32_t ctr = (uint32_t)(id[u] >> 32), act = (uint32_t)id[u];
                if (in) {
                    ctr_hi = ctr - 1u > ctr_hi ? ctr - 1u : ctr_hi; /* a counter of 0 wraps to the top */
                    act_hi = act > act_hi ? act : act_hi;
                }
                /* both id bitmaps in one atomic; a key beyond the header's bounds (flagged above) is kept inside the bitmap */
                const uint32_t key = ptx_min(ptx_mad24_su(ctr, ix.na1, act), keyspace - 1u);
                const uint32_t bit = 1u << (key & 31u);
                if (in) ptx_atomic_or64((unsigned long long*)&ix.ib[key >> 5], (unsigned long long)(c == 0u ? bit : 0u) | ((unsigned long long)bit << 32));
                if (c == 0u) { /* an insert: row and key into the LDS lists (anywhere inside the window when the header understates the rows) */
                    const uint32_t sl = ptx_min(slot[u], top16);
                    PTX_LDS_WILD_STORE16(&lds16[sl], i);
                    if (small_keys) PTX_LDS_WILD_STORE16(&lds16[ptx_min(sl + k_delta, top16)], key);
                } else if (c <= 5u) { /* a delete or a mark op: row | key << 16 into the park (anywhere inside the log's own span rows ...) */
                    park[ptx_min(slot[u], park_top)] = i | (small_keys ? key << 16 : 0u);
                }
                if ((add4 >> (8u * (uint32_t)u)) & 1u) {
                    const uint32_t k = ptx_min(slot[u] - mp0, K); /* K: the spare bit */
                    ptx_atomic_or(&maddbits[k >> 5], 1u << (k & 31u));
                }
            }
        };
        /* two register sets in turn (no copies from "next" to "current"): the even steps' rows arrive in id / a4 / mt4, the odd ones' in id_n / a4_n / mt4_n;
         * a wave that has no row left in a step (wave-uniform) has none in the later ones either */
#define PTX_P1_STEP(st_, id_, a_, mt_, idn_, an_, mtn_)                                                                      \
    {                                                                                                                        \
        const uint32_t g_ = PTX_G_OF(st_, p1_steps);                                                                          \
        if (PTX_WAVE_FIRST(g_) >= p1_groups) break;                                                                           \
        PTX_P1_LOAD(PTX_G_OF((st_) + 1u, p1_steps), idn_, an_, mtn_) /* the next step's rows are in flight while this one is processed */ \
        if (PTX_WAVE_FIRST(g_) + PTX_WS <= p1_full) p1_rows(std::false_type(), g_, PTX_U1, id_, a_, mt_);                     \
        else p1_rows(std::true_type(), g_, g_ * PTX_U1 < p1_N ? (p1_N - g_ * PTX_U1 < PTX_U1 ? p1_N - g_ * PTX_U1 : PTX_U1) : 0u, id_, a_, mt_); \
    }
#pragma nounroll
        for (uint32_t st = 0; st < p1_steps; st += 2u) {
            PTX_P1_STEP(st, id, a4, mt4, id_n, a4_n, mt4_n)
            if (st + 1u >= p1_steps) break;
            PTX_P1_STEP(st + 1u, id_n, a4_n, mt4_n, id, a4, mt4)
        }
#undef PTX_P1_STEP
#undef PTX_P1_LOAD
        if (err4 != 0u || ctr_hi >= ix.max_ctr || act_hi > ix.max_actor) ptx_atomic_or(&H->cur[7], 1u);
        /* the ONE full barrier of the list phase: the park entries were stored by whichever thread met the row and are read by others (P3a: the deletes,
         * P5: the marks), so the stores to HBM must have completed.  A header that understates the rows of a class parks junk: the census check below
         * rejects that log before anything is made of it. */
        PTX_SYNC_FULL();
        PTX_STAMP(11); /* end of the row loop; census, duplicate check and the prefix scan of the id bitmap follow */
        PTX_REMAT_AT(0, PTX_REMAT());
        PTX_P3A_DQ(0u, p3_dq) /* the lists are complete: the deletes of P3a's first step are on their way */
        if (H->cur[7] != 0u) {
            /* some row is malformed (unknown action or mark type, op id of counter 0 or beyond the header's bounds): the first one
             * in log order is the log's error — the rare path, one row per thread and step.  The malformed rows were listed under
             * some class by the pass above, so the census is taken again here, without them. */
            uint32_t* cnt6 = H->scan_tmp;
            PTX_LEADER {
                for (int c = 0; c < 6; ++c) cnt6[c] = 0;
            }
            PTX_SYNC_LDS();
            PTX_FOR(i, N) {
                const uint32_t ctr = (uint32_t)(op_id[i] >> 32), act = (uint32_t)op_id[i], a = action[i], mt = mark_type[i];
                const bool mark = a == PTX_ACT_ADDMARK || a == PTX_ACT_REMOVEMARK;
                if (a >= 8u || (mark && mt > 3u) || ctr - 1u >= ix.max_ctr || act > ix.max_actor) ptx_raise(H, i, 1, PTX_ERR_BAD_OP);
                else if (a == PTX_ACT_INSERT) ptx_atomic_add(&cnt6[0], 1u);
                else if (a == PTX_ACT_DELETE) ptx_atomic_add(&cnt6[1], 1u);
                else if (mark) ptx_atomic_add(&cnt6[2u + mt], 1u);
            }
            PTX_SYNC_LDS();
            PTX_LEADER {
                if (cnt6[0] != n || cnt6[1] != D || cnt6[2] != moff1 || cnt6[3] != moff2 - moff1 || cnt6[4] != moff3 - moff2 || cnt6[5] != K - moff3)
                    ptx_raise(H, 0, 0, PTX_ERR_BAD_OP);
            }
        } else {
            PTX_LEADER {
                /* the header must be the exact census of the rows */
                if (H->cur[0] != i_at + n || H->cur[1] != D || H->cur[2] != mp0 + moff1 || H->cur[3] != mp0 + moff2 || H->cur[4] != mp0 + moff3 || H->cur[5] != mp0 + K)
                    ptx_raise(H, 0, 0, PTX_ERR_BAD_OP);
            }
        }
        PTX_P3A_LOAD(0u, p3_i, p3_id, p3_ra, p3_di, p3_dra, p3_dq) /* P3a's first step: the gathers go out behind the duplicate check and the prefix scan */
        PTX_P3A_DQ(1u, p3_dq)
        {
            uint32_t distinct = 0;
            PTX_FOR(w, nw + 1) distinct += ptx_popc(ix.ib[w].pre);
            ptx_atomic_add(&H->cur[6], distinct);
        }
        PTX_SYNC_LDS();
        if (H->err == PTX_NO_ERR && H->cur[6] != N) {
            /* some opId occurs twice (every row had a well-formed id, so N distinct ids were expected): find the
             * first repeated row with a second, returning pass over a cleared bitmap — the rare path */
            PTX_SYNC_LDS();
            PTX_FOR(w, nw + 1) ix.ib[w].pre = 0;
            PTX_SYNC_LDS();
            PTX_FOR(i, N) {
                uint32_t key = 0;
                ptx_id_key(ix, op_id[i], key);
                const uint32_t bit = 1u << (key & 31);
                if (ptx_atomic_or(&ix.ib[key >> 5].pre, bit) & bit) ptx_raise(H, i, 1, PTX_ERR_DUPLICATE_OP);
            }
            /* (which of two equal ids is "the repeat" depends on the race; the status is what is reported) */
        }
        PTX_SYNC_LDS();
        ptx_bitwords_prefix<kThreads>(ix.ib, nw + 1, &H->scan_tmp[16]);
    }
    PTX_BAIL_IF_ERROR();
    PTX_STAMP(2);
    PTX_REMAT_AT(1, PTX_REMAT());

    /* P5a's loads.  The rows of the mark ops come straight from their park (the low halves of its entries from mp0 on; a lane's entries of a block are
     * consecutive words): the park entries of the first step go out at the start of P3c (two LDS-only phases ahead of their use: with nine logs per CU a trip to HBM takes 10-20 k cycles, and
     * P4 — 5 k cycles of LDS work — used to stand waiting for them), its gathers ahead of P4; in the loop
     * the park entries run two steps ahead of the step in work, the gathers one.  (Round 4: copying the list back into LDS first was ONE exposed trip to HBM per
     * log — 2 k cycles with a CU to itself, 28 k under load.) */
    PtxMarkBlocks MB = ptx_mark_blocks(moff1, moff2, moff3, K, PTX_JB_CAP);
    uint32_t m_steps = PTX_JB_STEPS(MB.B, PTX_UM);
#ifdef PTX_JB_LANE_IS_FIXED
    PtxMarkLaneKept MLK = ptx_mark_lane_kept(MB, PTX_JB_LANE(0u, 0, PTX_UM));
#define PTX_MARK_OF(st_, u_, k_) ptx_mark_of_kept(MLK, PTX_JB_BLOCK(st_, u_, PTX_UM), k_)
#else
#define PTX_MARK_OF(st_, u_, k_) ptx_mark_of(MB, PTX_JB_BLOCK(st_, u_, PTX_UM), PTX_JB_LANE(st_, u_, PTX_UM), k_)
#endif
    uint32_t kq[PTX_UM], kq_n[PTX_UM]; /* the thread's mark ops of a step (0xFFFFFFFF: none) */
    uint32_t mq[PTX_UM];               /* the park entries of the step whose gathers go out next */
#define PTX_MARK_PQ(st_, mq_)                                               \
    _Pragma("unroll") for (int u = 0; u < (int)PTX_UM; ++u) {               \
        uint32_t k_;                                                        \
        (void)PTX_MARK_OF(st_, u, k_);                                       \
        mq_[u] = ptx_coherent_load32(&park[K ? mp0 + k_ : 0u]);             \
    }
    uint32_t i[PTX_UM], sa[PTX_UM], sb[PTX_UM], pl[PTX_UM], i_n[PTX_UM], sa_n[PTX_UM], sb_n[PTX_UM], pl_n[PTX_UM];
    uint64_t ra[PTX_UM], rb[PTX_UM], ra_n[PTX_UM], rb_n[PTX_UM];
    /* rows of this thread's mark ops of a step (list read, then the column gathers; the payload only of the comment ops — their
     * id —, the others' is not needed before P5b, and then only the winners') */
#define PTX_MARK_LOAD(st_, kq_, i_, ra_, rb_, sa_, sb_, pl_, mq_)           \
    _Pragma("unroll") for (int u = 0; u < (int)PTX_UM; ++u) {               \
        uint32_t k_;                                                        \
        const bool has_ = PTX_MARK_OF(st_, u, k_);                         \
        kq_[u] = has_ ? k_ : 0xFFFFFFFFu;                                   \
        const uint32_t r_ = mq_[u] & 0xFFFFu;                               \
        i_[u] = r_ < N ? r_ : N - 1u;                                       \
        pl_[u] = has_ && k_ >= moff2 && k_ < moff3 ? 1u : 0u;               \
    }                                                                       \
    _Pragma("unroll") for (int u = 0; u < (int)PTX_UM; ++u) {               \
        rb_[u] = ref_b[i_[u]];                                              \
        ra_[u] = PTX_KO_P5A_RA ? rb_[u] : ref_a[i_[u]];                     \
        sa_[u] = side_a[i_[u]];                                             \
        sb_[u] = side_b[i_[u]];                                             \
        if (pl_[u]) pl_[u] = payload[i_[u]];                                \
    }
    /* ---- P3: causal tree of the inserts -> document position of every element ---- */
    {
        /* children per parent -> bucket starts -> bucket ends; 16-bit counters, two per atomically updated word */
        uint16_t* cnt = (uint16_t*)cntw;
        uint16_t* srt = ilist;       /* children of every parent, descending opId, parents ascending (ilist is dead after P3b) */
        uint16_t* seg = L;           /* bucket members in arrival order (klist is dead after P3b's checks) */
        uint16_t* nx = seg;          /* P3d: the successors (seg is dead by then: srt holds the sorted buckets) */
        uint32_t* R = RW;            /* the list ranking's {next, weight} words (srt and cnt are dead by then) */
#define PTX_REMAT_P3()           \
    do {                         \
        PTX_REMAT();             \
        cnt = (uint16_t*)cntw;   \
        srt = ilist;             \
        seg = L;                 \
        nx = seg;                \
        R = RW;                  \
    } while (0)

        PTX_FOR(p, (n + 2 + 1) / 2 + 1) cntw[p] = 0;
        PTX_SYNC_LDS();
        /* P3a: element index of every insert, its parent, children counts.  The deletes ride along: their gathers of ref_a hit
         * the lines the inserts of the same stretch of the log have just brought in (one trip to HBM instead of two), and their
         * round trips hide behind the inserts'.  What a delete cannot do yet is the application-order check (row_of is being
         * written): the thread leaves the target element in the slot of `ilist` and the delete's row in the slot of `klist` it has
         * just consumed — same thread, same index, no hazard — and the check runs below, from LDS alone.  Deletes beyond slot n
         * (more deletes than inserts) go the old way. */
        {
            const uint32_t jmax = n > d_fused ? n : d_fused;
            const uint32_t steps = PTX_JSTEPS(jmax);
            /* two register sets in turn (no copies from "next" to "current"): p3_* (step 0 was loaded at the end of P1) holds the even steps, *_n the odd ones;
             * p3_dq: the park entries of the deletes of the step whose gathers go out next */
            uint32_t i_n[PTX_U], di_n[PTX_U];
            uint64_t id_n[PTX_U], ra_n[PTX_U], dra_n[PTX_U]; /* id: the op id, or (small_keys) just its key from klist */
            auto p3a_step = [&](uint32_t st, const uint32_t (&i)[PTX_U], const uint64_t (&id)[PTX_U], const uint64_t (&ra)[PTX_U], const uint32_t (&di)[PTX_U],
                                const uint64_t (&dra)[PTX_U]) {
                /* the three id look-ups of an item (own key, parent, the delete's target) and those of the step's other items first, side by side: one LDS round
                 * trip for all of them, then the stores that depend on them (nested ifs ran them one after the other) */
                uint32_t e_[PTX_U];
                int p_[PTX_U], t_[PTX_U];
#pragma unroll
                for (int u = 0; u < PTX_U; ++u) {
                    uint32_t key = (uint32_t)id[u];
                    if (!small_keys) ptx_id_key(ix, id[u], key);
                    e_[u] = ptx_bitrank(ix.ib, key < keyspace ? key : 0u);
                    p_[u] = ptx_elem_lookup24(ix, ra[u]);
                    t_[u] = ptx_elem_lookup24(ix, dra[u]);
                }
#pragma unroll
                for (int u = 0; u < PTX_U; ++u) {
                    const uint32_t j = PTX_J_OF(st, u);
                    if (j < n) {
                        const uint32_t e = e_[u];
                        row_of[e] = (uint16_t)i[u];
                        uint32_t pe = n;
                        if (ra[u] != 0) {
                            const int p = p_[u];
                            if (p < 0) ptx_raise(H, i[u], 1, PTX_ERR_ELEM_NOT_FOUND); /* micromerge.ts:752 */
                            else pe = (uint32_t)p;
                        }
                        par[e] = (uint16_t)pe;
                        ptx_atomic_add(&cntw[pe >> 1], 1u << (16u * (pe & 1u)));
                    }
                    if (j < d_fused) {
                        /* the element must exist when the delete is applied (micromerge.ts:752); deleting twice is fine (:693) */
                        const int t = t_[u];
                        if (t < 0) ptx_raise(H, di[u], 1, PTX_ERR_ELEM_NOT_FOUND);
                        else ptx_atomic_or(&delbits[(uint32_t)t >> 5], 1u << ((uint32_t)t & 31u));
                        const uint32_t sl = j < n ? PTX_JX(j, n) : j; /* j <= n; slot n is P1's spare */
                        ilist[sl] = (uint16_t)(t < 0 ? 0xFFFF : t);
                        klist[sl] = (uint16_t)di[u];
                    }
                }
            };
#pragma nounroll
            for (uint32_t st = 0; st < steps; st += 2u) {
                PTX_P3A_LOAD(st + 1u, i_n, id_n, ra_n, di_n, dra_n, p3_dq) /* in flight while this step is processed */
                PTX_P3A_DQ(st + 2u, p3_dq)
                p3a_step(st, p3_i, p3_id, p3_ra, p3_di, p3_dra);
                if (st + 1u >= steps) break;
                PTX_P3A_LOAD(st + 2u, p3_i, p3_id, p3_ra, p3_di, p3_dra, p3_dq)
                PTX_P3A_DQ(st + 3u, p3_dq)
                p3a_step(st + 1u, i_n, id_n, ra_n, di_n, dra_n);
            }
#undef PTX_P3A_LOAD
#undef PTX_P3A_DQ
        }
        PTX_BAIL_IF_ERROR();
        PTX_STAMP(12); /* end of P3a */
        PTX_REMAT_AT(2, PTX_REMAT_P3());
        /* the deletes' application-order check, now that row_of is complete: target element and row left in ilist / klist by P3a */
        PTX_FORV(j0, d_fused, PTX_UV) {
            uint32_t t[PTX_UV], i[PTX_UV], rt[PTX_UV];
#pragma unroll
            for (int u = 0; u < PTX_UV; ++u) {
                const uint32_t j = PTX_IN(j0, u) ? PTX_IX(j0, u) : 0u;
                const uint32_t sl = j < n ? PTX_JX(j, n) : j;
                t[u] = PTX_IN(j0, u) ? ilist[sl] : 0xFFFFu;
                i[u] = klist[sl];
            }
#pragma unroll
            for (int u = 0; u < PTX_UV; ++u) rt[u] = row_of[t[u] != 0xFFFFu ? t[u] : 0u];
#pragma unroll
            for (int u = 0; u < PTX_UV; ++u)
                if (t[u] != 0xFFFFu) {
                    if (rt[u] >= i[u]) ptx_raise(H, i[u], 1, PTX_ERR_ELEM_NOT_FOUND);
                    if (out_refs && i[u] < N) out_refs[base + i[u]] = rt[u];
                }
        }
        if (D > d_fused) { /* more deletes than inserts + 1: the rest, two trips each (park, then the column) */
            PTX_FOR(jj, D - d_fused) {
                const uint32_t j = d_fused + jj;
                const uint32_t r = ptx_coherent_load32(&park[PTX_JX(j, D)]) & 0xFFFFu, i = r < N ? r : N - 1u;
                const int t = ptx_elem_lookup24(ix, ref_a[i]);
                if (t < 0 || row_of[t] >= i) ptx_raise(H, i, 1, PTX_ERR_ELEM_NOT_FOUND);
                else {
                    ptx_atomic_or(&delbits[(uint32_t)t >> 5], 1u << ((uint32_t)t & 31u));
                    if (out_refs) out_refs[base + i] = row_of[t];
                }
            }
        }
        ptx_scan_excl<uint16_t, 1, kThreads>(cnt, n + 2, H->scan_tmp, A.div_magic); /* cnt[p] = first slot of p's children (its barriers stand between the checks above, which read klist, and the scatter below, which writes the same words as seg) */
        /* P3b: scatter into the parent buckets; application-order checks of the inserts now that row_of is complete */
        PTX_FORU(e0, n) {
            uint32_t pe[PTX_U], re[PTX_U], rp[PTX_U];
#pragma unroll
            for (int u = 0; u < PTX_U; ++u)
                if (PTX_IN(e0, u)) {
                    pe[u] = par[PTX_IX(e0, u)];
                    re[u] = row_of[PTX_IX(e0, u)];
                }
#pragma unroll
            for (int u = 0; u < PTX_U; ++u)
                if (PTX_IN(e0, u)) rp[u] = pe[u] < n ? (uint32_t)row_of[pe[u]] : 0u;
#pragma unroll
            for (int u = 0; u < PTX_U; ++u)
                if (PTX_IN(e0, u)) {
                    /* the reference element must already exist when the op is applied (micromerge.ts:752) */
                    if (pe[u] < n && rp[u] >= re[u]) ptx_raise(H, re[u], 1, PTX_ERR_ELEM_NOT_FOUND);
                    /* now cnt[p] = END of p's bucket (no carry between the halves: a counter never exceeds n < 32767) */
                    seg[(ptx_atomic_add(&cntw[pe[u] >> 1], 1u << (16u * (pe[u] & 1u))) >> (16u * (pe[u] & 1u))) & 0xFFFFu] = (uint16_t)PTX_IX(e0, u);
                }
        }
        PTX_LEADER { H->cur_big = H->cur_med = 0; }
        PTX_BAIL_IF_ERROR();
        PTX_STAMP(3);
        PTX_REMAT_AT(3, PTX_REMAT_P3());
        PTX_MARK_PQ(0u, mq) /* P5a's first park entries: on their way while the tree is ordered and ranked (LDS only) */
        /* P3c: the children of every parent in descending element index == descending opId (the skip loop of micromerge.ts:630-635), one PARENT per lane.
         * Pass 1, all parents: an only child (most elements are one) is placed at once, a parent with more goes to a list.  Pass 2, the listed parents: a
         * bucket of up to PTX_SMALL_BUCKET members is sorted in registers by a 19-exchange network; up to PTX_HUGE_BUCKET members by PTX_G lanes per member,
         * larger ones (the children of HEAD in a document everybody types at the start of) through a bitmap over the element indices: rank = members with a
         * larger index.  (The network over ALL parents was measured: 2.6 k vector instructions per log, 10 % of the kernel's, for the fifth that need it.) */
        {
            const uint32_t steps1 = PTX_JSTEPS_U(n + 1u, PTX_UV);
#pragma nounroll
            for (uint32_t st = 0; st < steps1; ++st) { /* every thread runs every step: the list slots come from a wave-wide prefix sum */
                uint32_t pp[PTX_UV], s1[PTX_UV], t1[PTX_UV], x1[PTX_UV], many = 0;
#pragma unroll
                for (int u = 0; u < PTX_UV; ++u) {
                    const uint32_t j = PTX_J_OF_U(st, u, PTX_UV);
                    const bool in = j < n + 1u;
                    pp[u] = in ? PTX_JX(j, n + 1u) : 0u;
                    s1[u] = in && pp[u] ? (uint32_t)cnt[pp[u] - 1u] : 0u;
                    t1[u] = in ? (uint32_t)cnt[pp[u]] : 0u; /* (an item past the end: an empty bucket) */
                }
#pragma unroll
                for (int u = 0; u < PTX_UV; ++u) x1[u] = seg[s1[u] < n ? s1[u] : 0u];
#pragma unroll
                for (int u = 0; u < PTX_UV; ++u) {
                    if (t1[u] - s1[u] == 1u) srt[s1[u]] = (uint16_t)x1[u];
                    many += t1[u] - s1[u] >= 2u ? 1u : 0u;
                }
                uint32_t at = ptx_append_n(&H->cur_med, many);
#pragma unroll
                for (int u = 0; u < PTX_UV; ++u)
                    if (t1[u] - s1[u] >= 2u) plist[at++] = (uint16_t)pp[u];
            }
        }
        PTX_SYNC_LDS();
        PTX_FOR(h, H->cur_med) {
            const uint32_t p = plist[h];
            const uint32_t s = p ? cnt[p - 1] : 0u, t = cnt[p];
            const uint32_t m = t - s;
            static_assert(PTX_SMALL_BUCKET == 8u, "the exchange network below sorts eight");
            if (m <= PTX_SMALL_BUCKET) {
                uint32_t v[8]; /* element index + 1 (0 = no member: sorts to the end) */
#pragma unroll
                for (uint32_t k = 0; k < 8u; ++k) v[k] = k < m ? (uint32_t)seg[s + k] + 1u : 0u;
#define PTX_CE(a_, b_)                                    \
    {                                                     \
        const uint32_t hi_ = v[a_] > v[b_] ? v[a_] : v[b_]; \
        v[b_] = v[a_] > v[b_] ? v[b_] : v[a_];            \
        v[a_] = hi_;                                      \
    }
                PTX_CE(0, 1) PTX_CE(2, 3) PTX_CE(4, 5) PTX_CE(6, 7) PTX_CE(0, 2) PTX_CE(1, 3) PTX_CE(4, 6) PTX_CE(5, 7) PTX_CE(1, 2) PTX_CE(5, 6)
                PTX_CE(0, 4) PTX_CE(1, 5) PTX_CE(2, 6) PTX_CE(3, 7) PTX_CE(2, 4) PTX_CE(3, 5) PTX_CE(1, 2) PTX_CE(3, 4) PTX_CE(5, 6)
#undef PTX_CE
#pragma unroll
                for (uint32_t k = 0; k < 8u; ++k)
                    if (k < m) srt[s + k] = (uint16_t)(v[k] - 1u);
            } else {
                const uint32_t jb = ptx_append(&H->cur_big, true);
                bigp[jb] = (uint16_t)p;
            }
        }
        PTX_SYNC_LDS();
        {
            const uint32_t nb = H->cur_big;
            for (uint32_t h = 0; h < nb; ++h) { /* uniform: nb and the bucket bounds come from LDS after the barrier */
                const uint32_t p = bigp[h];
                const uint32_t s = p ? cnt[p - 1] : 0u, t = cnt[p];
                if (t - s <= PTX_HUGE_BUCKET) { /* a lane per member; the members it is compared with are the same words in every lane, eight reads in flight */
                    PTX_FOR(k, t - s) {
                        const uint32_t x = seg[s + k];
                        uint32_t c = 0, q = s;
#pragma nounroll
                        for (; q + 8u <= t; q += 8u) {
                            uint32_t y[8];
#pragma unroll
                            for (int u = 0; u < 8; ++u) y[u] = seg[q + (uint32_t)u];
#pragma unroll
                            for (int u = 0; u < 8; ++u) c += y[u] > x ? 1u : 0u;
                        }
                        for (; q < t; ++q) c += seg[q] > x ? 1u : 0u;
                        srt[s + c] = (uint16_t)x;
                    }
                    continue;
                }
                PTX_FOR(w, nwe + 1) {
                    PtxBitWord z;
                    z.bits = 0;
                    z.pre = 0;
                    hb[w] = z;
                }
                PTX_SYNC_LDS();
                PTX_FOR(k, t - s) {
                    const uint32_t x = seg[s + k];
                    ptx_atomic_or(&hb[x >> 5].bits, 1u << (x & 31));
                }
                PTX_SYNC_LDS();
                ptx_bitwords_prefix<kThreads>(hb, nwe + 1, &H->scan_tmp[17]);
                PTX_FOR(k, t - s) {
                    const uint32_t x = seg[s + k];
                    srt[s + (t - s - 1u - ptx_bitrank(hb, x))] = (uint16_t)x; /* members with a larger index come first */
                }
                PTX_SYNC_LDS();
            }
        }
        PTX_SYNC_LDS();
        PTX_STAMP(4);
        PTX_REMAT_AT(4, PTX_REMAT_P3());
        /* P3d: document order = pre-order of the tree.  Successor of an element x: its first child; a leaf's: `after(x)` = the next sibling, or — x being the
         * last child — after(parent).  after() of the last children is resolved by pointer jumping up the tree (a few rounds: the chains of last children are
         * short; the rounds stop when a pass finds nothing open); then the successor list of the n + 1 nodes (elements, HEAD = n; the terminal node n + 1)
         * is ranked by in-place pointer jumping over {next << 16 | elements in [node, next)} words. */
        const uint32_t term = n + 1u;
        PTX_FORV(j0, n, PTX_UV) {
            uint32_t x[PTX_UV], sib[PTX_UV], p[PTX_UV], e[PTX_UV];
#pragma unroll
            for (int u = 0; u < PTX_UV; ++u) {
                const uint32_t j = PTX_IN(j0, u) ? PTX_IX(j0, u) : 0u;
                x[u] = srt[j];
                sib[u] = srt[j + 1u < n ? j + 1u : j];
            }
#pragma unroll
            for (int u = 0; u < PTX_UV; ++u) p[u] = par[x[u]];
#pragma unroll
            for (int u = 0; u < PTX_UV; ++u) e[u] = cnt[p[u]];
#pragma unroll
            for (int u = 0; u < PTX_UV; ++u)
                if (PTX_IN(j0, u)) nx[x[u]] = (uint16_t)(PTX_IX(j0, u) + 1u < e[u] ? sib[u] : (p[u] == n ? term : 0x8000u | p[u])); /* 0x8000 | p: open, ask p */
        }
        uint32_t* open3 = &H->cur_big; /* cur_big, cur_med, cur_huge: "some node is still open" of round r in word r % 3.  The leader of round r clears the word of
                                          round r + 1; a thread that is slow to read round r's word cannot find it cleared: that is the leader of round r + 2's
                                          doing, who has passed the barrier of round r + 1 — behind the slow thread's read */
        PTX_LEADER {
            nx[n] = (uint16_t)term;
            open3[0] = 0;
        }
        PTX_SYNC_LDS();
#pragma nounroll
        for (uint32_t r = 0;; ++r) {
            uint32_t open = 0;
            /* branch-free: an item past the end plays HEAD (resolved from the start); a resolved node writes back what it read */
            PTX_FORV(x0, n, PTX_UV) {
                uint32_t ix_[PTX_UV], v[PTX_UV], w[PTX_UV];
#pragma unroll
                for (int u = 0; u < PTX_UV; ++u) {
                    ix_[u] = PTX_IN(x0, u) ? PTX_IX(x0, u) : n;
                    v[u] = nx[ix_[u]];
                }
#pragma unroll
                for (int u = 0; u < PTX_UV; ++u) w[u] = nx[v[u] & 0x7FFFu]; /* (of an open node) resolved, or a node further up: either way a valid state of x */
#pragma unroll
                for (int u = 0; u < PTX_UV; ++u) {
                    const uint32_t nv = (v[u] & 0x8000u) ? w[u] : v[u];
                    nx[ix_[u]] = (uint16_t)nv;
                    open |= nv & 0x8000u;
                }
            }
            if (open) ptx_atomic_or(&open3[r % 3u], 1u);
            PTX_LEADER { open3[(r + 1u) % 3u] = 0; }
            PTX_SYNC_LDS();
            if (!open3[r % 3u]) break;
        }
        PTX_STAMP(14); /* after() stands; the successor list and its ranking follow */
        PTX_REMAT_AT(5, PTX_REMAT_P3());
        PTX_FORV(x0, n + 1, PTX_UV) { /* the successor, in place: the first child where there is one */
            uint32_t s[PTX_UV], t[PTX_UV], f[PTX_UV];
#pragma unroll
            for (int u = 0; u < PTX_UV; ++u) {
                const uint32_t x = PTX_IN(x0, u) ? PTX_IX(x0, u) : 0u;
                s[u] = x ? cnt[x - 1] : 0u;
                t[u] = cnt[x];
            }
#pragma unroll
            for (int u = 0; u < PTX_UV; ++u) f[u] = srt[s[u] < n ? s[u] : 0u];
#pragma unroll
            for (int u = 0; u < PTX_UV; ++u)
                if (PTX_IN(x0, u) && t[u] > s[u]) nx[PTX_IX(x0, u)] = (uint16_t)f[u];
        }
        PTX_SYNC_LDS();
        PTX_STAMP(16); /* the successor list stands */
        PTX_REMAT_AT(6, PTX_REMAT_P3());
        /* List ranking, work-efficient (every pass touches a node once; Wyllie's doubling over all n nodes was measured: 4 x the instructions, no faster alone):
         * every S-th element, and HEAD, is a splitter.  A splitter walks to the next one, counts the elements it passes and leaves on each of them its own
         * index (in nx: only this walker ever reads nx[v], and it just did) and the count before it (in par, dead by now); the splitters are ranked by in-place
         * pointer jumping over {next splitter << 16 | elements of the segment}; one flat pass turns (splitter suffix, local count) into positions.
         * S: 8 while the splitters fit one pass of the workgroup (segments are geometric: mean S, the longest of them bounds the pass). */
        {
            uint32_t lgS = 3u;
            while ((n >> lgS) + 3u > PTX_NTHREADS && lgS < 15u) ++lgS;
            const uint32_t smask = (1u << lgS) - 1u;
            const uint32_t nsE = n ? ((n - 1u) >> lgS) + 1u : 0u; /* splitters 0 .. nsE - 1: the elements sp << lgS; nsE: HEAD; ns = nsE + 1: the terminal node */
            const uint32_t ns = nsE + 1u;
            PTX_FOR(sp, ns) {
                uint32_t v = sp < nsE ? sp << lgS : n, acc = 0;
                for (;;) {
                    const uint32_t nxt = nx[v];
                    if (v < n) {
                        par[v] = (uint16_t)acc; /* elements of this segment strictly before v */
                        nx[v] = (uint16_t)sp;
                        ++acc;
                    }
                    v = nxt;
                    if (v >= n || (v & smask) == 0u) break; /* the terminal node (n + 1; HEAD is nobody's successor) or the next splitter */
                }
                R[sp] = ((v < n ? v >> lgS : ns) << 16) | acc;
            }
            PTX_LEADER { R[ns] = ns << 16; } /* terminal: points at itself with weight 0 */
            PTX_SYNC_LDS();
            PTX_STAMP(17); /* the walks are over */
            const uint32_t rounds = ptx_ceil_log2(ns + 1);
            /* in-place pointer jumping: every intermediate {next, weight} word is a valid state (weight = elements in [node, next)), so reading a word
             * another lane already advanced this round only makes the jump longer.  The few splitters are ONE wave's work: its lanes talk through the LDS
             * in program order, so the rounds need no barrier (the other waves wait at the one below) */
            PTX_ONE_WAVE {
#pragma nounroll
                for (uint32_t r = 0; r < rounds; ++r) {
                    PTX_FOR_LANES(sp, ns) {
                        const uint32_t a = R[sp];
                        const uint32_t b = R[a >> 16];
                        R[sp] = (b & 0xFFFF0000u) | ((a + b) & 0xFFFFu);
                    }
                    PTX_WSYNC();
                }
            }
            PTX_SYNC_LDS();
            /* elements from x to the end of the document = suffix of x's splitter - elements before x in its segment */
            PTX_FORV(x0, n, PTX_UV) { /* document position incl. tombstones */
                uint32_t sp[PTX_UV], lc[PTX_UV], a[PTX_UV];
#pragma unroll
                for (int u = 0; u < PTX_UV; ++u) {
                    const uint32_t x = PTX_IN(x0, u) ? PTX_IX(x0, u) : 0u;
                    sp[u] = nx[x];
                    lc[u] = par[x];
                }
#pragma unroll
                for (int u = 0; u < PTX_UV; ++u) a[u] = R[sp[u] <= ns ? sp[u] : ns];
#pragma unroll
                for (int u = 0; u < PTX_UV; ++u)
                    if (PTX_IN(x0, u)) par[PTX_IX(x0, u)] = (uint16_t)(n - ((a[u] & 0xFFFFu) - lc[u]));
            }
        }
        PTX_SYNC_LDS();
    }
    PTX_STAMP(18); /* document positions stand; the mark list comes back from its park */
    PTX_REMAT_AT(7, PTX_REMAT());
    bp.off = mark_lds; /* release the tree scratch */
    PTX_STAMP(5);
    /* the lane's share of the mark runs, worked out again from the header as it has just been read again: two vector registers that need not be kept — in a build of
     * 64 of them, spilled to scratch memory — through the tree phases for the sake of the one early load above (round 6) */
    MB = ptx_mark_blocks(moff1, moff2, moff3, K, PTX_JB_CAP);
    m_steps = PTX_JB_STEPS(MB.B, PTX_UM);
#ifdef PTX_JB_LANE_IS_FIXED
    MLK = ptx_mark_lane_kept(MB, PTX_JB_LANE(0u, 0, PTX_UM));
#endif

    /* (P5a's loads: the park entries of its first step went out at the start of P3c) their gathers go out here, ahead of P4, and the second step's park entries */
    PTX_MARK_LOAD(0u, kq, i, ra, rb, sa, sb, pl, mq)
    PTX_MARK_PQ(1u, mq)
    /* ---- P4: tombstones -> visible index ---- */
    const uint32_t nwv = nwe; /* bit positions 0..n by document position */
    alive = ptx_alloc<PtxBitWord>(bp, nwv + 1);
    brkbits = ptx_alloc<uint32_t>(bp, nwe + 1); /* visible positions where a comment interval starts/ends */
    PTX_BAIL_CAPACITY();
    PTX_FOR(w, nwv + 1) {
        PtxBitWord z;
        z.bits = 0;
        z.pre = 0;
        alive[w] = z;
        brkbits[w] = 0;
    }
    PTX_SYNC_LDS();
    /* Over the SURVIVING elements.  Sparse documents (a document that has seen thousands of ops usually shows a few dozen characters) go word by word over
     * the tombstone bitmap, a lane per 32 elements, so that the work is per survivor and not per element ever inserted; documents that keep a good share of
     * their elements go element by element (a lane would loop over up to 32 survivors one after the other). */
#define PTX_LIVE_WORD(w_) ((~delbits[w_]) & ((w_) == (n >> 5) ? (1u << (n & 31u)) - 1u : 0xFFFFFFFFu))
    auto for_live = [&](bool sparse, auto body) {
        if (sparse) {
            PTX_FOR(w, nwe) {
                uint32_t live = PTX_LIVE_WORD(w);
                while (live) {
                    const uint32_t e = (w << 5) + (uint32_t)__builtin_ctz(live);
                    live &= live - 1u;
                    body(e);
                }
            }
        } else {
            PTX_FOR(e, n) {
                if (!ptx_bittest(delbits, e)) body(e);
            }
        }
    };
    for_live((n > D ? n - D : 0u) * 8u < n, [&](uint32_t e) { /* (n - D: the least number of survivors the header allows) */
        const uint32_t r = rnk[e];
        ptx_atomic_or(&alive[r >> 5].bits, 1u << (r & 31));
    });
    PTX_SYNC_LDS();
    const uint32_t V = ptx_bitwords_prefix<kThreads>(alive, nwv + 1, &H->scan_tmp[18]);
    /* the visible interval [lo, hi) of every mark op (the few later uses of an op's row — the ops that still cover a visible character — read it from the
     * park).  Until the marks are looked at, the space of `hi` and of the comment ids holds the rows of the visible elements (vrow) if they fit */
    mrk_lo = ptx_alloc<uint16_t>(bp, K + 1);
    const uint32_t mrk_at = bp.off;
    mrk_hi = ptx_alloc<uint16_t>(bp, K + 1);
    cid = ptx_alloc<uint16_t>(bp, Kc + 1);    /* comment mark -> doc-local comment id */
    PTX_BAIL_CAPACITY();
    const uint32_t mrk_bytes = bp.off - mrk_at; /* the two arrays stand back to back; nothing is stored in them before the marks are looked at */
    PTX_STAMP(6);
    PTX_REMAT_AT(8, PTX_REMAT());

    /* ---- P5a: visible values out; every mark op -> visible interval [lo, hi) ---- */
    /* (the one-wave build — bound by vector issue — carries a thread's digest share of the values on to the spans and flushes ONCE: a wave-wide 64-bit sum is ~70
     * vector instructions, three of them were 7 % of a 256-op log's; the builds of more waves keep their flush per pass: four more live VGPRs through P5a cost them more) */
    constexpr bool kOneFlush = kThreads == 64u;
    uint64_t carry_h1 = 0, carry_h2 = 0;
    {
        uint64_t h1 = 0, h2 = 0;
        /* the surviving elements again (a lane per word of the tombstone bitmap): their rows by visible index (= alive bits below the element's position) into
         * a dense list — the digest is ~100 vector instructions per item and wave, so it runs over a packed list (one wave pass per 64 visible characters),
         * never inside the sparse loop */
        const bool sparse = V * 8u < n;
        if (2u * (V + 1u) <= mrk_bytes) {
            uint16_t* vrow = mrk_hi;
            PTX_LDS_ALLOCATED(mrk_hi, mrk_bytes, mrk_bytes); /* (the list runs over the two arrays and the padding between them) */
            for_live(sparse, [&](uint32_t e) { vrow[ptx_bitrank(alive, rnk[e])] = row_of[e]; });
            PTX_SYNC_LDS();
            PTX_FOR(q, V) {
                const uint32_t row = vrow[q];
                const uint32_t v = payload[row < N ? row : N - 1u];
                out_values[q] = v;
                ptx_digest_item(h1, h2, 1u, q, v, 0u);
            }
        } else { /* (a log with hardly any mark op and many visible characters: no room for the list) */
            for_live(sparse, [&](uint32_t e) {
                const uint32_t row = row_of[e], q = ptx_bitrank(alive, rnk[e]);
                const uint32_t v = payload[row < N ? row : N - 1u];
                out_values[q] = v;
                ptx_digest_item(h1, h2, 1u, q, v, 0u);
            });
        }
        /* elem_rank (optional output): document position + tombstone flag of EVERY element, by the row that inserted it */
        if (out_rank) {
            PTX_FOR(e, n) {
                const uint32_t r = rnk[e];
                out_rank[base + row_of[e]] = r | (ptx_bittest(delbits, e) ? PTX_RANK_TOMBSTONE : 0u);
            }
        }
        if (kOneFlush) {
            carry_h1 = h1;
            carry_h2 = h2;
        } else if (V >= 48u) { /* uniform: most lanes of a wave carry a share */
            if (kThreads == 64u || kThreads == 128u) ptx_digest_flush_dense_dpp(H, h1, h2);
            else ptx_digest_flush_dense(H, h1, h2);
        }
        else ptx_digest_flush(H, h1, h2);
        PTX_SYNC_LDS(); /* vrow (= the head of mrk_hi) has been read by everyone before the first interval is stored */
    }
#undef PTX_LIVE_WORD
    PTX_STAMP(13); /* the values are out; the marks' intervals follow */
    PTX_REMAT_AT(9, PTX_REMAT());
    {
    /* two register sets in turn (no copies from "next" to "current"): kq / i / ra ... hold the even steps' mark ops, the *_n set the odd ones' */
    auto mark_step = [&](const uint32_t (&kq)[PTX_UM], const uint32_t (&i)[PTX_UM], const uint64_t (&ra)[PTX_UM], const uint64_t (&rb)[PTX_UM], const uint32_t (&sa)[PTX_UM],
                         const uint32_t (&sb)[PTX_UM], const uint32_t (&pl)[PTX_UM]) {
#pragma unroll
        for (int u = 0; u < (int)PTX_UM; ++u)
            if (kq[u] != 0xFFFFFFFFu) {
                const uint32_t k = kq[u];
                /* Both boundaries side by side, no branch: the two chains id -> element -> (row, position) -> visible rank are three dependent LDS round trips
                 * together (written as nested ifs they were eight, one after the other: round 6 found the kernel bound by its chain of fixed latencies, not by
                 * instruction issue).  start: only before/after(elem) can ever match a slot (peritext.ts:236); an element that is not in the list when the op
                 * is applied means the op never starts; an end that is not found is never reached: the mark runs to the end of the text. */
                const int js0 = ptx_elem_lookup24(ix, ra[u]), je0 = ptx_elem_lookup24(ix, rb[u]);
                const uint32_t ea = js0 < 0 ? 0u : (uint32_t)js0, eb = je0 < 0 ? 0u : (uint32_t)je0;
                const uint32_t row_a = row_of[ea], row_b = row_of[eb], rk_a = rnk[ea];
                uint32_t rk_b = rnk[eb];
                PTX_KEEP_VGPR(rk_b); /* (read now, with the other three: left alone the compiler sinks this read behind the test of the end's row) */
                const bool has_a = (sa[u] <= PTX_SIDE_AFTER) & (js0 >= 0) & (row_a < i[u]); /* (& not &&: a short-circuit puts the LDS reads back into branches) */
                const bool has_b = (sb[u] <= PTX_SIDE_AFTER) & (je0 >= 0) & (row_b < i[u]);
                const uint32_t slot_a = 2u * rk_a + (sa[u] == PTX_SIDE_AFTER ? 1u : 0u);
                uint32_t slot_b = has_b ? 2u * rk_b + (sb[u] == PTX_SIDE_AFTER ? 1u : 0u) : 0xFFFFFFFFu;
                /* same slot: the start test fires first and the end is never seen */
                if (slot_b == slot_a) slot_b = 0xFFFFFFFFu;
                const bool covers = has_a & (slot_b > slot_a);
                const uint32_t lo_rank = (slot_a + 1u) >> 1;
                const uint32_t hi_rank = slot_b == 0xFFFFFFFFu ? n : (slot_b + 1u) >> 1;
                const uint32_t lo_v = ptx_bitrank(alive, lo_rank), hi_v = ptx_bitrank(alive, hi_rank); /* (both ranks <= n whatever the lanes without a start hold) */
                const uint32_t lo = covers ? lo_v : 0u, hi = covers ? hi_v : 0u;
                const int js = has_a ? js0 : -1;
                (void)js;
                mrk_lo[k] = (uint16_t)lo;
                mrk_hi[k] = (uint16_t)hi;
                if (out_refs) {
                    /* both boundary slots, each on its own (the replay needs the end slot even where the op never starts) */
                    uint32_t va = 0xFFFFu, vb = 0xFFFFu;
                    if (js >= 0) va = 2u * rnk[js] + (sa[u] == PTX_SIDE_AFTER ? 1u : 0u);
                    if (sb[u] == PTX_SIDE_BEFORE || sb[u] == PTX_SIDE_AFTER) {
                        const int je = ptx_elem_lookup24(ix, rb[u]);
                        if (je >= 0 && row_of[je] < i[u]) vb = 2u * rnk[je] + (sb[u] == PTX_SIDE_AFTER ? 1u : 0u);
                    }
                    out_refs[base + i[u]] = va | (vb << 16);
                }
                if (k >= moff2 && k < moff3) {
                    if (pl[u] >= Kid) ptx_raise(H, i[u], 1, PTX_ERR_BAD_OP); /* beyond the id space the header declares */
                    cid[k - moff2] = (uint16_t)(pl[u] < Kid ? pl[u] : 0u);
                }
            }
    };
#pragma nounroll
    for (uint32_t st = 0; st < m_steps; st += 2u) {
        PTX_MARK_LOAD(st + 1u, kq_n, i_n, ra_n, rb_n, sa_n, sb_n, pl_n, mq) /* the next step's gathers are in flight while this step is processed */
        PTX_MARK_PQ(st + 2u, mq)
        mark_step(kq, i, ra, rb, sa, sb, pl);
        if (st + 1u >= m_steps) break;
        PTX_MARK_LOAD(st + 2u, kq, i, ra, rb, sa, sb, pl, mq)
        PTX_MARK_PQ(st + 3u, mq)
        mark_step(kq_n, i_n, ra_n, rb_n, sa_n, sb_n, pl_n);
    }
#undef PTX_MARK_LOAD
#undef PTX_MARK_PQ
#undef PTX_MARK_OF
    }
    PTX_LEADER { H->cur_med = 0; } /* (the cursor of the list of live mark ops, below; the barriers of the error check stand in between) */
    PTX_BAIL_IF_ERROR();
    if (H->adm != PTX_NO_ERR) { /* no op-level error anywhere: the failed admission is the log's error */
        lds_high = bp.high;
        return H->adm & 15u;
    }
    PTX_REMAT_AT(10, PTX_REMAT());
    const uint32_t mark2_lds = bp.off;
    /* the element-side arrays (id bitmap, row_of, positions, tombstones) are dead now: their storage is the first
     * choice for the scratch of the tail phases */
    PtxBump bd;
    bd.base = lds;
    bd.off = elem_lds;
    bd.cap = elem_end; /* (the add / remove bitmap of the mark ops stands between the element arrays and the phase scratch: it lives to the end) */
    bd.high = 0;
    bd.overflow = false;
    PTX_STAMP(7);
    /* the mark ops that still cover a visible character (a tenth of them in a document that has seen thousands of ops and shows a few dozen characters):
     * the comment rule and the LWW trees of a short document walk this list instead of every mark op */
    /* The list takes what the recycled element region has left once the worst case of the tail phases (the comments' tables, the trees) is set aside — so
     * that it never costs a byte of the log's LDS window; a log with more live mark ops than that (rare) walks all its mark ops as before. */
    uint32_t live_cap = 0;
    {
        const uint32_t t4 = n < PTX_TILE_4 ? (n ? 1u << ptx_ceil_log2(n) : 1u) : PTX_TILE_4;
        const uint32_t need_c = Kc ? 2u * (uint32_t)ptx_a16(4u * (Kid + 1u)) + (uint32_t)ptx_a16(8u * (Kc + 1u)) : 0u;
        const uint32_t need_4 = (uint32_t)(ptx_a16(K ? 4u * 4u * 2u * t4 : 0u) + ptx_a16(4u * (t4 + 1u)) + ptx_a16(8u * (t4 / 32u + 2u)));
        const uint32_t need_1 = n > PTX_TILE_4 ? (uint32_t)(ptx_a16(K ? 4u * 2u * PTX_TILE_1 : 0u) + ptx_a16(4u * (PTX_TILE_1 + 1u)) + ptx_a16(8u * (PTX_TILE_1 / 32u + 2u))) : 0u;
        uint32_t reserve = need_c > need_4 ? need_c : need_4;
        if (need_1 > reserve) reserve = need_1;
        const uint32_t room = elem_end - elem_lds;
        live_cap = room > reserve + 32u ? (room - reserve - 32u) / 2u : 0u;
        if (live_cap > K) live_cap = K;
    }
    uint16_t* live = live_cap ? ptx_alloc2<uint16_t>(bd, bp, live_cap + 1) : (uint16_t*)lds; /* (no room: no list, not a byte taken — and nothing read through it) */
    PTX_BAIL_CAPACITY();
    {
        const uint32_t steps = PTX_JSTEPS_U(K, PTX_UV);
#pragma nounroll
        for (uint32_t st = 0; st < steps; ++st) { /* every thread runs every step: the list slots come from a wave-wide prefix sum */
            uint32_t kk[PTX_UV], lo_[PTX_UV], hi_[PTX_UV], cnt_ = 0;
#pragma unroll
            for (int u = 0; u < PTX_UV; ++u) {
                const uint32_t j = PTX_J_OF_U(st, u, PTX_UV);
                kk[u] = j < K ? PTX_JX(j, K) : K; /* (entry K of the interval arrays is spare) */
                lo_[u] = mrk_lo[kk[u]];
                hi_[u] = mrk_hi[kk[u]];
            }
#pragma unroll
            for (int u = 0; u < PTX_UV; ++u) cnt_ += kk[u] < K && lo_[u] < hi_[u] ? 1u : 0u;
            uint32_t at = ptx_append_n(&H->cur_med, cnt_);
#pragma unroll
            for (int u = 0; u < PTX_UV; ++u)
                if (kk[u] < K && lo_[u] < hi_[u]) {
                    if (at < live_cap) live[at] = (uint16_t)kk[u];
                    ++at;
                }
        }
    }
    PTX_SYNC_LDS();
    PTX_REMAT_AT(11, PTX_REMAT());
    const bool use_live = H->cur_med <= live_cap; /* else: the list is incomplete, every mark op is looked at */
    const uint32_t nlive = use_live ? H->cur_med : 0u;
    const uint32_t live_lds = bd.off, live_top = bp.off; /* (the list stays until the end: the scratch of the tail phases is released down to here) */

    /* ---- P5c: comments: per id, presence intervals decided by the last-applied covering op ---- */
    if (Kc > 0) {
        uint32_t* ccnt = ptx_alloc2<uint32_t>(bd, bp, Kid + 1);
        uint32_t* ccur = ptx_alloc2<uint32_t>(bd, bp, Kid + 1);
        uint32_t* cicnt = ccur; /* intervals per id: reuses the scatter cursors once the entries are placed */
        PtxCEntry* cent = ptx_alloc2<PtxCEntry>(bd, bp, Kc + 1);
        PTX_BAIL_CAPACITY();
        PTX_FOR(c, Kid + 1) {
            ccnt[c] = 0;
            ccur[c] = 0;
        }
        PTX_SYNC_LDS();
        const uint32_t c_items = use_live ? nlive : Kc; /* the live mark ops, or (no list) every comment op */
        PTX_FOR(j, c_items) {
            const uint32_t k = use_live ? (uint32_t)live[j] : moff2 + j;
            if (k >= moff2 && k < moff3 && mrk_lo[k] < mrk_hi[k]) ptx_atomic_add(&ccnt[cid[k - moff2]], 1u);
        }
        PTX_SYNC_LDS();
        ptx_counts_prefix<kThreads>(ccnt, Kid + 1, &H->scan_tmp[20]); /* ccnt[c] = first entry of id c, ccnt[Kid] = total */
        PTX_FOR(j, c_items) {
            const uint32_t k = use_live ? (uint32_t)live[j] : moff2 + j;
            if (k >= moff2 && k < moff3 && mrk_lo[k] < mrk_hi[k]) {
                const uint32_t c = cid[k - moff2];
                const uint32_t pos = ccnt[c] + ptx_atomic_add(&ccur[c], 1u);
                PtxCEntry e;
                e.lo = mrk_lo[k];
                e.hi = mrk_hi[k];
                e.t = (uint16_t)ptx_coherent_load32(&park[mp0 + k]); /* application index = row in the log (from the park: the LDS copy has become mrk_lo) */
                e.add = ptx_bittest(maddbits, k) ? 1 : 0;
                cent[pos] = e;
            }
        }
        PTX_SYNC_LDS();
        PTX_FOR(c, Kid + 1) {
            cicnt[c] = c < Kid ? ptx_comment_sweep<kThreads == 0u || kThreads == 256u>(cent + ccnt[c], ccnt[c + 1] - ccnt[c], [](uint32_t, uint32_t) {}) : 0u;
        }
        PTX_SYNC_LDS();
        const uint32_t I = ptx_counts_prefix<kThreads>(cicnt, Kid + 1, &H->scan_tmp[21]);
        PTX_LEADER { H->I = I; }
        /* the interval rows: first into LDS by the per-id sweeps (a lane per id, ragged), then out — rows, break bits and digest — by a dense pass (the
         * digest is ~100 vector instructions per item and wave: it must not sit inside the ragged loop) */
        uint32_t* crow = ptx_try_alloc<uint32_t>(bd, 2 * I + 2); /* {id, start | end << 16}; from what is left of the recycled region, if it has room */
        uint64_t h1 = 0, h2 = 0;
        PTX_FOR(c, Kid) {
            uint32_t row = cicnt[c];
            ptx_comment_sweep<kThreads == 0u || kThreads == 256u>(cent + ccnt[c], ccnt[c + 1] - ccnt[c], [&](uint32_t s, uint32_t e) {
                if (crow) {
                    crow[2u * row] = c;
                    crow[2u * row + 1u] = s | (e << 16);
                } else { /* no room: the row goes out from here */
                    ptx_cinterval ci;
                    ci.id = c;
                    ci.start = s;
                    ci.end = e;
                    out_cints[row] = ci;
                    ptx_atomic_or(&brkbits[s >> 5], 1u << (s & 31));
                    ptx_atomic_or(&brkbits[e >> 5], 1u << (e & 31));
                    ptx_digest_item(h1, h2, 3u, c, s, e);
                }
                ++row;
            });
        }
        if (crow) {
            PTX_SYNC_LDS();
            PTX_FOR(r, I) {
                ptx_cinterval ci;
                ci.id = crow[2u * r];
                ci.start = crow[2u * r + 1u] & 0xFFFFu;
                ci.end = crow[2u * r + 1u] >> 16;
                out_cints[r] = ci;
                ptx_atomic_or(&brkbits[ci.start >> 5], 1u << (ci.start & 31));
                ptx_atomic_or(&brkbits[ci.end >> 5], 1u << (ci.end & 31)); /* end <= V: bit V is never read */
                ptx_digest_item(h1, h2, 3u, ci.id, ci.start, ci.end);
            }
        }
        ptx_digest_flush(H, h1, h2);
        PTX_SYNC_LDS();
    }
    bp.off = live_top; /* release the comment scratch */
    bd.off = live_lds;
    PTX_STAMP(8);
    PTX_REMAT_AT(12, PTX_REMAT());

    /* ---- P5b + P6: LWW winners per visible char, spans, digest — in tiles of the visible axis ----
     * Short documents: one tile, four trees (one per mark type) updated and queried in one pass.
     * Long documents: tiles of PTX_TILE_1 chars, one tree reused per mark type. */
    {
        /* (round 6, last session) a document of up to 2 PTX_TILE_4 characters whose mark ops are of FEWER than four types — BASELINE config #3: 158 characters,
         * strong and em only — takes the short-document form too where a tree per PRESENT type fits what the launch's window has free: the live list instead of
         * every mark op, one pass instead of one per type.  (Not in the three-wave lean build.) */
        const uint32_t present4 = (moff1 ? 1u : 0u) | (moff2 > moff1 ? 2u : 0u) | (moff3 > moff2 ? 4u : 0u) | (K > moff3 ? 8u : 0u); /* mark types with ops */
        bool mid = false;
        if (kThreads != 192u && V > PTX_TILE_4 && V <= 2u * PTX_TILE_4 && K != 0u && present4 != 15u) {
            const uint32_t T2 = 2u * PTX_TILE_4, nt = ptx_popc(present4);
            const uint64_t over = ptx_overflow3((uint64_t)bd.cap - bd.off, 4u * nt * 2u * T2, 4u * (T2 + 1u), 8u * (T2 / 32u + 2u));
            mid = (uint64_t)bp.off + over <= bp.cap;
        }
        const bool four = V <= PTX_TILE_4 || mid;
        /* tree of mark type ty in the short-document form: its own (four trees), or its rank among the present types */
        const uint32_t tslots = !mid ? 0x03020100u
                                     : (0u | (ptx_popc(present4 & 1u) << 8) | (ptx_popc(present4 & 3u) << 16) | (ptx_popc(present4 & 7u) << 24));
#define PTX_TSLOT(ty_) ((tslots >> (8u * (ty_))) & 255u)
        uint32_t TV = 1;
        if (four) {
            while (TV < V) TV <<= 1;
        } else {
            TV = PTX_TILE_1;
            /* (round 6) a document of more than one tile takes tiles of twice the size where the recycled element region and the free phase scratch hold
             * them: half the passes — each a handful of barriers and, for the links, two trips to HBM */
            if ((kThreads == 0u || kThreads == 256u) && V > PTX_TILE_1) {
                const uint32_t T2 = 2u * PTX_TILE_1;
                const uint64_t over = ptx_overflow3((uint64_t)bd.cap - bd.off, K ? 4u * 2u * T2 : 0u, 4u * (T2 + 1u), 8u * (T2 / 32u + 2u));
                if ((uint64_t)bp.off + over <= bp.cap) TV = T2;
            }
        }
        const uint32_t gstep = four ? 4u : 1u;                                /* mark types per pass */
        const uint32_t ntree = mid ? ptx_popc(present4) : four ? 4u : 1u;      /* trees in LDS */
        uint32_t* tree = ptx_alloc2<uint32_t>(bd, bp, K ? ntree * 2 * TV : 0u); /* a log without mark ops has one span per break: no trees */
        uint32_t* attr = ptx_alloc2<uint32_t>(bd, bp, TV + 1);
        PtxBitWord* st = ptx_alloc2<PtxBitWord>(bd, bp, TV / 32 + 2);
        PTX_BAIL_CAPACITY();
        const uint32_t kmask = (1u << kbits) - 1u;
        uint64_t h1 = kOneFlush ? carry_h1 : 0, h2 = kOneFlush ? carry_h2 : 0;
        uint32_t span_base = 0;
        uint32_t prev_attr = 0; /* marks of the last char of the previous tile */
        /* Long documents (several tiles, a pass per tile and mark type): the park words — row | id key — of the ops a thread takes in step 0 of every type's run are
         * read ONCE, here.  Read inside the pass they were a trip to HBM per tile and type — 16 of them, one after the other, for a document of 2 000 characters:
         * a third of the 160 k cycles this phase takes of such a log (round 6, `rich4k`; a run of up to PTX_UB ops per thread is one step, i.e. all of it). */
        uint32_t pk0[PTX_UB], pk1[PTX_UB], pk3[PTX_UB];
        /* (both long-document forms — this and the three-chars-per-step query below — only in the builds for any launch shape, which take the documents that keep
         * their text: in the lean builds of the three usual shapes they cost scalar registers and bought nothing, config #4 +0.7 %, #3 +0.5 %) */
        constexpr bool kLongDocs = kThreads == 0u || kThreads == 128u || kThreads == 256u; /* (the two-wave build too: a 1K-op log of BASELINE config #3 shows 158 characters, -0.9 %; 256: the four-wave lean build, which takes the documents that keep their text) */
        const bool pk_cached = kLongDocs && !four && K != 0u;
#pragma unroll
        for (int u = 0; u < (int)PTX_UB; ++u) {
            const uint32_t j = PTX_J_OF_U(0u, u, PTX_UB);
            const uint32_t kn0 = moff1, kn1 = moff2 - moff1, kn3 = K - moff3;
            pk0[u] = pk_cached && j < kn0 ? ptx_coherent_load32(&park[mp0 + PTX_JX(j, kn0)]) : 0u;
            pk1[u] = pk_cached && j < kn1 ? ptx_coherent_load32(&park[mp0 + moff1 + PTX_JX(j, kn1)]) : 0u;
            pk3[u] = pk_cached && j < kn3 ? ptx_coherent_load32(&park[mp0 + moff3 + PTX_JX(j, kn3)]) : 0u;
        }
        /* a log without a mark op (BASELINE config #2: inserts and deletes only): one span over the whole text, no attribute, no break — nothing to sweep
         * (round 6: the tile pass of such a log was 313 of the 2 825 vector instructions of a 256-op log, whose build is bound by their issue) */
        const uint32_t V_swept = K != 0u ? V : 0u;
        if (K == 0u && V != 0u) {
            PTX_LEADER {
                ptx_span sp;
                sp.start = 0;
                sp.attr = 0;
                out_spans[0] = sp;
                ptx_digest_item(h1, h2, 2u, 0u, 0u, 0u);
            }
            span_base = 1u;
        }
#pragma nounroll
        for (uint32_t t0 = 0; t0 < V_swept; t0 += TV) {
            const uint32_t tv = V - t0 < TV ? V - t0 : TV; /* chars in this tile */
            PTX_FOR(q, tv + 1) attr[q] = 0;
            PTX_FOR(w, TV / 32 + 2) {
                PtxBitWord z;
                z.bits = 0;
                z.pre = 0;
                st[w] = z;
            }
#pragma nounroll
            for (uint32_t g = 0; g < 4; g += gstep) {
                /* mark types [g, g + gstep) */
                const uint32_t k_lo = g == 0 ? 0u : g == 1 ? moff1 : g == 2 ? moff2 : moff3;
                const uint32_t k_hi = four ? K : (g == 0 ? moff1 : g == 1 ? moff2 : g == 2 ? moff3 : K);
                if (k_hi == k_lo) continue;
                PTX_FOR(p, ntree * 2 * TV) tree[p] = 0;
                PTX_SYNC_LDS();
                /* PTX_UB mark ops per thread and step: the opIds of those that still cover a visible char (LWW order = opId order;
                 * the comment tree only records "some comment op covers") are gathered together — one round trip to HBM per
                 * step instead of one per op */
                /* all four types at once: the live ops; one type at a time: its run, in row order */
                const bool by_list = four && use_live;
                const uint32_t kn = k_hi - k_lo, b_steps = by_list ? PTX_JSTEPS_U(nlive, PTX_UB) : PTX_JSTEPS_U(kn, PTX_UB);
                /* two register sets in turn: the opId gathers of the next step are in flight while this step's ranges go into the trees */
                uint32_t kq_a[PTX_UB], lo_a[PTX_UB], hi_a[PTX_UB], kq_b[PTX_UB], lo_b[PTX_UB], hi_b[PTX_UB];
                uint32_t idq_a[PTX_UB], idq_b[PTX_UB]; /* the park entry of the op: row | id key << 16 */
                auto lww_load = [&](uint32_t st, uint32_t (&kq)[PTX_UB], uint32_t (&lo)[PTX_UB], uint32_t (&hi)[PTX_UB], uint32_t (&idq)[PTX_UB]) {
#pragma unroll
                    for (int u = 0; u < (int)PTX_UB; ++u) {
                        uint32_t k;
                        bool has;
                        if (by_list) { /* a short document: the live mark ops, all four types at once */
                            const uint32_t j = PTX_J_OF_U(st, u, PTX_UB);
                            has = j < nlive;
                            k = live[has ? PTX_JX(j, nlive) : 0u];
                        } else {
                            const uint32_t j = PTX_J_OF_U(st, u, PTX_UB);
                            has = j < kn;
                            k = k_lo + (has ? PTX_JX(j, kn) : 0u);
                        }
                        uint32_t l = mrk_lo[k], h = mrk_hi[k];
                        l = l > t0 ? l - t0 : 0u;
                        h = h > t0 ? (h - t0 < tv ? h - t0 : tv) : 0u;
                        if (!has || st >= b_steps) l = h = 0u;
                        kq[u] = k;
                        lo[u] = l;
                        hi[u] = h;
                        idq[u] = 0;
                        if (l < h && PTX_TYPE_OF(k) != PTX_MARK_COMMENT)
                            idq[u] = pk_cached && st == 0u ? (g == 0u ? pk0[u] : g == 1u ? pk1[u] : pk3[u]) : ptx_coherent_load32(&park[mp0 + k]);
                    }
                };
                auto lww_put = [&](const uint32_t (&kq)[PTX_UB], const uint32_t (&lo)[PTX_UB], const uint32_t (&hi)[PTX_UB], const uint32_t (&idq)[PTX_UB]) {
#pragma unroll
                    for (int u = 0; u < (int)PTX_UB; ++u)
                        if (lo[u] < hi[u]) {
                            const uint32_t k = kq[u], ty = PTX_TYPE_OF(k);
                            uint32_t key = idq[u] >> 16; /* LWW order = opId order = order of the dense id keys */
                            if (!small_keys && ty != PTX_MARK_COMMENT) { /* (keys beyond 16 bits are not parked: a second trip, through the row) */
                                const uint32_t r = idq[u] & 0xFFFFu;
                                ptx_id_key(ix, op_id[r < N ? r : N - 1u], key);
                            }
                            /* the low bits say who won */
                            ptx_tree_chmax(tree + (four ? PTX_TSLOT(ty) : 0u) * 2 * TV, TV, lo[u], hi[u], ty == PTX_MARK_COMMENT ? 1u : ((key + 1u) << kbits) | k);
                        }
                };
                lww_load(0u, kq_a, lo_a, hi_a, idq_a);
#pragma nounroll
                for (uint32_t st = 0; st < b_steps; st += 2u) {
                    lww_load(st + 1u, kq_b, lo_b, hi_b, idq_b);
                    lww_put(kq_a, lo_a, hi_a, idq_a);
                    if (st + 1u >= b_steps) break;
                    lww_load(st + 2u, kq_a, lo_a, hi_a, idq_a);
                    lww_put(kq_b, lo_b, hi_b, idq_b);
                }
                PTX_SYNC_LDS();
                if (four || !kLongDocs) { /* a short document: one pass, four trees of a few levels, a char per lane (the lean builds: every document) */
                PTX_FOR(q, tv) {
                    uint32_t at = 0;
                    for (uint32_t ty = g; ty < g + gstep; ++ty) {
                        if (mid && !((present4 >> ty) & 1u)) continue; /* (no tree of its own: no op of that type) */
                        const uint32_t w = PTX_QUERY9 && kThreads != 192u && four ? ptx_tree_query9(tree + PTX_TSLOT(ty) * 2 * TV, TV, q) : ptx_tree_query(tree + (four ? PTX_TSLOT(ty) : 0u) * 2 * TV, TV, q);
                        if (w == 0) continue;
                        if (ty == PTX_MARK_COMMENT) at |= PTX_ATTR_COMMENT;
                        else {
                            const uint32_t k = w & kmask;
                            if (ptx_bittest(maddbits, k)) {
                                if (ty == PTX_MARK_STRONG) at |= PTX_ATTR_STRONG;
                                else if (ty == PTX_MARK_EM) at |= PTX_ATTR_EM;
                                else at |= PTX_ATTR_LINK | (payload[ptx_coherent_load32(&park[mp0 + k]) & 0xFFFFu] & PTX_ATTR_ID_MASK);
                            }
                        }
                    }
                    if (at) attr[q + 1] |= at; /* attr[0] = last char of the previous tile */
                }
                } else {
                /* three chars per thread and step: their tree look-ups first, then the park words of the link winners, then the urls — two trips to HBM per step
                 * whatever the chars (a char at a time they were two trips EACH, one after the other: the larger part of this phase's time in a document of
                 * thousands of characters, round 6) */
                PTX_FORV(q0, tv, 3) {
                    uint32_t at[3], lk[3], pw[3], url[3];
#pragma unroll
                    for (int u = 0; u < 3; ++u) {
                        at[u] = 0;
                        lk[u] = 0xFFFFFFFFu; /* the link op that wins at this char, if one does */
                        if (PTX_IN(q0, u)) {
                            const uint32_t q = PTX_IX(q0, u);
                            const uint32_t ty = g, w = ptx_tree_query_tile(tree, TV, q); /* (one type per pass) */
                            if (w != 0) {
                                if (ty == PTX_MARK_COMMENT) at[u] |= PTX_ATTR_COMMENT;
                                else {
                                    const uint32_t k = w & kmask;
                                    if (ptx_bittest(maddbits, k)) {
                                        if (ty == PTX_MARK_STRONG) at[u] |= PTX_ATTR_STRONG;
                                        else if (ty == PTX_MARK_EM) at[u] |= PTX_ATTR_EM;
                                        else lk[u] = k;
                                    }
                                }
                            }
                        }
                    }
#pragma unroll
                    for (int u = 0; u < 3; ++u) pw[u] = lk[u] != 0xFFFFFFFFu ? ptx_coherent_load32(&park[mp0 + lk[u]]) & 0xFFFFu : 0u;
#pragma unroll
                    for (int u = 0; u < 3; ++u) url[u] = lk[u] != 0xFFFFFFFFu ? payload[pw[u] < N ? pw[u] : N - 1u] : 0u;
#pragma unroll
                    for (int u = 0; u < 3; ++u)
                        if (PTX_IN(q0, u)) {
                            if (lk[u] != 0xFFFFFFFFu) at[u] |= PTX_ATTR_LINK | (url[u] & PTX_ATTR_ID_MASK);
                            if (at[u]) attr[PTX_IX(q0, u) + 1] |= at[u]; /* attr[0] = last char of the previous tile */
                        }
                }
                }
                PTX_SYNC_LDS();
            }
            PTX_LEADER { attr[0] = prev_attr; }
            PTX_SYNC_LDS();
            /* spans = maximal runs of equal marks over the visible chars (peritext.ts:438-455) */
            PTX_FOR(q, tv) {
                const uint32_t gq = t0 + q;
                const bool is_start = gq == 0 || attr[q + 1] != attr[q] || ptx_bittest(brkbits, gq);
                if (is_start) ptx_atomic_or(&st[q >> 5].bits, 1u << (q & 31));
            }
            PTX_SYNC_LDS();
            const uint32_t S_tile = ptx_bitwords_prefix<kThreads>(st, TV / 32 + 2, &H->scan_tmp[19]);
            PTX_FOR(q, tv) {
                const PtxBitWord w = st[q >> 5];
                if ((w.bits >> (q & 31)) & 1u) {
                    const uint32_t s = span_base + w.pre + ptx_popc(w.bits & ((1u << (q & 31)) - 1u));
                    ptx_span sp;
                    sp.start = t0 + q;
                    sp.attr = attr[q + 1];
                    out_spans[s] = sp;
                    ptx_digest_item(h1, h2, 2u, s, sp.start, sp.attr);
                }
            }
            span_base += S_tile;
            prev_attr = attr[tv];
            PTX_SYNC_LDS();
        }
        if (V >= 48u) { /* uniform: most lanes of a wave carry a share */
            if (kThreads == 64u || kThreads == 128u) ptx_digest_flush_dense_dpp(H, h1, h2);
            else ptx_digest_flush_dense(H, h1, h2);
        }
        else ptx_digest_flush(H, h1, h2);
        PTX_SYNC_LDS();
        PTX_LEADER {
            H->V = V;
            H->S = span_base;
        }
        PTX_FOR(t, 2u) { /* the two count items of the digest, a lane each */
            uint64_t g1 = 0, g2 = 0;
            ptx_digest_item(g1, g2, 4u, t, t ? H->I : V, t ? n : span_base);
            ptx_digest_flush(H, g1, g2);
        }
    }
    PTX_SYNC_LDS();
    PTX_STAMP(10);
    lds_high = bp.high;
    return PTX_OK;
}

/* kManyActors (0 / 1 / 2): include the admission paths for batches with more than three actors per document (1: the walk for up to seven + the table; 2: the walks for eight to fifteen + the table; they cost ~35
 * VGPRs, i.e. two waves per SIMD, so it lives in its own build of the kernel) */
template <int kManyActors, uint32_t kThreads, bool kDiag = false, bool kLean = false>
PTX_DEV void ptx_merge_log(const PtxMergeArgs& A, uint32_t log, uint8_t* lds) {
    uint32_t lds_high = 0;
    const uint32_t status = ptx_merge_log_body<kManyActors, kThreads, kDiag, kLean>(A, log, lds, lds_high);
    PTX_SYNC_LDS();
    ptx_write_result<kDiag>(A, log, (PtxHdr*)lds, status, lds_high);
}
