/*
 * biglog_core.h — replica logs that do not fit one CU's LDS (VERDICT r2 "missing" #3; SURVEY §7's multi-block path, here HBM-staged).
 *
 * ptx_merge_log (merge_core.h) keeps the whole working set of a log in LDS with 16-bit indices: at most 65 534 rows, 32 766 list elements and
 * what 160 KB hold.  The reference has no such bound (reference/src/micromerge.ts:614-672 works on arbitrary arrays): a 100 000-character essay
 * is one replica log.  Such logs are merged by THIS body instead: the same closed form, the same outputs
 * (values, spans, comment intervals, 128-bit digest, per-log status and failing row) — but
 *   - the working set lives in a slice of HBM scratch the host sizes per log (ptx_big_need), indices are 32 bits wide;
 *   - one workgroup of up to 1 024 threads per log, phases separated by FULL barriers (the waves talk through global memory);
 *   - the simplest parallel algorithm per phase, chosen for unbounded sizes rather than for the last cycle: element index = bitmap over the id
 *     keyspace + popcount prefix (as in the LDS kernel); children of every parent in descending opId by ONE bitonic sort of (parent, ~element)
 *     keys; document order by pointer jumping over the Euler tour ({next, weight} in one 64-bit word); LWW winners by range-chmax trees of
 *     64-bit (key, mark) words; the comment rule by the per-id sweep of merge_core.h.
 * The host routes a log here when its LDS need exceeds the CU (census_and_shape); everything else still takes the LDS kernel, which is an
 * order of magnitude faster per op.  Per-log errors mirror the reference's throw sites exactly as there (first failing row wins).
 *
 * Written with the same PTX_FOR / PTX_LEADER / PTX_SYNC vocabulary, so the CPU test-suite's emulation runs it too (tests/test_emu_biglog.py).
 */
#pragma once
#include "merge_core.h"

#ifndef PTX_BIG_SORT_TILE
#define PTX_BIG_SORT_TILE 4096u /* keys (a power of two) a workgroup sorts between two team barriers */
#endif
#ifndef PTX_BIG_COMMENT_OPS_PER_ID
#define PTX_BIG_COMMENT_OPS_PER_ID 1024u /* comment ops with a visible interval up to which ONE lane sweeps a comment id (quadratic); an id with more is swept by the whole team through a range-chmax tree (round 5) */
#endif
struct PtxBigEntry { /* one comment op that covers something: visible interval, application index (row), add / remove */
    uint32_t lo, hi, t, add;
};

PTX_HD uint64_t ptx_pow2_ge(uint64_t x) {
    uint64_t p = 1;
    while (p < x) p <<= 1;
    return p;
}
PTX_HD uint64_t ptx_a64(uint64_t x) { return (x + 63) & ~63ull; }

/* bytes of HBM scratch ptx_big_merge_log takes for a log (mirrors its allocations, in order) */
PTX_HD uint64_t ptx_big_need(uint64_t N, const ptx_log_hdr& h, uint64_t C, uint64_t na, uint64_t threads) {
    const uint64_t n = h.n_ins, D = h.n_del, K = (uint64_t)h.n_mark[0] + h.n_mark[1] + h.n_mark[2] + h.n_mark[3], Kc = h.n_mark[PTX_MARK_COMMENT];
    const uint64_t Kid = Kc ? h.n_comment_ids : 0;
    const uint64_t ks = ((uint64_t)h.max_counter + 1) * ((uint64_t)h.max_actor + 1), nw = (ks + 31) / 32, nwe = (n >> 5) + 2;
    const uint64_t P2 = ptx_pow2_ge(n + 1), PV = ptx_pow2_ge(n + 1);
    (void)N;
    uint64_t b = 0;
    b += ptx_a64(sizeof(PtxHdr));                                      /* the header (a cooperative launch keeps it here, not in LDS) */
    b += ptx_a64(4 * (threads / PTX_WS + 3));                          /* scan partials: one per wave of the team */
    b += C ? ptx_a64(4 * (na + 2)) + 2 * ptx_a64(4 * (C + 1)) : 0;     /* admission: first[], tbl[], crow[] */
    b += ptx_a64(8 * (nw + 1)) + ptx_a64(4 * (nw + 1));                /* id bitmap {bits, prefix}, all-ids bitmap */
    b += ptx_a64(4 * (n + 1)) + ptx_a64(4 * (D + 1)) + 4 * ptx_a64(4 * (K + 1)); /* ilist, dlist, mlist, mflag, mrk_lo, mrk_hi */
    b += 3 * ptx_a64(4 * (n + 2)) + ptx_a64(4 * nwe);                  /* row_of, par, pos, delbits */
    b += ptx_a64(8 * P2) + ptx_a64(4 * (n + 3));                       /* sort keys, bucket starts */
    b += ptx_a64(8 * (2 * n + 3));                                     /* Euler tour */
    b += ptx_a64(8 * (nwe + 1));                                       /* alive {bits, prefix} */
    b += 2 * ptx_a64(4 * (Kid + 2)) + ptx_a64(16 * (Kc + 1)) + ptx_a64(4 * (Kc + 1)) + ptx_a64(4 * (Kc / PTX_BIG_COMMENT_OPS_PER_ID + 2)); /* comments: per-id counters, entries, ids, the heavy ids */
    b += 4 * ptx_a64(8 * 2 * PV);                                      /* four LWW trees */
    b += ptx_a64(4 * (n + 2)) + ptx_a64(8 * (nwe + 1)) + ptx_a64(4 * (nwe + 1)); /* attr, span starts, comment breaks */
    return b + 256;
}

/* exclusive scan of a[0..m) (stride STRIDE) in GLOBAL memory by every thread of the log's team (one workgroup, or — kGrid — the workgroups of a cooperative
 * launch): per-thread chunks, the chunk sums scanned inside each wave (DPP, no memory), the wave totals (part[0 .. waves]) by ONE wave.  (Round 4 had the
 * leader add up one partial per THREAD: a thousand dependent trips to global memory per scan.) */
template <bool kGrid, class T, int STRIDE>
PTX_DEV uint32_t ptx_big_scan(T* a, uint32_t m, uint32_t* part, uint32_t* _gbar) {
    constexpr uint32_t kThreads = 0u; /* (the loop macros of the GPU platform read the workgroup size at run time when this is 0) */
    (void)kThreads;
    (void)_gbar;
    const uint32_t TT = PTX_BNT, chunk = (m + TT - 1u) / TT, nwaves = (TT + PTX_WS - 1u) / PTX_WS;
    uint32_t mine = 0, incl_mine = 0; /* (every thread runs its one iteration of the two loops over t: what it found in the first is still in its registers) */
    PTX_BFOR(t, TT) {
        const uint32_t lo = t * chunk < m ? t * chunk : m, hi = lo + chunk < m ? lo + chunk : m;
        uint32_t s = 0;
        for (uint32_t j = lo; j < hi; ++j) s += (uint32_t)a[(uint64_t)j * STRIDE];
        const uint32_t incl = ptx_wave_incl_scan(s);
        mine = s;
        incl_mine = incl;
        PTX_BIG_KEEP(t, s, incl)
        if ((t & (PTX_WS - 1u)) == PTX_WS - 1u || t == TT - 1u) part[t / PTX_WS] = incl; /* the wave's total */
    }
    PTX_BSYNC();
    PTX_BFIRST_WAVE { /* the wave totals -> exclusive prefixes, 64 at a time (one wave: its lanes need no barrier) */
        uint32_t run = 0;
        for (uint32_t w0 = 0; w0 < nwaves; w0 += PTX_WS) {
            const uint32_t w = w0 + PTX_BLANE;
            const uint32_t v = w < nwaves ? part[w] : 0u;
            const uint32_t in = ptx_wave_incl_scan(v);
            if (w < nwaves) part[w] = run + in - v;
            run += ptx_wave_last(in);
        }
        if (PTX_BLANE == 0u) part[nwaves] = run;
    }
    PTX_BSYNC();
    PTX_BFOR(t, TT) {
        const uint32_t lo = t * chunk < m ? t * chunk : m, hi = lo + chunk < m ? lo + chunk : m;
        PTX_BIG_RECALL(t, mine, incl_mine)
        uint32_t run = part[t / PTX_WS] + incl_mine - mine;
        for (uint32_t j = lo; j < hi; ++j) {
            const uint32_t v = (uint32_t)a[(uint64_t)j * STRIDE];
            a[(uint64_t)j * STRIDE] = (T)run;
            run += v;
        }
    }
    PTX_BSYNC();
    const uint32_t total = part[nwaves];
    PTX_BSYNC();
    return total;
}

PTX_DEV void ptx_big_chmax(unsigned long long* tree, uint32_t P, uint32_t lo, uint32_t hi, unsigned long long val) {
    uint32_t l = lo + P, r = hi + P;
    while (l < r) {
        if (l & 1u) ptx_atomic_max64(&tree[l++], val);
        if (r & 1u) ptx_atomic_max64(&tree[--r], val);
        l >>= 1;
        r >>= 1;
    }
}
PTX_DEV unsigned long long ptx_big_query(const unsigned long long* tree, uint32_t P, uint32_t q) {
    unsigned long long w = 0;
    for (uint32_t p = q + P; p >= 1; p >>= 1) w = tree[p] > w ? tree[p] : w;
    return w;
}

#define PTX_BIG_BAIL_IF_ERROR()                                                                  \
    do {                                                                                         \
        PTX_BSYNC();                                                                              \
        uint32_t _st = H->err;                                                                   \
        if (_st != PTX_NO_ERR && H->adm < _st) _st = H->adm;                                     \
        PTX_BSYNC();                                                                              \
        if (_st != PTX_NO_ERR) return _st & 15u;                                                 \
    } while (0)

/* Applies log `log` with its working set in `win` (win_bytes of HBM scratch); returns the status.  H: the PtxHdr in LDS. */
template <bool kGrid>
PTX_DEV uint32_t ptx_big_merge_body(const PtxMergeArgs& A, uint32_t log, uint8_t* win, uint64_t win_bytes, PtxHdr* H) {
    constexpr uint32_t kThreads = 0u;
    (void)kThreads;
    uint32_t* const _gbar = A.grid_bar; /* (kGrid: the barrier words of the cooperative launch) */
    (void)_gbar;
    const uint64_t base = A.log_off[log];
    const uint64_t N64 = A.log_off[log + 1] - base;
    PTX_BLEADER {
        H->err = PTX_NO_ERR;
        H->adm = PTX_NO_ERR;
        H->n_ins = H->n_applied = 0;
        H->V = H->S = H->I = 0;
        H->h1 = H->h2 = 0;
        for (int k = 0; k < 8; ++k) H->cur[k] = 0;
    }
    PTX_BSYNC();
    if (N64 > 0x7FFFFFF0ull) return PTX_ERR_CAPACITY;
    const uint32_t N = (uint32_t)N64;
    const uint64_t* op_id = A.op_id + base;
    const uint64_t* ref_a = A.ref_a + base;
    const uint64_t* ref_b = A.ref_b + base;
    const uint32_t* payload = A.payload + base;
    const uint8_t* action = A.action + base;
    const uint8_t* mark_type = A.mark_type + base;
    if (N == 0) {
        PTX_BLEADER {
            uint64_t g1 = 0, g2 = 0;
            ptx_digest_item(g1, g2, 4u, 0u, 0u, 0u);
            ptx_digest_item(g1, g2, 4u, 1u, 0u, 0u);
            H->h1 += g1;
            H->h2 += g2;
        }
        PTX_BSYNC();
        return PTX_OK;
    }
    const ptx_log_hdr hd = A.log_hdr[log];
    const uint32_t n = hd.n_ins, D = hd.n_del, K = hd.n_mark[0] + hd.n_mark[1] + hd.n_mark[2] + hd.n_mark[3], Kc = hd.n_mark[PTX_MARK_COMMENT];
    const uint32_t Kid = Kc ? hd.n_comment_ids : 0u;
    if ((uint64_t)n + D + K > N) return PTX_ERR_BAD_OP;
    PtxElemIndex ix;
    ix.max_ctr = hd.max_counter;
    ix.max_actor = hd.max_actor;
    ix.na1 = ix.max_actor + 1u;
    const uint64_t ks64 = ((uint64_t)ix.max_ctr + 1u) * ix.na1;
    if (ix.max_actor > 4095u || ks64 > (1ull << 30) || n > 0x3FFFFFF0u || Kid > 0x0FFFFFFFu) return PTX_ERR_CAPACITY;
    const uint32_t keyspace = (uint32_t)ks64, nw = (keyspace + 31u) / 32u, nwe = (n >> 5) + 2u;
    const uint32_t TT = PTX_BNT;

    /* ---- the scratch slice, carved in the order ptx_big_need counts it ---- */
    uint64_t off = 0;
    bool over = false;
    auto take = [&](uint64_t bytes) -> uint8_t* {
        uint8_t* p = win + off;
        off += ptx_a64(bytes);
        if (off > win_bytes) over = true;
        return p;
    };
    uint32_t* part = (uint32_t*)take(4ull * (TT / PTX_WS + 3u));
    uint32_t C = 0, na = 0;
    bool narrow_long = false;
    uint32_t *first = nullptr, *tbl = nullptr, *crow = nullptr;
    const uint32_t* c_hdr = nullptr;
    const uint16_t *c_env = nullptr, *c_env_hi = nullptr;
    uint32_t estride = 0;
    /* value k of the envelope of this log: exact (lo | hi << 16) when the batch carries the wide column */
    auto env_at = [&](uint64_t k) -> uint32_t { return (uint32_t)c_env[k] | (c_env_hi ? (uint32_t)c_env_hi[k] << 16 : 0u); };
    if (A.chg_off) {
        const uint64_t c0 = A.chg_off[log], C64 = A.chg_off[log + 1] - c0;
        na = A.max_actors;
        if (C64 > 0x7FFFFFF0ull) return PTX_ERR_CAPACITY;
        if (na == 0u || na > 4096u) return PTX_ERR_BAD_OP;
        C = (uint32_t)C64;
        estride = PTX_ENV_STRIDE(na);
        c_hdr = A.chg_hdr + c0;
        c_env = A.chg_env + c0 * estride;
        c_env_hi = A.chg_env_hi ? A.chg_env_hi + c0 * estride : nullptr;
        /* seq / deps beyond 16 bits need the wide column; without it the values of a long log saturate at 65535.  A log of more than 65533 changes whose
         * narrow values all stay below that sentinel (several actors, each with fewer than 65535 changes) is exact as it stands and is admitted; one that
         * holds a saturated value cannot be represented: a capacity report, never a spurious sequence gap (checked below, once the header words are free) */
        narrow_long = C > 65533u && !c_env_hi;
        first = (uint32_t*)take(4ull * (na + 2));
        tbl = (uint32_t*)take(4ull * (C + 1));
        crow = (uint32_t*)take(4ull * (C + 1));
    }
    PtxBitWord* ib = (PtxBitWord*)take(8ull * (nw + 1));
    uint32_t* allb = (uint32_t*)take(4ull * (nw + 1));
    uint32_t* ilist = (uint32_t*)take(4ull * (n + 1));
    uint32_t* dlist = (uint32_t*)take(4ull * (D + 1));
    uint32_t* mlist = (uint32_t*)take(4ull * (K + 1));
    uint32_t* mflag = (uint32_t*)take(4ull * (K + 1)); /* mark type | addMark << 2 | comment id << 3 */
    uint32_t* mrk_lo = (uint32_t*)take(4ull * (K + 1));
    uint32_t* mrk_hi = (uint32_t*)take(4ull * (K + 1));
    uint32_t* row_of = (uint32_t*)take(4ull * (n + 2));
    uint32_t* par = (uint32_t*)take(4ull * (n + 2));
    uint32_t* pos = (uint32_t*)take(4ull * (n + 2));
    uint32_t* delbits = (uint32_t*)take(4ull * nwe);
    const uint32_t P2 = (uint32_t)ptx_pow2_ge((uint64_t)n + 1);
    unsigned long long* skey = (unsigned long long*)take(8ull * P2);
    uint32_t* bstart = (uint32_t*)take(4ull * (n + 3));
    unsigned long long* tour = (unsigned long long*)take(8ull * (2ull * n + 3));
    PtxBitWord* alive = (PtxBitWord*)take(8ull * (nwe + 1));
    uint32_t* ccnt = (uint32_t*)take(4ull * (Kid + 2));
    uint32_t* ccur = (uint32_t*)take(4ull * (Kid + 2));
    PtxBigEntry* cent = (PtxBigEntry*)take(16ull * (Kc + 1));
    uint32_t* cidx = (uint32_t*)take(4ull * (Kc + 1)); /* comment-mark ordinal -> mark index */
    const uint32_t hvy_cap = Kc / PTX_BIG_COMMENT_OPS_PER_ID + 2u;
    uint32_t* hvy = (uint32_t*)take(4ull * hvy_cap); /* the comment ids with more ops than one lane sweeps */
    const uint32_t PV = (uint32_t)ptx_pow2_ge((uint64_t)n + 1);
    unsigned long long* tree[4];
    for (int ty = 0; ty < 4; ++ty) tree[ty] = (unsigned long long*)take(8ull * 2 * PV);
    uint32_t* attr = (uint32_t*)take(4ull * (n + 2));
    PtxBitWord* st = (PtxBitWord*)take(8ull * (nwe + 1));
    uint32_t* brkbits = (uint32_t*)take(4ull * (nwe + 1));
    if (over) return PTX_ERR_CAPACITY;
    ix.ib = ib;

    if (narrow_long) { /* (uniform) a long log without the wide column: exact iff no seq / dep sits at the 16-bit sentinel */
        uint32_t sat = 0;
        PTX_BFOR(c, C) {
            for (uint32_t b = 0; b <= na; ++b) sat |= c_env[(uint64_t)c * estride + b] == 0xFFFFu ? 1u : 0u;
        }
        if (sat) ptx_atomic_or(&H->cur[6], 1u);
        PTX_BSYNC();
        const bool saturated = H->cur[6] != 0u;
        PTX_BSYNC();
        if (saturated) return PTX_ERR_CAPACITY;
    }
    /* ---- P0: causal admission (micromerge.ts:499-511): the (actor, seq) -> change table of merge_core.h's many-actor path, 32-bit ---- */
    if (A.chg_off) {
        PTX_BFOR(a, na + 2) first[a] = 0;
        PTX_BFOR(c, C + 1) {
            tbl[c] = 0xFFFFFFFFu;
            crow[c] = c < C ? (c_hdr[c] & PTX_CHG_NOPS) : 0u;
        }
        PTX_BSYNC();
        const uint32_t rows = ptx_big_scan<kGrid, uint32_t, 1>(crow, C + 1, part, _gbar); /* crow[c] = first row of change c */
        if (rows != N) return PTX_ERR_BAD_OP; /* the changes must tile the rows of the log exactly */
        PTX_BFOR(c, C) {
            const uint32_t a = c_hdr[c] >> PTX_CHG_ACTOR_SHIFT;
            if (a >= na) ptx_atomic_min(&H->adm, ((crow[c] * 2u) << 4) | PTX_ERR_BAD_OP);
            else if (na <= 8u) { /* a handful of actors: one atomic per wave and actor instead of one per change on a handful of addresses */
                for (uint32_t ab = 0; ab < na; ++ab)
                    if (a == ab) (void)ptx_append(&first[ab], true);
            } else ptx_atomic_add(&first[a], 1u);
        }
        PTX_BSYNC();
        if (H->adm != PTX_NO_ERR) return PTX_ERR_BAD_OP;
        PTX_BLEADER {
            uint32_t run = 0;
            for (uint32_t a = 0; a < na + 2u; ++a) {
                const uint32_t v = first[a];
                first[a] = run;
                run += v;
            }
        }
        PTX_BSYNC();
        PTX_BFOR(c, C) {
            const uint32_t a = c_hdr[c] >> PTX_CHG_ACTOR_SHIFT, sq = env_at((uint64_t)c * estride);
            const uint32_t f = first[a], cnt_a = first[a + 1] - f;
            if (sq - 1u < cnt_a) ptx_atomic_min(&tbl[f + sq - 1u], c);
        }
        PTX_BSYNC();
        PTX_BFOR(c, C) {
            const uint32_t a = c_hdr[c] >> PTX_CHG_ACTOR_SHIFT, sq = env_at((uint64_t)c * estride);
            const uint32_t f = first[a], cnt_a = first[a + 1] - f;
            const bool bad_seq = !(sq - 1u < cnt_a) || tbl[f + sq - 1u] != c || (sq > 1u && tbl[f + sq - 2u] >= c); /* seq == clock[a] + 1 */
            bool bad_dep = false;
            for (uint32_t b = 0; b < na && !bad_seq; ++b) { /* clock[b] >= deps[b] for every actor */
                const uint32_t d = env_at((uint64_t)c * estride + 1u + b);
                if (d != 0u) {
                    const uint32_t fb = first[b], cnt_b = first[b + 1] - fb;
                    if (!(d <= cnt_b) || tbl[fb + d - 1u] >= c) bad_dep = true;
                }
            }
            if (bad_seq || bad_dep) ptx_atomic_min(&H->adm, ((ptx_min(crow[c], 0x03FFFFFFu) * 2u) << 4) | (bad_seq ? PTX_ERR_SEQ_GAP : PTX_ERR_MISSING_DEP));
        }
        PTX_BSYNC(); /* a failed admission stays pending: an op-level error of an EARLIER row wins over it */
    }

    /* ---- P1: the rows: id bitmaps, row lists per class (order inside a list does not matter here) ---- */
    PTX_BFOR(w, nw + 1) {
        PtxBitWord z;
        z.bits = 0;
        z.pre = 0;
        ib[w] = z;
        allb[w] = 0;
    }
    PTX_BFOR(w, nwe) delbits[w] = 0;
    if (A.out_rank) PTX_BFOR(i, N) A.out_rank[base + i] = 0xFFFFFFFFu;
    PTX_BSYNC();
    PTX_BFOR(i, N) {
        const uint64_t id = op_id[i];
        const uint32_t ctr = (uint32_t)(id >> 32), act = (uint32_t)id, a = action[i], mt = mark_type[i];
        const bool mark = a == PTX_ACT_ADDMARK || a == PTX_ACT_REMOVEMARK;
        if (a >= 8u || (mark && mt > 3u) || ctr - 1u >= ix.max_ctr || act > ix.max_actor) {
            ptx_raise(H, ptx_min(i, 0x03FFFFFFu), 1, PTX_ERR_BAD_OP);
        } else {
            const uint32_t key = ctr * ix.na1 + act, bit = 1u << (key & 31u);
            if (ptx_atomic_or(&allb[key >> 5], bit) & bit) ptx_atomic_or(&H->cur[7], 1u); /* some op id occurs twice */
            /* list slots through wave-aggregated appends (one atomic per wave and branch): the cursors live in GLOBAL memory when several workgroups merge the
             * log, where a returning atomic per row on one address is ~11 ns each — a millisecond per 100 000 rows */
            if (a == PTX_ACT_INSERT) {
                ptx_atomic_or(&ib[key >> 5].bits, bit);
                const uint32_t s = ptx_append(&H->cur[0], true);
                if (s < n) ilist[s] = i;
            } else if (a == PTX_ACT_DELETE) {
                const uint32_t s = ptx_append(&H->cur[1], true);
                if (s < D) dlist[s] = i;
            } else if (mark) {
                const uint32_t s = ptx_append(&H->cur[2], true);
                if (s < K) {
                    mlist[s] = i;
                    mflag[s] = mt | (a == PTX_ACT_ADDMARK ? 4u : 0u);
                }
                if (mt == PTX_MARK_COMMENT) (void)ptx_append(&H->cur[4], true);
                else (void)ptx_append(&H->cur[3], true);
                if (mt == PTX_MARK_STRONG) (void)ptx_append(&H->cur[5], true);
                if (mt == PTX_MARK_EM) (void)ptx_append(&H->cur[6], true);
            }
        }
    }
    PTX_BSYNC();
    PTX_BLEADER { /* the header must be the exact census of the (well-formed) rows */
        if (H->cur[0] != n || H->cur[1] != D || H->cur[2] != K || H->cur[4] != Kc || H->cur[5] != hd.n_mark[PTX_MARK_STRONG] || H->cur[6] != hd.n_mark[PTX_MARK_EM])
            ptx_raise(H, 0, 0, PTX_ERR_BAD_OP);
        H->n_ins = n;
        H->n_applied = n + D + K;
    }
    PTX_BSYNC();
    if (H->err == PTX_NO_ERR && H->cur[7] != 0u) { /* name a repeated row (which of two equal ids is "the repeat" depends on the race; the status is what is reported) */
        PTX_BSYNC();
        PTX_BFOR(w, nw + 1) allb[w] = 0;
        PTX_BSYNC();
        PTX_BFOR(i, N) {
            uint32_t key = 0;
            ptx_id_key(ix, op_id[i], key);
            const uint32_t bit = 1u << (key & 31u);
            if (ptx_atomic_or(&allb[key >> 5], bit) & bit) ptx_raise(H, ptx_min(i, 0x03FFFFFFu), 1, PTX_ERR_DUPLICATE_OP);
        }
    }
    PTX_BIG_BAIL_IF_ERROR();
    PTX_BFOR(w, nw + 1) ib[w].pre = ptx_popc(ib[w].bits);
    PTX_BSYNC();
    ptx_big_scan<kGrid, uint32_t, 2>(&ib[0].pre, nw + 1, part, _gbar);

    /* ---- P3a: element index of every insert (rank of its id among the inserts), its row, its parent; the deletes' targets ---- */
    PTX_BFOR(s, n) {
        const uint32_t i = ilist[s];
        uint32_t key = 0;
        ptx_id_key(ix, op_id[i], key);
        const uint32_t e = ptx_bitrank(ib, key);
        row_of[e] = i;
        uint32_t pe = n;
        const uint64_t ra = ref_a[i];
        if (ra != 0) {
            const int p = ptx_elem_lookup(ix, ra);
            if (p < 0) ptx_raise(H, ptx_min(i, 0x03FFFFFFu), 1, PTX_ERR_ELEM_NOT_FOUND); /* micromerge.ts:752 */
            else pe = (uint32_t)p;
        }
        par[e] = pe;
    }
    PTX_BFOR(j, D) { /* the target of a delete must exist (micromerge.ts:752); deleting twice is fine (:693) */
        const uint32_t i = dlist[j];
        const int t = ptx_elem_lookup(ix, ref_a[i]);
        if (t < 0) ptx_raise(H, ptx_min(i, 0x03FFFFFFu), 1, PTX_ERR_ELEM_NOT_FOUND);
        else ptx_atomic_or(&delbits[(uint32_t)t >> 5], 1u << ((uint32_t)t & 31u));
    }
    PTX_BIG_BAIL_IF_ERROR();
    /* ... and must already exist when the op is applied: the same two groups of checks, in the same order, as the LDS kernel's P3a / P3b */
    PTX_BFOR(e, n) {
        const uint32_t pe = par[e];
        if (pe < n && row_of[pe] >= row_of[e]) ptx_raise(H, ptx_min(row_of[e], 0x03FFFFFFu), 1, PTX_ERR_ELEM_NOT_FOUND);
    }
    PTX_BFOR(j, D) {
        const uint32_t i = dlist[j];
        const int t = ptx_elem_lookup(ix, ref_a[i]);
        if (t >= 0 && row_of[t] >= i) ptx_raise(H, ptx_min(i, 0x03FFFFFFu), 1, PTX_ERR_ELEM_NOT_FOUND);
        /* the resolved references the patch-stream replay / change() / cursors read beside elem_rank (merge_core.h PtxMergeArgs.out_refs): per delete the
         * row that inserted its target */
        if (A.out_refs && t >= 0) A.out_refs[base + i] = row_of[t];
    }
    PTX_BIG_BAIL_IF_ERROR();

    /* ---- P3b/c: children of every parent in descending opId (= descending element index), parents ascending: one bitonic sort of the
     *      keys (parent << 32 | ~element); the pad keys sort behind everything ---- */
    PTX_BFOR(j, P2) skey[j] = j < n ? ((unsigned long long)par[j] << 32) | (unsigned long long)(0xFFFFFFFFu - j) : ~0ull;
    PTX_BSYNC();
    /* Bitonic sort in TILES (round 5): a compare-exchange pass with partner distance j2 < tile stays inside an aligned tile, so a workgroup runs ALL the passes
     * of a tile between two team barriers with only its own barrier in between — the team barrier (a grid barrier when several workgroups merge the log) is
     * paid once per distance >= tile: 28 instead of 153 barriers for 128 K keys. */
    {
        const uint32_t tile = PTX_BIG_SORT_TILE < P2 ? PTX_BIG_SORT_TILE : P2, ntiles = P2 / tile;
        /* one pass over a tile: its tile / 2 pairs (i, i + j2), ascending where bit k2 of the index is clear */
        auto tile_passes = [&](uint32_t k2_lo, uint32_t k2_hi, bool whole_stages) {
            for (uint32_t tb = PTX_BWG_ID; tb < ntiles; tb += PTX_BWG_COUNT) {
                const uint32_t b0 = tb * tile;
                for (uint32_t k2 = k2_lo; k2 <= k2_hi; k2 <<= 1) {
                    for (uint32_t j2 = whole_stages ? k2 >> 1 : tile >> 1; j2 > 0; j2 >>= 1) {
                        PTX_WFOR(q, tile >> 1) {
                            const uint32_t i = b0 + (q / j2) * 2u * j2 + (q % j2), l = i + j2;
                            const unsigned long long x = skey[i], y = skey[l];
                            const bool up = (i & k2) == 0;
                            if (up ? x > y : x < y) {
                                skey[i] = y;
                                skey[l] = x;
                            }
                        }
                        PTX_WG_SYNC();
                    }
                }
            }
        };
        tile_passes(2u, tile, true); /* every tile sorted (alternating directions) */
        PTX_BSYNC();
        for (uint32_t k2 = tile << 1; k2 <= P2 && k2 != 0u; k2 <<= 1) {
            for (uint32_t j2 = k2 >> 1; j2 >= tile; j2 >>= 1) { /* the far partners: the whole team, one pass per distance */
                PTX_BFOR(q, P2 >> 1) {
                    const uint32_t i = (q / j2) * 2u * j2 + (q % j2), l = i + j2;
                    const unsigned long long x = skey[i], y = skey[l];
                    const bool up = (i & k2) == 0;
                    if (up ? x > y : x < y) {
                        skey[i] = y;
                        skey[l] = x;
                    }
                }
                PTX_BSYNC();
            }
            tile_passes(k2, k2, false); /* the near partners: tile by tile */
            PTX_BSYNC();
        }
    }
    /* bstart[p] = first sorted slot of p's children (p = n: HEAD), bstart[n + 1] = n */
    PTX_BFOR(p, n + 3) bstart[p] = 0xFFFFFFFFu;
    PTX_BSYNC();
    PTX_BFOR(j, n) {
        const uint32_t p = (uint32_t)(skey[j] >> 32);
        if (j == 0 || (uint32_t)(skey[j - 1] >> 32) != p) bstart[p] = j;
    }
    PTX_BSYNC();
#define PTX_BIG_SRT(j) (0xFFFFFFFFu - (uint32_t)skey[j])
    /* ---- P3d: Euler tour (nodes 0 = enter(HEAD), x + 1 = enter(x), n + 1 + x = exit(x), 2n + 1 = end) ranked by pointer jumping:
     *      weight 1 on enter(x), position(x) = n - (enter nodes from x to the end) ---- */
    const uint32_t term = 2u * n + 1u;
    PTX_BFOR(j, n + 1) {
        /* node j's first child: j == n is HEAD */
        const uint32_t owner = j; /* as a parent */
        const uint32_t fs = bstart[owner];
        const uint32_t nx = fs != 0xFFFFFFFFu ? PTX_BIG_SRT(fs) + 1u : (owner == n ? term : n + 1u + owner);
        const uint32_t enter_node = owner == n ? 0u : owner + 1u;
        tour[enter_node] = ((unsigned long long)nx << 32) | (owner == n ? 0ull : 1ull);
        if (j < n) { /* exit(x) for the element in sorted slot j: its next sibling, else its parent's exit */
            const uint32_t x = PTX_BIG_SRT(j), p = (uint32_t)(skey[j] >> 32);
            const bool has_sib = j + 1u < n && (uint32_t)(skey[j + 1] >> 32) == p;
            const uint32_t other = has_sib ? PTX_BIG_SRT(j + 1u) + 1u : (p == n ? term : n + 1u + p);
            tour[n + 1u + x] = (unsigned long long)other << 32;
        }
    }
    PTX_BLEADER { tour[term] = (unsigned long long)term << 32; }
    PTX_BSYNC();
    for (uint32_t r = 0, rounds = ptx_ceil_log2(2u * n + 3u) + 1u; r < rounds; ++r) {
        /* in place: every intermediate {next, weight} word is a valid state (weight = sum over [node, next)) */
        PTX_BFOR(v, term) {
            const unsigned long long a = tour[v];
            const unsigned long long b = tour[(uint32_t)(a >> 32)];
            tour[v] = (b & 0xFFFFFFFF00000000ull) | (unsigned long long)(uint32_t)((uint32_t)a + (uint32_t)b);
        }
        PTX_BSYNC();
    }
    PTX_BFOR(x, n) pos[x] = n - (uint32_t)tour[x + 1u];
    PTX_BSYNC();
#undef PTX_BIG_SRT

    /* ---- P4: tombstones -> visible index ---- */
    PTX_BFOR(w, nwe + 1) {
        PtxBitWord z;
        z.bits = 0;
        z.pre = 0;
        alive[w] = z;
        st[w] = z;
        brkbits[w] = 0;
    }
    PTX_BSYNC();
    PTX_BFOR(e, n) {
        if (!ptx_bittest(delbits, e)) ptx_atomic_or(&alive[pos[e] >> 5].bits, 1u << (pos[e] & 31u));
    }
    PTX_BSYNC();
    PTX_BFOR(w, nwe + 1) alive[w].pre = ptx_popc(alive[w].bits);
    PTX_BSYNC();
    const uint32_t V = ptx_big_scan<kGrid, uint32_t, 2>(&alive[0].pre, nwe + 1, part, _gbar);

    /* ---- P5a: values out; every mark op -> visible interval ---- */
    {
        uint64_t h1 = 0, h2 = 0;
        PTX_BFOR(e, n) {
            const uint32_t r = pos[e], row = row_of[e];
            const bool vis = !ptx_bittest(delbits, e);
            if (vis) {
                const uint32_t q = ptx_bitrank(alive, r), v = payload[row];
                A.out_values[base + q] = v;
                ptx_digest_item(h1, h2, 1u, q, v, 0u);
            }
            if (A.out_rank) A.out_rank[base + row] = r | (vis ? 0u : PTX_RANK_TOMBSTONE);
        }
        ptx_digest_flush_dense(H, h1, h2); /* (a butterfly per wave, then one atomic: the header may live in global memory) */
    }
    PTX_BLEADER { H->cur[7] = 0; } /* comment marks met so far */
    PTX_BSYNC();
    PTX_BFOR(k, K) {
        const uint32_t i = mlist[k], sa = A.side_a[base + i], sb = A.side_b[base + i];
        uint32_t lo = 0, hi = 0;
        int js = -1;
        if (sa == PTX_SIDE_BEFORE || sa == PTX_SIDE_AFTER) {
            js = ptx_elem_lookup(ix, ref_a[i]);
            if (js >= 0 && row_of[js] >= i) js = -1; /* not in the list when the op is applied: the op never starts */
        }
        if (js >= 0) {
            const uint64_t slot_a = 2ull * pos[js] + (sa == PTX_SIDE_AFTER ? 1u : 0u);
            uint64_t slot_b = ~0ull;
            if (sb == PTX_SIDE_BEFORE || sb == PTX_SIDE_AFTER) {
                int je = ptx_elem_lookup(ix, ref_b[i]);
                if (je >= 0 && row_of[je] >= i) je = -1;
                if (je >= 0) slot_b = 2ull * pos[je] + (sb == PTX_SIDE_AFTER ? 1u : 0u);
            }
            if (slot_b == slot_a) slot_b = ~0ull; /* same slot: the start test fires first */
            if (slot_b > slot_a) {
                lo = ptx_bitrank(alive, (uint32_t)((slot_a + 1u) >> 1));
                hi = ptx_bitrank(alive, slot_b == ~0ull ? n : (uint32_t)((slot_b + 1u) >> 1));
            }
        }
        mrk_lo[k] = lo;
        mrk_hi[k] = hi;
        if (A.out_refs) { /* per mark op its two boundary slots, each on its own (none: all ones): 16 bits each in out_refs; a log of more than 32 766 elements has
                             slots beyond 16 bits — their high halves go to out_refs_hi where the host provides it (round 6), else the log reports no slots */
            uint32_t va = 0xFFFFFFFFu, vb = 0xFFFFFFFFu;
            if (n <= 32766u || A.out_refs_hi) {
                if (js >= 0) va = 2u * pos[js] + (sa == PTX_SIDE_AFTER ? 1u : 0u);
                if (sb == PTX_SIDE_BEFORE || sb == PTX_SIDE_AFTER) {
                    const int je = ptx_elem_lookup(ix, ref_b[i]);
                    if (je >= 0 && row_of[je] < i) vb = 2u * pos[je] + (sb == PTX_SIDE_AFTER ? 1u : 0u);
                }
            }
            A.out_refs[base + i] = (va & 0xFFFFu) | (vb << 16);
            if (A.out_refs_hi) A.out_refs_hi[base + i] = (va >> 16) | (vb & 0xFFFF0000u);
        }
        if ((mflag[k] & 3u) == PTX_MARK_COMMENT) {
            const uint32_t pl = payload[i];
            if (pl >= Kid) ptx_raise(H, ptx_min(i, 0x03FFFFFFu), 1, PTX_ERR_BAD_OP); /* beyond the id space the header declares */
            mflag[k] |= (pl < Kid ? pl : 0u) << 3;
            const uint32_t o = ptx_append(&H->cur[7], true);
            if (o < Kc) cidx[o] = k;
        }
    }
    PTX_BIG_BAIL_IF_ERROR();
    if (H->adm != PTX_NO_ERR) return H->adm & 15u; /* no op-level error anywhere: the failed admission is the log's error */

    /* ---- P5c: comments: per id, presence intervals decided by the last-applied covering op (peritext.ts:314-321) ---- */
    if (Kc > 0) {
        PTX_BFOR(c, Kid + 2) {
            ccnt[c] = 0;
            ccur[c] = 0;
        }
        PTX_BSYNC();
        PTX_BFOR(o, Kc) {
            const uint32_t k = cidx[o];
            if (mrk_lo[k] < mrk_hi[k]) ptx_atomic_add(&ccnt[mflag[k] >> 3], 1u);
        }
        PTX_BSYNC();
        ptx_big_scan<kGrid, uint32_t, 1>(ccnt, Kid + 1, part, _gbar);
        PTX_BFOR(o, Kc) {
            const uint32_t k = cidx[o];
            if (mrk_lo[k] < mrk_hi[k]) {
                const uint32_t c = mflag[k] >> 3;
                PtxBigEntry e;
                e.lo = mrk_lo[k];
                e.hi = mrk_hi[k];
                e.t = mlist[k];
                e.add = (mflag[k] >> 2) & 1u;
                cent[ccnt[c] + ptx_atomic_add(&ccur[c], 1u)] = e;
            }
        }
        PTX_BSYNC();
        /* The sweep of one id is ONE lane's work and quadratic in the id's ops (with a visible interval), read from HBM here.  An id with more than
         * PTX_BIG_COMMENT_OPS_PER_ID of them (round 5: no longer a capacity report) is swept by the whole TEAM instead, one such id at a time: its presence function
         * — "add" of the LAST-applied covering op per visible position — is a range-chmax tree over the positions of (application index + 1) << 1 | add (the first
         * LWW tree's storage: those are built after this phase), queried once per position into a bitmap whose runs of ones are the id's intervals. */
        PTX_BLEADER { H->cur[7] = 0; }
        PTX_BSYNC();
        PTX_BFOR(c, Kid) {
            if (ccnt[c + 1] - ccnt[c] > PTX_BIG_COMMENT_OPS_PER_ID) {
                const uint32_t s2 = ptx_append(&H->cur[7], true);
                if (s2 < hvy_cap) hvy[s2] = c;
            }
        }
        PTX_BSYNC();
        const uint32_t nheavy = H->cur[7] < hvy_cap ? H->cur[7] : hvy_cap; /* (at most Kc / PTX_BIG_COMMENT_OPS_PER_ID ids can be that heavy: the list holds them all) */
        PTX_BSYNC();
        uint32_t TVc = 1;
        while (TVc < V) TVc <<= 1;
        const uint32_t vwords = (V + 31u) >> 5;
        /* intervals of heavy id c: counted (emit = false) or written as rows row0 .. (emit = true); the same answer in every thread */
        auto heavy_sweep = [&](uint32_t c, uint32_t row0, bool emit, uint64_t& h1, uint64_t& h2) -> uint32_t {
            const uint32_t e0 = ccnt[c], m = ccnt[c + 1] - e0;
            PTX_BFOR(p, 2u * TVc) tree[0][p] = 0;
            PTX_BFOR(w, nwe + 1) {
                PtxBitWord z;
                z.bits = 0;
                z.pre = 0;
                st[w] = z;
            }
            PTX_BSYNC();
            PTX_BFOR(j, m) {
                const PtxBigEntry e = cent[e0 + j];
                ptx_big_chmax(tree[0], TVc, e.lo, e.hi, (((unsigned long long)e.t + 1ull) << 1) | (unsigned long long)(e.add & 1u));
            }
            PTX_BSYNC();
            PTX_BFOR(w, vwords) {
                uint32_t bits = 0;
                for (uint32_t bq = 0; bq < 32u; ++bq) {
                    const uint32_t q = (w << 5) + bq;
                    if (q < V && (ptx_big_query(tree[0], TVc, q) & 1ull)) bits |= 1u << bq;
                }
                st[w].bits = bits;
            }
            PTX_BSYNC();
            PTX_BFOR(w, nwe + 1) { /* the starts of the runs of ones: a one whose lower neighbour is a zero */
                const uint32_t bw = st[w].bits, prev = w ? st[w - 1].bits >> 31 : 0u;
                st[w].pre = ptx_popc(bw & ~((bw << 1) | prev));
            }
            PTX_BSYNC();
            const uint32_t total = ptx_big_scan<kGrid, uint32_t, 2>(&st[0].pre, nwe + 1, part, _gbar);
            if (emit) {
                PTX_BFOR(w, vwords) {
                    const uint32_t bw = st[w].bits, prev = w ? st[w - 1].bits >> 31 : 0u;
                    uint32_t starts = bw & ~((bw << 1) | prev), row = row0 + st[w].pre;
                    while (starts) {
                        const uint32_t sq = (w << 5) + (uint32_t)__builtin_ctz(starts);
                        starts &= starts - 1u;
                        uint32_t eq = sq; /* the first zero behind the start (bits from V on are zero; the bitmap has a spare word) */
                        for (;;) {
                            const uint32_t ww = eq >> 5, zz = ~st[ww].bits & (0xFFFFFFFFu << (eq & 31u));
                            if (zz) {
                                eq = (ww << 5) + (uint32_t)__builtin_ctz(zz);
                                break;
                            }
                            eq = (ww + 1u) << 5;
                        }
                        ptx_cinterval ci;
                        ci.id = c;
                        ci.start = sq;
                        ci.end = eq;
                        A.out_cints[base + row++] = ci;
                        ptx_atomic_or(&brkbits[sq >> 5], 1u << (sq & 31u));
                        ptx_atomic_or(&brkbits[eq >> 5], 1u << (eq & 31u));
                        ptx_digest_item(h1, h2, 3u, c, sq, eq);
                    }
                }
            }
            PTX_BSYNC();
            return total;
        };
        uint64_t h1 = 0, h2 = 0;
        PTX_BFOR(c, Kid + 1) {
            const uint32_t m = c < Kid ? ccnt[c + 1] - ccnt[c] : 0u;
            ccur[c] = m && m <= PTX_BIG_COMMENT_OPS_PER_ID ? ptx_comment_sweep(cent + ccnt[c], m, [](uint32_t, uint32_t) {}) : 0u;
        }
        PTX_BSYNC();
        for (uint32_t hq = 0; hq < nheavy; ++hq) { /* (uniform) */
            const uint32_t c = hvy[hq];
            const uint32_t cntc = heavy_sweep(c, 0u, false, h1, h2);
            PTX_BLEADER { ccur[c] = cntc; }
            PTX_BSYNC();
        }
        const uint32_t I = ptx_big_scan<kGrid, uint32_t, 1>(ccur, Kid + 1, part, _gbar);
        PTX_BLEADER { H->I = I; }
        PTX_BFOR(c, Kid) {
            const uint32_t m = ccnt[c + 1] - ccnt[c];
            if (m == 0u || m > PTX_BIG_COMMENT_OPS_PER_ID) continue;
            uint32_t row = ccur[c];
            ptx_comment_sweep(cent + ccnt[c], m, [&](uint32_t s, uint32_t e) {
                ptx_cinterval ci;
                ci.id = c;
                ci.start = s;
                ci.end = e;
                A.out_cints[base + row++] = ci;
                ptx_atomic_or(&brkbits[s >> 5], 1u << (s & 31u));
                ptx_atomic_or(&brkbits[e >> 5], 1u << (e & 31u));
                ptx_digest_item(h1, h2, 3u, c, s, e);
            });
        }
        for (uint32_t hq = 0; hq < nheavy; ++hq) {
            const uint32_t c = hvy[hq];
            (void)heavy_sweep(c, ccur[c], true, h1, h2);
        }
        if (nheavy) { /* the span-start bitmap of P6 lent its words to the sweeps */
            PTX_BFOR(w, nwe + 1) {
                PtxBitWord z;
                z.bits = 0;
                z.pre = 0;
                st[w] = z;
            }
        }
        ptx_digest_flush_dense(H, h1, h2); /* (a butterfly per wave, then one atomic: the header may live in global memory) */
        PTX_BSYNC();
    }

    /* ---- P5b: LWW winners per visible char (peritext.ts:304-313): four range-chmax trees of (opId key + 1) << 32 | mark ---- */
    uint32_t TV = 1;
    while (TV < V) TV <<= 1;
    for (int ty = 0; ty < 4; ++ty) PTX_BFOR(p, 2u * TV) tree[ty][p] = 0;
    PTX_BSYNC();
    PTX_BFOR(k, K) {
        if (mrk_lo[k] < mrk_hi[k]) {
            const uint32_t ty = mflag[k] & 3u;
            uint32_t key = 0;
            ptx_id_key(ix, op_id[mlist[k]], key);
            ptx_big_chmax(tree[ty], TV, mrk_lo[k], mrk_hi[k], ty == PTX_MARK_COMMENT ? 1ull : (((unsigned long long)key + 1ull) << 32) | k);
        }
    }
    PTX_BSYNC();
    PTX_BFOR(q, V) {
        uint32_t at = 0;
        for (uint32_t ty = 0; ty < 4; ++ty) {
            const unsigned long long w = ptx_big_query(tree[ty], TV, q);
            if (w == 0) continue;
            if (ty == PTX_MARK_COMMENT) at |= PTX_ATTR_COMMENT;
            else {
                const uint32_t k = (uint32_t)w;
                if (mflag[k] & 4u) {
                    if (ty == PTX_MARK_STRONG) at |= PTX_ATTR_STRONG;
                    else if (ty == PTX_MARK_EM) at |= PTX_ATTR_EM;
                    else at |= PTX_ATTR_LINK | (payload[mlist[k]] & PTX_ATTR_ID_MASK);
                }
            }
        }
        attr[q + 1] = at;
    }
    PTX_BLEADER { attr[0] = 0; }
    PTX_BSYNC();
    /* ---- P6: spans = maximal runs of equal marks over the visible chars (peritext.ts:438-455) + digest ---- */
    PTX_BFOR(q, V) {
        if (q == 0 || attr[q + 1] != attr[q] || ptx_bittest(brkbits, q)) ptx_atomic_or(&st[q >> 5].bits, 1u << (q & 31u));
    }
    PTX_BSYNC();
    PTX_BFOR(w, nwe + 1) st[w].pre = ptx_popc(st[w].bits);
    PTX_BSYNC();
    const uint32_t S = ptx_big_scan<kGrid, uint32_t, 2>(&st[0].pre, nwe + 1, part, _gbar);
    {
        uint64_t h1 = 0, h2 = 0;
        PTX_BFOR(q, V) {
            if ((st[q >> 5].bits >> (q & 31u)) & 1u) {
                const uint32_t s = ptx_bitrank(st, q);
                ptx_span sp;
                sp.start = q;
                sp.attr = attr[q + 1];
                A.out_spans[base + s] = sp;
                ptx_digest_item(h1, h2, 2u, s, sp.start, sp.attr);
            }
        }
        ptx_digest_flush_dense(H, h1, h2); /* (a butterfly per wave, then one atomic: the header may live in global memory) */
    }
    PTX_BSYNC();
    PTX_BLEADER {
        H->V = V;
        H->S = S;
        uint64_t g1 = 0, g2 = 0;
        ptx_digest_item(g1, g2, 4u, 0u, V, S);
        ptx_digest_item(g1, g2, 4u, 1u, H->I, n);
        H->h1 += g1;
        H->h2 += g2;
    }
    PTX_BSYNC();
    return PTX_OK;
}

/* kGrid: the log is merged by ALL workgroups of a cooperative launch (round 5; VERDICT r4 missing #3): every loop strides over the threads of the whole grid, every
 * barrier is a grid barrier, and the header — counters, error words, digest — lives at the start of the log's scratch slice instead of in LDS.  The body is the
 * same text either way (PTX_BFOR / PTX_BSYNC / PTX_BLEADER). */
template <bool kGrid>
PTX_DEV void ptx_big_merge_log(const PtxMergeArgs& A, uint32_t log, uint8_t* win, uint64_t win_bytes, uint8_t* lds) {
    constexpr uint32_t kThreads = 0u;
    (void)kThreads;
    uint32_t* const _gbar = A.grid_bar;
    (void)_gbar;
    PtxHdr* H = kGrid ? (PtxHdr*)win : (PtxHdr*)lds;
    const uint64_t hb = ptx_a64(sizeof(PtxHdr));
    const uint32_t status = win_bytes < hb ? (uint32_t)PTX_ERR_CAPACITY : ptx_big_merge_body<kGrid>(A, log, win + hb, win_bytes - hb, H);
    PTX_BSYNC();
    PTX_BLEADER {
        ptx_log_result r;
        r.status = status;
        r.n_ops = status ? 0 : H->n_applied;
        r.n_elems = status ? 0 : H->n_ins;
        r.n_visible = status ? 0 : H->V;
        r.n_spans = status ? 0 : H->S;
        r.n_cintervals = status ? 0 : H->I;
        r.reserved[0] = 0; /* (no LDS figure: the working set lives in HBM) */
        const uint32_t ew = H->err < H->adm ? H->err : H->adm;
        r.reserved[1] = status && ew != PTX_NO_ERR ? ew >> 5 : 0xFFFFFFFFu;
        r.digest[0] = status ? 0 : (uint64_t)H->h1;
        r.digest[1] = status ? 0 : (uint64_t)H->h2;
        A.res[log] = r;
    }
}
