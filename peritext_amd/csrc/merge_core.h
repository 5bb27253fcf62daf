/*
 * merge_core.h — the per-replica-log merge algorithm of the MI355X engine.
 *
 * One workgroup applies ONE replica op log and materialises its formatted document, entirely in
 * LDS: the op columns are read from HBM once, the outputs are written once.
 *
 * It replaces, for one log, the reference's sequential
 *     for change of log: doc.applyChange(change)        reference/src/micromerge.ts:499-514
 *         applyListInsert  (:614-672)   applyListUpdate (:677-724)   findListElement (:731-755)
 *         applyAddRemoveMark            reference/src/peritext.ts:154-249
 *     doc.getTextWithFormatting(["text"])               reference/src/peritext.ts:337-395, opsToMarks :294-326
 * with the order-independent closed form of SURVEY.md Appendix A.3/A.5/A.7:
 *   A  opId -> dense Lamport rank: a bitmap over (counter<<actorBits | actor) + popcount prefix
 *      (compareOpIds order, micromerge.ts:812-827); id -> op row lookup for every elemId reference
 *   B  RGA causal tree: element order = pre-order DFS, children by DESCENDING opId (equivalent to the
 *      skip loop at micromerge.ts:630-635): sort inserts by (parent, rank desc), link first-child /
 *      next-sibling, pre-order successor by pointer jumping, list ranking (Wyllie)
 *   C  tombstones: delete flags by target, popcount-prefix over "alive by rank" -> visible index
 *      (the `visible` counters of micromerge.ts:747-750)
 *   D  marks: boundary slots 2*rank+side -> element-rank interval -> visible interval; per visible
 *      char the max-opId covering op per non-multi mark type (LWW, peritext.ts:304-313) through a
 *      range-chmax tree; comments (allowMultiple, :314-321) as per-id presence intervals decided by
 *      the LAST-APPLIED covering op of that id
 *   E  spans = maximal runs of visible chars with equal marks (peritext.ts:438-455) + 128-bit digest
 *
 * The code is written as phases of `PTX_FOR` (a parallel loop over the workgroup) separated by
 * `PTX_SYNC()`; no iteration reads what another iteration of the same phase writes except through
 * commutative atomics.  That discipline lets the SAME source be compiled two ways:
 *   - by hipcc for gfx950 as the body of the kernel in merge_kernel.hip (the product), and
 *   - by g++ with -DPTX_EMU as a single-threaded emulation used ONLY by the CPU test-suite
 *     (tests/emu) to check the kernel's logic where no GPU exists.  The emulation is not linked
 *     into libperitext_hip.so and is never a fallback for the product path.
 */
#pragma once
#include <stdint.h>
#include "../../include/peritext_hip.h"

#ifdef PTX_EMU
#include <string.h>
#define PTX_DEV static inline
#define PTX_SYNC() ((void)0)
extern int ptx_emu_reverse; /* 1: run every parallel loop backwards (order-independence check) */
#define PTX_FOR(i, n)                                                                              \
    for (uint32_t _n = (n), _k = 0, i = (ptx_emu_reverse ? _n - 1 : 0); _k < _n;                  \
         ++_k, i = (ptx_emu_reverse ? _n - 1 - _k : _k))
#define PTX_LEADER if (true)
PTX_DEV uint32_t ptx_atomic_or(uint32_t* p, uint32_t v) { uint32_t o = *p; *p = o | v; return o; }
PTX_DEV uint32_t ptx_atomic_add(uint32_t* p, uint32_t v) { uint32_t o = *p; *p = o + v; return o; }
PTX_DEV uint32_t ptx_atomic_max(uint32_t* p, uint32_t v) { uint32_t o = *p; if (v > o) *p = v; return o; }
PTX_DEV uint32_t ptx_popc(uint32_t x) { return (uint32_t)__builtin_popcount(x); }
#else
#include <hip/hip_runtime.h>
#define PTX_DEV __device__ __forceinline__
#define PTX_SYNC() __syncthreads()
#define PTX_FOR(i, n) for (uint32_t i = threadIdx.x, _n = (n); i < _n; i += blockDim.x)
#define PTX_LEADER if (threadIdx.x == 0)
PTX_DEV uint32_t ptx_atomic_or(uint32_t* p, uint32_t v) { return atomicOr(p, v); }
PTX_DEV uint32_t ptx_atomic_add(uint32_t* p, uint32_t v) { return atomicAdd(p, v); }
PTX_DEV uint32_t ptx_atomic_max(uint32_t* p, uint32_t v) { return atomicMax(p, v); }
PTX_DEV uint32_t ptx_popc(uint32_t x) { return (uint32_t)__popc(x); }
#endif

/* kernel arguments: device pointers (host pointers under PTX_EMU) */
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
    ptx_log_result* res;
    uint32_t* out_values;
    ptx_span* out_spans;
    ptx_cinterval* out_cints;
    uint32_t* out_rank;
    uint32_t n_logs;
    uint32_t lds_bytes;
};

#define PTX_NONE 0xFFFFu

/* ---- digest: 128-bit multiset hash of the canonical output (restated in peritext_amd/canon.py) ---- */
PTX_DEV uint64_t ptx_fmix64(uint64_t x) {
    x ^= x >> 30;
    x *= 0xbf58476d1ce4e5b9ull;
    x ^= x >> 27;
    x *= 0x94d049bb133111ebull;
    x ^= x >> 31;
    return x;
}
PTX_DEV void ptx_digest_item(uint64_t& h1, uint64_t& h2, uint32_t tag, uint32_t a, uint32_t b, uint32_t c) {
    const uint64_t x = ((uint64_t)tag << 60) ^ ((uint64_t)a << 32) ^ (uint64_t)b;
    const uint64_t y = ptx_fmix64(x) ^ ((uint64_t)c * 0x9E3779B97F4A7C15ull);
    h1 += ptx_fmix64(y);
    h2 += ptx_fmix64(y ^ 0xD6E8FEB86659FD93ull);
}

/* ---- LDS header ---- */
struct PtxHdr {
    uint32_t status;
    uint32_t max_ctr, max_actor;
    uint32_t n_ins, n_marks, n_applied;
    uint32_t n_type[4]; /* mark ops per mark type */
    uint32_t cur_a, cur_b;
    uint32_t V, S, I;
    uint32_t pad;
    uint32_t scan_tmp[36];
    unsigned long long h1, h2;
};

PTX_DEV void ptx_digest_flush(PtxHdr* H, uint64_t h1, uint64_t h2) {
#ifdef PTX_EMU
    H->h1 += h1;
    H->h2 += h2;
#else
    /* wave-level butterfly first: one LDS atomic per wave instead of per lane */
    for (int d = 32; d >= 1; d >>= 1) {
        h1 += (uint64_t)__shfl_xor((unsigned long long)h1, d, 64);
        h2 += (uint64_t)__shfl_xor((unsigned long long)h2, d, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&H->h1, (unsigned long long)h1);
        atomicAdd(&H->h2, (unsigned long long)h2);
    }
#endif
}

/* ---- block-wide exclusive scan of an LDS array, in place; returns the total (all threads call it) ---- */
template <class T>
PTX_DEV uint32_t ptx_scan_excl(T* a, uint32_t m, uint32_t* tmp /* >= 36 u32 in LDS */) {
#ifdef PTX_EMU
    uint32_t run = 0;
    for (uint32_t j = 0; j < m; ++j) {
        uint32_t v = a[j];
        a[j] = (T)run;
        run += v;
    }
    (void)tmp;
    return run;
#else
    const uint32_t T_ = blockDim.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = (T_ + 63) >> 6;
    const uint32_t chunk = (m + T_ - 1) / T_;
    const uint32_t lo = tid * chunk < m ? tid * chunk : m;
    const uint32_t hi = lo + chunk < m ? lo + chunk : m;
    uint32_t sum = 0;
    for (uint32_t j = lo; j < hi; ++j) sum += a[j];
    uint32_t incl = sum;
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t v = __shfl_up(incl, d, 64);
        if ((int)lane >= d) incl += v;
    }
    if (lane == 63) tmp[wave] = incl;
    __syncthreads();
    if (wave == 0) {
        uint32_t w = lane < nwaves ? tmp[lane] : 0;
        uint32_t wi = w;
        for (int d = 1; d < 64; d <<= 1) {
            uint32_t v = __shfl_up(wi, d, 64);
            if ((int)lane >= d) wi += v;
        }
        if (lane < nwaves) tmp[lane] = wi - w;
        if (lane == 63) tmp[35] = wi;
    }
    __syncthreads();
    uint32_t run = tmp[wave] + incl - sum;
    for (uint32_t j = lo; j < hi; ++j) {
        uint32_t v = a[j];
        a[j] = (T)run;
        run += v;
    }
    const uint32_t total = tmp[35];
    __syncthreads();
    return total;
#endif
}

/* ---- bit-rank: bits[] + exclusive popcount prefix per 32-bit word ---- */
struct PtxBitRank {
    uint32_t* bits;
    uint16_t* pre;
};
PTX_DEV uint32_t ptx_bitrank(const PtxBitRank& b, uint32_t pos) { /* # set bits strictly below pos */
    const uint32_t w = pos >> 5, s = pos & 31;
    return (uint32_t)b.pre[w] + ptx_popc(b.bits[w] & ((1u << s) - 1u));
}
PTX_DEV bool ptx_bittest(const uint32_t* bits, uint32_t pos) { return (bits[pos >> 5] >> (pos & 31)) & 1u; }

/* ---- opId -> row index ---- */
struct PtxIdIndex {
    PtxBitRank br;
    uint16_t* by_rank;
    uint32_t abits, max_ctr, max_actor;
};
PTX_DEV bool ptx_id_key(const PtxIdIndex& ix, uint64_t id, uint32_t& key) {
    const uint32_t ctr = (uint32_t)(id >> 32), actor = (uint32_t)id;
    if (ctr == 0 || ctr > ix.max_ctr || actor > ix.max_actor) return false;
    key = (ctr << ix.abits) | actor;
    return true;
}
/* row of the op with this id inside the log, or -1 */
PTX_DEV int ptx_id_lookup(const PtxIdIndex& ix, uint64_t id) {
    uint32_t key;
    if (!ptx_id_key(ix, id, key)) return -1;
    if (!ptx_bittest(ix.br.bits, key)) return -1;
    return (int)ix.by_rank[ptx_bitrank(ix.br, key)];
}

/* ---- LDS bump allocator ---- */
struct PtxBump {
    uint8_t* base;
    uint32_t off, cap;
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
    return p;
}

PTX_DEV uint32_t ptx_ceil_log2(uint32_t x) { /* smallest k with (1<<k) >= x, x>=1 */
    uint32_t k = 0;
    while ((1u << k) < x) ++k;
    return k;
}

PTX_DEV void ptx_write_result(const PtxMergeArgs& A, uint32_t log, PtxHdr* H, uint32_t status) {
    PTX_LEADER {
        ptx_log_result r;
        r.status = status;
        r.n_ops = status ? 0 : H->n_applied;
        r.n_elems = status ? 0 : H->n_ins;
        r.n_visible = status ? 0 : H->V;
        r.n_spans = status ? 0 : H->S;
        r.n_cintervals = status ? 0 : H->I;
        r.reserved[0] = r.reserved[1] = 0;
        r.digest[0] = status ? 0 : (uint64_t)H->h1;
        r.digest[1] = status ? 0 : (uint64_t)H->h2;
        A.res[log] = r;
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
template <class F>
PTX_DEV uint32_t ptx_comment_sweep(const PtxCEntry* ent, uint32_t m, F emit) {
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

/* Uniform early exit on a per-log error.  The status word is sampled between two barriers so that a
 * later phase's error write can never be seen by a thread that is still at this check point. */
#define PTX_BAIL_IF_ERROR()                         \
    do {                                            \
        PTX_SYNC();                                 \
        const uint32_t _st = H->status;             \
        PTX_SYNC();                                 \
        if (_st) {                                  \
            ptx_write_result(A, log, H, _st);       \
            return;                                 \
        }                                           \
    } while (0)

/* ================================================================================================ */
PTX_DEV void ptx_merge_log(const PtxMergeArgs& A, uint32_t log, uint8_t* lds) {
    const uint64_t base = A.log_off[log];
    const uint64_t N64 = A.log_off[log + 1] - base;
    PtxHdr* H = (PtxHdr*)lds;
    PTX_LEADER {
        H->status = 0;
        H->max_ctr = H->max_actor = 0;
        H->n_ins = H->n_marks = H->n_applied = 0;
        H->n_type[0] = H->n_type[1] = H->n_type[2] = H->n_type[3] = 0;
        H->cur_a = H->cur_b = 0;
        H->V = H->S = H->I = 0;
        H->h1 = H->h2 = 0;
    }
    PTX_SYNC();
    if (N64 > 65534u) {
        ptx_write_result(A, log, H, PTX_ERR_CAPACITY);
        return;
    }
    const uint32_t N = (uint32_t)N64;
    const uint64_t* op_id = A.op_id + base;
    const uint64_t* ref_a = A.ref_a + base;
    const uint64_t* ref_b = A.ref_b + base;
    const uint32_t* payload = A.payload + base;

    PtxBump bp;
    bp.base = lds;
    bp.off = (uint32_t)((sizeof(PtxHdr) + 15u) & ~15u);
    bp.cap = A.lds_bytes;
    bp.overflow = false;

    uint8_t* kind = ptx_alloc<uint8_t>(bp, N);        /* action | mark_type << 4, per op row */
    uint16_t* by_rank = ptx_alloc<uint16_t>(bp, N);   /* Lamport rank -> op row */
    uint32_t* delbits = ptx_alloc<uint32_t>(bp, (N + 31) / 32 + 1); /* tombstone flag per op row */
    uint16_t* rnk = ptx_alloc<uint16_t>(bp, N);       /* document position (incl. tombstones) per insert row */
    if (bp.overflow) {
        ptx_write_result(A, log, H, PTX_ERR_CAPACITY);
        return;
    }

    /* ---- A0: load, classify, reduce ---- */
    {
        uint32_t mc = 0, ma = 0, ni = 0, nm = 0, nap = 0, bad = 0;
        uint32_t nt0 = 0, nt1 = 0, nt2 = 0, nt3 = 0;
        PTX_FOR(i, N) {
            const uint64_t id = op_id[i];
            const uint32_t ctr = (uint32_t)(id >> 32), act = (uint32_t)id;
            const uint32_t a = A.action[base + i], mt = A.mark_type[base + i];
            kind[i] = (uint8_t)((a & 15u) | ((mt & 15u) << 4));
            mc = ctr > mc ? ctr : mc;
            ma = act > ma ? act : ma;
            if (ctr == 0 || a > PTX_ACT_NOP) bad = 1;
            if (a == PTX_ACT_INSERT) ni++;
            if (a == PTX_ACT_ADDMARK || a == PTX_ACT_REMOVEMARK) {
                nm++;
                if (mt > 3) bad = 1;
                else if (mt == 0) nt0++;
                else if (mt == 1) nt1++;
                else if (mt == 2) nt2++;
                else nt3++;
            }
            if (a != PTX_ACT_MAKELIST && a != PTX_ACT_NOP) nap++;
        }
        PTX_FOR(w, (N + 31) / 32 + 1) delbits[w] = 0;
        ptx_atomic_max(&H->max_ctr, mc);
        ptx_atomic_max(&H->max_actor, ma);
        ptx_atomic_add(&H->n_ins, ni);
        ptx_atomic_add(&H->n_marks, nm);
        ptx_atomic_add(&H->n_applied, nap);
        ptx_atomic_add(&H->n_type[0], nt0);
        ptx_atomic_add(&H->n_type[1], nt1);
        ptx_atomic_add(&H->n_type[2], nt2);
        ptx_atomic_add(&H->n_type[3], nt3);
        if (bad) ptx_atomic_max(&H->status, PTX_ERR_BAD_OP);
    }
    PTX_BAIL_IF_ERROR();

    /* ---- A1..A3: Lamport rank of every op id ---- */
    PtxIdIndex ix;
    ix.max_ctr = H->max_ctr;
    ix.max_actor = H->max_actor;
    ix.abits = ptx_ceil_log2(ix.max_actor + 1);
    ix.by_rank = by_rank;
    if (ix.abits > 12 || ix.max_ctr >= (1u << 19)) { /* keyspace must stay far below 2^31 bits */
        ptx_write_result(A, log, H, PTX_ERR_CAPACITY);
        return;
    }
    const uint32_t keyspace = (ix.max_ctr + 1u) << ix.abits;
    const uint32_t nw = (keyspace + 31) / 32;
    ix.br.bits = ptx_alloc<uint32_t>(bp, nw + 1);
    ix.br.pre = ptx_alloc<uint16_t>(bp, nw + 1);
    if (bp.overflow) {
        ptx_write_result(A, log, H, PTX_ERR_CAPACITY);
        return;
    }
    PTX_FOR(w, nw + 1) ix.br.bits[w] = 0;
    PTX_SYNC();
    PTX_FOR(i, N) {
        uint32_t key = 0;
        ptx_id_key(ix, op_id[i], key);
        const uint32_t bit = 1u << (key & 31);
        if (ptx_atomic_or(&ix.br.bits[key >> 5], bit) & bit) ptx_atomic_max(&H->status, PTX_ERR_DUPLICATE_OP);
    }
    PTX_SYNC();
    PTX_FOR(w, nw + 1) ix.br.pre[w] = (uint16_t)ptx_popc(ix.br.bits[w]);
    PTX_SYNC();
    ptx_scan_excl(ix.br.pre, nw + 1, H->scan_tmp);
    PTX_BAIL_IF_ERROR();
    PTX_FOR(i, N) {
        uint32_t key = 0;
        ptx_id_key(ix, op_id[i], key);
        by_rank[ptx_bitrank(ix.br, key)] = (uint16_t)i;
    }
    PTX_SYNC();

    const uint32_t n = H->n_ins;
    const uint32_t mark_lds = bp.off; /* everything above this mark is phase scratch */

    /* ---- B: causal tree of the inserts, tombstone flags ---- */
    {
        const uint32_t M = (N + 2u) & ~1u; /* nodes: op rows 0..N-1 plus ROOT = N; even for alignment */
        uint16_t* parent = ptx_alloc<uint16_t>(bp, M);
        uint16_t* fc = ptx_alloc<uint16_t>(bp, M);
        uint16_t* ns = ptx_alloc<uint16_t>(bp, M);
        uint16_t* X = ptx_alloc<uint16_t>(bp, 4 * M); /* 4 work arrays; the sort keys alias them */
        if (bp.overflow) {
            ptx_write_result(A, log, H, PTX_ERR_CAPACITY);
            return;
        }
        uint16_t* X1 = X;
        uint16_t* X2 = X + M;
        uint16_t* X3 = X + 2 * M;
        uint16_t* X4 = X + 3 * M;
        uint32_t* keys = (uint32_t*)X;
        uint32_t P2 = 1;
        while (P2 < n) P2 <<= 1; /* 4*P2 < 8*N <= sizeof(X) */

        PTX_FOR(i, N + 1) {
            fc[i] = PTX_NONE;
            ns[i] = PTX_NONE;
        }
        PTX_FOR(k, P2) keys[k] = 0xFFFFFFFFu;
        PTX_SYNC();
        PTX_FOR(i, N) {
            const uint32_t a = kind[i] & 15u;
            if (a == PTX_ACT_INSERT) {
                const uint64_t ra = ref_a[i];
                int p = (int)N;
                if (ra != 0) {
                    p = ptx_id_lookup(ix, ra);
                    /* the reference element must already exist when the op is applied (micromerge.ts:752) */
                    if (p < 0 || (uint32_t)p >= i || (kind[p] & 15u) != PTX_ACT_INSERT) {
                        ptx_atomic_max(&H->status, PTX_ERR_ELEM_NOT_FOUND);
                        p = (int)N;
                    }
                }
                parent[i] = (uint16_t)p;
                uint32_t key = 0;
                ptx_id_key(ix, op_id[i], key);
                const uint32_t r = ptx_bitrank(ix.br, key);
                keys[ptx_atomic_add(&H->cur_a, 1u)] = ((uint32_t)p << 16) | (0xFFFFu - r);
            } else if (a == PTX_ACT_DELETE) {
                const int t = ptx_id_lookup(ix, ref_a[i]);
                if (t < 0 || (uint32_t)t >= i || (kind[t] & 15u) != PTX_ACT_INSERT) {
                    ptx_atomic_max(&H->status, PTX_ERR_ELEM_NOT_FOUND);
                } else {
                    ptx_atomic_or(&delbits[t >> 5], 1u << (t & 31));
                }
            }
        }
        PTX_BAIL_IF_ERROR();

        /* siblings: ascending (parent, 0xFFFF - rank) == per parent, DESCENDING opId */
        for (uint32_t k = 2; k <= P2; k <<= 1) {
            for (uint32_t j = k >> 1; j > 0; j >>= 1) {
                PTX_FOR(i, P2) {
                    const uint32_t l = i ^ j;
                    if (l > i) {
                        const uint32_t x = keys[i], y = keys[l];
                        const bool asc = (i & k) == 0;
                        if ((x > y) == asc) {
                            keys[i] = y;
                            keys[l] = x;
                        }
                    }
                }
                PTX_SYNC();
            }
        }
        PTX_FOR(k, n) {
            const uint32_t key = keys[k];
            const uint32_t p = key >> 16;
            const uint32_t x = by_rank[0xFFFFu - (key & 0xFFFFu)];
            if (k == 0 || (keys[k - 1] >> 16) != p) fc[p] = (uint16_t)x;
            uint32_t nx = PTX_NONE;
            if (k + 1 < n && (keys[k + 1] >> 16) == p) nx = by_rank[0xFFFFu - (keys[k + 1] & 0xFFFFu)];
            ns[x] = (uint16_t)nx;
        }
        PTX_SYNC();

        if (n > 0) {
            /* nearest ancestor-or-self that has a next sibling (ROOT is a fixed point) */
            uint16_t* upA = X1;
            uint16_t* upB = X2;
            PTX_FOR(x, N + 1) {
                if (x == N) upA[x] = (uint16_t)N;
                else if ((kind[x] & 15u) == PTX_ACT_INSERT) upA[x] = ns[x] != PTX_NONE ? (uint16_t)x : parent[x];
            }
            PTX_SYNC();
            for (uint32_t span = 1; span < n + 1; span <<= 1) {
                PTX_FOR(x, N + 1) {
                    if (x == N) upB[x] = (uint16_t)N;
                    else if ((kind[x] & 15u) == PTX_ACT_INSERT) upB[x] = upA[upA[x]];
                }
                PTX_SYNC();
                uint16_t* t = upA;
                upA = upB;
                upB = t;
            }
            /* pre-order successor, then distance to the end of the list by pointer jumping */
            uint16_t* nxA = X3;
            uint16_t* dA = X4;
            PTX_FOR(x, N) {
                if ((kind[x] & 15u) == PTX_ACT_INSERT) {
                    uint32_t s = fc[x];
                    if (s == PTX_NONE) s = ns[upA[x]]; /* ns[ROOT] == NONE: end of list */
                    nxA[x] = (uint16_t)s;
                    dA[x] = s == PTX_NONE ? 0 : 1;
                }
            }
            PTX_SYNC();
            uint16_t* nxB = X1; /* the two `up` arrays are dead now: reuse them as the ping-pong halves */
            uint16_t* dB = X2;
            for (uint32_t span = 1; span < n; span <<= 1) {
                PTX_FOR(x, N) {
                    if ((kind[x] & 15u) == PTX_ACT_INSERT) {
                        const uint32_t nx = nxA[x];
                        if (nx != PTX_NONE) {
                            dB[x] = (uint16_t)(dA[x] + dA[nx]);
                            nxB[x] = nxA[nx];
                        } else {
                            dB[x] = dA[x];
                            nxB[x] = PTX_NONE;
                        }
                    }
                }
                PTX_SYNC();
                uint16_t* t = nxA;
                nxA = nxB;
                nxB = t;
                t = dA;
                dA = dB;
                dB = t;
            }
            PTX_FOR(x, N) {
                if ((kind[x] & 15u) == PTX_ACT_INSERT) rnk[x] = (uint16_t)(n - 1u - dA[x]);
            }
            PTX_SYNC();
        }
    }
    bp.off = mark_lds; /* release the tree scratch */

    /* ---- C: tombstone-aware visible index ---- */
    const uint32_t nwv = n / 32 + 1; /* bit positions 0..n */
    PtxBitRank alive;
    alive.bits = ptx_alloc<uint32_t>(bp, nwv + 1);
    alive.pre = ptx_alloc<uint16_t>(bp, nwv + 1);
    const uint32_t K = H->n_marks;
    uint16_t* mrk_op = ptx_alloc<uint16_t>(bp, K + 1);
    uint16_t* mrk_lo = ptx_alloc<uint16_t>(bp, K + 1);
    uint16_t* mrk_hi = ptx_alloc<uint16_t>(bp, K + 1);
    uint16_t* mrk_r = ptx_alloc<uint16_t>(bp, K + 1);
    if (bp.overflow) {
        ptx_write_result(A, log, H, PTX_ERR_CAPACITY);
        return;
    }
    PTX_FOR(w, nwv + 1) alive.bits[w] = 0;
    PTX_SYNC();
    PTX_FOR(x, N) {
        if ((kind[x] & 15u) == PTX_ACT_INSERT && !ptx_bittest(delbits, x)) {
            const uint32_t r = rnk[x];
            ptx_atomic_or(&alive.bits[r >> 5], 1u << (r & 31));
        }
    }
    PTX_SYNC();
    PTX_FOR(w, nwv + 1) alive.pre[w] = (uint16_t)ptx_popc(alive.bits[w]);
    PTX_SYNC();
    const uint32_t V = ptx_scan_excl(alive.pre, nwv + 1, H->scan_tmp);
    PTX_SYNC();
    {
        uint64_t h1 = 0, h2 = 0;
        PTX_FOR(x, N) {
            uint32_t rr = 0xFFFFFFFFu;
            if ((kind[x] & 15u) == PTX_ACT_INSERT) {
                rr = rnk[x];
                if (!ptx_bittest(delbits, x)) {
                    const uint32_t q = ptx_bitrank(alive, rr);
                    const uint32_t v = payload[x];
                    A.out_values[base + q] = v;
                    ptx_digest_item(h1, h2, 1u, q, v, 0u);
                }
            }
            if (A.out_rank) A.out_rank[base + x] = rr;
        }
        ptx_digest_flush(H, h1, h2);
    }

    /* ---- D1: every mark op -> visible interval [lo, hi) ---- */
    PTX_FOR(i, N) {
        const uint32_t a = kind[i] & 15u;
        if (a == PTX_ACT_ADDMARK || a == PTX_ACT_REMOVEMARK) {
            const uint32_t k = ptx_atomic_add(&H->cur_b, 1u);
            const uint32_t sa = A.side_a[base + i], sb = A.side_b[base + i];
            uint32_t lo = 0, hi = 0;
            /* start: only before/after(elem) can ever match a slot (peritext.ts:236); an element that is
               not in the list when the op is applied means the op never starts */
            int js = -1;
            if (sa == PTX_SIDE_BEFORE || sa == PTX_SIDE_AFTER) {
                js = ptx_id_lookup(ix, ref_a[i]);
                if (js >= 0 && ((uint32_t)js >= i || (kind[js] & 15u) != PTX_ACT_INSERT)) js = -1;
            }
            if (js >= 0) {
                const uint32_t slot_a = 2u * rnk[js] + (sa == PTX_SIDE_AFTER ? 1u : 0u);
                uint32_t slot_b = 0xFFFFFFFFu; /* never reached: runs to the end of the text */
                if (sb == PTX_SIDE_BEFORE || sb == PTX_SIDE_AFTER) {
                    int je = ptx_id_lookup(ix, ref_b[i]);
                    if (je >= 0 && ((uint32_t)je >= i || (kind[je] & 15u) != PTX_ACT_INSERT)) je = -1;
                    if (je >= 0) slot_b = 2u * rnk[je] + (sb == PTX_SIDE_AFTER ? 1u : 0u);
                }
                /* same slot: the start test fires first and the end is never seen */
                if (slot_b == slot_a) slot_b = 0xFFFFFFFFu;
                if (slot_b > slot_a) {
                    const uint32_t lo_rank = (slot_a + 1u) >> 1;
                    const uint32_t hi_rank = slot_b == 0xFFFFFFFFu ? n : (slot_b + 1u) >> 1;
                    lo = ptx_bitrank(alive, lo_rank);
                    hi = ptx_bitrank(alive, hi_rank);
                }
            }
            uint32_t key = 0;
            ptx_id_key(ix, op_id[i], key);
            mrk_op[k] = (uint16_t)i;
            mrk_lo[k] = (uint16_t)lo;
            mrk_hi[k] = (uint16_t)hi;
            mrk_r[k] = (uint16_t)ptx_bitrank(ix.br, key);
        }
    }
    PTX_SYNC();

    /* ---- D2: per visible char, the winning op of each non-multi mark type (LWW by opId) ---- */
    uint32_t P2V = 1;
    while (P2V < V) P2V <<= 1;
    uint32_t* tree = ptx_alloc<uint32_t>(bp, 2 * P2V);
    uint32_t* attr = ptx_alloc<uint32_t>(bp, V + 1);
    const uint32_t nwq = V / 32 + 1;
    uint32_t* brkbits = ptx_alloc<uint32_t>(bp, nwq + 1);
    PtxBitRank st;
    st.bits = ptx_alloc<uint32_t>(bp, nwq + 1);
    st.pre = ptx_alloc<uint16_t>(bp, nwq + 1);
    if (bp.overflow) {
        ptx_write_result(A, log, H, PTX_ERR_CAPACITY);
        return;
    }
    PTX_FOR(q, V + 1) attr[q] = 0;
    PTX_FOR(w, nwq + 1) {
        brkbits[w] = 0;
        st.bits[w] = 0;
    }
    PTX_SYNC();
    for (uint32_t pass = 0; pass < 4; ++pass) {
        /* pass = mark type; the comment pass only asks "is any comment op covering" (key present) */
        if (V == 0 || H->n_type[pass] == 0) continue;
        PTX_FOR(p, 2 * P2V) tree[p] = 0;
        PTX_SYNC();
        PTX_FOR(k, K) {
            const uint32_t i = mrk_op[k];
            if ((uint32_t)(kind[i] >> 4) == pass && mrk_lo[k] < mrk_hi[k]) {
                ptx_tree_chmax(tree, P2V, mrk_lo[k], mrk_hi[k], pass == PTX_MARK_COMMENT ? 1u : (uint32_t)mrk_r[k] + 1u);
            }
        }
        PTX_SYNC();
        PTX_FOR(q, V) {
            const uint32_t w = ptx_tree_query(tree, P2V, q);
            if (w != 0) {
                if (pass == PTX_MARK_COMMENT) {
                    attr[q] |= PTX_ATTR_COMMENT;
                } else {
                    const uint32_t i = by_rank[w - 1u];
                    if ((kind[i] & 15u) == PTX_ACT_ADDMARK) {
                        if (pass == PTX_MARK_STRONG) attr[q] |= PTX_ATTR_STRONG;
                        else if (pass == PTX_MARK_EM) attr[q] |= PTX_ATTR_EM;
                        else attr[q] |= PTX_ATTR_LINK | (payload[i] & PTX_ATTR_ID_MASK);
                    }
                }
            }
        }
        PTX_SYNC();
    }

    /* ---- D3: comments: per id, presence intervals decided by the last-applied covering op ---- */
    const uint32_t Kc = H->n_type[PTX_MARK_COMMENT];
    if (Kc > 0) {
        uint32_t* ccnt = ptx_alloc<uint32_t>(bp, Kc + 1);
        uint32_t* ccur = ptx_alloc<uint32_t>(bp, Kc + 1);
        uint32_t* cicnt = ptx_alloc<uint32_t>(bp, Kc + 1);
        PtxCEntry* cent = ptx_alloc<PtxCEntry>(bp, Kc + 1);
        if (bp.overflow) {
            ptx_write_result(A, log, H, PTX_ERR_CAPACITY);
            return;
        }
        PTX_FOR(c, Kc + 1) {
            ccnt[c] = 0;
            ccur[c] = 0;
            cicnt[c] = 0;
        }
        PTX_SYNC();
        PTX_FOR(k, K) {
            const uint32_t i = mrk_op[k];
            if ((uint32_t)(kind[i] >> 4) == PTX_MARK_COMMENT) {
                const uint32_t c = payload[i];
                if (c >= Kc) ptx_atomic_max(&H->status, PTX_ERR_BAD_OP); /* ids must be dense per doc */
                else if (mrk_lo[k] < mrk_hi[k]) ptx_atomic_add(&ccnt[c], 1u);
            }
        }
        PTX_BAIL_IF_ERROR();
        ptx_scan_excl(ccnt, Kc + 1, H->scan_tmp); /* ccnt[c] = first entry of id c, ccnt[Kc] = total */
        PTX_SYNC();
        PTX_FOR(k, K) {
            const uint32_t i = mrk_op[k];
            if ((uint32_t)(kind[i] >> 4) == PTX_MARK_COMMENT && mrk_lo[k] < mrk_hi[k]) {
                const uint32_t c = payload[i];
                const uint32_t pos = ccnt[c] + ptx_atomic_add(&ccur[c], 1u);
                PtxCEntry e;
                e.lo = mrk_lo[k];
                e.hi = mrk_hi[k];
                e.t = (uint16_t)i;
                e.add = (kind[i] & 15u) == PTX_ACT_ADDMARK ? 1 : 0;
                cent[pos] = e;
            }
        }
        PTX_SYNC();
        PTX_FOR(c, Kc) {
            cicnt[c] = ptx_comment_sweep(cent + ccnt[c], ccnt[c + 1] - ccnt[c], [](uint32_t, uint32_t) {});
        }
        PTX_SYNC();
        const uint32_t I = ptx_scan_excl(cicnt, Kc + 1, H->scan_tmp);
        PTX_SYNC();
        PTX_LEADER { H->I = I; }
        {
            uint64_t h1 = 0, h2 = 0;
            PTX_FOR(c, Kc) {
                uint32_t row = cicnt[c];
                ptx_comment_sweep(cent + ccnt[c], ccnt[c + 1] - ccnt[c], [&](uint32_t s, uint32_t e) {
                    ptx_cinterval ci;
                    ci.id = c;
                    ci.start = s;
                    ci.end = e;
                    A.out_cints[base + row++] = ci;
                    ptx_atomic_or(&brkbits[s >> 5], 1u << (s & 31));
                    ptx_atomic_or(&brkbits[e >> 5], 1u << (e & 31)); /* e <= V: bit V is never read */
                    ptx_digest_item(h1, h2, 3u, c, s, e);
                });
            }
            ptx_digest_flush(H, h1, h2);
        }
        PTX_SYNC();
    }

    /* ---- E: spans = maximal runs of equal marks over the visible chars ---- */
    PTX_FOR(q, V) {
        const bool is_start = q == 0 || attr[q] != attr[q - 1] || ptx_bittest(brkbits, q);
        if (is_start) ptx_atomic_or(&st.bits[q >> 5], 1u << (q & 31));
    }
    PTX_SYNC();
    PTX_FOR(w, nwq + 1) st.pre[w] = (uint16_t)ptx_popc(st.bits[w]);
    PTX_SYNC();
    const uint32_t S = ptx_scan_excl(st.pre, nwq + 1, H->scan_tmp);
    PTX_SYNC();
    {
        uint64_t h1 = 0, h2 = 0;
        PTX_FOR(q, V) {
            if (ptx_bittest(st.bits, q)) {
                const uint32_t s = ptx_bitrank(st, q);
                ptx_span sp;
                sp.start = q;
                sp.attr = attr[q];
                A.out_spans[base + s] = sp;
                ptx_digest_item(h1, h2, 2u, s, q, sp.attr);
            }
        }
        ptx_digest_flush(H, h1, h2);
    }
    PTX_SYNC();
    PTX_LEADER {
        H->V = V;
        H->S = S;
        uint64_t h1 = 0, h2 = 0;
        ptx_digest_item(h1, h2, 4u, 0u, V, S);
        ptx_digest_item(h1, h2, 4u, 1u, H->I, n);
        H->h1 += h1;
        H->h2 += h2;
    }
    PTX_SYNC();
    ptx_write_result(A, log, H, PTX_OK);
}
