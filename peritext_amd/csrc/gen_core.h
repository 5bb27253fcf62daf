/*
 * gen_core.h — on-device `change()`: index-based InputOperations -> id-based ops, one document per wave.
 *
 * What it replaces: `Micromerge.change` (reference/src/micromerge.ts:308-441) with `getListElementId` incl.
 * `lookAfterTombstones` (:762-805) and `changeMark` (src/peritext.ts:458-501), driven by the workload of the reference's
 * own fuzzer (test/fuzz.ts:115-205: random insert / delete / addMark / removeMark on a random replica, pairwise syncs with
 * `applyChange` and retry on the causal RangeError, final full sync) in its seeded restatement PTXGEN
 * (oracle/ptxgen.js: mulberry32 draws, same decision order).  Output = the replica logs in application order as the SoA
 * op-log columns + Change envelope of include/peritext_hip.h, written straight into HBM: a generated batch goes to
 * ptx_merge without ever visiting the host.
 *
 * State per DOCUMENT, in LDS (round 5: ONE element list for all its replicas — the RGA order of two elements does not depend on the replica, SURVEY A.3, so a
 * replica's list is the document's list restricted to the elements that replica has applied; rounds 1-4 kept a list per replica: three times the LDS, and every
 * delivered insert opened a gap again):
 *   the list of ALL elements made so far, in document order (every insert enters it when it is made, by the reference's own skip rule, micromerge.ts:630-635),
 *   key = counter << 2 | actor   (integer order == compareOpIds order, micromerge.ts:812-827; actors "doc1".."doc4"), u16 while
 *         the counters of the document stay below 2^14 (ptx_gen_small_keys), else u32: 16 bytes = 8 or 4 keys per lane and read
 * and per replica two bit planes by list position: `dead` = the element is NOT VISIBLE in this replica (not applied yet, or deleted), `after` = the element's
 *         `after` slot is a defined one (markOpsAfter !== undefined: set when a non-inclusive mark op ends on it, peritext.ts:240) — what `lookAfterTombstones`
 *         looks at (an element the replica has not applied has no such slot and is walked over like a tombstone without one).  The k-th visible element is
 *         one popcount scan over the `dead` plane; applying a REMOTE insert is one look-up and one bit.
 * plus clock / maxOp / visible length.  Marks need no other state to GENERATE ops.
 * Everything is uniform control flow over ONE wave: lanes talk through LDS with PTX_WSYNC (a compiler fence: the LDS serves a wave's
 * accesses in order); the full barrier, which also waits for the wave's outstanding stores to HBM, only stands where rows, change
 * records or comment ids written earlier are read back (deliver, the first change, `known`, the final remap).
 * Uniform control flow throughout; the list primitives (find an id, select the k-th visible element, open a
 * gap) live in ptx_platform_gfx950.h.  change_core.h keeps the older one-word-per-element form (ptx_gen_select & co below).
 */
#pragma once
#include "merge_core.h"

#define PTX_GEN_MAX_R 8u /* replicas per document (round 5: 8; "doc1" .. "doc8": their string order is their number's).  The document code is compiled for 4 and for 8
                           * (kMaxR): the per-replica state of a document of up to four replicas — the usual case — keeps round 4's size and registers */
#define PTX_GK_DEAD 0x40000000u
#define PTX_GK_AFTER 0x80000000u
#define PTX_GK_KEY 0x3FFFFFFFu

/* one made change: where its rows sit in its author's log + what applyChange checks */
template <uint32_t kMaxR>
struct PtxGenChangeT {
    uint32_t rowoff;
    uint32_t nops_start; /* nops << 24 | startOp */
    uint32_t deps[kMaxR / 2]; /* deps[2 j] | deps[2 j + 1] << 16 */
};
PTX_HD uint32_t ptx_gen_max_r(uint64_t R) { return R <= 4 ? 4u : PTX_GEN_MAX_R; } /* the build (kMaxR) that generates documents of R replicas */
PTX_HD uint64_t ptx_gen_change_bytes(uint64_t R) { return R <= 4 ? sizeof(PtxGenChangeT<4>) : sizeof(PtxGenChangeT<PTX_GEN_MAX_R>); }

struct PtxGenArgs {
    uint32_t n_docs, first_doc, seed;
    uint32_t R, ops_per_log;
    uint32_t mix0, mix01, mix012; /* cumulative percent thresholds: insert | delete | addMark | removeMark */
    uint32_t n_mark_types;
    uint8_t mark_types[4];        /* PTX_MARK_* in the config's order */
    uint32_t init_len;
    uint8_t init_text[16];
    uint32_t rows_per_log;        /* ops_per_log + 1 (the makeList) */
    uint32_t list_cap;            /* elements per replica list the LDS holds */
    uint32_t lds_bytes;
    /* outputs: log l = doc * R + r owns rows [l * rows_per_log, (l + 1) * rows_per_log) of every column and the same
     * range of the envelope columns (compacted afterwards by the host library) */
    uint64_t* op_id;
    uint64_t* ref_a;
    uint64_t* ref_b;
    uint32_t* payload;
    uint8_t* action;
    uint8_t* mark_type;
    uint8_t* side_a;
    uint8_t* side_b;
    uint32_t* chg_hdr;    /* actor << PTX_CHG_ACTOR_SHIFT | nops */
    uint16_t* chg_env;    /* rows of PTX_ENV_STRIDE(R) u16: seq, deps[R] */
    uint32_t* n_changes;  /* [n_docs * R] */
    uint32_t* n_comments; /* [n_docs] comment ids "comment-0" .. "comment-(C-1)" the document uses */
    uint32_t* status;     /* [n_docs] PTX_OK / PTX_ERR_CAPACITY */
    /* scratch in HBM, per document: change table [R][rows_per_log] and known-comment lists [R][rows_per_log] */
    void* ctab;           /* PtxGenChangeT<ptx_gen_max_r(R)> [n_docs][R][rows_per_log] */
    uint16_t* known;
};

template <uint32_t kMaxR>
struct PtxGenHdrT {
    uint32_t n[kMaxR];      /* n[0]: length of the document's list (every element made so far) */
    uint32_t vis[kMaxR];    /* visible length */
    uint32_t clock[kMaxR][kMaxR];
    uint32_t max_op[kMaxR];
    uint32_t rows[kMaxR];   /* rows written to the replica's log */
    uint32_t chgs[kMaxR];   /* changes written to its envelope */
    uint32_t nknown[kMaxR];
    uint32_t plo[kMaxR], phi[kMaxR]; /* pending changes of a delivery, per actor */
    uint32_t tmp;
    uint32_t overflow;
    uint32_t scan_tmp[36];
};

/* actor bits of a key: 2 up to four replicas, 3 up to eight */
PTX_HD uint32_t ptx_gen_actor_bits(uint64_t R) { return R <= 4 ? 2u : 3u; }
/* u16 keys while every counter of the document fits 14 (13) bits (a document makes R * ops_per_log + 1 ops) */
PTX_HD bool ptx_gen_small_keys(uint64_t R, uint64_t rows_per_log) { return R * rows_per_log + 8 < (1ull << (16 - ptx_gen_actor_bits(R))); }
PTX_HD uint64_t ptx_gen_list_stride(uint64_t list_cap) { return (list_cap + 64 + 15) & ~15ull; } /* keys per replica: whole 16-byte blocks + slack */
PTX_HD uint64_t ptx_gen_plane_words(uint64_t list_cap) { return (ptx_gen_list_stride(list_cap) >> 5) + 2; }
PTX_HD uint64_t ptx_gen_lds_need(uint64_t R, uint64_t list_cap, uint64_t rows_per_log) {
    const uint64_t kb = ptx_gen_small_keys(R, rows_per_log) ? 2 : 4;
    return ptx_a16(R <= 4 ? sizeof(PtxGenHdrT<4>) : sizeof(PtxGenHdrT<PTX_GEN_MAX_R>)) + ptx_a16(kb * ptx_gen_list_stride(list_cap)) + 2 * ptx_a16(4 * R * ptx_gen_plane_words(list_cap)) +
           ptx_a16(4 * ((rows_per_log >> 5) + 2));
}

/* 64-wide ballots (PTX_BALLOT64), PTX_LANE0, PTX_GEN_FOR: ptx_platform_gfx950.h */
PTX_DEV uint32_t ptx_ffs64(uint64_t m) { return (uint32_t)__builtin_ctzll(m); }      /* m != 0 */
PTX_DEV uint32_t ptx_fls64(uint64_t m) { return 63u - (uint32_t)__builtin_clzll(m); } /* m != 0 */
PTX_DEV uint32_t ptx_select64(uint64_t m, uint32_t k) { /* position of the k-th (0-based) set bit; popcount(m) > k */
    for (uint32_t i = 0; i < k; ++i) m &= m - 1ull;
    return ptx_ffs64(m);
}

/* mulberry32 (oracle/ptxgen.js:31-40) and randInt = floor(next() * n / 2^32) */
PTX_DEV uint32_t ptx_gen_rand(uint32_t& a, uint32_t n) {
    a += 0x6d2b79f5u;
    uint32_t t = a;
    t = (t ^ (t >> 15)) * (t | 1u);
    t ^= t + (t ^ (t >> 7)) * (t | 61u);
    const uint32_t x = t ^ (t >> 14);
    return (uint32_t)(((uint64_t)x * (uint64_t)n) >> 32);
}

/* position of the element with this key, or 0xFFFFFFFF */
PTX_DEV uint32_t ptx_gen_find(const uint32_t* lst, uint32_t n, uint32_t key) {
    for (uint32_t base = 0; base < n; base += 64u) {
        PTX_BALLOT64(m, l, base + l < n && (lst[base + l] & PTX_GK_KEY) == key)
        if (m) return base + ptx_ffs64(m);
    }
    return 0xFFFFFFFFu;
}
/* position of the visible element number `index` (getListElementId's walk, micromerge.ts:771-777), or 0xFFFFFFFF */
PTX_DEV uint32_t ptx_gen_select(const uint32_t* lst, uint32_t n, uint32_t index) {
    uint32_t seen = 0;
    for (uint32_t base = 0; base < n; base += 64u) {
        PTX_BALLOT64(m, l, base + l < n && !(lst[base + l] & PTX_GK_DEAD))
        const uint32_t c = (uint32_t)__builtin_popcountll(m);
        if (index < seen + c) return base + ptx_select64(m, index - seen);
        seen += c;
    }
    return 0xFFFFFFFFu;
}
/* lookAfterTombstones (micromerge.ts:778-793): from the visible element at `pos`, over the directly following
 * tombstones: the LAST one whose `after` slot is a defined one, else the element itself */
PTX_DEV uint32_t ptx_gen_after_tombstones(const uint32_t* lst, uint32_t n, uint32_t pos) {
    uint32_t pick = pos;
    for (uint32_t base = pos + 1u; base < n; base += 64u) {
        PTX_BALLOT64(alive, l, base + l >= n || !(lst[base + l] & PTX_GK_DEAD))
        PTX_BALLOT64(marked, l2, base + l2 < n && (lst[base + l2] & PTX_GK_DEAD) && (lst[base + l2] & PTX_GK_AFTER))
        const uint64_t below = alive ? ((1ull << ptx_ffs64(alive)) - 1ull) : ~0ull; /* the tombstones before the next visible element */
        if (marked & below) pick = base + ptx_fls64(marked & below);
        if (alive) break;
    }
    return pick;
}


/* one op row */
struct PtxGenRow {
    uint64_t op_id, ref_a, ref_b;
    uint32_t payload;
    uint8_t action, mark_type, side_a, side_b;
};
/* key = counter << kb | actor, kb = 2 bits of actor up to four replicas, 3 up to eight (ptx_gen_actor_bits): integer order == compareOpIds order */
PTX_DEV uint32_t ptx_gen_key_of(uint64_t id, uint32_t kb) { return ((uint32_t)(id >> 32) << kb) | ((uint32_t)id & ((1u << kb) - 1u)); } /* 0 for HEAD (id 0) */
PTX_DEV uint64_t ptx_gen_id_of(uint32_t key, uint32_t kb) { return ((uint64_t)(key >> kb) << 32) | (uint64_t)(key & ((1u << kb) - 1u)); }

template <uint32_t kMaxR>
PTX_DEV uint32_t ptx_gen_dep(const PtxGenChangeT<kMaxR>& c, uint32_t b) {
    if (kMaxR == 4u) return ((b < 2u ? c.deps[0] : c.deps[1]) >> (16u * (b & 1u))) & 0xFFFFu;
    uint32_t w = c.deps[0];
#pragma unroll
    for (uint32_t j = 1; j < kMaxR / 2u; ++j) w = (b >> 1) == j ? c.deps[j] : w;
    return (w >> (16u * (b & 1u))) & 0xFFFFu;
}
/* is the decimal string of j smaller than that of k (string order, j != k) */
PTX_DEV uint32_t ptx_gen_digits(uint32_t v) {
    uint32_t n = 1;
    while (v >= 10u) {
        v /= 10u;
        ++n;
    }
    return n;
}
PTX_DEV bool ptx_gen_str_less(uint32_t j, uint32_t k) {
    const uint32_t nj = ptx_gen_digits(j), nk = ptx_gen_digits(k);
    if (nj == nk) return j < k;
    if (nj < nk) {
        uint32_t kp = k;
        for (uint32_t q = nj; q < nk; ++q) kp /= 10u;
        return j != kp ? j < kp : true; /* a proper prefix sorts first */
    }
    uint32_t jp = j;
    for (uint32_t q = nk; q < nj; ++q) jp /= 10u;
    return jp != k ? jp < k : false;
}

template <uint32_t kThreads, class KeyT, uint32_t kMaxR>
struct PtxGenDoc {
    typedef PtxGenChangeT<kMaxR> PtxGenChange;
    const PtxGenArgs& A;
    PtxGenHdrT<kMaxR>* H;
    KeyT* key0;          /* the document's list */
    uint32_t* dead0;     /* replica r's planes = dead0 / after0 + r * plane_words (no pointer table: nothing of this kernel lives in scratch) */
    uint32_t* after0;
    uint32_t lst_stride, plane_words;
    uint32_t* done;   /* pending-change bitmap of a delivery */
    uint16_t* crank;  /* comment counter -> doc-local rank */
    uint64_t row0;    /* first row of replica 0's log */
    PtxGenChange* ctab;
    uint16_t* known;
    uint32_t cap;
    static constexpr uint32_t kb = kMaxR <= 4u ? 2u : 3u; /* actor bits of a key (= ptx_gen_actor_bits of the replica counts this build generates) */

    PTX_MEM uint64_t log_base(uint32_t r) const { return row0 + (uint64_t)r * A.rows_per_log; }
    PTX_MEM KeyT* keys(uint32_t) const { return key0; }
    PTX_MEM uint32_t* dead(uint32_t r) const { return dead0 + (uint64_t)r * plane_words; }
    PTX_MEM uint32_t* after(uint32_t r) const { return after0 + (uint64_t)r * plane_words; }
    /* id of the element at position p of replica r's list */
    PTX_MEM uint64_t id_at(uint32_t r, uint32_t p) const { return ptx_gen_id_of(keys(r)[p], kb); }

    /* applyOp for replica r (micromerge.ts:614-640 insert, :677-695 delete; for marks only the `after` flag).  An insert r MAKES (it carries r as its actor
     * and is applied by r first) enters the document's list; one that arrives from another replica is already there: r only starts to see it. */
    PTX_MEM void apply(uint32_t r, const PtxGenRow& o) {
        KeyT* L = key0;
        const uint32_t n = H->n[0];
        if (o.action == PTX_ACT_INSERT) {
            const uint32_t key = ptx_gen_key_of(o.op_id, kb);
            if (((uint32_t)o.op_id & (kMaxR - 1u)) != r) { /* a delivered insert: its element is in the list since its author made it (causal delivery) */
                const uint32_t p = ptx_list_find<KeyT>(L, n, key);
                if (PTX_LANE0 && p != 0xFFFFFFFFu && ((dead(r)[p >> 5] >> (p & 31u)) & 1u)) {
                    dead(r)[p >> 5] &= ~(1u << (p & 31u));
                    H->vis[r] += 1u;
                }
                PTX_WSYNC();
                return;
            }
            uint32_t at = 0;
            if (o.ref_a != 0) at = ptx_list_find<KeyT>(L, n, ptx_gen_key_of(o.ref_a, kb)) + 1u; /* the reference element exists (causal delivery) */
            /* skip the elements with a greater id (concurrent inserts at the same spot, :630-635): over the document's list, which already holds what the
             * other replicas made concurrently — the position every replica will agree on */
            for (;;) {
                PTX_BALLOT64(stop, l, at + l >= n || (uint32_t)L[at + l] < key)
                if (stop) {
                    at += ptx_ffs64(stop);
                    break;
                }
                at += 64u;
            }
            if (n + 1u > cap || key > (uint32_t)(KeyT)~(KeyT)0) { /* the list, or the key type, is too small for this document */
                if (PTX_LANE0) H->overflow = 1;
                PTX_WSYNC();
                return;
            }
            /* open the gap in the keys and in every replica's planes: the new element is visible to its author, not yet to the others; its `after` slot undefined */
            ptx_list_shift_up<KeyT>(L, at, n);
            for (uint32_t q = 0; q < A.R; ++q) {
                ptx_plane_shift_up(dead(q), at, n);
                ptx_plane_shift_up(after(q), at, n);
            }
            if (PTX_LANE0) {
                L[at] = (KeyT)key;
                for (uint32_t q = 0; q < A.R; ++q)
                    if (q != r) dead(q)[at >> 5] |= 1u << (at & 31u);
                H->n[0] = n + 1u;
                H->vis[r] += 1u;
            }
            PTX_WSYNC();
        } else if (o.action == PTX_ACT_DELETE) {
            const uint32_t p = ptx_list_find<KeyT>(L, n, ptx_gen_key_of(o.ref_a, kb));
            if (PTX_LANE0 && p != 0xFFFFFFFFu && !((dead(r)[p >> 5] >> (p & 31u)) & 1u)) {
                dead(r)[p >> 5] |= 1u << (p & 31u);
                H->vis[r] -= 1u;
            }
            PTX_WSYNC();
        } else if ((o.action == PTX_ACT_ADDMARK || o.action == PTX_ACT_REMOVEMARK) && o.side_b == PTX_SIDE_AFTER) {
            const uint32_t p = ptx_list_find<KeyT>(L, n, ptx_gen_key_of(o.ref_b, kb));
            if (PTX_LANE0 && p != 0xFFFFFFFFu) after(r)[p >> 5] |= 1u << (p & 31u);
            PTX_WSYNC();
        }
    }

    PTX_MEM void write_row(uint32_t r, const PtxGenRow& o) {
        const uint32_t k = H->rows[r];
        if (PTX_LANE0 && k < A.rows_per_log) {
            const uint64_t at = log_base(r) + k;
            A.op_id[at] = o.op_id;
            A.ref_a[at] = o.ref_a;
            A.ref_b[at] = o.ref_b;
            A.payload[at] = o.payload;
            A.action[at] = o.action;
            A.mark_type[at] = o.mark_type;
            A.side_a[at] = o.side_a;
            A.side_b[at] = o.side_b;
        }
        PTX_WSYNC();
        if (PTX_LANE0) H->rows[r] = k + 1u;
        PTX_WSYNC();
    }
    PTX_MEM PtxGenRow read_row(uint32_t r, uint32_t k) const {
        const uint64_t at = log_base(r) + k;
        PtxGenRow o;
        o.op_id = A.op_id[at];
        o.ref_a = A.ref_a[at];
        o.ref_b = A.ref_b[at];
        o.payload = A.payload[at];
        o.action = A.action[at];
        o.mark_type = A.mark_type[at];
        o.side_a = A.side_a[at];
        o.side_b = A.side_b[at];
        return o;
    }
    /* envelope entry of a change applied by replica r + what `record` keeps (ptxgen.js:98-107) */
    PTX_MEM void record(uint32_t r, uint32_t actor, uint32_t seq, const PtxGenChange& c) {
        const uint32_t k = H->chgs[r], nops = c.nops_start >> 24;
        if (PTX_LANE0 && k < A.rows_per_log) {
            const uint64_t at = log_base(r) + k;
            const uint32_t es = PTX_ENV_STRIDE(A.R);
            A.chg_hdr[at] = (actor << PTX_CHG_ACTOR_SHIFT) | nops;
            for (uint32_t b = 0; b < es; ++b) A.chg_env[at * es + b] = (uint16_t)(b == 0 ? seq : b <= A.R ? ptx_gen_dep(c, b - 1u) : 0u);
            H->chgs[r] = k + 1u;
        }
        PTX_WSYNC();
    }
    PTX_MEM void note_comment(uint32_t r, const PtxGenRow& o) {
        if (o.action == PTX_ACT_ADDMARK && o.mark_type == PTX_MARK_COMMENT) {
            if (PTX_LANE0) {
                const uint32_t k = H->nknown[r];
                if (k < A.rows_per_log) known[(uint64_t)r * A.rows_per_log + k] = (uint16_t)o.payload;
                H->nknown[r] = k + 1u;
            }
            PTX_WSYNC();
        }
    }

    /* applyChange of change (actor, s) on replica dst (micromerge.ts:499-514): false = the causal RangeError */
    PTX_MEM bool deliver_one(uint32_t dst, uint32_t actor, uint32_t s) {
        const PtxGenChange c = ctab[(uint64_t)actor * A.rows_per_log + s];
        if (s + 1u != H->clock[dst][actor] + 1u) return false;
        for (uint32_t b = 0; b < A.R; ++b)
            if (H->clock[dst][b] < ptx_gen_dep(c, b)) return false;
        const uint32_t nops = c.nops_start >> 24, start = c.nops_start & 0xFFFFFFu;
        for (uint32_t k = 0; k < nops; ++k) {
            const PtxGenRow o = read_row(actor, c.rowoff + k);
            apply(dst, o);
            write_row(dst, o);
            note_comment(dst, o);
        }
        if (PTX_LANE0) {
            H->clock[dst][actor] = s + 1u;
            const uint32_t last = start + nops - 1u;
            if (last > H->max_op[dst]) H->max_op[dst] = last;
        }
        PTX_WSYNC();
        record(dst, actor, s + 1u, c);
        return true;
    }
    /* deliver to dst everything src has applied and dst has not (ptxgen.js:122-147): the pending queue with its
     * push-back-on-failure discipline = repeated passes, in order, over the not-yet-applied entries */
    PTX_MEM void deliver(uint32_t src, uint32_t dst) {
        uint32_t total = 0;
        for (uint32_t a = 0; a < A.R; ++a) {
            const uint32_t l = H->clock[dst][a], h = H->clock[src][a] > l ? H->clock[src][a] : l;
            total += h - l;
        }
        if (total == 0) return;
        PTX_SYNC();
        if (PTX_LANE0)
            for (uint32_t a = 0; a < A.R; ++a) { /* the pending list is fixed when the delivery starts */
                const uint32_t l = H->clock[dst][a];
                H->plo[a] = l;
                H->phi[a] = H->clock[src][a] > l ? H->clock[src][a] : l;
            }
        PTX_GEN_FOR(w, (total >> 5) + 1u) done[w] = 0;
        PTX_WSYNC();
        uint32_t remaining = total;
        for (uint32_t pass = 0; remaining && pass <= total; ++pass) {
            uint32_t idx = 0;
            for (uint32_t a = 0; a < A.R; ++a)
                for (uint32_t s = H->plo[a]; s < H->phi[a]; ++s, ++idx) {
                    if ((done[idx >> 5] >> (idx & 31)) & 1u) continue;
                    if (deliver_one(dst, a, s)) {
                        if (PTX_LANE0) done[idx >> 5] |= 1u << (idx & 31);
                        PTX_WSYNC();
                        --remaining;
                    }
                }
        }
    }

    /* one emitted op of change(): id = maxOp + 1, applied locally at once (micromerge.ts:483-493) */
    PTX_MEM void emit(uint32_t k, PtxGenRow o, uint32_t& nops) {
        const uint32_t ctr = H->max_op[k] + 1u;
        PTX_WSYNC();
        if (PTX_LANE0) H->max_op[k] = ctr;
        PTX_WSYNC();
        o.op_id = ((uint64_t)ctr << 32) | k;
        apply(k, o);
        write_row(k, o);
        note_comment(k, o);
        ++nops;
    }
};

template <uint32_t kThreads, class KeyT, uint32_t kMaxR>
PTX_DEV void ptx_gen_doc_keyed(const PtxGenArgs& A, uint32_t doc_local, uint8_t* lds) {
    typedef PtxGenChangeT<kMaxR> PtxGenChange;
    typedef PtxGenHdrT<kMaxR> PtxGenHdr;
    PtxGenHdr* H = (PtxGenHdr*)lds;
    const uint32_t R = A.R, N = A.rows_per_log;
    PtxBump bp;
    bp.base = lds;
    bp.off = (uint32_t)ptx_a16(sizeof(PtxGenHdr));
    bp.cap = A.lds_bytes;
    bp.high = bp.off;
    bp.overflow = false;
    PtxGenDoc<kThreads, KeyT, kMaxR> G{A, H, nullptr, nullptr, nullptr, 0, 0, nullptr, nullptr, 0, nullptr, nullptr, A.list_cap};
    G.lst_stride = (uint32_t)ptx_gen_list_stride(A.list_cap);
    G.plane_words = (uint32_t)ptx_gen_plane_words(A.list_cap);
    G.key0 = ptx_alloc<KeyT>(bp, G.lst_stride);
    G.dead0 = ptx_alloc<uint32_t>(bp, G.plane_words * R);
    G.after0 = ptx_alloc<uint32_t>(bp, G.plane_words * R);
    G.done = ptx_alloc<uint32_t>(bp, (N >> 5) + 2);
    G.crank = (uint16_t*)G.key0; /* needed once the documents are finished and the lists (keys + planes, contiguous) dead: see the end */
    G.row0 = (uint64_t)doc_local * R * N;
    G.ctab = (PtxGenChange*)A.ctab + (uint64_t)doc_local * R * N;
    G.known = A.known + (uint64_t)doc_local * R * N;
    if (bp.overflow || R == 0 || R > kMaxR || N > 0xFFFFFFu) {
        if (PTX_LANE0) {
            A.status[doc_local] = PTX_ERR_CAPACITY;
            A.n_comments[doc_local] = 0;
            for (uint32_t r = 0; r < R && r < kMaxR; ++r) A.n_changes[doc_local * R + r] = 0;
        }
        return;
    }
    if (PTX_LANE0) {
        for (uint32_t r = 0; r < kMaxR; ++r) {
            H->n[r] = H->vis[r] = H->max_op[r] = H->rows[r] = H->chgs[r] = H->nknown[r] = 0;
            for (uint32_t a = 0; a < kMaxR; ++a) H->clock[r][a] = 0;
        }
        H->overflow = 0;
        H->tmp = 0;
    }
    PTX_GEN_FOR(w, G.plane_words * R) {
        G.dead0[w] = 0;
        G.after0[w] = 0;
    }
    PTX_WSYNC();

    /* docSeed (ptxgen.js:62-64) */
    uint32_t rng = (0x5eed0000u + (A.first_doc + doc_local)) ^ (A.seed * 0x9e3779b1u);
    uint32_t comment_counter = 0;

    /* one change() of replica k: `body` emits the ops */
#define PTX_GEN_CHANGE_BEGIN(k_)                                                                 \
    const uint32_t ck_ = (k_);                                                                   \
    uint32_t nops_ = 0;                                                                          \
    PtxGenChange c_;                                                                             \
    c_.deps[0] = H->clock[ck_][0] | (H->clock[ck_][1] << 16);                                    \
    c_.deps[1] = H->clock[ck_][2] | (H->clock[ck_][3] << 16);                                    \
    _Pragma("unroll") for (uint32_t j_ = 2; j_ < kMaxR / 2u; ++j_)                               \
        c_.deps[j_] = H->clock[ck_][2u * j_] | (H->clock[ck_][2u * j_ + 1u] << 16);              \
    c_.rowoff = H->rows[ck_];                                                                    \
    const uint32_t seq_ = H->clock[ck_][ck_] + 1u, start_ = H->max_op[ck_] + 1u;                 \
    PTX_WSYNC();                                                                                 \
    if (PTX_LANE0) H->clock[ck_][ck_] = seq_;                                                    \
    PTX_WSYNC();
#define PTX_GEN_CHANGE_END()                                                                     \
    c_.nops_start = (nops_ << 24) | start_;                                                      \
    if (PTX_LANE0) G.ctab[(uint64_t)ck_ * N + (seq_ - 1u)] = c_;                                 \
    PTX_WSYNC();                                                                                 \
    G.record(ck_, ck_, seq_, c_);

    /* generateDocs (ptxgen.js:109-121): doc1 creates the list + the initial text, everybody applies it */
    {
        PTX_GEN_CHANGE_BEGIN(0u)
        PtxGenRow o;
        o.op_id = 0;
        o.ref_a = o.ref_b = 0;
        o.payload = 0;
        o.action = PTX_ACT_MAKELIST;
        o.mark_type = o.side_a = o.side_b = 0;
        G.emit(0u, o, nops_);
        uint64_t ref = 0;
        for (uint32_t i = 0; i < A.init_len; ++i) {
            o.action = PTX_ACT_INSERT;
            o.ref_a = ref;
            o.payload = A.init_text[i];
            G.emit(0u, o, nops_);
            ref = ((uint64_t)H->max_op[0] << 32) | 0u;
        }
        PTX_GEN_CHANGE_END()
    }
    PTX_SYNC(); /* the first change is read back from HBM */
    for (uint32_t i = 1; i < R; ++i) G.deliver_one(i, 0u, 0u);
    uint32_t ops_so_far = A.init_len;

    while (ops_so_far < A.ops_per_log && !H->overflow) {
        const uint32_t k = ptx_gen_rand(rng, R);
        const uint32_t len = H->vis[k];
        const uint32_t budget = A.ops_per_log - ops_so_far;
        const uint32_t x = ptx_gen_rand(rng, 100u);
        uint32_t kind = x < A.mix0 ? 0u : x < A.mix01 ? 1u : x < A.mix012 ? 2u : 3u;
        if (kind == 1u && len < 2u) kind = 0u;
        if (kind >= 2u && (len < 1u || A.n_mark_types == 0u)) kind = 0u;
        PTX_GEN_CHANGE_BEGIN(k)
        PtxGenRow o;
        o.op_id = 0;
        o.ref_a = o.ref_b = 0;
        o.payload = 0;
        o.mark_type = o.side_a = o.side_b = 0;
        if (kind == 0u) {
            /* insert (micromerge.ts:335-345): after the element at index-1, past its tombstones */
            const uint32_t index = ptx_gen_rand(rng, len + 1u);
            uint32_t nvals = 1u + ptx_gen_rand(rng, 2u);
            if (nvals > budget) nvals = budget;
            uint64_t ref = 0;
            if (index != 0u) {
                const uint32_t p = ptx_plane_select0(G.dead(k), H->n[0], index - 1u);
                ref = G.id_at(k, ptx_plane_after_tombstones(G.dead(k), G.after(k), H->n[0], p));
            }
            for (uint32_t v = 0; v < nvals; ++v) {
                const uint32_t h = ptx_gen_rand(rng, 16u);
                o.action = PTX_ACT_INSERT;
                o.ref_a = ref;
                o.payload = h < 10u ? 48u + h : 87u + h; /* "0123456789abcdef" */
                G.emit(k, o, nops_);
                ref = ((uint64_t)H->max_op[k] << 32) | k;
            }
        } else if (kind == 1u) {
            /* delete (micromerge.ts:346-352): always the same visible index */
            const uint32_t index = 1u + ptx_gen_rand(rng, len - 1u);
            const uint32_t room = len - index;
            uint32_t count = 1u + ptx_gen_rand(rng, room < 3u ? room : 3u);
            if (count > budget) count = budget;
            for (uint32_t q = 0; q < count; ++q) {
                const uint32_t p = ptx_plane_select0(G.dead(k), H->n[0], index);
                o.action = PTX_ACT_DELETE;
                o.ref_a = G.id_at(k, p);
                G.emit(k, o, nops_);
            }
        } else {
            /* changeMark (peritext.ts:458-501) */
            const uint32_t start_index = ptx_gen_rand(rng, len);
            const uint32_t end_index = start_index + 1u + ptx_gen_rand(rng, len - start_index);
            const uint32_t mt = A.mark_types[ptx_gen_rand(rng, A.n_mark_types)];
            bool add = kind == 2u;
            uint32_t pay = 0;
            if (mt == PTX_MARK_LINK) {
                const uint32_t u = ptx_gen_rand(rng, 26u); /* drawn for removeMark too */
                if (add) pay = u;
            } else if (mt == PTX_MARK_COMMENT) {
                if (!add && H->nknown[k] == 0u) add = true;
                if (add) pay = comment_counter++;
                else {
                    PTX_SYNC(); /* `known` is written by this wave (note_comment) */
                    pay = G.known[(uint64_t)k * N + ptx_gen_rand(rng, H->nknown[k])];
                }
            }
            const bool inclusive = mt == PTX_MARK_STRONG || mt == PTX_MARK_EM; /* schema.ts:45-96 */
            o.action = add ? PTX_ACT_ADDMARK : PTX_ACT_REMOVEMARK;
            o.mark_type = (uint8_t)mt;
            o.payload = pay;
            o.side_a = PTX_SIDE_BEFORE;
            o.ref_a = G.id_at(k, ptx_plane_select0(G.dead(k), H->n[0], start_index));
            if (inclusive && end_index >= len) {
                o.side_b = PTX_SIDE_END_OF_TEXT;
                o.ref_b = 0;
            } else if (inclusive) {
                o.side_b = PTX_SIDE_BEFORE;
                o.ref_b = G.id_at(k, ptx_plane_select0(G.dead(k), H->n[0], end_index));
            } else {
                o.side_b = PTX_SIDE_AFTER;
                o.ref_b = G.id_at(k, ptx_plane_select0(G.dead(k), H->n[0], end_index - 1u));
            }
            G.emit(k, o, nops_);
        }
        PTX_GEN_CHANGE_END()
        ops_so_far += nops_;
        if (R > 1u) {
            const uint32_t left = ptx_gen_rand(rng, R);
            uint32_t right = ptx_gen_rand(rng, R - 1u);
            if (right >= left) ++right;
            G.deliver(left, right);
            G.deliver(right, left);
        }
    }
    /* final full sync (ptxgen.js:203-206) */
    for (uint32_t round = 0; round < R + 1u; ++round)
        for (uint32_t i = 0; i < R; ++i)
            for (uint32_t j = 0; j < R; ++j)
                if (i != j) G.deliver(i, j);
#undef PTX_GEN_CHANGE_BEGIN
#undef PTX_GEN_CHANGE_END

    /* comment ids "comment-<k>": the wire format wants their rank in string order inside the document
     * (peritext.ts:318 keeps comment arrays id-sorted): decimal strings compared digit by digit */
    const uint32_t C = comment_counter;
    const uint64_t crank_room = (uint64_t)((uint8_t*)(G.after0 + (uint64_t)G.plane_words * R) - (uint8_t*)G.key0) / 2u; /* u16 entries the dead lists hold */
    if (C > crank_room) {
        if (PTX_LANE0) H->overflow = 1;
        PTX_WSYNC();
    }
    PTX_GEN_FOR(kk, C <= crank_room ? C : 0u) {
        uint32_t rank = 0;
        for (uint32_t j = 0; j < C; ++j) rank += j != kk && ptx_gen_str_less(j, kk) ? 1u : 0u;
        G.crank[kk] = (uint16_t)rank;
    }
    PTX_SYNC();
    for (uint32_t r = 0; r < R; ++r) {
        const uint64_t b0 = G.log_base(r);
        const uint32_t nr = H->rows[r] < N ? H->rows[r] : N;
        PTX_GEN_FOR(i, nr) {
            if (C <= crank_room && A.mark_type[b0 + i] == PTX_MARK_COMMENT && (A.action[b0 + i] == PTX_ACT_ADDMARK || A.action[b0 + i] == PTX_ACT_REMOVEMARK))
                A.payload[b0 + i] = G.crank[A.payload[b0 + i]];
        }
    }
    PTX_WSYNC();
    if (PTX_LANE0) {
        bool bad = H->overflow != 0;
        for (uint32_t r = 0; r < R; ++r) {
            A.n_changes[doc_local * R + r] = H->chgs[r];
            if (H->rows[r] != N || H->chgs[r] > N) bad = true;
        }
        A.n_comments[doc_local] = C;
        A.status[doc_local] = bad ? (uint32_t)PTX_ERR_CAPACITY : (uint32_t)PTX_OK;
    }
}

/* one document: the key width follows from the size of the document */
template <uint32_t kThreads, uint32_t kMaxR>
PTX_DEV void ptx_gen_doc(const PtxGenArgs& A, uint32_t doc_local, uint8_t* lds) {
    if (ptx_gen_small_keys(A.R, A.rows_per_log)) ptx_gen_doc_keyed<kThreads, uint16_t, kMaxR>(A, doc_local, lds);
    else ptx_gen_doc_keyed<kThreads, uint32_t, kMaxR>(A, doc_local, lds);
}
