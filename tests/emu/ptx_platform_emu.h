/*
 * ptx_platform_emu.h — TEST INFRASTRUCTURE.  The interface of peritext_amd/csrc/ptx_platform_gfx950.h, played by ONE host
 * thread: a "parallel" loop runs its iterations one after the other, in the order ptx_emu_reverse selects (forward, backward,
 * a fixed pseudo-random permutation), so that tests/emu can check the kernels' logic — and its independence of the iteration
 * order — in a container without a GPU.  Never linked into libperitext_hip.so; never a fallback of the product path.
 */
#pragma once
#include <string.h>
#define PTX_HD static inline
#define PTX_DEV static inline
#define PTX_SYNC() ((void)0)
#define PTX_SYNC_LDS() ((void)0)
#define PTX_SYNC_FULL() ((void)0)
/* sanitizer build of the emulation (g++ -fsanitize=address): the padding behind every array of the LDS bump allocator is poisoned,
 * so that an off-by-one of the kernel logic is reported instead of landing silently in the next array's slack */
#if defined(__SANITIZE_ADDRESS__)
#include <sanitizer/asan_interface.h>
#define PTX_LDS_ALLOCATED(p, used_bytes, total_bytes)                                                       \
    do {                                                                                                    \
        ASAN_UNPOISON_MEMORY_REGION((p), (used_bytes));                                                     \
        ASAN_POISON_MEMORY_REGION((const uint8_t*)(p) + (used_bytes), (total_bytes) - (used_bytes));        \
    } while (0)
#else
#define PTX_LDS_ALLOCATED(p, used_bytes, total_bytes) ((void)0)
#endif
/* P1's list stores may land anywhere inside the log's window when a header understates the rows (the log is rejected afterwards):
 * the one store the sanitizer build must not check against the padding marks (the clamp to the window is checked by the tests) */
#if defined(__SANITIZE_ADDRESS__)
__attribute__((no_sanitize_address, noinline)) static void ptx_emu_wild_store16(uint16_t* p, uint16_t v) { *p = v; }
#define PTX_LDS_WILD_STORE16(p, v) ptx_emu_wild_store16((p), (uint16_t)(v))
#else
#define PTX_LDS_WILD_STORE16(p, v) (*(p) = (uint16_t)(v))
#endif
#define PTX_SYNC_T() ((void)0)
extern int ptx_emu_reverse; /* order of every emulated parallel loop: 0 forward, 1 backward, 2 a fixed pseudo-random permutation (order-independence checks) */
static inline uint32_t ptx_emu_ix(uint32_t k, uint32_t n) { /* k-th iteration runs index ...; 104729 is a prime above any loop length here */
    return ptx_emu_reverse == 0 ? k : ptx_emu_reverse == 1 ? n - 1u - k : (uint32_t)(((uint64_t)k * 104729ull + 7ull) % (n ? n : 1u));
}
#define PTX_FOR(i, n)                                                                              \
    for (uint32_t _n = (n), _k = 0, i = (_n ? ptx_emu_ix(0, _n) : 0); _k < _n;                    \
         ++_k, i = (_k < _n ? ptx_emu_ix(_k, _n) : 0))
#define PTX_LEADER if (true)
#define PTX_ONE_WAVE if (true)
#define PTX_LANE_ID 0u
#define PTX_FOR_LANES(i, n) PTX_FOR(i, n)
PTX_DEV uint32_t ptx_atomic_or(uint32_t* p, uint32_t v) { uint32_t o = *p; *p = o | v; return o; }
PTX_DEV uint32_t ptx_atomic_and(uint32_t* p, uint32_t v) { uint32_t o = *p; *p = o & v; return o; }
PTX_DEV uint32_t ptx_atomic_add(uint32_t* p, uint32_t v) { uint32_t o = *p; *p = o + v; return o; }
PTX_DEV uint32_t ptx_atomic_max(uint32_t* p, uint32_t v) { uint32_t o = *p; if (v > o) *p = v; return o; }
PTX_DEV uint32_t ptx_atomic_min(uint32_t* p, uint32_t v) { uint32_t o = *p; if (v < o) *p = v; return o; }
PTX_DEV unsigned long long ptx_atomic_add64(unsigned long long* p, unsigned long long v) { unsigned long long o = *p; *p = o + v; return o; }
PTX_DEV void ptx_atomic_or64(unsigned long long* p, unsigned long long v) { *p |= v; }
PTX_DEV void ptx_atomic_max64(unsigned long long* p, unsigned long long v) { if (v > *p) *p = v; }
PTX_DEV uint32_t ptx_popc(uint32_t x) { return (uint32_t)__builtin_popcount(x); }
PTX_DEV uint16_t ptx_coherent_load16(const uint16_t* p) { return *p; }
PTX_DEV void ptx_coherent_store16(uint16_t* p, uint16_t v) { *p = v; }
PTX_DEV uint32_t ptx_coherent_load32(const uint32_t* p) { return *p; }
PTX_DEV void ptx_coherent_store32(uint32_t* p, uint32_t v) { *p = v; }
PTX_DEV uint64_t ptx_coherent_load64(const uint64_t* p) { return *p; }
PTX_DEV void ptx_coherent_store64(uint64_t* p, uint64_t v) { *p = v; }
#define PTX_U32(x) ((uint32_t)(x))
PTX_DEV uint32_t ptx_brev(uint32_t x) {
    uint32_t r = 0;
    for (int i = 0; i < 32; ++i) r |= ((x >> i) & 1u) << (31 - i);
    return r;
}
PTX_DEV void ptx_global_stores_done() {}
/* append to a list: index of this element (valid only where pred) */
PTX_DEV uint32_t ptx_append(uint32_t* cursor, bool pred) { return pred ? (*cursor)++ : 0u; }
PTX_DEV uint32_t ptx_append_n(uint32_t* cursor, uint32_t count) { const uint32_t b = *cursor; *cursor += count; return b; }
PTX_DEV uint64_t ptx_clock() { return 0; }
#define PTX_G 1u
PTX_DEV uint32_t ptx_group_sum(uint32_t c) { return c; }
/* batched parallel loop: PTX_U iterations per step so that their loads are all in flight together */
#define PTX_FORU(i0, n) for (uint32_t i0 = 0, _n = (n), _T = 1; i0 < _n; i0 += PTX_U)
#define PTX_FORV(i0, n, U) for (uint32_t i0 = 0, _n = (n), _T = 1; i0 < _n; i0 += (U))
#define PTX_IX(i0, u) ((i0) + (uint32_t)(u) < _n ? ptx_emu_ix((i0) + (uint32_t)(u), _n) : (i0) + (uint32_t)(u))

/* ---- list slots for a batch of rows: rows of class c < 6 get consecutive slots from cursor[c], in ROW order
 *      within the wave (lane-major, each lane holding PTX_U consecutive rows), so that the lists stay (nearly)
 *      sorted by row and later gathers through them stay (nearly) coalesced.  One LDS atomic per wave and batch
 *      (6 lanes, 6 distinct cursors); the ranking itself is a DPP prefix sum in registers.  Every lane of the
 *      wave must call it (uniform control flow). ---- */
template <int U>
PTX_DEV void ptx_wave_slots(uint32_t* cursor, const uint32_t* cls, uint32_t* slot) {
    for (int u = 0; u < U; ++u) slot[u] = cls[u] < 6u ? cursor[cls[u]]++ : 0xFFFFFFFFu;
}

/* ---- packed 16-bit arithmetic and byte permutes (plain C restatements of v_pk_max_u16, v_pk_sub_u16 clamp, v_perm_b32) ---- */
PTX_DEV uint32_t ptx_pk_max_u16(uint32_t a, uint32_t b) {
    const uint32_t l = (a & 0xFFFFu) > (b & 0xFFFFu) ? a & 0xFFFFu : b & 0xFFFFu, h = (a >> 16) > (b >> 16) ? a >> 16 : b >> 16;
    return l | (h << 16);
}
PTX_DEV uint32_t ptx_pk_subsat_u16(uint32_t a, uint32_t b) {
    const uint32_t l = (a & 0xFFFFu) > (b & 0xFFFFu) ? (a & 0xFFFFu) - (b & 0xFFFFu) : 0u, h = (a >> 16) > (b >> 16) ? (a >> 16) - (b >> 16) : 0u;
    return l | (h << 16);
}
PTX_DEV uint32_t ptx_perm(uint32_t hi, uint32_t lo, uint32_t sel) {
    const uint64_t pool = ((uint64_t)hi << 32) | lo;
    uint32_t r = 0;
    for (int k = 0; k < 4; ++k) {
        const uint32_t q = (sel >> (8 * k)) & 255u;
        const uint32_t byte = q < 8u ? (uint32_t)(pool >> (8u * q)) & 255u : (q <= 12u ? 0u : 255u); /* 8..11 (sign fills) are never selected by the kernels */
        r |= byte << (8 * k);
    }
    return r;
}
PTX_DEV uint64_t ptx_shl64(uint64_t x, uint32_t s) { return x << (s & 63u); }
PTX_DEV uint64_t ptx_shr64(uint64_t x, uint32_t s) { return x >> (s & 63u); }
PTX_DEV bool ptx_wave_pick(bool pred, uint32_t value, uint32_t& out) {
    if (pred) out = value;
    return pred;
}
PTX_DEV uint32_t ptx_wave_pk_max_u16(uint32_t v) { return v; }

PTX_DEV uint32_t ptx_mul24(uint32_t a, uint32_t b) { return (a & 0xFFFFFFu) * (b & 0xFFFFFFu); }
PTX_DEV uint32_t ptx_mad24(uint32_t a, uint32_t b, uint32_t c) { return (a & 0xFFFFFFu) * (b & 0xFFFFFFu) + c; }
PTX_DEV uint32_t ptx_mad24_su(uint32_t a, uint32_t b, uint32_t c) { return ptx_mad24(a, b, c); }
PTX_DEV uint32_t ptx_min(uint32_t a, uint32_t b) { return a < b ? a : b; }
/* the same with the classes of the rows in the bytes of c4 and absolute cursors; rows of class 6 / 7 go to the dump */
template <int U>
PTX_DEV void ptx_wave_slots4(uint32_t* cursor, uint32_t dump, uint32_t c4, uint32_t* slot) {
    for (int u = 0; u < U; ++u) {
        const uint32_t c = (c4 >> (8 * u)) & 255u;
        slot[u] = c < 6u ? cursor[c]++ : dump + (uint32_t)(u & 3);
    }
}

/* Software-pipelined uniform loops: step st of `steps` handles group PTX_G_OF(st); the loads of step st+1 are
 * issued before step st is processed (the caller keeps two register sets).  Every thread runs every step; a
 * group index past the end means "no work" (its loads are clamped to valid addresses, its effects masked). */
#define PTX_STEPS(groups) (groups)
#define PTX_WHOLE_STEPS(groups) true /* (one group per step here) */
#define PTX_G_OF(st, steps) ((st) < (steps) ? ptx_emu_ix((st), (steps)) : (steps) + ((st) - (steps)))

/* the same for loops over list items, PTX_U items per thread and step, lanes on consecutive items:
 * step st, slot u handles item PTX_J_OF(st, u) (past the end = no work); PTX_JX maps it for the emulation's
 * reversed order */
#define PTX_JSTEPS_U(n, U) (((n) + (U)-1u) / (U))
#define PTX_J_OF_U(st, u, U) ((st) * (U) + (uint32_t)(u))
#define PTX_JX(j, n) ((j) < (n) ? ptx_emu_ix((j), (n)) : (j))
/* loops over BLOCKS of items: the emulation plays blocks of 8 lanes, one lane per step, in the selected order */
#define PTX_JB_CAP 8u
#define PTX_KEEP_VGPR(x) ((void)(x))

/* (the gfx950 header has DPP forms of the wave reductions beside the butterflies; here they are the same functions) */
#define ptx_wave_pk_max_u16_dpp ptx_wave_pk_max_u16
#define ptx_wave_max_dpp ptx_wave_max
#define ptx_reduce_add32_dpp ptx_reduce_add32
#define ptx_reduce_add64_dpp ptx_reduce_add64
#define PTX_CONST_LOAD(p) (*(p))
#define PTX_FRESH_ARGS(A) (A) /* (the GPU reads its kernel arguments again, scalar loads; here they are where they were) */
#define PTX_JB_STEPS(B, U) (((B) * PTX_JB_CAP + (U)-1u) / (U))
#define PTX_JB_BLOCK(st, u, U) (((st) * (U) + (uint32_t)(u)) / PTX_JB_CAP)
#define PTX_JB_LANE(st, u, U) (ptx_emu_ix(((st) * (U) + (uint32_t)(u)) % PTX_JB_CAP, PTX_JB_CAP))

/* wave-explicit loops: every wave runs the body once with its wave index `w` and lane index `lane`; the
 * emulation plays three one-lane waves in turn */
#define PTX_WAVE_FIRST(g) (g)
#define PTX_WS 1u
#define PTX_NWAVES 3u
#define PTX_FOR_WAVE(w, lane) for (uint32_t w = 0, lane = 0; w < PTX_NWAVES; ++w)
PTX_DEV uint32_t ptx_wave_incl_scan(uint32_t v) { return v; }
PTX_DEV uint32_t ptx_wave_last(uint32_t incl) { return incl; }
PTX_DEV uint32_t ptx_wave_total(uint32_t v) { return v; }
PTX_DEV uint32_t ptx_wave_min(uint32_t v) { return v; }
PTX_DEV uint32_t ptx_wave_max(uint32_t v) { return v; }

#define PTX_NTHREADS 5u /* the emulation splits per-thread runs five ways so that the run/prefix logic is exercised */
/* biglog_core.h's team vocabulary: one host thread plays the whole team (a workgroup, or the workgroups of a cooperative launch) */
#define PTX_BFOR(i, n) PTX_FOR(i, n)
#define PTX_BSYNC() ((void)0)
#define PTX_BLEADER if (true)
#define PTX_BNT PTX_NTHREADS
#define PTX_BFIRST_WAVE if (true)
#define PTX_WFOR(i, n) PTX_FOR(i, n)
#define PTX_WG_SYNC() ((void)0)
#define PTX_BWG_ID 0u
#define PTX_BWG_COUNT 1u
#define PTX_BLANE 0u
static uint32_t ptx_emu_keep[2][PTX_NTHREADS];
#define PTX_BIG_KEEP(t, s, incl) ptx_emu_keep[0][t] = (s), ptx_emu_keep[1][t] = (incl);
#define PTX_BIG_RECALL(t, s, incl) (s) = ptx_emu_keep[0][t], (incl) = ptx_emu_keep[1][t];

#define PTX_FORA(i0, n) for (uint32_t i0 = 0, _n = (n), _T = 1; i0 < _n; i0 += PTX_UA)

/* uniform loop over groups of PTX_U consecutive items: every thread runs every step (g may be past the end) */
#define PTX_FORG(g, groups) \
    for (uint32_t _ng = (groups), _k = 0, g = (_ng ? ptx_emu_ix(0, _ng) : 0); _k < _ng; ++_k, g = (_k < _ng ? ptx_emu_ix(_k, _ng) : 0))

/* sum / max over the workgroup into LDS words (every thread calls them) */
PTX_DEV void ptx_reduce_add64(unsigned long long* dst, unsigned long long v) {
    *dst += v;
}

PTX_DEV void ptx_reduce_add32(uint32_t* dst, uint32_t v) { /* every lane of the wave calls it: one LDS atomic per wave */
    *dst += v;
}

PTX_DEV void ptx_reduce_max32(uint32_t* dst, uint32_t v) {
    if (v > *dst) *dst = v;
}

/* ---- block-wide exclusive scan of an LDS array (element k at a[k*STRIDE]), in place; returns the
 *      total (all threads call it; ends with a barrier) ---- */
template <class T, int STRIDE, uint32_t kThreads>
PTX_DEV uint32_t ptx_scan_excl(T* a, uint32_t m, uint32_t* tmp /* >= 36 u32 in LDS */, uint32_t div_magic = 0 /* as PTX_DIV_T; needed when kThreads == 0 */) {
    uint32_t run = 0;
    for (uint32_t j = 0; j < m; ++j) {
        uint32_t v = a[j * STRIDE];
        a[j * STRIDE] = (T)run;
        run += v;
    }
    (void)tmp;
    (void)div_magic;
    return run;
}

PTX_DEV void ptx_flush_clocks(unsigned long long*, unsigned long long*, int) {}

/* logs whose one-pass admission check failed and were walked again by the exact code (the tests assert that valid logs never are) */
extern unsigned long long ptx_emu_exact_walks;
#define PTX_NOTE_EXACT_WALK() (++ptx_emu_exact_walks)

/* phase stamps of the diagnostic build: nothing to stamp here */
#define PTX_STAMP(k) ((void)0)

/* PTX_AC consecutive headers / envelope rows of a lane; indices past `hi` are clamped, their effects masked.
 * The library pads its copies of both columns, so the 16-byte loads may run past the last change. */
#define PTX_ADM_HDRS(dst_, cl_) \
    for (uint32_t u_ = 0; u_ < PTX_AC; ++u_) dst_[u_] = c_hdr[(cl_) + u_ < C ? (cl_) + u_ : C - 1u];
#define PTX_ADM_ENVS32(e0_, e1_, cl_)                                                                         \
    for (uint32_t u_ = 0; u_ < PTX_AC; ++u_) {                                                                \
        const uint16_t* row_ = c_env + (uint64_t)((cl_) + u_ < C ? (cl_) + u_ : C - 1u) * 4u;                 \
        e0_[u_] = (uint32_t)row_[0] | ((uint32_t)row_[1] << 16);                                              \
        e1_[u_] = (uint32_t)row_[2] | ((uint32_t)row_[3] << 16);                                              \
    }
#define PTX_ADM_HDRSN(dst_, cl_, AC_) \
    for (uint32_t u_ = 0; u_ < (AC_); ++u_) dst_[u_] = c_hdr[(cl_) + u_ < C ? (cl_) + u_ : C - 1u];
#define PTX_ADM_ROWSN(e_, cl_, W_, AC_)                                                                          \
    for (uint32_t u_ = 0; u_ < (AC_); ++u_) {                                                                \
        const uint16_t* row_ = c_env + (uint64_t)((cl_) + u_ < C ? (cl_) + u_ : C - 1u) * (2u * (W_));        \
        for (uint32_t j_ = 0; j_ < (W_); ++j_) e_[u_][j_] = (uint32_t)row_[2u * j_] | ((uint32_t)row_[2u * j_ + 1u] << 16); \
    }
#define PTX_ADM_ENVS(dst_, cl_)                                                              \
    for (uint32_t u_ = 0; u_ < PTX_AC; ++u_)                                                 \
        for (uint32_t b_ = 0; b_ < 4u; ++b_) dst_[u_][b_] = c_env[(uint64_t)((cl_) + u_ < C ? (cl_) + u_ : C - 1u) * 4u + b_];

/* the action / mark_type bytes of a thread's PTX_U1 consecutive rows from r0_ on, one byte each in dst_ (uses N) */
#define PTX_P1_IDS(dst_, ptr_) for (uint32_t u_ = 0; u_ < PTX_U1; ++u_) dst_[u_] = (ptr_)[u_];
#define PTX_P1_BYTES(col_, r0_, dst_, n_)                                \
    dst_ = 0;                                                            \
    for (uint32_t u_ = 0; u_ < PTX_U1; ++u_) dst_ |= (uint32_t)col_[(r0_) + u_ < (n_) ? (r0_) + u_ : (n_) - 1u] << (8u * u_);

/* ---- gen_core.h / change_core.h: ONE wave per workgroup; the ballots are built lane by lane ---- */
#define PTX_BALLOT64(mask_, lane_, expr)                 \
    uint64_t mask_ = 0;                                  \
    for (uint32_t lane_ = 0; lane_ < 64u; ++lane_)       \
        if (expr) mask_ |= 1ull << lane_;
#define PTX_LANE0 true
#define PTX_GEN_FOR(i, n) for (uint32_t i = 0, _gn = (n); i < _gn; ++i)
#define PTX_MEM inline
#define PTX_WSYNC() ((void)0)
PTX_DEV void ptx_shift_up64(uint32_t* L, uint32_t lo, uint32_t hi) {
    uint32_t chunk[64];
    for (uint32_t i = lo; i < hi; ++i) chunk[i - lo] = L[i];
    for (uint32_t i = lo; i < hi; ++i) L[i + 1u] = chunk[i - lo];
}

/* ---- element lists of the one-wave kernels (gen_core.h): plain loops ---- */
template <class KeyT>
PTX_DEV uint32_t ptx_list_find(const KeyT* keys, uint32_t n, uint32_t key) {
    for (uint32_t i = 0; i < n; ++i)
        if (keys[i] == key) return i;
    return 0xFFFFFFFFu;
}
template <class KeyT>
PTX_DEV void ptx_list_shift_up(KeyT* keys, uint32_t at, uint32_t n) {
    for (uint32_t i = n; i > at; --i) keys[i] = keys[i - 1u];
}
PTX_DEV bool ptx_emu_bit(const uint32_t* plane, uint32_t i) { return (plane[i >> 5] >> (i & 31u)) & 1u; }
PTX_DEV void ptx_emu_setbit(uint32_t* plane, uint32_t i, bool v) {
    if (v) plane[i >> 5] |= 1u << (i & 31u);
    else plane[i >> 5] &= ~(1u << (i & 31u));
}
PTX_DEV void ptx_plane_shift_up(uint32_t* plane, uint32_t at, uint32_t n) {
    for (uint32_t i = n; i > at; --i) ptx_emu_setbit(plane, i, ptx_emu_bit(plane, i - 1u));
    ptx_emu_setbit(plane, at, false);
}
PTX_DEV uint32_t ptx_last_set_below(const uint32_t* bits, uint32_t lim) {
    for (uint32_t s = lim; s-- > 0u;)
        if (ptx_emu_bit(bits, s)) return s + 1u;
    return 0u;
}
PTX_DEV uint32_t ptx_plane_select0(const uint32_t* plane, uint32_t n, uint32_t k) {
    for (uint32_t i = 0; i < n; ++i)
        if (!ptx_emu_bit(plane, i) && k-- == 0u) return i;
    return 0xFFFFFFFFu;
}
PTX_DEV uint32_t ptx_plane_after_tombstones(const uint32_t* dead, const uint32_t* after, uint32_t n, uint32_t pos) {
    uint32_t pick = pos;
    for (uint32_t i = pos + 1u; i < n && ptx_emu_bit(dead, i); ++i)
        if (ptx_emu_bit(after, i)) pick = i;
    return pick;
}
