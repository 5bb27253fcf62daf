/*
 * ptx_platform_gfx950.h — what the kernel sources (merge_core.h, replay_core.h, gen_core.h, change_core.h, cursor_core.h) need
 * from the machine: the parallel-loop macros over a workgroup, LDS atomics, wave-level scans / reductions (DPP, shuffles, ballots),
 * the block-wide scan, the wide loads of the row pass and of the admission walk, the phase stamps of the diagnostic build.
 * This is the ONLY platform the product is built for: MI355X, gfx950, wave64.
 *
 * The CPU test-suite compiles the same kernel sources against tests/emu/ptx_platform_emu.h instead (one thread playing the
 * workgroup, in three iteration orders) to check their logic where no GPU exists; merge_core.h picks the header, nothing else
 * in csrc/ knows about the emulation.  Tunables (PTX_U, PTX_U1, PTX_AC ...) are defined by merge_core.h before this file.
 */
#pragma once
#include <hip/hip_runtime.h>
#define PTX_HD __host__ __device__ static inline
#define PTX_DEV __device__ __forceinline__
#define PTX_SYNC() __syncthreads()
/* the full barrier of merge_core.h (global stores of one phase are read by other lanes in the next): a one-wave build waits for its own outstanding accesses only */
#define PTX_SYNC_FULL() do { if (kThreads == 64u) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); PTX_WSYNC(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); } else __syncthreads(); } while (0)
/* the barrier between two phases that talk through LDS only: the wave's LDS operations are complete (lgkmcnt), its loads from and stores to HBM stay in
 * flight across it.  __syncthreads() also waits for every outstanding global access (vmcnt(0)): the software-pipelined gathers issued ahead of a barrier
 * would be waited for at the barrier, and every output store would stand in the critical path of its phase. */
#define PTX_SYNC_LDS_WG()                                                   \
    do {                                                                    \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");     \
        __builtin_amdgcn_s_barrier();                                       \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");     \
    } while (0)
/* (in a build whose workgroup is known to be ONE wave — kThreads == 64 — the phases are separated by the wave's own program order: no s_barrier at all) */
#define PTX_SYNC_LDS() do { if (kThreads == 64u) PTX_WSYNC(); else PTX_SYNC_LDS_WG(); } while (0)
#define PTX_LDS_ALLOCATED(p, used_bytes, total_bytes) ((void)0) /* a hook of the bump allocator (the CPU test-suite's sanitizer build marks the padding) */
/* P1's list stores: inside the log's LDS window, but — when a header understates the rows of a class — not necessarily inside the list
 * (the log is rejected afterwards); a hook for the CPU test-suite's sanitizer build, which poisons the padding between the arrays */
#define PTX_LDS_WILD_STORE16(p, v) (*(p) = (uint16_t)(v))
/* lanes of ONE wave talking through LDS: the LDS serves a wave's accesses in issue order, so only the compiler has to be kept from
 * moving or forwarding them — no s_barrier and, above all, no wait for the wave's outstanding stores to HBM (__syncthreads has one) */
#define PTX_WSYNC()                                              \
    do {                                                         \
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   \
        __builtin_amdgcn_wave_barrier();                         \
    } while (0)
/* the barrier of a phase in a kernel that may run as ONE wave (kThreads == 64 known at compile time) */
#define PTX_SYNC_T() PTX_SYNC_LDS()
/* threads per workgroup: a compile-time constant in the builds specialised for the usual launch shapes (kThreads != 0:
 * the per-phase loop bounds and strides then fold, which removes a quarter of the scalar instructions), else blockDim.x */
#define PTX_BLOCKDIM (kThreads ? kThreads : blockDim.x)
#define PTX_FOR(i, n) _Pragma("nounroll") for (uint32_t i = threadIdx.x, _n = (n); i < _n; i += PTX_BLOCKDIM)
#define PTX_LEADER if (threadIdx.x == 0)
/* a section that ONE wave of the workgroup runs (the others go on to the next barrier), and its loop over items, a lane each: for the phases of a few dozen
 * items, whose steps then need no s_barrier — PTX_WSYNC() orders the wave's own LDS accesses */
#define PTX_ONE_WAVE if (threadIdx.x < 64u)
#define PTX_LANE_ID (threadIdx.x & 63u)
#define PTX_FOR_LANES(i, n) _Pragma("nounroll") for (uint32_t i = threadIdx.x, _n = (n); i < _n; i += 64u)
PTX_DEV uint32_t ptx_atomic_or(uint32_t* p, uint32_t v) { return atomicOr(p, v); }
PTX_DEV uint32_t ptx_atomic_and(uint32_t* p, uint32_t v) { return atomicAnd(p, v); }
PTX_DEV uint32_t ptx_atomic_add(uint32_t* p, uint32_t v) { return atomicAdd(p, v); }
PTX_DEV uint32_t ptx_atomic_max(uint32_t* p, uint32_t v) { return atomicMax(p, v); }
PTX_DEV uint32_t ptx_atomic_min(uint32_t* p, uint32_t v) { return atomicMin(p, v); }
PTX_DEV unsigned long long ptx_atomic_add64(unsigned long long* p, unsigned long long v) { return atomicAdd(p, v); }
PTX_DEV void ptx_atomic_or64(unsigned long long* p, unsigned long long v) { (void)atomicOr(p, v); }
PTX_DEV void ptx_atomic_max64(unsigned long long* p, unsigned long long v) { (void)atomicMax(p, v); }
PTX_DEV uint32_t ptx_popc(uint32_t x) { return (uint32_t)__popc(x); }
/* global memory that lanes of ONE workgroup hand to each other: workgroup-scope accesses (the waves of a workgroup share their CU's L1, which stores
 * write through and update) and the wait for the wave's outstanding stores that must stand between a store and another lane's load of it */
PTX_DEV uint16_t ptx_coherent_load16(const uint16_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
PTX_DEV void ptx_coherent_store16(uint16_t* p, uint16_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
PTX_DEV uint32_t ptx_coherent_load32(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
PTX_DEV void ptx_coherent_store32(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
PTX_DEV uint64_t ptx_coherent_load64(const uint64_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
PTX_DEV void ptx_coherent_store64(uint64_t* p, uint64_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
PTX_DEV uint32_t ptx_brev(uint32_t x) { return __builtin_bitreverse32(x); } /* v_bfrev_b32 */
/* a value that is the same in every lane of the wave (read from one LDS address, say): tell the compiler, so that what depends on it is scalar code and scalar branches */
#define PTX_U32(x) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(x)))
PTX_DEV void ptx_global_stores_done() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); }
/* wave-aggregated append: ONE LDS atomic per wave, lanes get consecutive slots.  May be called in
 * divergent control flow (the ballot covers the active lanes only). */
PTX_DEV uint32_t ptx_append(uint32_t* cursor, bool pred) {
    const unsigned long long m = __ballot(pred);
    if (m == 0) return 0u;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t leader = (uint32_t)__ffsll((long long)m) - 1u;
    uint32_t b = 0;
    if (lane == leader) b = atomicAdd(cursor, (uint32_t)__popcll(m));
    b = (uint32_t)__shfl((int)b, (int)leader, 64);
    return b + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
}
/* the same for a COUNT per lane (0 .. ): the lane's first slot; one DPP prefix sum and one LDS atomic per wave.  Every lane of the wave must call it
 * (uniform control flow). */
PTX_DEV uint32_t ptx_wave_incl_scan(uint32_t v);
PTX_DEV uint32_t ptx_append_n(uint32_t* cursor, uint32_t count) {
    const uint32_t incl = ptx_wave_incl_scan(count);
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    uint32_t b = 0;
    if ((threadIdx.x & 63u) == 0u && total) b = atomicAdd(cursor, total);
    b = (uint32_t)__builtin_amdgcn_readfirstlane((int)b);
    return b + incl - count;
}
PTX_DEV uint64_t ptx_clock() { return (uint64_t)__builtin_readcyclecounter(); }
/* ---- biglog_core.h: the team that merges ONE large log is a workgroup or — kGrid, a compile-time constant in scope — all the workgroups of a cooperative launch ---- */
/* grid barrier (MI355X_MICROARCH.md "barrier-counter"): ONE monotonic arrival counter — arrival a belongs to barrier a / workgroups, which is over once the
 * counter reaches the next multiple; no reset, no generation word.  One lane per workgroup talks to memory: release fence at agent scope (writes back the XCD's
 * dirty L2 lines: the L2 slices are not coherent with each other) + an explicit wait for it (the compiler may drop the fence's own when it believes the wave's
 * vector-memory scoreboard empty), a relaxed arrival, a RELAXED poll (acquire polls cost 2-3 x), then ONE acquire fence (a CU's L1 is never refreshed by another
 * CU's stores).  Every workgroup must be resident: the host launches such a kernel with hipLaunchCooperativeKernel. */
__device__ __forceinline__ void ptx_grid_sync(uint32_t* bar) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint32_t mine = __hip_atomic_fetch_add(&bar[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t target = (mine / gridDim.x + 1u) * gridDim.x;
        while ((int32_t)(__hip_atomic_load(&bar[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}
#define PTX_BTID (kGrid ? blockIdx.x * blockDim.x + threadIdx.x : threadIdx.x)
#define PTX_BNT (kGrid ? gridDim.x * blockDim.x : blockDim.x)
#define PTX_BFOR(i, n) _Pragma("nounroll") for (uint32_t i = PTX_BTID, _n = (n), _bs = PTX_BNT; i < _n; i += _bs)
#define PTX_BLEADER if (PTX_BTID == 0u)
#define PTX_BSYNC() do { if (kGrid) ptx_grid_sync(_gbar); else __syncthreads(); } while (0)
#define PTX_BFIRST_WAVE if (PTX_BTID < 64u)
/* inside the team: the workgroup's own loop and barrier (its waves talk through global memory: the CU's L1 serves them all), its index and the team's workgroups */
#define PTX_WFOR(i, n) _Pragma("nounroll") for (uint32_t i = threadIdx.x, _wn = (n); i < _wn; i += blockDim.x)
#define PTX_WG_SYNC() __syncthreads()
#define PTX_BWG_ID (kGrid ? blockIdx.x : 0u)
#define PTX_BWG_COUNT (kGrid ? gridDim.x : 1u)
#define PTX_BLANE (threadIdx.x & 63u)
#define PTX_BIG_KEEP(t, s, incl)   /* (every thread runs exactly one iteration of the scan's loops over the team's threads: its registers keep what it found) */
#define PTX_BIG_RECALL(t, s, incl)
#define PTX_G 8u /* lanes that share one member of a large child bucket */
PTX_DEV uint32_t ptx_group_sum(uint32_t c) {
    c += (uint32_t)__shfl_xor((int)c, 1, 64);
    c += (uint32_t)__shfl_xor((int)c, 2, 64);
    c += (uint32_t)__shfl_xor((int)c, 4, 64);
    return c;
}
#define PTX_FORU(i0, n) for (uint32_t i0 = threadIdx.x, _n = (n), _T = PTX_BLOCKDIM; i0 < _n; i0 += PTX_U * _T)
/* the same with U items per thread and step (item u of a step: PTX_IX(i0, u), it exists iff PTX_IN(i0, u)): the loops whose iterations are chains of dependent
 * LDS reads run U chains at once instead of one after the other (PTX_FOR is deliberately not unrolled) */
#define PTX_FORV(i0, n, U) _Pragma("nounroll") for (uint32_t i0 = threadIdx.x, _n = (n), _T = PTX_BLOCKDIM; i0 < _n; i0 += (U) * _T)
#define PTX_IX(i0, u) ((i0) + (uint32_t)(u) * _T)

/* ---- list slots for a batch of rows: rows of class c < 6 get consecutive slots from cursor[c], in ROW order
 *      within the wave (lane-major, each lane holding PTX_U consecutive rows), so that the lists stay (nearly)
 *      sorted by row and later gathers through them stay (nearly) coalesced.  One LDS atomic per wave and batch
 *      (6 lanes, 6 distinct cursors); the ranking itself is a DPP prefix sum in registers.  Every lane of the
 *      wave must call it (uniform control flow). ---- */
PTX_DEV uint32_t ptx_wave_incl_scan(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false); /* row_shr:1 */
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false); /* row_shr:2 */
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false); /* row_shr:4 */
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false); /* row_shr:8 */
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false); /* row_bcast:15 -> rows 1,3 */
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false); /* row_bcast:31 -> rows 2,3 */
    return v;
}
template <int U>
PTX_DEV void ptx_wave_slots(uint32_t* cursor, const uint32_t* cls, uint32_t* slot) {
    /* per-lane counts, 10 bits per class: classes 0..2 in w0, 3..5 in w1 (a wave holds at most 64 * U <= 1023 rows) */
    uint32_t w0 = 0, w1 = 0, off[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const uint32_t c = cls[u];
        const uint32_t sh = (c >= 3u ? c - 3u : c) * 10u;
        off[u] = ((c >= 3u ? w1 : w0) >> sh) & 1023u;
        w0 += (c < 3u ? 1u : 0u) << sh;
        w1 += (c >= 3u && c < 6u ? 1u : 0u) << sh;
    }
    const uint32_t i0 = ptx_wave_incl_scan(w0), i1 = ptx_wave_incl_scan(w1);
    const uint32_t t0 = (uint32_t)__builtin_amdgcn_readlane((int)i0, 63), t1 = (uint32_t)__builtin_amdgcn_readlane((int)i1, 63);
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t basev = 0;
    if (lane < 6u) {
        const uint32_t cnt = ((lane >= 3u ? t1 : t0) >> ((lane >= 3u ? lane - 3u : lane) * 10u)) & 1023u;
        basev = atomicAdd(&cursor[lane], cnt);
    }
    const uint32_t e0 = i0 - w0, e1 = i1 - w1; /* exclusive prefix over the lower lanes */
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const uint32_t c = cls[u];
        const uint32_t sh = (c >= 3u ? c - 3u : c) * 10u;
        const uint32_t b = (uint32_t)__shfl((int)basev, (int)(c & 7u), 64);
        slot[u] = c < 6u ? b + (((c >= 3u ? e1 : e0) >> sh) & 1023u) + off[u] : 0xFFFFFFFFu;
    }
}

/* ---- packed 16-bit arithmetic and byte permutes of the admission walk and the row pass (v_pk_max_u16, v_pk_sub_u16 clamp,
 *      v_perm_b32, v_lshlrev_b64 / v_lshrrev_b64) ---- */
typedef uint16_t ptx_u16x2 __attribute__((ext_vector_type(2)));
PTX_DEV uint32_t ptx_pk_max_u16(uint32_t a, uint32_t b) { /* per 16-bit half: max */
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(ptx_u16x2, a), __builtin_bit_cast(ptx_u16x2, b)));
}
PTX_DEV uint32_t ptx_pk_subsat_u16(uint32_t a, uint32_t b) { /* per 16-bit half: a - b, 0 where b > a */
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(__builtin_bit_cast(ptx_u16x2, a), __builtin_bit_cast(ptx_u16x2, b)));
}
/* byte k of the result = byte sel.k of the eight bytes {hi, lo} (0..3 = lo, 4..7 = hi), 0x0c = the constant 0 */
PTX_DEV uint32_t ptx_perm(uint32_t hi, uint32_t lo, uint32_t sel) { return __builtin_amdgcn_perm(hi, lo, sel); }
PTX_DEV uint64_t ptx_shl64(uint64_t x, uint32_t s) { return x << (s & 63u); } /* the hardware takes the low six bits of the amount */
PTX_DEV uint64_t ptx_shr64(uint64_t x, uint32_t s) { return x >> (s & 63u); }
/* the value of SOME lane whose `pred` holds (false: no lane's does); the same answer in every lane */
PTX_DEV bool ptx_wave_pick(bool pred, uint32_t value, uint32_t& out) {
    const unsigned long long m = __ballot(pred);
    if (m == 0) return false;
    out = (uint32_t)__builtin_amdgcn_readlane((int)value, (int)(__ffsll((long long)m) - 1));
    return true;
}
/* Wave-wide reductions as DPP prefix steps (row_shr 1 / 2 / 4 / 8, row_bcast 15 / 31: lane 63 ends up with the reduction of all 64, read back by v_readlane) — a
 * dozen vector instructions and no LDS.  As butterflies of __shfl_xor they were six ds_bpermute round trips EACH, one after the other: the five reductions that end
 * a wave's admission walk were 30 dependent LDS trips, more than the walk of a one-step log itself (round 6: the kernel is bound by its chain of fixed latencies). */
#define PTX_DPP_STEPS(OP_)                                                                                         \
    v = OP_(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false)); /* row_shr:1 */           \
    v = OP_(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false)); /* row_shr:2 */           \
    v = OP_(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false)); /* row_shr:4 */           \
    v = OP_(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false)); /* row_shr:8 */           \
    v = OP_(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false)); /* row_bcast:15 -> rows 1,3 */ \
    v = OP_(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false)); /* row_bcast:31 -> rows 2,3 */
PTX_DEV uint32_t ptx_max_u32_(uint32_t a, uint32_t b) { return a > b ? a : b; }
PTX_DEV uint32_t ptx_wave_pk_max_u16_dpp(uint32_t v) { /* per 16-bit half, the same value in every lane (a lane without a source reads 0: the identity of an unsigned maximum) */
    PTX_DPP_STEPS(ptx_pk_max_u16)
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
PTX_DEV uint32_t ptx_wave_pk_max_u16(uint32_t v) { /* (the butterfly: what the builds of three and more waves per log keep — measured 0.9 % faster there, their LDS trips hide behind the other waves) */
    for (int d = 32; d >= 1; d >>= 1) v = ptx_pk_max_u16(v, (uint32_t)__shfl_xor((int)v, d, 64));
    return v;
}

PTX_DEV uint32_t ptx_mul24(uint32_t a, uint32_t b) { return __umul24(a, b); } /* the low 24 bits of both factors */
PTX_DEV uint32_t ptx_mad24(uint32_t a, uint32_t b, uint32_t c) { /* (a & 0xFFFFFF) * (b & 0xFFFFFF) + c in ONE full-rate instruction */
    uint32_t r;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
PTX_DEV uint32_t ptx_mad24_su(uint32_t a, uint32_t b, uint32_t c) { /* the same with b the same in every lane (a scalar register) */
    uint32_t r;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(b), "v"(c));
    return r;
}
PTX_DEV uint32_t ptx_min(uint32_t a, uint32_t b) { return __builtin_elementwise_min(a, b); }
/* List slots for the rows of a wave's step, U consecutive rows per lane, their classes (0..7) in the bytes of c4: rows of class
 * c < 6 get consecutive values from cursor[c], in ROW order within the wave (lane-major), so that the lists stay (nearly) sorted
 * by row and later gathers through them stay (nearly) coalesced; rows of class 6 / 7 get dump .. dump + 3.  One LDS atomic per
 * wave and step (6 lanes, 6 distinct cursors); the ranking is a DPP prefix sum over counters of 10 bits per class packed in two
 * words (classes 0..2 at bits 0 / 10 / 20, classes 3..5 at bits 32 / 42 / 52, the unlisted ones in the two bits left over, which
 * may overflow).  Every lane of the wave must call it (uniform control flow). */
template <int U>
PTX_DEV void ptx_wave_slots4(uint32_t* cursor, uint32_t dump, uint32_t c4, uint32_t* slot) {
    const uint32_t sh4 = ptx_perm(0x3E3E342Au, 0x20140A00u, c4); /* class -> bit position of its counter */
    const uint32_t ix4 = c4 << 2;                                /* class -> byte address of lane `class` for ds_bpermute */
    uint32_t wl[U], wh[U], tl = 0, th = 0;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const uint64_t one = ptx_shl64(1ull, sh4 >> (8 * u));
        wl[u] = tl;
        wh[u] = th;
        tl += (uint32_t)one;
        th += (uint32_t)(one >> 32);
    }
    const uint32_t il = ptx_wave_incl_scan(tl), ih = ptx_wave_incl_scan(th);
    const uint32_t t0 = (uint32_t)__builtin_amdgcn_readlane((int)il, 63), t1 = (uint32_t)__builtin_amdgcn_readlane((int)ih, 63);
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t basev = dump;
    if (lane < 6u) {
        const uint32_t cnt = ((lane >= 3u ? t1 : t0) >> ((lane >= 3u ? lane - 3u : lane) * 10u)) & 1023u;
        basev = atomicAdd(&cursor[lane], cnt);
    }
    const uint32_t el = il - tl, eh = ih - th; /* exclusive prefix over the lower lanes */
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const uint64_t x = ((uint64_t)(eh + wh[u]) << 32) | (uint64_t)(el + wl[u]);
        const uint32_t b = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((ix4 >> (8 * u)) & 255u), (int)basev);
        slot[u] = b + ((uint32_t)ptx_shr64(x, sh4 >> (8 * u)) & 1023u);
    }
}

/* Software-pipelined uniform loops: step st of `steps` handles group PTX_G_OF(st); the loads of step st+1 are
 * issued before step st is processed (the caller keeps two register sets).  Every thread runs every step; a
 * group index past the end means "no work" (its loads are clamped to valid addresses, its effects masked). */
/* x / threads-per-workgroup without a division: the host passes magic = floor(2^32 / T) + 1, exact for x * T < 2^32
 * (x is a row count + T here; a uniform integer division costs ~25 instructions per wave, and a log has a dozen) */
#define PTX_DIV_T(x) (kThreads ? (uint32_t)(x) / (kThreads ? kThreads : 1u) : (uint32_t)__umulhi((uint32_t)(x), A.div_magic))
#define PTX_STEPS(groups) PTX_DIV_T((groups) + PTX_BLOCKDIM - 1u)
#define PTX_G_OF(st, steps) (threadIdx.x + (st) * PTX_BLOCKDIM)
#define PTX_WHOLE_STEPS(groups) (PTX_DIV_T(groups) * PTX_BLOCKDIM == (uint32_t)(groups)) /* the groups fill whole steps of the workgroup: no thread idles in the last one */

/* the same for loops over list items, PTX_U items per thread and step, lanes on consecutive items:
 * step st, slot u handles item PTX_J_OF(st, u) (past the end = no work); PTX_JX maps it for the emulation's
 * reversed order */
#define PTX_JSTEPS_U(n, U) ((PTX_DIV_T((n) + PTX_BLOCKDIM - 1u) + (U)-1u) / (U)) /* = ceil(n / (U * T)) */
#define PTX_J_OF_U(st, u, U) (((st) * (U) + (uint32_t)(u)) * PTX_BLOCKDIM + threadIdx.x)
#define PTX_JX(j, n) (j)
/* loops over BLOCKS of items: a block is ONE wave's work (64 lanes), every wave takes U blocks per step.  The marks' loops use them: a block holds the mark
 * ops of one stretch of the log, so the cache lines a wave's gathers touch are touched by no other wave (with a block per workgroup the three waves of a
 * step asked for the same lines one after the other, and the L1 had often dropped them in between: 1.6 x the lines the columns hold) */
#define PTX_JB_CAP 64u
#define PTX_JB_STEPS(B, U) (((B) + (U) * PTX_NWAVES - 1u) / ((U) * PTX_NWAVES))
#define PTX_JB_BLOCK(st, u, U) (((st) * (U) + (uint32_t)(u)) * PTX_NWAVES + (threadIdx.x >> 6))
#define PTX_JB_LANE(st, u, U) (threadIdx.x & 63u)
#define PTX_JB_LANE_IS_FIXED 1 /* a thread's lane of a block is the same in every step: its share of the runs is worked out once per log */
/* the kernel arguments read AGAIN from the kernarg segment (scalar loads): what a phase derives from them is a new value to the compiler, so the
 * values an earlier phase derived need not stay in scalar registers (or in the VGPR lanes they are spilled to) across the phases in between */
template <class ArgsT>
PTX_DEV const ArgsT& ptx_fresh_args(const ArgsT&) {
    const __attribute__((address_space(4))) ArgsT* p = (const __attribute__((address_space(4))) ArgsT*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return *(const ArgsT*)p;
}
#define PTX_FRESH_ARGS(A) ptx_fresh_args(A)
/* a word of the batch that no kernel of the call writes (offsets, headers), at an address that is the same in every lane: read through the constant address space,
 * i.e. by the scalar unit (left to itself the compiler only does so where it can prove that no store of the kernel so far may alias the word) */
template <class T>
PTX_DEV T ptx_const_load(const T* p) { return *(const __attribute__((address_space(4))) T*)p; }
#define PTX_CONST_LOAD(p) ptx_const_load(p)
#define PTX_KEEP_VGPR(x) asm volatile("" : "+v"(x)) /* the value stays in its register: the compiler neither works it out again at each use nor treats it as a constant */

/* wave-explicit loops: every wave runs the body once with its wave index `w` and lane index `lane`; the
 * emulation plays three one-lane waves in turn */
#define PTX_WAVE_FIRST(g) ((uint32_t)__builtin_amdgcn_readfirstlane((int)((g) - (threadIdx.x & 63u)))) /* index handled by lane 0 of this wave (readfirstlane: the compiler then knows it is the same in every lane) */
#define PTX_WS 64u
#define PTX_NWAVES (PTX_BLOCKDIM >> 6)
/* (the wave index through readfirstlane: the compiler then knows that what derives from it is the same in every lane) */
#define PTX_FOR_WAVE(w, lane) for (uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63u, _once = 1; _once; _once = 0)
PTX_DEV uint32_t ptx_wave_last(uint32_t incl) { return (uint32_t)__builtin_amdgcn_readlane((int)incl, 63); }
PTX_DEV uint32_t ptx_wave_total(uint32_t v) { return ptx_wave_last(ptx_wave_incl_scan(v)); }
PTX_DEV uint32_t ptx_wave_min(uint32_t v) { /* the same value in every lane */
    for (int d = 32; d >= 1; d >>= 1) {
        const uint32_t o = (uint32_t)__shfl_xor((int)v, d, 64);
        v = o < v ? o : v;
    }
    return v;
}
PTX_DEV uint32_t ptx_wave_max_dpp(uint32_t v) { /* the same value in every lane */
    PTX_DPP_STEPS(ptx_max_u32_)
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
PTX_DEV uint32_t ptx_wave_max(uint32_t v) {
    for (int d = 32; d >= 1; d >>= 1) {
        const uint32_t o = (uint32_t)__shfl_xor((int)v, d, 64);
        v = o > v ? o : v;
    }
    return v;
}

#define PTX_NTHREADS PTX_BLOCKDIM

#define PTX_FORA(i0, n) for (uint32_t i0 = threadIdx.x, _n = (n), _T = PTX_BLOCKDIM; i0 < _n; i0 += PTX_UA * _T)

/* uniform loop over groups of PTX_U consecutive items: every thread runs every step (g may be past the end) */
#define PTX_FORG(g, groups) for (uint32_t _ng = (groups), _g0 = 0, g = threadIdx.x; _g0 < _ng; _g0 += PTX_BLOCKDIM, g += PTX_BLOCKDIM)

/* sum / max over the workgroup into LDS words (every thread calls them) */
PTX_DEV void ptx_reduce_add64(unsigned long long* dst, unsigned long long v) {
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(dst, v);
}
PTX_DEV void ptx_reduce_add64_dpp(unsigned long long* dst, unsigned long long v) {
    /* the 64-bit sum of the wave from four 16-bit limbs, each a 32-bit DPP prefix sum (64 x 65 535 fits 22 bits), put together again: no LDS round trip
     * (the butterfly of 64-bit shuffles was twelve ds_bpermute trips one after the other) */
    const uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    const unsigned long long l0 = ptx_wave_last(ptx_wave_incl_scan(lo & 0xFFFFu)), l1 = ptx_wave_last(ptx_wave_incl_scan(lo >> 16));
    const unsigned long long l2 = ptx_wave_last(ptx_wave_incl_scan(hi & 0xFFFFu)), l3 = ptx_wave_last(ptx_wave_incl_scan(hi >> 16));
    const unsigned long long t = l0 + (l1 << 16) + (l2 << 32) + (l3 << 48);
    if ((threadIdx.x & 63) == 0) atomicAdd(dst, t);
}

PTX_DEV void ptx_reduce_add32_dpp(uint32_t* dst, uint32_t v) { /* every lane of the wave calls it: one LDS atomic per wave */
    v = ptx_wave_last(ptx_wave_incl_scan(v)); /* (a DPP prefix sum, its last lane: the total) */
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(dst, v);
}
PTX_DEV void ptx_reduce_add32(uint32_t* dst, uint32_t v) { /* every lane of the wave calls it: one LDS atomic per wave */
    for (int d = 32; d >= 1; d >>= 1) v += (uint32_t)__shfl_xor((int)v, d, 64);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(dst, v);
}

PTX_DEV void ptx_reduce_max32(uint32_t* dst, uint32_t v) {
    v = ptx_wave_max(v);
    if ((threadIdx.x & 63) == 0) atomicMax(dst, v);
}

/* ---- block-wide exclusive scan of an LDS array (element k at a[k*STRIDE]), in place; returns the
 *      total (all threads call it; ends with a barrier) ---- */
template <class T, int STRIDE, uint32_t kThreads>
PTX_DEV uint32_t ptx_scan_excl(T* a, uint32_t m, uint32_t* tmp /* >= 36 u32 in LDS */, uint32_t div_magic = 0 /* as PTX_DIV_T; needed when kThreads == 0 */) {
    const uint32_t T_ = PTX_BLOCKDIM, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = (T_ + 63) >> 6;
    if (kThreads == 64 && m <= 64u) { /* one wave, one element per lane: a DPP prefix sum and nothing else */
        const uint32_t v = lane < m ? (uint32_t)a[lane * STRIDE] : 0u;
        const uint32_t in = ptx_wave_incl_scan(v);
        if (lane < m) a[lane * STRIDE] = (T)(in - v);
        PTX_SYNC_T();
        return (uint32_t)__builtin_amdgcn_readlane((int)in, 63);
    }
    const uint32_t chunk = kThreads ? (m + T_ - 1) / (kThreads ? kThreads : 1u) : (uint32_t)__umulhi(m + T_ - 1, div_magic);
    const uint32_t lo = tid * chunk < m ? tid * chunk : m;
    const uint32_t hi = lo + chunk < m ? lo + chunk : m;
    uint32_t sum = 0;
    for (uint32_t j = lo; j < hi; ++j) sum += a[j * STRIDE];
    const uint32_t incl = ptx_wave_incl_scan(sum); /* DPP prefix sum: no LDS traffic */
    uint32_t wbase = 0, total = 0;
    if (kThreads == 64) { /* one wave: the total is lane 63's prefix, no trip through the LDS */
        total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    } else {
        if (lane == 63) tmp[wave] = incl;
        PTX_SYNC_T();
#pragma nounroll
        for (uint32_t w = 0; w < nwaves; ++w) { /* a workgroup has at most 16 waves: every thread sums the few wave totals itself */
            const uint32_t t = tmp[w];
            wbase += w < wave ? t : 0u;
            total += t;
        }
    }
    uint32_t run = wbase + incl - sum;
    for (uint32_t j = lo; j < hi; ++j) {
        uint32_t v = a[j * STRIDE];
        a[j * STRIDE] = (T)run;
        run += v;
    }
    PTX_SYNC_T();
    return total;
}

/* diagnostic build only: phase k = time from stamp k to the stamp recorded next IN TIME (the stamp indices are not in execution order), summed over the logs */
PTX_DEV void ptx_flush_clocks(unsigned long long* clocks, unsigned long long* clk, int nclk) {
    if (!clocks) return;
    clk[nclk] = ptx_clock();
#pragma nounroll
    for (int k = 0; k < nclk; ++k) {
        if (clk[k] == 0) continue;
        unsigned long long next = clk[nclk];
#pragma nounroll
        for (int j = 0; j < nclk; ++j)
            if (clk[j] > clk[k] && clk[j] < next) next = clk[j];
        atomicAdd(&clocks[k], next - clk[k]);
    }
}

/* logs whose one-pass admission check failed and were walked again: counted by the diagnostic build only (slot PTX_CLK_EXACT_WALKS of the
 * phase clocks; tools/phase_profile.py prints it) */
#define PTX_NOTE_EXACT_WALK()                                                                              \
    do {                                                                                                   \
        if (kDiag && A.clocks && threadIdx.x == 0) atomicAdd(&A.clocks[PTX_CLK_EXACT_WALKS], 1ull);               \
    } while (0)

/* only in the diagnostic build of the kernel (kDiag): phase cycle stamps and the early exit of the per-phase PMC runs */
#ifdef PTX_DIAG /* the early exits cost the diagnostic kernel 35 VGPRs (its stamps would then be taken at 4 waves per SIMD): only in the builds that ask for them */
#define PTX_STOP_AFTER(k)                                              \
    if ((k) != 0 && A.stop_after == (k)) {                             \
        lds_high = bp.high;                                            \
        return PTX_OK;                                                 \
    }
#else
#define PTX_STOP_AFTER(k)
#endif
#define PTX_STAMP(k)                                                   \
    do {                                                               \
        if (kDiag) {                                                   \
            if (A.clocks && threadIdx.x == 0) H->clk[k] = ptx_clock(); \
            PTX_STOP_AFTER(k)                                          \
        }                                                              \
    } while (0)

/* PTX_AC consecutive headers / envelope rows of a lane; indices past `hi` are clamped, their effects masked.
 * The library pads its copies of both columns, so the 16-byte loads may run past the last change. */
#ifndef PTX_NT
#define PTX_NT 0 /* 1: the columns that are read exactly once (Change envelope, op ids / action / mark type of the row pass) are loaded with the non-temporal hint */
#endif
typedef uint32_t ptx_u32x4 __attribute__((ext_vector_type(4)));
typedef ptx_u32x4 ptx_u32x4_a4 __attribute__((aligned(4)));
typedef uint32_t ptx_u32x2 __attribute__((ext_vector_type(2)));
typedef ptx_u32x2 ptx_u32x2_a4 __attribute__((aligned(4)));
typedef uint64_t ptx_u64x2 __attribute__((ext_vector_type(2)));
typedef ptx_u64x2 ptx_u64x2_a8 __attribute__((aligned(8)));
typedef uint32_t ptx_u32_a1 __attribute__((aligned(1)));
#if PTX_NT
#define PTX_STREAM_LOAD(p_) __builtin_nontemporal_load(p_)
#else
#define PTX_STREAM_LOAD(p_) (*(p_))
#endif
#define PTX_ADM_HDRS(dst_, cl_)                                                              \
    {                                                                                        \
        static_assert(PTX_AC == 4, "one 16-byte load of headers");                           \
        const ptx_u32x4 q_ = PTX_STREAM_LOAD((const ptx_u32x4_a4*)(c_hdr + (cl_)));          \
        dst_[0] = q_.x;                                                                      \
        dst_[1] = q_.y;                                                                      \
        dst_[2] = q_.z;                                                                      \
        dst_[3] = q_.w;                                                                      \
    }
/* the same rows as dwords: e0_ = seq | deps[0] << 16, e1_ = deps[1] | deps[2] << 16 (rows of four u16) */
#define PTX_ADM_ENVS32(e0_, e1_, cl_)                                                        \
    {                                                                                        \
        const ptx_u32x4_a4* p_ = (const ptx_u32x4_a4*)(c_env + (uint64_t)(cl_) * 4u);        \
        const ptx_u32x4 qa_ = PTX_STREAM_LOAD(p_), qb_ = PTX_STREAM_LOAD(p_ + 1);            \
        e0_[0] = qa_.x;                                                                      \
        e1_[0] = qa_.y;                                                                      \
        e0_[1] = qa_.z;                                                                      \
        e1_[1] = qa_.w;                                                                      \
        e0_[2] = qb_.x;                                                                      \
        e1_[2] = qb_.y;                                                                      \
        e0_[3] = qb_.z;                                                                      \
        e1_[3] = qb_.w;                                                                      \
    }
/* AC_ consecutive headers (AC_ = 4: one 16-byte load, 2: one of 8 bytes) */
#define PTX_ADM_HDRSN(dst_, cl_, AC_)                                                        \
    {                                                                                        \
        if ((AC_) == 4u) {                                                                   \
            const ptx_u32x4 q_ = PTX_STREAM_LOAD((const ptx_u32x4_a4*)(c_hdr + (cl_)));      \
            dst_[0] = q_.x;                                                                  \
            dst_[1] = q_.y;                                                                  \
            dst_[(AC_) - 2u] = q_.z;                                                         \
            dst_[(AC_) - 1u] = q_.w;                                                         \
        } else {                                                                             \
            const ptx_u32x2 q_ = PTX_STREAM_LOAD((const ptx_u32x2_a4*)(c_hdr + (cl_)));      \
            dst_[0] = q_.x;                                                                  \
            dst_[1] = q_.y;                                                                  \
        }                                                                                    \
    }
/* AC_ consecutive envelope rows of W_ dwords (W_ = 4, 6, 8: four to fifteen actors) : e_[u][j] = halves 2 j, 2 j + 1 of the row of change cl_ + u; loads of 16 (+ 8) bytes */
#define PTX_ADM_ROWSN(e_, cl_, W_, AC_)                                                           \
    {                                                                                        \
        const uint32_t* p_ = (const uint32_t*)(c_env + (uint64_t)(cl_) * (2u * (W_)));       \
        _Pragma("unroll") for (uint32_t u_ = 0; u_ < (AC_); ++u_) {                         \
            _Pragma("unroll") for (uint32_t j_ = 0; j_ + 4u <= (W_); j_ += 4u) {             \
                const ptx_u32x4 q_ = PTX_STREAM_LOAD((const ptx_u32x4_a4*)(p_ + u_ * (W_) + j_)); \
                e_[u_][j_] = q_.x;                                                           \
                e_[u_][j_ + 1u] = q_.y;                                                      \
                e_[u_][j_ + 2u] = q_.z;                                                      \
                e_[u_][j_ + 3u] = q_.w;                                                      \
            }                                                                                \
            if ((W_) % 4u == 2u) {                                                           \
                const ptx_u32x2 q_ = PTX_STREAM_LOAD((const ptx_u32x2_a4*)(p_ + u_ * (W_) + ((W_) - 2u))); \
                e_[u_][(W_) - 2u] = q_.x;                                                    \
                e_[u_][(W_) - 1u] = q_.y;                                                    \
            }                                                                                \
        }                                                                                    \
    }
#define PTX_ADM_ENVS(dst_, cl_)                                                              \
    {                                                                                        \
        struct __attribute__((packed, aligned(4))) PtxE4 { uint16_t v[PTX_AC][4]; };         \
        const PtxE4 q_ = *(const PtxE4*)(c_env + (uint64_t)(cl_) * 4u);                      \
        _Pragma("unroll") for (uint32_t u_ = 0; u_ < PTX_AC; ++u_)                           \
            _Pragma("unroll") for (uint32_t b_ = 0; b_ < 4u; ++b_) dst_[u_][b_] = q_.v[u_][b_]; \
    }

/* the action / mark_type bytes of a thread's PTX_U1 consecutive rows from r0_ on, one byte each in dst_ (uses N): ONE unaligned
 * 4-byte load; the library pads its copies of the byte columns by PTX_BYTE_PAD */
/* the ids of a thread's PTX_U1 consecutive rows, all of which exist: one address, loads of 16 + 8 bytes */
#define PTX_P1_IDS(dst_, ptr_)                                                         \
    {                                                                                  \
        static_assert(PTX_U1 == 3 || PTX_U1 == 4, "16 + 8 or 16 + 16 bytes");          \
        const uint64_t* p_ = (ptr_);                                                   \
        const ptx_u64x2 q_ = PTX_STREAM_LOAD((const ptx_u64x2_a8*)p_);                 \
        dst_[0] = q_.x;                                                                \
        dst_[1] = q_.y;                                                                \
        if (PTX_U1 == 4) {                                                             \
            const ptx_u64x2 r_ = PTX_STREAM_LOAD((const ptx_u64x2_a8*)(p_ + 2));       \
            dst_[2] = r_.x;                                                            \
            dst_[PTX_U1 - 1] = r_.y;                                                   \
        } else {                                                                       \
            dst_[2] = PTX_STREAM_LOAD(p_ + 2);                                         \
        }                                                                              \
    }
#define PTX_P1_BYTES(col_, r0_, dst_, n_) dst_ = PTX_STREAM_LOAD((const ptx_u32_a1*)(col_ + ((r0_) < (n_) ? (r0_) : (n_) - 1u)));

/* ---- gen_core.h / change_core.h: ONE wave per workgroup ---- */
/* 64-wide ballot over lanes: `expr` may use `lane_` */
#define PTX_BALLOT64(mask_, lane_, expr)                 \
    uint64_t mask_;                                      \
    {                                                    \
        const uint32_t lane_ = threadIdx.x & 63u;        \
        mask_ = __ballot(expr);                          \
    }
#define PTX_LANE0 (threadIdx.x == 0)
#define PTX_GEN_FOR(i, n) for (uint32_t i = threadIdx.x, _gn = (n); i < _gn; i += 64u)
#define PTX_MEM __device__ __forceinline__
/* L[lo + 1 .. hi] = L[lo .. hi - 1] for hi - lo <= 64: every lane has read its element before any lane writes */
PTX_DEV void ptx_shift_up64(uint32_t* L, uint32_t lo, uint32_t hi) {
    const uint32_t i = lo + (threadIdx.x & 63u);
    const uint32_t v = i < hi ? L[i] : 0u;
    PTX_SYNC();
    if (i < hi) L[i + 1u] = v;
}

/* ---- element lists of the one-wave kernels (gen_core.h): keys in document order (KeyT = u16 or u32, 16-byte aligned, read
 *      16 bytes per lane) + one bit per position in planes.  Every lane of the wave calls these. ---- */
template <class KeyT>
PTX_DEV uint32_t ptx_list_key_at(const uint32_t (&w)[4], uint32_t j) { /* element j of a 16-byte block */
    return sizeof(KeyT) == 2 ? (w[j >> 1] >> (16u * (j & 1u))) & 0xFFFFu : w[j];
}
/* position of `key` among keys[0 .. n), or 0xFFFFFFFF (keys are unique) */
template <class KeyT>
PTX_DEV uint32_t ptx_list_find(const KeyT* keys, uint32_t n, uint32_t key) {
    constexpr uint32_t W = 16u / (uint32_t)sizeof(KeyT);
    const uint32_t lane = threadIdx.x & 63u;
    for (uint32_t base = 0; base < n; base += 64u * W) {
        const uint32_t i0 = base + lane * W;
        uint32_t hit = 0xFFFFFFFFu;
        if (i0 < n) {
            const uint4 v = *(const uint4*)(keys + i0);
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (uint32_t j = 0; j < W; ++j)
                if (i0 + j < n && ptx_list_key_at<KeyT>(w, j) == key) hit = j;
        }
        const unsigned long long m = __ballot(hit != 0xFFFFFFFFu);
        if (m) {
            const uint32_t l = (uint32_t)__ffsll((long long)m) - 1u;
            return base + l * W + (uint32_t)__shfl((int)hit, (int)l, 64);
        }
    }
    return 0xFFFFFFFFu;
}
/* keys[at + 1 .. n] = keys[at .. n - 1] (keys[at] is left for the caller): whole 16-byte blocks, 64 of them per step, from the
 * top down; a block's new content = its old one moved up by one key, the key that enters from below read separately */
template <class KeyT>
PTX_DEV void ptx_list_shift_up(KeyT* keys, uint32_t at, uint32_t n) {
    constexpr uint32_t W = 16u / (uint32_t)sizeof(KeyT);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t b_at = at / W, b_top = n / W; /* position n (the new last element) lives in block b_top */
    for (uint32_t bh = b_top + 1u; bh > b_at;) {
        const uint32_t bl = bh - b_at > 64u ? bh - 64u : b_at;
        const uint32_t b = bl + lane;
        const bool act = b < bh;
        uint32_t w[4] = {0u, 0u, 0u, 0u}, prev = 0u;
        if (act) {
            const uint4 v = *(const uint4*)(keys + b * W);
            w[0] = v.x;
            w[1] = v.y;
            w[2] = v.z;
            w[3] = v.w;
            if (b * W > at) prev = keys[b * W - 1u];
        }
        PTX_WSYNC(); /* every lane has read before any lane writes; a lower chunk is only rewritten after the chunks above it */
        if (act) {
            uint32_t o[4] = {0u, 0u, 0u, 0u};
#pragma unroll
            for (uint32_t j = 0; j < W; ++j) {
                const uint32_t old_j = ptx_list_key_at<KeyT>(w, j), below = j ? ptx_list_key_at<KeyT>(w, j - 1u) : prev;
                const uint32_t nj = b * W + j > at ? below : old_j;
                if (sizeof(KeyT) == 2) o[j >> 1] |= nj << (16u * (j & 1u));
                else o[j] = nj;
            }
            *(uint4*)(keys + b * W) = make_uint4(o[0], o[1], o[2], o[3]);
        }
        PTX_WSYNC();
        bh = bl;
    }
}
/* the same for a bit plane: bits at + 1 .. n = old bits at .. n - 1, bit `at` = 0 */
PTX_DEV void ptx_plane_shift_up(uint32_t* plane, uint32_t at, uint32_t n) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t w0 = at >> 5, w1 = n >> 5;
    for (uint32_t wh = w1 + 1u; wh > w0;) {
        const uint32_t wl = wh - w0 > 64u ? wh - 64u : w0;
        const uint32_t w = wl + lane;
        const bool act = w < wh;
        const uint32_t cur = act ? plane[w] : 0u, prv = act && w > w0 ? plane[w - 1u] : 0u;
        PTX_WSYNC();
        if (act) {
            const uint32_t low = (1u << (at & 31u)) - 1u;
            plane[w] = w == w0 ? (cur & low) | ((cur & ~low) << 1) : (cur << 1) | (prv >> 31);
        }
        PTX_WSYNC();
        wh = wl;
    }
}
/* position + 1 of the highest set bit of `bits` strictly below `lim`, 0 when there is none — the same value in every lane (one-wave workgroups; every lane
 * calls it).  Chunks of 64 words from the top down, one word per lane: a ballot finds the highest lane whose word has a bit, ONE readlane fetches that word. */
PTX_DEV uint32_t ptx_last_set_below(const uint32_t* bits, uint32_t lim) {
    const uint32_t lane = threadIdx.x & 63u;
    if (lim == 0u) return 0u;
    { /* usually the word that holds slot lim - 1 has it (one LDS address for the wave: a broadcast, then scalar code) */
        const uint32_t w = (lim - 1u) >> 5;
        uint32_t m = bits[w];
        if (lim & 31u) m &= (1u << (lim & 31u)) - 1u;
        m = (uint32_t)__builtin_amdgcn_readfirstlane((int)m);
        if (m) return (w << 5) + 32u - (uint32_t)__builtin_clz(m);
        lim = w << 5; /* the words below it */
    }
    const uint32_t words = lim >> 5;
    for (uint32_t top = (words + 63u) & ~63u; top != 0u; top -= 64u) {
        const uint32_t w = top - 64u + lane;
        uint32_t m = 0;
        if (w < words) {
            m = bits[w];
            if ((w << 5) + 32u > lim) m &= (1u << (lim & 31u)) - 1u;
        }
        const unsigned long long nz = __ballot(m != 0u);
        if (nz) {
            const uint32_t l = 63u - (uint32_t)__builtin_clzll(nz);
            const uint32_t mm = (uint32_t)__shfl((int)m, (int)l, 64);
            return ((top - 64u + l) << 5) + (31u - (uint32_t)__builtin_clz(mm)) + 1u;
        }
    }
    return 0u;
}
/* position of the k-th (0-based) ZERO bit of plane among positions [0, n), or 0xFFFFFFFF */
PTX_DEV uint32_t ptx_plane_select0(const uint32_t* plane, uint32_t n, uint32_t k) {
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t seen = 0;
    for (uint32_t bw = 0; (bw << 5) < n; bw += 64u) {
        const uint32_t w = bw + lane;
        uint32_t a = 0;
        if ((w << 5) < n) {
            a = ~plane[w];
            if ((w << 5) + 32u > n) a &= (1u << (n & 31u)) - 1u;
        }
        const uint32_t c = ptx_popc(a), incl = ptx_wave_incl_scan(c), total = ptx_wave_last(incl);
        if (k < seen + total) {
            const unsigned long long m = __ballot(seen + incl > k);
            const uint32_t l = (uint32_t)__ffsll((long long)m) - 1u;
            uint32_t x = (uint32_t)__shfl((int)a, (int)l, 64);
            const uint32_t before = seen + (uint32_t)__shfl((int)(incl - c), (int)l, 64);
            for (uint32_t r = k - before; r; --r) x &= x - 1u;
            return ((bw + l) << 5) + (uint32_t)__builtin_ctz(x);
        }
        seen += total;
    }
    return 0xFFFFFFFFu;
}
/* lookAfterTombstones on planes: from the visible element at `pos`, over the directly following tombstones (dead bits): the
 * LAST one whose `after` bit is set, else pos.  Both planes are zero from position n on. */
PTX_DEV uint32_t ptx_plane_after_tombstones(const uint32_t* dead, const uint32_t* after, uint32_t n, uint32_t pos) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t w0 = pos >> 5;
    uint32_t pick = pos;
    for (uint32_t bw = w0; (bw << 5) < n; bw += 64u) {
        const uint32_t w = bw + lane;
        uint32_t d = 0, vm = 0;
        if ((w << 5) < n) {
            d = dead[w];
            vm = (w << 5) + 32u <= n ? 0xFFFFFFFFu : (1u << (n & 31u)) - 1u;
        }
        const uint32_t above = w == w0 ? ~((2u << (pos & 31u)) - 1u) : 0xFFFFFFFFu; /* strictly above pos */
        const uint32_t alive = ~d & vm & above;
        uint32_t mk = (w << 5) < n ? d & after[w] & above : 0u;
        const unsigned long long ma = __ballot(alive != 0u);
        if (ma) { /* the next visible element ends the run of tombstones */
            const uint32_t l = (uint32_t)__ffsll((long long)ma) - 1u;
            const uint32_t nb = (uint32_t)__builtin_ctz((uint32_t)__shfl((int)alive, (int)l, 64));
            if (lane > l) mk = 0;
            if (lane == l) mk &= (1u << nb) - 1u;
        }
        const unsigned long long mm = __ballot(mk != 0u);
        if (mm) {
            const uint32_t l2 = 63u - (uint32_t)__builtin_clzll(mm);
            pick = ((bw + l2) << 5) + 31u - (uint32_t)__builtin_clz((uint32_t)__shfl((int)mk, (int)l2, 64));
        }
        if (ma) break;
    }
    return pick;
}
