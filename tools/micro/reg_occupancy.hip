// Resident workgroups per CU of 192-thread workgroups with little LDS as a function of the registers the kernel claims (inline asm clobbers):
// what does a wave of 68 VGPRs / 106 SGPRs (the merge kernel) cost in wave slots on gfx950?
// hipcc --offload-arch=gfx950 -O3 -o tools/micro/bin/reg_occupancy tools/micro/reg_occupancy.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <algorithm>
#define BODY(...)                                                                                       \
    extern __shared__ unsigned lds[];                                                                   \
    unsigned hw, xcc;                                                                                   \
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));                                    \
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));                                  \
    const unsigned cu = ((xcc & 0xF) << 8) | ((hw >> 8) & 0xFF);                                        \
    __shared__ unsigned now;                                                                            \
    if (threadIdx.x == 0) {                                                                             \
        now = atomicAdd(&counters[cu], 1u) + 1u;                                                        \
        atomicMax(&resident_max[cu], now);                                                              \
    }                                                                                                   \
    lds[threadIdx.x] = threadIdx.x;                                                                     \
    __syncthreads();                                                                                    \
    unsigned v = lds[(threadIdx.x + 1) % blockDim.x] + now;                                             \
    asm volatile("" ::: __VA_ARGS__);                                                                     \
    for (unsigned i = 0; i < spin; ++i) v = v * 1664525u + 1013904223u;                                 \
    if (v == 0xDEADBEEFu) resident_max[0] = v;                                                          \
    __syncthreads();                                                                                    \
    if (threadIdx.x == 0) atomicSub(&counters[cu], 1u);
__global__ void __launch_bounds__(1024) k_small(unsigned* resident_max, unsigned* counters, unsigned spin) { BODY("memory") }
__global__ void __launch_bounds__(1024) k_v64(unsigned* resident_max, unsigned* counters, unsigned spin) { BODY("v63") }
__global__ void __launch_bounds__(1024) k_v68(unsigned* resident_max, unsigned* counters, unsigned spin) { BODY("v67") }
__global__ void __launch_bounds__(1024) k_v72(unsigned* resident_max, unsigned* counters, unsigned spin) { BODY("v71") }
__global__ void __launch_bounds__(1024) k_v80(unsigned* resident_max, unsigned* counters, unsigned spin) { BODY("v79") }
__global__ void __launch_bounds__(1024) k_s100(unsigned* resident_max, unsigned* counters, unsigned spin) { BODY("s99") }
__global__ void __launch_bounds__(1024) k_v68s100(unsigned* resident_max, unsigned* counters, unsigned spin) { BODY("v67", "s99") }
__global__ void __launch_bounds__(1024) k_v64s80(unsigned* resident_max, unsigned* counters, unsigned spin) { BODY("v63", "s79") }
__global__ void __launch_bounds__(1024) k_s82(unsigned* resident_max, unsigned* counters, unsigned spin) { BODY("s81") }
__global__ void __launch_bounds__(1024) k_s84(unsigned* resident_max, unsigned* counters, unsigned spin) { BODY("s83") }
__global__ void __launch_bounds__(1024) k_s86(unsigned* resident_max, unsigned* counters, unsigned spin) { BODY("s85") }
__global__ void __launch_bounds__(1024) k_s88(unsigned* resident_max, unsigned* counters, unsigned spin) { BODY("s87") }
__global__ void __launch_bounds__(1024) k_s90(unsigned* resident_max, unsigned* counters, unsigned spin) { BODY("s89") }
__global__ void __launch_bounds__(1024) k_s92(unsigned* resident_max, unsigned* counters, unsigned spin) { BODY("s91") }
__global__ void __launch_bounds__(1024) k_s94(unsigned* resident_max, unsigned* counters, unsigned spin) { BODY("s93") }
__global__ void __launch_bounds__(1024) k_s96(unsigned* resident_max, unsigned* counters, unsigned spin) { BODY("s95") }
__global__ void __launch_bounds__(1024) k_s98(unsigned* resident_max, unsigned* counters, unsigned spin) { BODY("s97") }
__global__ void __launch_bounds__(1024) k_s102(unsigned* resident_max, unsigned* counters, unsigned spin) { BODY("s101") }
typedef void (*kern_t)(unsigned*, unsigned*, unsigned);
int main() {
    unsigned *d_max, *d_cnt;
    hipMalloc(&d_max, 4096 * 4);
    hipMalloc(&d_cnt, 4096 * 4);
    struct { const char* name; kern_t k; } ks[] = {{"small", k_small}, {"v64", k_v64}, {"v68", k_v68}, {"v72", k_v72}, {"v80", k_v80}, {"s100", k_s100}, {"v68_s100", k_v68s100}, {"v64_s80", k_v64s80}, {"s82", k_s82}, {"s84", k_s84}, {"s86", k_s86}, {"s88", k_s88}, {"s90", k_s90}, {"s92", k_s92}, {"s94", k_s94}, {"s96", k_s96}, {"s98", k_s98}, {"s102", k_s102}};
    for (auto& e : ks)
        for (unsigned threads : {192u, 64u}) {
            int blocks = 0;
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, (const void*)e.k, (int)threads, 1280);
            hipFuncAttributes fa;
            hipFuncGetAttributes(&fa, (const void*)e.k);
            hipMemset(d_max, 0, 4096 * 4);
            hipMemset(d_cnt, 0, 4096 * 4);
            hipLaunchKernelGGL(e.k, dim3(65536), dim3(threads), 1280, 0, d_max, d_cnt, 20000u);
            hipDeviceSynchronize();
            static unsigned h[4096];
            hipMemcpy(h, d_max, sizeof(h), hipMemcpyDeviceToHost);
            unsigned mx = 0;
            for (unsigned i = 0; i < 4096; ++i) mx = std::max(mx, h[i]);
            printf("{\"kernel\": \"%s\", \"threads\": %u, \"numRegs\": %d, \"runtime_says_blocks_per_cu\": %d, \"max_resident_workgroups_seen\": %u, \"waves_per_cu\": %u}\n", e.name, threads, fa.numRegs, blocks, mx,
                   mx * ((threads + 63) / 64));
        }
    return 0;
}
