// What the runtime says about resident workgroups per CU for the merge kernel's register budget at several LDS windows, and what the chip does:
// a kernel in which every workgroup records the SMID-like CU id (s_getreg HW_ID) and overlaps are counted by a per-CU resident counter.
// hipcc --offload-arch=gfx950 -O3 -o tools/micro/bin/occupancy_query tools/micro/occupancy_query.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <algorithm>
__global__ void __launch_bounds__(1024) occ_kernel(unsigned* resident_max, unsigned* counters, unsigned spin) {
    extern __shared__ unsigned lds[];
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    // HW_ID: cu_id bits [11:8], sh_id [12], se_id [15:13]; XCC_ID in a different register on gfx94x+: use (se, sh, cu) + xcc
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const unsigned cu = ((xcc & 0xF) << 8) | ((hw >> 8) & 0xFF);
    __shared__ unsigned now;
    if (threadIdx.x == 0) {
        now = atomicAdd(&counters[cu], 1u) + 1u;
        atomicMax(&resident_max[cu], now);
    }
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    unsigned v = lds[(threadIdx.x + 1) % blockDim.x] + now;
    for (unsigned i = 0; i < spin; ++i) v = v * 1664525u + 1013904223u;
    if (v == 0xDEADBEEFu) resident_max[0] = v;
    __syncthreads();
    if (threadIdx.x == 0) atomicSub(&counters[cu], 1u);
}
int main() {
    unsigned *d_max, *d_cnt;
    hipMalloc(&d_max, 4096 * 4);
    hipMalloc(&d_cnt, 4096 * 4);
    hipFuncSetAttribute((const void*)occ_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const unsigned shapes[][2] = {{192, 23392}, {192, 19728}, {192, 18176}, {192, 16384}, {192, 14848}, {192, 12288}, {64, 4096}, {64, 3584}, {128, 10752}, {512, 43616}};
    for (auto& sh : shapes) {
        int blocks = 0;
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, (const void*)occ_kernel, (int)sh[0], sh[1]);
        hipMemset(d_max, 0, 4096 * 4);
        hipMemset(d_cnt, 0, 4096 * 4);
        hipLaunchKernelGGL(occ_kernel, dim3(65536), dim3(sh[0]), sh[1], 0, d_max, d_cnt, 20000u);
        hipDeviceSynchronize();
        static unsigned h[4096];
        hipMemcpy(h, d_max, sizeof(h), hipMemcpyDeviceToHost);
        unsigned mx = 0, cus = 0;
        unsigned long long sum = 0;
        for (unsigned i = 0; i < 4096; ++i)
            if (h[i]) {
                mx = std::max(mx, h[i]);
                sum += h[i];
                ++cus;
            }
        printf("{\"threads\": %u, \"lds\": %u, \"runtime_says_blocks_per_cu\": %d, \"cus_seen\": %u, \"max_resident_seen\": %u, \"mean_of_per_cu_max\": %.2f}\n", sh[0], sh[1], blocks, cus, mx, cus ? (double)sum / cus : 0.0);
    }
    return 0;
}
