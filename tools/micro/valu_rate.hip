// Integer VALU issue rate on MI355X: independent chains of full-rate integer ops, W waves per SIMD.  Prints instructions per cycle per CU.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
template <int KIND>
__global__ void k(uint32_t* out, uint32_t n, uint32_t seed) {
    uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 + 11, a5 = a0 + 13, a6 = a0 ^ 17, a7 = a0 ^ 19;
    for (uint32_t i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (KIND == 0) { a0 += a1; a1 += a2; a2 += a3; a3 += a4; a4 += a5; a5 += a6; a6 += a7; a7 += a0; }            // v_add_u32
            if (KIND == 1) { a0 = (a0 << 3) + a1; a1 = (a1 << 3) + a2; a2 = (a2 << 3) + a3; a3 = (a3 << 3) + a4; a4 = (a4 << 3) + a5; a5 = (a5 << 3) + a6; a6 = (a6 << 3) + a7; a7 = (a7 << 3) + a0; } // v_lshl_add_u32
            if (KIND == 2) { a0 = __popc(a0) + a1; a1 = __popc(a1) + a2; a2 = __popc(a2) + a3; a3 = __popc(a3) + a4; a4 = __popc(a4) + a5; a5 = __popc(a5) + a6; a6 = __popc(a6) + a7; a7 = __popc(a7) + a0; } // v_bcnt
            if (KIND == 3) { a0 = a0 * a1; a1 = a1 * a2; a2 = a2 * a3; a3 = a3 * a4; a4 = a4 * a5; a5 = a5 * a6; a6 = a6 * a7; a7 = a7 * a0; } // v_mul_lo_u32
            if (KIND == 4) { a0 = a0 & a1; a1 = a1 | a2; a2 = a2 ^ a3; a3 = a3 & a4; a4 = a4 | a5; a5 = a5 ^ a6; a6 = a6 & a7; a7 = a7 ^ a0; }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
template <int KIND>
void run(const char* name, int waves_per_simd) {
    uint32_t* d; hipMalloc(&d, 256 * 2048 * 4 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const uint32_t n = 4096;
    const int threads = 256, blocks = 256 * waves_per_simd;  // 256 threads = 4 waves = one per SIMD; blocks per CU = waves_per_simd
    k<KIND><<<blocks, threads>>>(d, 16, 1); hipDeviceSynchronize();
    hipEventRecord(e0); k<KIND><<<blocks, threads>>>(d, n, 1); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_wave = (double)n * 16 * 8;
    const double waves_per_cu = 4.0 * waves_per_simd;
    const double cycles = ms * 1e-3 * 2.4e9;
    printf("%-14s waves/SIMD %d: %.3f ms -> %.2f VALU instr / cycle / CU at 2.4 GHz (%.2f cycles per instr per SIMD)\n", name, waves_per_simd, ms, instr_per_wave * waves_per_cu / cycles, cycles / (instr_per_wave * waves_per_simd));
    hipFree(d);
}
int main() {
    for (int w : {1, 2, 4, 8}) run<0>("v_add_u32", w);
    for (int w : {4, 8}) run<1>("v_lshl_add_u32", w);
    for (int w : {4, 8}) run<2>("v_bcnt+add", w);
    for (int w : {4, 8}) run<3>("v_mul_lo_u32", w);
    for (int w : {4, 8}) run<4>("and/or/xor", w);
    return 0;
}
