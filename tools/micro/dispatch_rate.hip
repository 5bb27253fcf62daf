// How fast does the chip start and retire workgroups of the merge kernel's shape (192 threads, 19 728 B of dynamic LDS) that do next to nothing?
// hipcc --offload-arch=gfx950 -O3 -o tools/micro/bin/dispatch_rate tools/micro/dispatch_rate.hip   (the binary travels to the GPU box with the snapshot; tools/micro/bin is git-ignored)
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void __launch_bounds__(1024) empty_wg(unsigned* out, unsigned spin) {
    extern __shared__ unsigned lds[];
    lds[threadIdx.x] = threadIdx.x + blockIdx.x;
    __syncthreads();
    unsigned v = lds[(threadIdx.x + 1) % blockDim.x];
    for (unsigned i = 0; i < spin; ++i) v = v * 1664525u + 1013904223u;  // spin: a workgroup that lives for a while
    if (v == 0xDEADBEEFu) out[0] = v;
}
int main() {
    unsigned* d;
    hipMalloc(&d, 4);
    hipFuncSetAttribute((const void*)empty_wg, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    const unsigned grids[] = {196608, 524288};
    const unsigned shapes[][2] = {{192, 19728}, {64, 4096}, {512, 43616}};
    for (auto& sh : shapes)
        for (unsigned g : grids)
            for (unsigned spin : {0u, 2000u, 20000u}) {
                hipLaunchKernelGGL(empty_wg, dim3(g), dim3(sh[0]), sh[1], 0, d, spin);
                hipDeviceSynchronize();
                hipEventRecord(a);
                for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(empty_wg, dim3(g), dim3(sh[0]), sh[1], 0, d, spin);
                hipEventRecord(b);
                hipEventSynchronize(b);
                float ms;
                hipEventElapsedTime(&ms, a, b);
                printf("{\"threads\": %u, \"lds\": %u, \"grid\": %u, \"spin\": %u, \"ms_per_launch\": %.4f, \"ns_per_workgroup\": %.2f}\n", sh[0], sh[1], g, spin, ms / 5, ms / 5 * 1e6 / g);
            }
    return 0;
}
