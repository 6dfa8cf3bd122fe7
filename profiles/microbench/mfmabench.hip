// Scratch micro-benchmark: the sustained issue rate of v_mfma_f32_32x32x16_bf16 on this box ("re-verify
// the datasheet peak on the box").  Every wave issues independent MFMAs on 4 accumulators from registers, no memory.
// build: hipcc --offload-arch=gfx950 -O3 profiles/microbench/probes/mfmabench.hip -o gpurun_out/mfmabench ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void mfma_loop(float *out, int iters, float seed) {
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) {
        a[e] = (__bf16)(seed + threadIdx.x * 0.001f + e);
        b[e] = (__bf16)(seed - e * 0.5f);
    }
    f32x16 acc[4];
    for (int q = 0; q < 4; ++q)
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[q], 0, 0, 0);
    }
    float s = 0.f;
    for (int q = 0; q < 4; ++q)
        for (int r = 0; r < 16; ++r) s += acc[q][r];
    if (s == 123.456f) out[0] = s;
}

template <int WAVES>
static void run(int blocks_per_cu, int iters) {
    float *out;
    hipMalloc(&out, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int grid = 256 * blocks_per_cu;
    hipLaunchKernelGGL(mfma_loop<WAVES>, dim3(grid), dim3(64 * WAVES), 0, 0, out, 100, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(mfma_loop<WAVES>, dim3(grid), dim3(64 * WAVES), 0, 0, out, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)grid * WAVES * iters * 16 * 2.0 * 32 * 32 * 16;
    printf("waves/block %d, blocks/CU %d (waves/SIMD %.1f): %.3f ms, %.1f TFLOP/s dense bf16\n", WAVES, blocks_per_cu,
           WAVES * blocks_per_cu / 4.0, ms, flops / ms / 1e9);
    hipFree(out);
}

int main() {
    run<4>(1, 20000);
    run<4>(2, 20000);
    run<8>(1, 20000);
    run<4>(1, 200000);
    return 0;
}
