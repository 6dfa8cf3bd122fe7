// Micro-benchmark: sustained issue cost of the VALU instructions the lazy-Adam replay is made of, on this box:
// v_fma_f32, v_pk_fma_f32 (two floats per lane), v_rcp_f32, and the replay's own step (packed over two rows / unpacked).
// Every lane runs 8 independent chains from registers, no memory; 8 waves per SIMD.  Reports SIMD clocks per
// wave-instruction at the nominal 2.4 GHz (the chip may clock lower under load: compare the rows with each other).
// build: hipcc --offload-arch=gfx950 -O3 profiles/microbench/valubench.hip -o gpurun_out/valubench ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void valu_loop(float *out, int iters, float seed) {
    float x[8], y[8];
    f32x2 px[4], py[4], pz[4];
    for (int e = 0; e < 8; ++e) {
        x[e] = seed + threadIdx.x * 1e-3f + e;
        y[e] = 1.0f + e * 1e-3f;
    }
    for (int e = 0; e < 4; ++e) {
        px[e] = f32x2{x[2 * e], x[2 * e + 1]};
        py[e] = f32x2{y[2 * e], y[2 * e + 1]};
        pz[e] = f32x2{0.5f + e, 0.25f + e};
    }
    const float c = 0.9995f, a = -1000.5f, b = -1e-5f, c1 = 0.1f;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {  // 8 x v_fma_f32
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = __builtin_fmaf(x[e], c, y[e]);
        } else if (MODE == 1) {  // 4 x v_pk_fma_f32
#pragma unroll
            for (int e = 0; e < 4; ++e) px[e] = __builtin_elementwise_fma(px[e], f32x2{c, c}, py[e]);
        } else if (MODE == 2) {  // 8 x v_rcp_f32 (+ 8 v_add to keep the values in range)
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = __builtin_amdgcn_rcpf(x[e]) + y[e];
        } else if (MODE == 3) {  // 8 x v_add_f32 (the baseline of mode 2)
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = x[e] + y[e];
        } else if (MODE == 4) {  // the replay step, unpacked, 4 elements: m, s, denom, rcp, p
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float m = x[e], s = y[e], p = x[4 + e];
                m = __builtin_fmaf(-m, c1, m);
                s = s * c;
                p = __builtin_fmaf(m, __builtin_amdgcn_rcpf(__builtin_fmaf(s, a, b)), p);
                x[e] = m;
                y[e] = s;
                x[4 + e] = p;
            }
        } else {  // the replay step, packed over two rows, 2 pairs = 4 elements
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                f32x2 m = px[e], s = py[e], p = pz[e];
                m = __builtin_elementwise_fma(-m, f32x2{c1, c1}, m);
                s = s * c;
                const f32x2 d = __builtin_elementwise_fma(s, f32x2{a, a}, f32x2{b, b});
                p = __builtin_elementwise_fma(m, f32x2{__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)}, p);
                px[e] = m;
                py[e] = s;
                pz[e] = p;
            }
        }
    }
    float s = 0.f;
    for (int e = 0; e < 8; ++e) s += x[e] + y[e];
    for (int e = 0; e < 4; ++e) s += px[e].x + px[e].y + py[e].x + pz[e].y;
    if (s == 123.456f) out[0] = s;
}

template <int MODE>
static void run(const char *name, double insts_per_iter, int iters) {
    float *out;
    hipMalloc(&out, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int grid = 256 * 8;  // 8 blocks of 4 waves per CU = 8 waves per SIMD
    hipLaunchKernelGGL(valu_loop<MODE>, dim3(grid), dim3(256), 0, 0, out, 100, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(valu_loop<MODE>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: 8 waves x iters x insts_per_iter wave-instructions in ms
    const double clk = ms * 1e-3 * 2.4e9 / (8.0 * iters * insts_per_iter);
    printf("%-44s %.3f ms  %.2f SIMD clocks per wave-instruction (at 2.4 GHz), %.2f per loop body\n", name, ms, clk,
           clk * insts_per_iter);
    hipFree(out);
}

int main() {
    const int it = 200000;
    run<0>("v_fma_f32 x8", 8, it);
    run<1>("v_pk_fma_f32 x4 (8 floats)", 4, it);
    run<3>("v_add_f32 x8", 8, it);
    run<2>("v_rcp_f32 x8 + v_add_f32 x8", 16, it);
    run<4>("replay step unpacked, 4 elements (20 insts)", 20, it);
    run<5>("replay step packed, 4 elements (8 pk + 4 rcp)", 12, it);
    return 0;
}
