// Scratch micro-benchmark: which global->LDS tile access pattern streams a row-major [M, ld] fp32 matrix fastest?
// build: hipcc --offload-arch=gfx950 -O3 profiles/microbench/probes/membench.hip -o gpurun_out/membench ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// block: 256 threads, tile BM rows x BK floats; thread -> (row = t / (BK/4) + j*RPP, col4 = t % (BK/4))
template <int BM, int BK, int P, bool SYNC2>
__global__ __launch_bounds__(256) void tile_stream(const float *__restrict__ A, int64_t lda, int64_t M, int K,
                                                   float *__restrict__ out) {
    constexpr int TPR = BK / 4;        // threads per row
    constexpr int RPP = 256 / TPR;     // rows per pass
    constexpr int NJ = BM / RPP;       // loads per thread per tile
    __shared__ f32x4 lds[BM * TPR];
    const int t = threadIdx.x;
    const int r0 = t / TPR, c4 = t % TPR;
    const int64_t m0 = (int64_t)blockIdx.x * BM;
    f32x4 reg[P][NJ];
    const int nk = K / BK;
    auto load = [&](int kt, f32x4 (&dst)[NJ]) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) dst[j] = *reinterpret_cast<const f32x4 *>(A + (m0 + r0 + j * RPP) * lda + kt * BK + c4 * 4);
    };
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int p = 0; p < P; ++p)
        if (p < nk) load(p, reg[p]);
    for (int kt = 0; kt < nk; kt += P) {
#pragma unroll
        for (int p = 0; p < P; ++p) {
            if (kt + p >= nk) break;
            __syncthreads();
#pragma unroll
            for (int j = 0; j < NJ; ++j) lds[(r0 + j * RPP) * TPR + c4] = reg[p][j];
            if (SYNC2) __syncthreads();
            if (kt + p + P < nk) load(kt + p + P, reg[p]);
            // consume: every thread reads 4 vectors back (stands in for fragment reads)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc += lds[((t + j * 64) % (BM * TPR))];
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}

__global__ __launch_bounds__(256) void linear_stream(const f32x4 *__restrict__ A, int64_t n4, float *out) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) acc += A[i];
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}

template <typename F>
float timeit(F f, int reps = 20) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    f();
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}

int main() {
    const int64_t M = 65536, ld = 1696;
    const int K = 1664;  // multiple of 128
    float *A, *out;
    hipMalloc(&A, M * ld * 4);
    hipMalloc(&out, 4);
    hipMemset(A, 0, M * ld * 4);
    const double bytes = (double)M * K * 4;
    {
        float ms = timeit([&] { hipLaunchKernelGGL(linear_stream, dim3(8192), dim3(256), 0, 0, (const f32x4 *)A, M * ld / 4, out); });
        printf("linear stream            : %.1f us  %.2f TB/s\n", ms * 1e3, (double)M * ld * 4 / ms / 1e9);
    }
#define RUN(BM, BK, P, S2)                                                                                          \
    {                                                                                                               \
        float ms = timeit([&] {                                                                                     \
            hipLaunchKernelGGL((tile_stream<BM, BK, P, S2>), dim3(M / BM), dim3(256), 0, 0, A, ld, M, K, out);      \
        });                                                                                                         \
        printf("BM=%3d BK=%3d P=%d sync2=%d : %.1f us  %.2f TB/s\n", BM, BK, P, (int)S2, ms * 1e3, bytes / ms / 1e9); \
    }
    RUN(128, 32, 1, true)
    RUN(128, 32, 2, true)
    RUN(128, 32, 3, true)
    RUN(64, 32, 1, true)
    RUN(64, 32, 2, true)
    RUN(64, 32, 4, true)
    RUN(128, 64, 1, true)
    RUN(128, 64, 2, true)
    RUN(64, 64, 1, true)
    RUN(64, 64, 2, true)
    RUN(32, 128, 1, true)
    RUN(32, 128, 2, true)
    RUN(64, 128, 1, true)
    RUN(64, 128, 2, true)
    RUN(32, 64, 2, true)
    RUN(32, 64, 4, true)
    RUN(128, 32, 2, false)
    RUN(64, 64, 2, false)
    return 0;
}
