// Scratch: which store pattern writes a row-major [M, ld] fp32 matrix fastest?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// MFMA-epilogue pattern: wave writes rows r and r+4 (lanes>=32), 32 lanes x 4 B = 128 B per row, 16 regs
__global__ __launch_bounds__(256) void store_mfma(float *C, int64_t ldc, int64_t M, int ntiles) {
    const int t = threadIdx.x, w = t >> 6, l = t & 63, i = l & 31, h = l >> 5;
    const int64_t m0 = (int64_t)blockIdx.x * 128;
    for (int tile = blockIdx.y * ntiles; tile < (blockIdx.y + 1) * ntiles; ++tile)
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t m = m0 + 32 * w + (r & 3) + 8 * (r >> 2) + 4 * h;
                C[m * ldc + tile * 64 + nt * 32 + i] = (float)r;
            }
}
// same tile walk, but each lane writes a float4: 16 lanes cover 256 B of one row, 4 rows per instruction
__global__ __launch_bounds__(256) void store_vec4(float *C, int64_t ldc, int64_t M, int ntiles) {
    const int t = threadIdx.x, w = t >> 6, l = t & 63;
    const int64_t m0 = (int64_t)blockIdx.x * 128;
    for (int tile = blockIdx.y * ntiles; tile < (blockIdx.y + 1) * ntiles; ++tile)
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int64_t m = m0 + 32 * w + r * 4 + (l >> 4);
            f32x4 v = {1.f, 2.f, 3.f, (float)r};
            *reinterpret_cast<f32x4 *>(C + m * ldc + tile * 64 + (l & 15) * 4) = v;
        }
}
// block writes its 128 rows fully contiguous (row after row), float4 per lane
__global__ __launch_bounds__(256) void store_rows(float *C, int64_t ldc, int64_t M, int ncols) {
    const int t = threadIdx.x;
    const int64_t m0 = (int64_t)blockIdx.x * 128;
    for (int r = 0; r < 128; ++r)
        for (int c = t * 4; c < ncols; c += 1024) {
            f32x4 v = {1.f, 2.f, 3.f, (float)r};
            *reinterpret_cast<f32x4 *>(C + (m0 + r) * ldc + c) = v;
        }
}
template <typename F> float timeit(F f, int reps = 20) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize(); hipEventRecord(a);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / reps;
}
int main() {
    const int64_t M = 65536, ld = 1696; const int N = 1664;
    float *C; hipMalloc(&C, M * ld * 4);
    const double bytes = (double)M * N * 4;
    for (int ys : {1, 2, 26}) {
        float ms = timeit([&] { hipLaunchKernelGGL(store_mfma, dim3(M / 128, ys), dim3(256), 0, 0, C, ld, M, 26 / ys); });
        printf("mfma-epilogue 128B segs, ysplit=%2d : %.1f us %.2f TB/s\n", ys, ms * 1e3, bytes / ms / 1e9);
        ms = timeit([&] { hipLaunchKernelGGL(store_vec4, dim3(M / 128, ys), dim3(256), 0, 0, C, ld, M, 26 / ys); });
        printf("float4 256B segs,        ysplit=%2d : %.1f us %.2f TB/s\n", ys, ms * 1e3, bytes / ms / 1e9);
    }
    float ms = timeit([&] { hipLaunchKernelGGL(store_rows, dim3(M / 128), dim3(256), 0, 0, C, ld, M, N); });
    printf("row-contiguous float4              : %.1f us %.2f TB/s\n", ms * 1e3, bytes / ms / 1e9);
    ms = timeit([&] { hipMemsetAsync(C, 0, M * ld * 4, 0); });
    printf("hipMemset                          : %.1f us %.2f TB/s\n", ms * 1e3, (double)M * ld * 4 / ms / 1e9);
    return 0;
}
