// Micro-benchmark (round 6): the ceiling the three dominant launches of the DeepFM step are actually
// against — RANDOM ROW GATHERS.  N random rows of ROWB bytes (256 = one fp32 D = 64 table row, 128 = its bf16 image, 512 = a
// [dH | S] record) from
//   * an 8.6 GB arena             (HBM-resident: the table gathers of the forward / catch-up / first-layer backward)
//   * a 33.5 MB buffer            (L2-miss, Infinity-Cache-hit: the dH / S rows of the first-layer backward)
//   * a 2 MB buffer               (L2-resident)
// swept over row loads in flight per lane (INF) and waves per CU, 16 lanes x 16 bytes per 256-byte row (the layout every
// kernel of the library uses).  Also: gather + scatter (read a random row, write a random row: the sample-major backward's
// singleton-row pattern), and "7 rows from 4 arenas" (the lazy Adam catch-up: read p, m, s, g of a row, write p, m, s).
// Reports rows/s and TB/s of row bytes.  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 profiles/microbench/rowgather.hip -o gpurun_out/rowgather && gpurun_out/rowgather
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) {                                                            \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));      \
            exit(1);                                                                       \
        }                                                                                  \
    } while (0)

// LPR lanes x 16 bytes per row (ROWB = 16 * LPR); a wave covers 64 / LPR rows per load instruction; INF independent row
// loads in flight per lane; WPS = waves per SIMD asked of the register allocator (grid is sized to fill them)
template <int LPR, int INF, int WPS>
__global__ __launch_bounds__(256, WPS) void gather_rows(const f32x4 *__restrict__ src, const int32_t *__restrict__ idx,
                                                        int64_t n, float *__restrict__ out) {
    constexpr int RPW = 64 / LPR;           // rows per wave-instruction
    constexpr int RPB = 4 * RPW * INF;      // rows per workgroup iteration
    const int t = threadIdx.x, lane = t & (LPR - 1), rg = t / LPR;  // rg: row slot of the workgroup (0 .. 4 RPW - 1)
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int64_t r0 = (int64_t)blockIdx.x * RPB; r0 < n; r0 += (int64_t)gridDim.x * RPB) {
        int32_t k[INF];
#pragma unroll
        for (int j = 0; j < INF; ++j) {
            const int64_t q = r0 + (int64_t)j * (4 * RPW) + rg;
            k[j] = idx[q < n ? q : n - 1];
        }
        f32x4 v[INF];
#pragma unroll
        for (int j = 0; j < INF; ++j) v[j] = src[(int64_t)k[j] * LPR + lane];
#pragma unroll
        for (int j = 0; j < INF; ++j) acc += v[j];
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}

// read a random row, write it to another random row (one writer per destination row: idx2 is a permutation)
template <int INF, int WPS>
__global__ __launch_bounds__(256, WPS) void gather_scatter_rows(const f32x4 *__restrict__ src, f32x4 *__restrict__ dst,
                                                                const int32_t *__restrict__ idx, const int32_t *__restrict__ idx2,
                                                                int64_t n) {
    constexpr int LPR = 16, RPW = 4, RPB = 4 * RPW * INF;
    const int t = threadIdx.x, lane = t & 15, rg = t >> 4;
    for (int64_t r0 = (int64_t)blockIdx.x * RPB; r0 < n; r0 += (int64_t)gridDim.x * RPB) {
        int32_t k[INF], k2[INF];
#pragma unroll
        for (int j = 0; j < INF; ++j) {
            const int64_t q = r0 + (int64_t)j * 16 + rg;
            k[j] = idx[q < n ? q : n - 1];
            k2[j] = idx2[q < n ? q : n - 1];
        }
        f32x4 v[INF];
#pragma unroll
        for (int j = 0; j < INF; ++j) v[j] = src[(int64_t)k[j] * LPR + lane];
#pragma unroll
        for (int j = 0; j < INF; ++j) dst[(int64_t)k2[j] * LPR + lane] = v[j] * 1.0001f;
    }
}

// the catch-up's pattern: p, m, s, g of a random row from four arenas (or ONE 1 KB record when `rec`), p, m, s written back
template <int INF, int WPS, bool REC>
__global__ __launch_bounds__(256, WPS) void catchup_rows(f32x4 *__restrict__ p, f32x4 *__restrict__ m, f32x4 *__restrict__ s,
                                                         const f32x4 *__restrict__ g, const int32_t *__restrict__ idx, int64_t n) {
    constexpr int RPB = 16 * INF;
    const int t = threadIdx.x, lane = t & 15, rg = t >> 4;
    for (int64_t r0 = (int64_t)blockIdx.x * RPB; r0 < n; r0 += (int64_t)gridDim.x * RPB) {
        int64_t o[INF][4];
#pragma unroll
        for (int j = 0; j < INF; ++j) {
            const int64_t q = r0 + (int64_t)j * 16 + rg;
            const int64_t k = idx[q < n ? q : n - 1];
            if (REC) {
                o[j][0] = k * 64 + lane, o[j][1] = o[j][0] + 16, o[j][2] = o[j][0] + 32, o[j][3] = o[j][0] + 48;
            } else {
                o[j][0] = o[j][1] = o[j][2] = o[j][3] = k * 16 + lane;
            }
        }
        f32x4 vp[INF], vm[INF], vs[INF], vg[INF];
#pragma unroll
        for (int j = 0; j < INF; ++j) {
            vp[j] = (REC ? p : p)[o[j][0]];
            vm[j] = (REC ? p : m)[o[j][1]];
            vs[j] = (REC ? p : s)[o[j][2]];
            vg[j] = (REC ? (const f32x4 *)p : g)[o[j][3]];
        }
#pragma unroll
        for (int j = 0; j < INF; ++j) {
            const f32x4 nm = 0.9f * vm[j] + 0.1f * vg[j];
            const f32x4 ns = 0.999f * vs[j] + 0.001f * vg[j] * vg[j];
            (REC ? p : p)[o[j][0]] = vp[j] - 0.001f * nm;
            (REC ? p : m)[o[j][1]] = nm;
            (REC ? p : s)[o[j][2]] = ns;
        }
    }
}

template <typename F>
static float timeit(F f, int reps = 10) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    f();
    f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    CK(hipEventDestroy(a));
    CK(hipEventDestroy(b));
    return ms / reps;
}

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rng() {
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return rng_state;
}

int main(int argc, char **argv) {
    const int64_t N = 1 << 21;  // rows gathered per launch (~ the 1.7 M (sample, field) pairs of a Criteo batch)
    const int64_t arena_rows = argc > 1 ? atoll(argv[1]) : 33762603;
    float *out;
    CK(hipMalloc(&out, 256));
    int32_t *idx, *idx2;
    CK(hipMalloc(&idx, N * 4));
    CK(hipMalloc(&idx2, N * 4));
    std::vector<int32_t> h(N), h2(N);
    char *arena;
    const size_t arena_bytes = (size_t)arena_rows * 256;
    CK(hipMalloc(&arena, arena_bytes));
    CK(hipMemset(arena, 0, arena_bytes));
    int cus = 256;
    {
        hipDeviceProp_t pr;
        CK(hipGetDeviceProperties(&pr, 0));
        cus = pr.multiProcessorCount;
        printf("device: %s, %d CUs; %lld rows per launch; arena %lld rows x 256 B = %.2f GB\n", pr.name, cus, (long long)N,
               (long long)arena_rows, arena_bytes / 1e9);
    }
    struct Src {
        const char *name;
        int64_t bytes;
    } srcs[] = {{"8.6GB arena (HBM)", (int64_t)arena_bytes}, {"33.5MB buffer (Infinity Cache)", 33554432}, {"2MB buffer (L2)", 2097152}};
    printf("\n== random row gathers: rows/s and TB/s of row bytes ==\n");
    printf("%-32s %5s %4s %4s %10s %8s %8s\n", "source", "rowB", "inf", "w/S", "us", "Grow/s", "TB/s");
    for (const Src &sc : srcs) {
#define GATHER(LPR, INF, WPS)                                                                                             \
    {                                                                                                                     \
        const int rowb = 16 * LPR;                                                                                        \
        const int64_t rows = sc.bytes / rowb;                                                                             \
        for (int64_t i = 0; i < N; ++i) h[i] = (int32_t)(rng() % (uint64_t)rows);                                          \
        CK(hipMemcpy(idx, h.data(), N * 4, hipMemcpyHostToDevice));                                                       \
        const unsigned grid = (unsigned)(cus * WPS);                                                                      \
        const float ms = timeit([&] {                                                                                     \
            hipLaunchKernelGGL((gather_rows<LPR, INF, WPS>), dim3(grid), dim3(256), 0, 0, (const f32x4 *)arena, idx, N, out); \
        });                                                                                                               \
        printf("%-32s %5d %4d %4d %10.1f %8.2f %8.2f\n", sc.name, rowb, INF, WPS, ms * 1e3, N / ms / 1e6,                 \
               (double)N * rowb / ms / 1e9);                                                                              \
    }
        GATHER(16, 1, 8)
        GATHER(16, 2, 8)
        GATHER(16, 4, 8)
        GATHER(16, 8, 8)
        GATHER(16, 8, 4)
        GATHER(16, 16, 4)
        GATHER(16, 8, 2)
        GATHER(16, 16, 2)
        GATHER(16, 24, 2)
        GATHER(16, 8, 1)
        GATHER(16, 24, 1)
        GATHER(8, 4, 8)
        GATHER(8, 8, 8)
        GATHER(8, 16, 4)
        GATHER(8, 16, 2)
        GATHER(32, 2, 8)
        GATHER(32, 4, 8)
        GATHER(32, 8, 4)
        GATHER(32, 8, 2)
        GATHER(32, 12, 2)
    }
    printf("\n== gather + scatter of 256-B rows inside the 8.6 GB arena (TB/s counts read + write) ==\n");
    {
        // destination rows: distinct (a stride walk of the upper half of the arena), source rows random
        const int64_t half = arena_rows / 2;
        for (int64_t i = 0; i < N; ++i) {
            h[i] = (int32_t)(rng() % (uint64_t)half);
            h2[i] = (int32_t)(half + (i * 7919) % half);
        }
        CK(hipMemcpy(idx, h.data(), N * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(idx2, h2.data(), N * 4, hipMemcpyHostToDevice));
#define GS(INF, WPS)                                                                                                      \
    {                                                                                                                     \
        const float ms = timeit([&] {                                                                                     \
            hipLaunchKernelGGL((gather_scatter_rows<INF, WPS>), dim3((unsigned)(cus * WPS)), dim3(256), 0, 0,              \
                               (const f32x4 *)arena, (f32x4 *)arena, idx, idx2, N);                                       \
        });                                                                                                               \
        printf("inf %2d w/S %d: %8.1f us  %6.2f Grow/s  %5.2f TB/s\n", INF, WPS, ms * 1e3, N / ms / 1e6, 2.0 * N * 256 / ms / 1e9); \
    }
        GS(4, 8)
        GS(8, 8)
        GS(8, 4)
        GS(16, 4)
        GS(16, 2)
    }
    printf("\n== catch-up pattern: 4 row reads + 3 row writes per row (TB/s counts 7 x 256 B) ==\n");
    {
        const int64_t quarter = arena_rows / 4;
        const int64_t NU = 530843;  // unique rows of a Criteo batch
        for (int64_t i = 0; i < NU; ++i) h[i] = (int32_t)((i * 15485863ll) % quarter);  // distinct rows
        CK(hipMemcpy(idx, h.data(), NU * 4, hipMemcpyHostToDevice));
        f32x4 *p = (f32x4 *)arena, *m = p + quarter * 16, *s = m + quarter * 16, *g = s + quarter * 16;
#define CU_(INF, WPS, REC)                                                                                                \
    {                                                                                                                     \
        const float ms = timeit([&] {                                                                                     \
            hipLaunchKernelGGL((catchup_rows<INF, WPS, REC>), dim3((unsigned)(cus * WPS)), dim3(256), 0, 0, p, m, s, g, idx, NU); \
        });                                                                                                               \
        printf("%-22s inf %d w/S %d: %8.1f us  %6.2f Grow/s  %5.2f TB/s\n", REC ? "one 1 KB record" : "four arenas", INF, WPS, \
               ms * 1e3, NU / ms / 1e6, 7.0 * NU * 256 / ms / 1e9);                                                       \
    }
        CU_(1, 8, false)
        CU_(2, 8, false)
        CU_(4, 4, false)
        CU_(1, 8, true)
        CU_(2, 8, true)
        CU_(4, 4, true)
    }
    return 0;
}
