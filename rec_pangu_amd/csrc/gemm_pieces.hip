// nn.Linear forward on PRE-SPLIT bf16 operands (round 4; VERDICT r3 item 3 "pre-split bf16 residency"), gfx950.
//
// The split-bf16 kernels of gemm.hip read fp32 operands and cut every k-tile into bf16 pieces on the way into LDS:
// global load -> VGPR -> 2 conversions + a subtraction per piece -> ds_write.  On the matrix-core-bound (wide) layers
// that staging work is what a wave spends its issue slots on (gemm.hip, linear_fwd_bf16_wide_kernel: 220 instructions per
// k-tile, 24 of them MFMAs; the matrix pipe is busy 40 % of the time).  Here the operands ARRIVE as pieces — written
// once by whoever produces them (rp_pieces_pack, or a producer's epilogue) — and a k-tile goes from HBM/L2 to LDS by
// LDS-DMA (global_load_lds_dwordx4) without touching a VGPR: the inner loop is fragment reads and MFMAs.
//
// Piece layout ("interleaved pieces"): row r of an operand is a sequence of 128-byte k-tiles,
//     NP = 2 (bf16x3: hi + lo, products hi.lo + lo.hi + hi.hi):  [ 32 x bf16 hi | 32 x bf16 lo ]  = 32 values of K
//     NP = 1 (plain bf16):                                       [ 64 x bf16 ]                    = 64 values of K
// with K zero-padded to whole tiles, so one k-tile of one row is ONE 128-byte line whatever the mode, a wave-wide DMA
// (64 lanes x 16 B) brings 8 rows, and the kernel below has a single loader.  hi = RN_bf16(x), lo = RN_bf16(x - hi): the
// same pieces the in-kernel split makes, and the MFMA sequence per accumulator is the same as in
// linear_fwd_bf16_wide_kernel, so for NP = 2 the results are BIT-IDENTICAL to rp_linear_fwd in the bf16x3 mode
// (tests/test_hip_kernels.py::test_linear_fwd_pieces_*).
//
// LDS image: [row][128 B], the 16-byte chunk c of row r stored at chunk position c ^ ((r >> 1) & 7).  The DMA writes
// lane-linearly (wave-uniform base + lane * 16), so the permutation is applied to the SOURCE address (lane L of the
// 8-row group fetches chunk (L & 7) ^ ((row >> 1) & 7) of row L >> 3 — the same 128-byte line, still one coalesced
// request per row) and again on the fragment read.  A ds_read_b128 is served per 16 lanes = 16 consecutive rows at one
// logical chunk: ((r & 1) * 8 + (c ^ (r >> 1))) takes 16 distinct values — every lane its own 16-byte slot of the
// 256-byte bank row, conflict free.
//
// Tile 256 x 256, 8 waves as 2 (M) x 4 (N), each 128 x 64 = 4 x 2 MFMA tiles of 32 x 32 (128 accumulator registers);
// two LDS buffers of 64 KB; per k-tile and wave: 8 DMA instructions, 24 (NP = 2) / 24 (NP = 1) fragment reads,
// 48 / 32 MFMAs, one barrier.  The loads of tile t + 1 are issued before the matrix work of tile t and waited for after
// it.  Workgroup -> tile mapping is XCD-aware as in linear_fwd_bf16_wide_kernel.
#include "bfsplit.h"
#include <cstdlib>

#define PC_BM 256
#define PC_BN 256
#define PC_ROWB 128                        // bytes per row and k-tile
#define PC_OPB (256 * PC_ROWB)             // one operand tile: 32 KB
#define PC_STAGE (2 * PC_OPB)              // A + W

typedef __attribute__((address_space(1))) const void *pc_gptr;
typedef __attribute__((address_space(3))) void *pc_lptr;

template <int NP>
__global__ __launch_bounds__(512) void linear_fwd_pieces_kernel(const char *__restrict__ A, int64_t lda_b,
                                                                 const char *__restrict__ Wp, int64_t ldw_b,
                                                                 const float *__restrict__ bias, float *__restrict__ C,
                                                                 int64_t ldc, int64_t M, int N, int nkt, int act,
                                                                 const float *__restrict__ aux, int64_t ldaux,
                                                                 int mblocks, int nblocks, int dbg) {
    constexpr int NPROD = NP == 2 ? 3 : 1;
    constexpr int KS = NP == 2 ? 2 : 4;  // 16-deep MFMA steps per k-tile
    __shared__ __attribute__((aligned(1024))) char smem[2 * PC_STAGE];  // (ONE LDS object: see the guide's glds traps)
    const int t = threadIdx.x;
    int mb, nb;
    {
        const int id = blockIdx.x, xcd = id & 7, j = id >> 3;
        const int g = j / nblocks;
        nb = j - g * nblocks;
        mb = g * 8 + xcd;
        if (mb >= mblocks) return;  // (the grid is padded to whole groups of eight M blocks)
    }
    const int64_t m0 = (int64_t)mb * PC_BM;
    const int n0 = nb * PC_BN;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6), l = t & 63, i = l & 31, h = l >> 5;
    const int wm = w & 1, wn = w >> 1;

    // ---- loader: wave w brings rows [32 w, 32 w + 32) of both operand tiles, 8 rows per DMA -------------------------
    const char *asrc[4], *wsrc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = 32 * w + 8 * j + (l >> 3);
        const int c = (l & 7) ^ ((row >> 1) & 7);
        int64_t m = m0 + row;
        if (m >= M) m = M - 1;  // (clamped, never stored)
        int n = n0 + row;
        if (n >= N) n = N - 1;
        asrc[j] = A + m * lda_b + c * 16;
        wsrc[j] = Wp + (int64_t)n * ldw_b + c * 16;
    }
    auto issue = [&](int kt, int buf) {
        char *base = smem + buf * PC_STAGE + (32 * w) * PC_ROWB;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            __builtin_amdgcn_global_load_lds((pc_gptr)(asrc[j] + (int64_t)kt * PC_ROWB), (pc_lptr)(base + 8 * j * PC_ROWB), 16, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            __builtin_amdgcn_global_load_lds((pc_gptr)(wsrc[j] + (int64_t)kt * PC_ROWB),
                                             (pc_lptr)(base + PC_OPB + 8 * j * PC_ROWB), 16, 0, 0);
    };

    // ---- fragment addresses: row R = 128 wm + 32 mi + i (A), 64 wn + 32 ni + i (W); (R >> 1) & 7 = (i >> 1) & 7 ------
    const int sw = (i >> 1) & 7;
    const int arow = (128 * wm + i) * PC_ROWB, brow = PC_OPB + (64 * wn + i) * PC_ROWB;

    f32x16 acc[4][2];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    auto compute = [&](int buf) {
        const char *sb = smem + buf * PC_STAGE;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            bf16x8 a[4][NP], b[2][NP];
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                const int c = (NP == 2 ? q * 4 + ks * 2 + h : ks * 2 + h) ^ sw;
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    b[ni][q] = *reinterpret_cast<const bf16x8 *>(sb + brow + ni * 32 * PC_ROWB + c * 16);
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
                    a[mi][q] = *reinterpret_cast<const bf16x8 *>(sb + arow + mi * 32 * PC_ROWB + c * 16);
            }
#pragma unroll
            for (int pr = 0; pr < NPROD; ++pr)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi][BfProd<NPROD>::pa(pr)],
                                                                              b[ni][BfProd<NPROD>::pb(pr)], acc[mi][ni], 0, 0, 0);
        }
    };

    issue(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nkt && dbg != 1) issue(kt + 1, buf ^ 1);
        if (dbg != 2) compute(buf);
        // the next tile has landed (this wave's DMAs) and every wave is done reading this one
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    // C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const int n = n0 + 64 * wn + 32 * ni + i;
        const bool nok = n < N;
        const float bv = (bias != nullptr && nok) ? bias[n] : 0.f;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t m = m0 + 128 * wm + 32 * mi + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (!nok || m >= M) continue;
                float v = acc[mi][ni][r] + bv;
                if (act == RP_ACT_RELU)
                    v = v > 0.f ? v : 0.f;
                else if (act == RP_ACT_MASK)
                    v = (aux[m * ldaux + n] > 0.f) ? v : 0.f;
                C[m * ldc + n] = v;
            }
        }
    }
}

// fp32 [M, K] -> interleaved pieces [M, ldo]: one thread per 4 consecutive values of K (8 B per piece)
template <int NP>
__global__ __launch_bounds__(256) void pieces_pack_kernel(const float *__restrict__ in, int64_t ld, int64_t M, int K,
                                                          __bf16 *__restrict__ out, int64_t ldo, int chunks) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= M * chunks) return;
    const int64_t m = e / chunks;
    const int k = (int)(e - m * chunks) * 4;
    const bool vec = ((ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(in) & 15u) == 0);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    const float *p = in + m * ld + k;
    if (vec && k + 4 <= K) {
        v = *reinterpret_cast<const f32x4 *>(p);
    } else {
        if (k < K) v.x = p[0];
        if (k + 1 < K) v.y = p[1];
        if (k + 2 < K) v.z = p[2];
        if (k + 3 < K) v.w = p[3];
    }
    bf16x4 pc[NP];
    bf_split4<NP>(v, pc);
    if (NP == 2) {
        __bf16 *o = out + m * ldo + (k >> 5) * 64 + (k & 31);
        *reinterpret_cast<bf16x4 *>(o) = pc[0];
        *reinterpret_cast<bf16x4 *>(o + 32) = pc[NP - 1];
    } else {
        *reinterpret_cast<bf16x4 *>(out + m * ldo + k) = pc[0];
    }
}

extern "C" int64_t rp_pieces_ld(int K, int np) {
    if (K <= 0 || (np != 1 && np != 2)) return 0;
    return np == 2 ? rp_cdiv(K, 32) * 64 : rp_cdiv(K, 64) * 64;
}

extern "C" int rp_pieces_pack(const float *in, int64_t ld, int64_t M, int K, int np, void *out, int64_t ldo,
                              void *stream) {
    RP_REQUIRE(in && out && M > 0 && K > 0, "pieces_pack: null operand or empty shape");
    RP_REQUIRE(np == 1 || np == 2, "pieces_pack: np must be 1 (bf16) or 2 (bf16x3 pieces), got %d", np);
    RP_REQUIRE(ld >= K, "pieces_pack: ld %lld < K %d", (long long)ld, K);
    RP_REQUIRE(ldo >= rp_pieces_ld(K, np) && (ldo % 64) == 0 && rp_aligned16(out),
               "pieces_pack: ldo %lld must be a multiple of 64 and >= rp_pieces_ld(K, np) = %lld; out 16-byte aligned",
               (long long)ldo, (long long)rp_pieces_ld(K, np));
    hipStream_t s = (hipStream_t)stream;
    const int chunks = (int)(rp_pieces_ld(K, np) / np / 4);  // 4 values of K per thread, the zero padding included
    const int64_t nthr = M * chunks;
    if (np == 2)
        hipLaunchKernelGGL((pieces_pack_kernel<2>), dim3((unsigned)rp_cdiv(nthr, 256)), dim3(256), 0, s, in, ld, M, K,
                           (__bf16 *)out, ldo, chunks);
    else
        hipLaunchKernelGGL((pieces_pack_kernel<1>), dim3((unsigned)rp_cdiv(nthr, 256)), dim3(256), 0, s, in, ld, M, K,
                           (__bf16 *)out, ldo, chunks);
    RP_LAUNCH_CHECK("pieces_pack");
    return RP_OK;
}

extern "C" int rp_linear_fwd_pieces(const void *a, int64_t lda, const void *w, int64_t ldw, const float *bias, float *out,
                                    int64_t ldo, int64_t M, int N, int K, int np, int act, const float *aux,
                                    int64_t ldaux, void *stream) {
    RP_REQUIRE(a && w && out && M > 0 && N > 0 && K > 0, "linear_fwd_pieces: null operand or empty shape");
    RP_REQUIRE(np == 1 || np == 2, "linear_fwd_pieces: np must be 1 or 2, got %d", np);
    const int64_t need = rp_pieces_ld(K, np);
    RP_REQUIRE(lda >= need && ldw >= need && (lda % 8) == 0 && (ldw % 8) == 0 && rp_aligned16(a) && rp_aligned16(w),
               "linear_fwd_pieces: piece rows must hold rp_pieces_ld(K, np) = %lld elements (lda %lld, ldw %lld) and be 16-byte "
               "aligned", (long long)need, (long long)lda, (long long)ldw);
    RP_REQUIRE(act == RP_ACT_NONE || act == RP_ACT_RELU || (act == RP_ACT_MASK && aux != nullptr),
               "linear_fwd_pieces: unknown activation %d (or a mask without aux)", act);
    RP_REQUIRE(ldo >= N, "linear_fwd_pieces: ldo %lld < N %d", (long long)ldo, N);
    hipStream_t s = (hipStream_t)stream;
    static const int dbg = getenv("RP_PIECES_DEBUG") ? atoi(getenv("RP_PIECES_DEBUG")) : 0;  // 1: no loads, 2: no matrix work (timing probes)
    const int mblocks = (int)rp_cdiv(M, PC_BM), nblocks = (int)rp_cdiv(N, PC_BN);
    const int nkt = (int)(need / 64);
    const unsigned grid = (unsigned)(rp_cdiv(mblocks, 8) * 8 * nblocks);
    if (np == 2)
        hipLaunchKernelGGL((linear_fwd_pieces_kernel<2>), dim3(grid), dim3(512), 0, s, (const char *)a, lda * 2,
                           (const char *)w, ldw * 2, bias, out, ldo, M, N, nkt, act, aux, ldaux, mblocks, nblocks, dbg);
    else
        hipLaunchKernelGGL((linear_fwd_pieces_kernel<1>), dim3(grid), dim3(512), 0, s, (const char *)a, lda * 2,
                           (const char *)w, ldw * 2, bias, out, ldo, M, N, nkt, act, aux, ldaux, mblocks, nblocks, dbg);
    RP_LAUNCH_CHECK("linear_fwd_pieces");
    return RP_OK;
}
