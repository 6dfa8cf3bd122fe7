// nn.Linear forward / dgrad / wgrad on the exact-fp32 matrix core (v_mfma_f32_32x32x2_f32), gfx950.
//
// Why the f32 MFMA: the parity gate is fp32 (logits/loss within 1e-4 of the reference's ATen fp32)
// and this instruction is bit-for-bit an fp32 FMA chain at 157 TFLOP/s, 2.4x a VALU GEMM.  With the
// reference's default MLP [64,64,64] every one of these GEMMs is HBM-bound anyway (69 flop per
// input byte): the first layer streams the [B, 1677] activation once and that is
// the roofline that matters.
//
// linear_fwd  (NT):  out[M,N] = act(A[M,K] . W[N,K]^T + bias)        tile 128x64x32, 4 waves
//   wave w owns rows 32w..32w+31 and both 32-column halves (2 accumulators = 32 VGPRs).
//   LDS rows are padded to 36 floats: the ds_read_b128 fragment reads (lane -> row lane&31,
//   k-quad lane>>5) then touch 16 distinct 16-B slots per 16-lane group (conflict free), and the
//   ds_write_b128 staging writes cover a full row per 8 lanes.  One b128 read per operand feeds
//   4 MFMAs: lanes <32 carry k0..k0+3, lanes >=32 carry k0+4..k0+7, MFMA e contracts
//   {k0+e, k0+4+e}; the contraction is a sum so the k pairing is free to choose.
//   Global->register prefetch of tile t+1 overlaps the MFMAs of tile t (single LDS buffer).
// The dgrad dX = dY . W reuses this kernel with the transposed weight (rp_transpose) as "W".
//
// linear_wgrad (TN): dW[N,K] = dY[M,N]^T . X[M,K], contraction over the batch.  Both operands are
//   row-major with the contraction index outermost, which is exactly the f32 MFMA's operand
//   shape (lane -> column lane&31 of row lane>>5): fragments are conflict-free ds_read_b32 of
//   consecutive floats.  The batch is split over gridDim.z; partials go to a workspace and a
//   second kernel sums them in fixed order (deterministic, no float atomics).  Column sums of dY
//   (the bias gradient) ride along in the blocks of the first k tile.
#include "common.h"
#include <cstdlib>

#define BM 128
#define BN 64
#define BK 32
#define LDS_LD 36

__device__ __forceinline__ f32x4 load4_guard(const float *p, int valid, bool vec) {
    // valid = number of readable floats at p (may be <= 0 or >= 4)
    if (vec && valid >= 4) return *reinterpret_cast<const f32x4 *>(p);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (valid > 0) v.x = p[0];
    if (valid > 1) v.y = p[1];
    if (valid > 2) v.z = p[2];
    if (valid > 3) v.w = p[3];
    return v;
}

template <bool VEC_A, bool VEC_W>
__global__ __launch_bounds__(256) void linear_fwd_kernel(const float *__restrict__ A, int64_t lda,
                                                         const float *__restrict__ W, int64_t ldw,
                                                         const float *__restrict__ bias, float *__restrict__ C,
                                                         int64_t ldc, int64_t M, int N, int K, int act,
                                                         const float *__restrict__ aux, int64_t ldaux) {
    __shared__ __attribute__((aligned(16))) float smem[(BM + BN) * LDS_LD];
    float *As = smem;
    float *Ws = smem + BM * LDS_LD;
    const int t = threadIdx.x;
    const int64_t m0 = (int64_t)blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;
    const int lr = t >> 3, lc = (t & 7) * 4;
    const int w = t >> 6, l = t & 63, i = l & 31, h = l >> 5;
    const bool two = (n0 + 32 < N);

    f32x4 ra[4], rw[2];
    auto load_tile = [&](int k0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t m = m0 + lr + 32 * j;
            ra[j] = (m < M) ? load4_guard(A + m * lda + k0 + lc, K - (k0 + lc), VEC_A) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + lr + 32 * j;
            rw[j] = (n < N) ? load4_guard(W + (int64_t)n * ldw + k0 + lc, K - (k0 + lc), VEC_W)
                            : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };

    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        acc0[r] = 0.f;
        acc1[r] = 0.f;
    }
    const int nk = (K + BK - 1) / BK;
    load_tile(0);
    for (int kt = 0; kt < nk; ++kt) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4 *>(&As[(lr + 32 * j) * LDS_LD + lc]) = ra[j];
#pragma unroll
        for (int j = 0; j < 2; ++j) *reinterpret_cast<f32x4 *>(&Ws[(lr + 32 * j) * LDS_LD + lc]) = rw[j];
        __syncthreads();
        if (kt + 1 < nk) load_tile((kt + 1) * BK);
#pragma unroll
        for (int kq = 0; kq < 4; ++kq) {
            const f32x4 a4 = *reinterpret_cast<const f32x4 *>(&As[(32 * w + i) * LDS_LD + kq * 8 + 4 * h]);
            const f32x4 b0 = *reinterpret_cast<const f32x4 *>(&Ws[i * LDS_LD + kq * 8 + 4 * h]);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[e], b0[e], acc0, 0, 0, 0);
            if (two) {
                const f32x4 b1 = *reinterpret_cast<const f32x4 *>(&Ws[(32 + i) * LDS_LD + kq * 8 + 4 * h]);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[e], b1[e], acc1, 0, 0, 0);
            }
        }
    }
    // C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int n = n0 + nt * 32 + i;
        if (n >= N) continue;
        const float bv = (bias != nullptr) ? bias[n] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t m = m0 + 32 * w + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (m >= M) continue;
            float v = (nt == 0 ? acc0[r] : acc1[r]) + bv;
            if (act == RP_ACT_MASK)
                v = (aux[m * ldaux + n] > 0.f) ? v : 0.f;
            else
                v = rp_act_apply(act, v);
            C[m * ldc + n] = v;
        }
    }
}

#define TN_BN 64
#define TN_BK 128
#define TN_BM 32

template <bool VEC_Y, bool VEC_X>
__global__ __launch_bounds__(256) void linear_wgrad_partial_kernel(const float *__restrict__ dY, int64_t lddy,
                                                                   const float *__restrict__ X, int64_t ldx,
                                                                   float *__restrict__ P, float *__restrict__ Pb,
                                                                   int64_t M, int N, int K, int64_t rows_per_split) {
    __shared__ __attribute__((aligned(16))) float smem[TN_BM * (TN_BN + TN_BK)];
    float *Ys = smem;                   // [32][64]
    float *Xs = smem + TN_BM * TN_BN;   // [32][128]
    const int t = threadIdx.x;
    const int k0 = blockIdx.x * TN_BK;
    const int n0 = blockIdx.y * TN_BN;
    const int64_t mbeg = (int64_t)blockIdx.z * rows_per_split;
    int64_t mend = mbeg + rows_per_split;
    if (mend > M) mend = M;
    const int w = t >> 6, l = t & 63, i = l & 31, h = l >> 5;
    const int nt = (w & 1) * 32, kt = (w >> 1) * 64;
    const int yr = t >> 4, yc = (t & 15) * 4;  // Y tile: rows yr, yr+16
    const int xr = t >> 5, xc = (t & 31) * 4;  // X tile: rows xr + 8j

    f32x4 ry[2], rx[4];
    auto load_tile = [&](int64_t mm) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int64_t m = mm + yr + 16 * j;
            ry[j] = (m < mend) ? load4_guard(dY + m * lddy + n0 + yc, N - (n0 + yc), VEC_Y) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t m = mm + xr + 8 * j;
            rx[j] = (m < mend) ? load4_guard(X + m * ldx + k0 + xc, K - (k0 + xc), VEC_X) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };

    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        acc0[r] = 0.f;
        acc1[r] = 0.f;
    }
    float bsum = 0.f;
    const bool do_bias = (Pb != nullptr) && (blockIdx.x == 0) && (t < TN_BN);
    if (mbeg < mend) load_tile(mbeg);
    for (int64_t mm = mbeg; mm < mend; mm += TN_BM) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 2; ++j) *reinterpret_cast<f32x4 *>(&Ys[(yr + 16 * j) * TN_BN + yc]) = ry[j];
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4 *>(&Xs[(xr + 8 * j) * TN_BK + xc]) = rx[j];
        __syncthreads();
        if (mm + TN_BM < mend) load_tile(mm + TN_BM);
#pragma unroll
        for (int s = 0; s < TN_BM / 2; ++s) {
            const int mr = 2 * s + h;
            const float a = Ys[mr * TN_BN + nt + i];
            const float b0 = Xs[mr * TN_BK + kt + i];
            const float b1 = Xs[mr * TN_BK + kt + 32 + i];
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc1, 0, 0, 0);
        }
        if (do_bias) {
#pragma unroll
            for (int r = 0; r < TN_BM; ++r) bsum += Ys[r * TN_BN + t];
        }
    }
    float *Pz = P + (int64_t)blockIdx.z * N * K;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int k = k0 + kt + kk * 32 + i;
        if (k >= K) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = n0 + nt + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (n < N) Pz[(int64_t)n * K + k] = (kk == 0 ? acc0[r] : acc1[r]);
        }
    }
    if (do_bias && n0 + t < N) Pb[(int64_t)blockIdx.z * N + n0 + t] = bsum;
}

// ================================================================================================
// Split-bf16 GEMMs on the CDNA4 bf16 matrix core (v_mfma_f32_32x32x16_bf16, 16x the rate of the f32-input MFMA).
//
// An fp32 operand x is split on the fly, while it is staged into LDS, into bf16 pieces
//     x = hi + mid + lo (+ r),   hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid)      (the subtractions are exact)
// and a product a*b is evaluated as a sum of bf16 x bf16 MFMA products accumulated in fp32:
//     NPROD = 6 : hi*hi + hi*mid + mid*hi + mid*mid + hi*lo + lo*hi      relative error ~2^-23  (fp32-faithful; default)
//     NPROD = 3 : hi*hi + hi*mid + mid*hi                                  relative error ~2^-16
//     NPROD = 1 : hi*hi                                                    plain bf16 inputs, ~2^-9
// Even at NPROD = 6 the matrix-core work is 6/16 of the f32 MFMA's, which turns the first DeepFM layer
// ([B,1677] x [1677,64]: 32 flop per streamed byte, above the f32-MFMA ridge of ~20) back into an HBM-bound kernel.
// Fragment layout: lane l carries row/col (l & 31) and the 8 consecutive k = 8*(l >> 5) .. +7 of a 16-deep k step
// for A and B alike (the contraction is a sum, so any k pairing shared by both operands is valid).
// LDS rows hold 32 bf16 + 8 pad (80 B = 5 x 16 B, odd) so the ds_read_b128 fragment reads are conflict-free.
// ================================================================================================
#include "bfsplit.h"

// NT:  out[M,N] = act(A[M,K] . W[N,K]^T + bias).  Block tile (32*WM) x 64 x 32, 4 waves:
//   WM = 4: wave w owns rows 32w.. and both 32-column halves;   WM = 2: waves are 2 (rows) x 2 (column halves) — used
//   when the 128-row grid would leave the chip under-filled (skinny N).
// Wave tiling is a template: WM x WN waves (WM*WN = 4), each owning MI x NI MFMA tiles of 32 x 32:
//   <4,1,1,2>  128 x 64   rows split over the waves, both column halves per wave           (skinny N, large grid)
//   <2,2,1,1>   64 x 64   used when the 128-row grid would leave the chip under-filled
//   <4,2,1,2>  128 x 128  EIGHT waves of 32 x 64 (four per SIMD: hipcc schedules a stage as "convert + LDS, then the MFMAs
//                         back to back", so the overlap has to come from other waves; 8.3 -> 7.9 ms on the wide MLP
//                         against <2,2,2,2>, four waves of 64 x 64) — the wide
//                         (compute-bound) layers, where LDS fragment traffic per MFMA is what limits the matrix core
template <int NPROD, int WM, int WN, int MI, int NI, int PF, bool VEC_A, bool VEC_W>
__global__ __launch_bounds__(64 * WM * WN) void linear_fwd_bf16_kernel(const float *__restrict__ A, int64_t lda,
                                                              const float *__restrict__ W, int64_t ldw,
                                                              const float *__restrict__ bias, float *__restrict__ C,
                                                              int64_t ldc, int64_t M, int N, int K, int act,
                                                              const float *__restrict__ aux, int64_t ldaux) {
    static_assert(WM * WN == 4 || WM * WN == 8, "4 or 8 waves per workgroup");
    constexpr int RPP = 8 * WM * WN;  // rows staged per pass: 8 threads (32 floats) per row
    constexpr int NP = BfProd<NPROD>::NP;
    constexpr int BMT = 32 * MI * WM;
    constexpr int BNT = 32 * NI * WN;
    constexpr int RA = BMT / RPP;  // A rows staged per thread
    constexpr int RW = BNT / RPP;  // W rows staged per thread
    __shared__ __attribute__((aligned(16))) __bf16 As[NP][BMT][BF_LD];
    __shared__ __attribute__((aligned(16))) __bf16 Ws[NP][BNT][BF_LD];
    const int t = threadIdx.x;
    const int64_t m0 = (int64_t)blockIdx.x * BMT;
    const int n0 = blockIdx.y * BNT;
    const int lr = t >> 3, lc = (t & 7) * 4;
    const int w = t >> 6, l = t & 63, i = l & 31, h = l >> 5;
    const int wm = w % WM, wn = w / WM;
    const int arow0 = 32 * MI * wm, bcol0 = 32 * NI * wn;

    // PF tiles of global loads are kept in flight per workgroup (a register ring): one tile per workgroup is a few
    // KB, and at 2-4 resident workgroups per CU that is far too little to cover HBM latency at full bandwidth.
    // The ring only works if the compiler can COUNT the loads in flight: gfx950 has one in-order counter (vmcnt) for
    // all global loads, and a load behind a branch (a bounds guard) makes every wait a vmcnt(0) — i.e. each tile
    // would wait for the loads issued for the tile PF steps ahead, exposing the full memory latency per tile.  So an
    // interior workgroup (all rows / columns in range, 16-byte aligned operands) runs its full k tiles through a
    // straight-line loop with unguarded loads; the guarded loop below finishes the last tiles and serves the edges.
    f32x4 ra[PF][RA], rw[PF][RW];
    auto load_tile = [&](int k0, f32x4 (&da)[RA], f32x4 (&dw)[RW]) {
#pragma unroll
        for (int j = 0; j < RA; ++j) {
            const int64_t m = m0 + lr + RPP * j;
            da[j] = (m < M) ? load4_guard(A + m * lda + k0 + lc, K - (k0 + lc), VEC_A) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int j = 0; j < RW; ++j) {
            const int n = n0 + lr + RPP * j;
            dw[j] = (n < N) ? load4_guard(W + (int64_t)n * ldw + k0 + lc, K - (k0 + lc), VEC_W)
                            : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    const float *a_row[RA];
    const float *w_row[RW];
#pragma unroll
    for (int j = 0; j < RA; ++j) a_row[j] = A + (m0 + lr + RPP * j) * lda + lc;
#pragma unroll
    for (int j = 0; j < RW; ++j) w_row[j] = W + (int64_t)(n0 + lr + RPP * j) * ldw + lc;
    auto load_tile_full = [&](int k0, f32x4 (&da)[RA], f32x4 (&dw)[RW]) {
#pragma unroll
        for (int j = 0; j < RA; ++j) da[j] = *reinterpret_cast<const f32x4 *>(a_row[j] + k0);
#pragma unroll
        for (int j = 0; j < RW; ++j) dw[j] = *reinterpret_cast<const f32x4 *>(w_row[j] + k0);
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    auto stage = [&](const f32x4 (&sa)[RA], const f32x4 (&sw)[RW]) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < RA; ++j) {
            bf16x4 pc[NP];
            bf_split4<NP>(sa[j], pc);
#pragma unroll
            for (int q = 0; q < NP; ++q) *reinterpret_cast<bf16x4 *>(&As[q][lr + RPP * j][lc]) = pc[q];
        }
#pragma unroll
        for (int j = 0; j < RW; ++j) {
            bf16x4 pc[NP];
            bf_split4<NP>(sw[j], pc);
#pragma unroll
            for (int q = 0; q < NP; ++q) *reinterpret_cast<bf16x4 *>(&Ws[q][lr + RPP * j][lc]) = pc[q];
        }
        __syncthreads();
    };
    auto compute = [&]() {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 a[MI][NP];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int q = 0; q < NP; ++q)
                    a[mi][q] = *reinterpret_cast<const bf16x8 *>(&As[q][arow0 + 32 * mi + i][ks * 16 + 8 * h]);
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                const int col0 = bcol0 + 32 * ni;
                if (n0 + col0 >= N) continue;  // wave-uniform
                bf16x8 b[NP];
#pragma unroll
                for (int q = 0; q < NP; ++q) b[q] = *reinterpret_cast<const bf16x8 *>(&Ws[q][col0 + i][ks * 16 + 8 * h]);
#pragma unroll
                for (int pr = 0; pr < NPROD; ++pr)
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi][BfProd<NPROD>::pa(pr)],
                                                                              b[BfProd<NPROD>::pb(pr)], acc[mi][ni], 0, 0, 0);
            }
        }
    };

    const int nk = (K + BK - 1) / BK;
    const int nkf = K / BK;  // k tiles that need no column guard
    int kbeg = 0;
    if (VEC_A && VEC_W && (m0 + BMT <= M) && (n0 + BNT <= N) && nkf >= 2 * PF) {
#pragma unroll
        for (int p = 0; p < PF; ++p) load_tile_full(p * BK, ra[p], rw[p]);
        const int nloop = (nkf - PF) / PF;
        for (int it = 0; it < nloop; ++it) {
#pragma unroll
            for (int p = 0; p < PF; ++p) {
                stage(ra[p], rw[p]);
                load_tile_full((it * PF + p + PF) * BK, ra[p], rw[p]);
                compute();
            }
        }
        kbeg = nloop * PF;  // tiles kbeg .. kbeg+PF-1 are in the ring
    } else {
#pragma unroll
        for (int p = 0; p < PF; ++p)
            if (p < nk) load_tile(p * BK, ra[p], rw[p]);
    }
    for (int kt0 = kbeg; kt0 < nk; kt0 += PF) {
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            const int kt = kt0 + p;
            if (kt >= nk) break;
            stage(ra[p], rw[p]);
            if (kt + PF < nk) load_tile((kt + PF) * BK, ra[p], rw[p]);
            compute();
        }
    }
    // C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        const int n = n0 + bcol0 + 32 * ni + i;
        if (n >= N) continue;
        const float bv = (bias != nullptr) ? bias[n] : 0.f;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t m = m0 + arow0 + 32 * mi + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (m >= M) continue;
                float v = acc[mi][ni][r] + bv;
                if (act == RP_ACT_MASK)
                    v = (aux[m * ldaux + n] > 0.f) ? v : 0.f;
                else
                    v = rp_act_apply(act, v);
                C[m * ldc + n] = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// NT for WIDE layers (matrix-core bound: N >= 256, long K):  256 x 128 workgroup tiles, EIGHT waves of 64 x 64.
//
// What bounds the 128 x 128 / 32 x 64-wave kernel above on these shapes is LDS read bandwidth, not the matrix core: a
// 32 x 64 wave tile reads 3 fragments per 2 MFMA tiles, 96 KB of ds_read_b128 per 32-deep k-tile and workgroup against
// 768 matrix-core cycles.  Here a wave owns 64 x 64 (2 x 2 MFMA tiles): 4 fragment reads feed 4 tiles, 0.67 KB of LDS
// reads per MFMA instead of 1 KB, and per k-tile a CU spends 1024 LDS cycles against 1536 matrix-core cycles.
//   * LDS is double buffered, ONE (LDS-only) barrier per k-tile: tile t+1 is converted and written while tile t is on
//     the matrix core; global loads run three tiles ahead through a register ring (counted vmcnt waits: the main loop is
//     straight-line code with unguarded loads, the K tail and the last tiles run a guarded epilogue).
//     (Measured at [65536, 1677] x [1024, 1677], bf16x3: 0.82 ms = 274 TFLOP/s algorithmic against 1.17 ms on the
//     128 x 128 kernel; SQ counters: the matrix pipe is busy 40 % of the kernel at the ~2.0 GHz it clocks to, a wave
//     spends 33 % of its cycles issuing (~220 instructions per k-tile, 24 of them MFMAs), 39 % stalled on the busy
//     pipe and 28 % parked at the barrier / on waitcnt.  Running the two waves of a SIMD in opposite stage / compute
//     order ("ping-pong") spilled registers and lost 4 %; plain column-fastest tile order loses 6 %, M-fastest 20 %.)
//   * workgroup -> tile mapping is XCD-aware: the hardware deals consecutive workgroup ids round-robin to the 8 XCDs
//     (each with its own 4 MB L2); ids are remapped so that one XCD walks all column blocks of an M block back to back
//     and the [256, K] activation slab (the operand that does not fit) is fetched from HBM once and then served by
//     that XCD's L2, instead of once per column block.
// Instantiated for the split modes with two pieces or fewer (bf16x3 / bf16: 120 KB of LDS; six products would need
// 180 KB double buffered and keep the kernel above).
template <int NPROD>
__global__ __launch_bounds__(512) void linear_fwd_bf16_wide_kernel(const float *__restrict__ A, int64_t lda,
                                                                    const float *__restrict__ W, int64_t ldw,
                                                                    const float *__restrict__ bias, float *__restrict__ C,
                                                                    int64_t ldc, int64_t M, int N, int K, int act,
                                                                    const float *__restrict__ aux, int64_t ldaux,
                                                                    int mblocks, int nblocks) {
    constexpr int NP = BfProd<NPROD>::NP;
    static_assert(NP <= 2, "double-buffered 256 x 128 tiles fit the LDS with at most two pieces per operand");
    constexpr int WBM = 256, WBN = 128;
    __shared__ __attribute__((aligned(16))) __bf16 As[2][NP][WBM][BF_LD];
    __shared__ __attribute__((aligned(16))) __bf16 Ws[2][NP][WBN][BF_LD];
    const int t = threadIdx.x;
    // XCD-aware mapping: id = xcd + 8 * j;  the 8 * nblocks ids {xcd + 8 * (g * nblocks + n)} of group g run the
    // column blocks n of M block (8 g + xcd) one after the other on XCD `xcd`
    int mb, nb;
    {
        const int id = blockIdx.x, xcd = id & 7, j = id >> 3;
        const int g = j / nblocks;
        nb = j - g * nblocks;
        mb = g * 8 + xcd;
        if (mb >= mblocks) return;  // (the grid is padded to whole groups)
    }
    const int64_t m0 = (int64_t)mb * WBM;
    const int n0 = nb * WBN;
    const int lr = t >> 3, lc = (t & 7) * 4;   // staging: 64 rows per pass, 8 threads (32 floats) per row
    const int w = t >> 6, l = t & 63, i = l & 31, h = l >> 5;
    const int wm = w & 3, wn = w >> 2;         // 4 (rows) x 2 (columns) waves of 64 x 64
    const int arow0 = 64 * wm, bcol0 = 64 * wn;

    const float *a_row[4];
    const float *w_row[2];
#pragma unroll
    for (int j = 0; j < 4; ++j) a_row[j] = A + (m0 + lr + 64 * j) * lda + lc;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        int n = n0 + lr + 64 * j;
        if (n >= N) n = N - 1;  // clamped, never stored (keeps the loads unguarded)
        w_row[j] = W + (int64_t)n * ldw + lc;
    }
    f32x4 ra[3][4], rw[3][2];
    auto load_full = [&](int kt, f32x4 (&da)[4], f32x4 (&dw)[2]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) da[j] = *reinterpret_cast<const f32x4 *>(a_row[j] + kt * BK);
#pragma unroll
        for (int j = 0; j < 2; ++j) dw[j] = *reinterpret_cast<const f32x4 *>(w_row[j] + kt * BK);
    };
    auto load_guard = [&](int kt, f32x4 (&da)[4], f32x4 (&dw)[2]) {
        const int k0 = kt * BK + lc;
#pragma unroll
        for (int j = 0; j < 4; ++j) da[j] = load4_guard(a_row[j] + kt * BK, K - k0, true);
#pragma unroll
        for (int j = 0; j < 2; ++j) dw[j] = load4_guard(w_row[j] + kt * BK, K - k0, true);
    };
    auto stage = [&](int buf, const f32x4 (&sa)[4], const f32x4 (&sw)[2]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            bf16x4 pc[NP];
            bf_split4<NP>(sa[j], pc);
#pragma unroll
            for (int q = 0; q < NP; ++q) *reinterpret_cast<bf16x4 *>(&As[buf][q][lr + 64 * j][lc]) = pc[q];
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            bf16x4 pc[NP];
            bf_split4<NP>(sw[j], pc);
#pragma unroll
            for (int q = 0; q < NP; ++q) *reinterpret_cast<bf16x4 *>(&Ws[buf][q][lr + 64 * j][lc]) = pc[q];
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
    auto compute = [&](int buf) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 a[2][NP], b[2][NP];
#pragma unroll
            for (int q = 0; q < NP; ++q) {
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
                    a[mi][q] = *reinterpret_cast<const bf16x8 *>(&As[buf][q][arow0 + 32 * mi + i][ks * 16 + 8 * h]);
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    b[ni][q] = *reinterpret_cast<const bf16x8 *>(&Ws[buf][q][bcol0 + 32 * ni + i][ks * 16 + 8 * h]);
            }
#pragma unroll
            for (int pr = 0; pr < NPROD; ++pr)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi][BfProd<NPROD>::pa(pr)],
                                                                              b[ni][BfProd<NPROD>::pb(pr)], acc[mi][ni], 0, 0, 0);
        }
    };
#define RP_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
    const int nk = (K + BK - 1) / BK;
    const int nkf = K / BK;  // k-tiles that need no column guard
    // ring slot of tile kt: kt % 3.  Prologue: tiles 0..2 in flight, tile 0 staged.
    int kt = 0;
    if (nkf >= 6) {
        load_full(0, ra[0], rw[0]);
        load_full(1, ra[1], rw[1]);
        load_full(2, ra[2], rw[2]);
        stage(0, ra[0], rw[0]);
        load_full(3, ra[0], rw[0]);
        RP_LDS_BARRIER();
        // steady state, unrolled by 6 (ring of 3 x double buffer of 2): at the top of step kt the LDS buffer kt & 1
        // holds tile kt, the ring holds tiles kt+1, kt+2, kt+3
        const int nmain = (nkf - 4) / 6 * 6;  // steps that can prefetch tile kt + 4 unguarded
        for (; kt < nmain; kt += 6) {
#pragma unroll
            for (int u = 0; u < 6; ++u) {
                const int slot = (u + 1) % 3;             // ring slot of tile kt + u + 1
                stage((u + 1) & 1, ra[slot], rw[slot]);
                load_full(kt + u + 4, ra[slot], rw[slot]);
                compute(u & 1);
                RP_LDS_BARRIER();
            }
        }
        // drain: tiles kt .. nk-1; tiles kt+1 .. kt+3 are in the ring (full tiles), later ones are loaded guarded
        // (kt is a multiple of 6 here, so LDS buffer parity and ring slots restart at 0)
        for (int u = 0; kt + u < nk; ++u) {
            const int nxt = kt + u + 1;
            if (nxt < nk) {
                const int slot = nxt % 3;
                if (u >= 3) {  // not in the ring any more
                    if (slot == 0) load_guard(nxt, ra[0], rw[0]);
                    else if (slot == 1) load_guard(nxt, ra[1], rw[1]);
                    else load_guard(nxt, ra[2], rw[2]);
                }
                if (slot == 0) stage(nxt & 1, ra[0], rw[0]);
                else if (slot == 1) stage(nxt & 1, ra[1], rw[1]);
                else stage(nxt & 1, ra[2], rw[2]);
            }
            compute((kt + u) & 1);
            RP_LDS_BARRIER();
        }
    } else {
        for (; kt < nk; ++kt) {
            load_guard(kt, ra[0], rw[0]);
            RP_LDS_BARRIER();
            stage(0, ra[0], rw[0]);
            RP_LDS_BARRIER();
            compute(0);
        }
    }
#undef RP_LDS_BARRIER
    // C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const int n = n0 + bcol0 + 32 * ni + i;
        if (n0 + bcol0 + 32 * ni >= N) continue;  // wave-uniform (N is a multiple of 32 on this path)
        const float bv = (bias != nullptr) ? bias[n] : 0.f;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t m = m0 + arow0 + 32 * mi + (r & 3) + 8 * (r >> 2) + 4 * h;
                float v = acc[mi][ni][r] + bv;
                if (act == RP_ACT_MASK)
                    v = (aux[m * ldaux + n] > 0.f) ? v : 0.f;
                else
                    v = rp_act_apply(act, v);
                C[m * ldc + n] = v;
            }
        }
    }
}

// NT with a short contraction (K <= 64: the dgrad of a wide layer, the 64 -> 64 hidden layers, the 64 -> 1 head).
// The [128, K] A tile is read ONCE, split, and kept in registers as MFMA fragments (4 k-steps x NP pieces); the
// workgroup then walks over `ntiles` 64-column tiles of W, double-buffered through LDS with one barrier per tile
// and the next W tile's global loads in flight, writing a 128 x 64 block of the output per step.  For the dgrad
// dX[B,1677] = dH[B,64] . W this makes the kernel a pure stream of output writes.
//
// FULL = every tile is interior (M % 128 == 0, N % 64 == 0, 16-byte aligned rows, K % 4 == 0): the loop body is
// straight-line code.  That matters: gfx950 counts loads and stores on ONE in-order counter (vmcnt), the W prefetch
// of tile t+1 is issued before the stores of tile t, and the compiler only emits the counted wait vmcnt(#stores)
// for it when no branch separates them — any guarded load/store in the loop turns it into vmcnt(0), a full store
// round trip (microseconds under a saturated write stream) per tile.  Edges run the guarded instantiation.
#define SK_LD 72
// ROWADD (interior tiles, no activation): out[m, n] += rs[m] * radd[m, n % 64] for the first `nadd` 64-column tiles —
// the DeepFM gather backward's FM term g_fm[b] * S[b, :] folded into the dgrad that produces dX (D = 64: one tile =
// one field), so the segmented reduce reads one row per (sample, field) pair instead of two.
template <int NPROD, int ACT, bool FULL, bool ROWADD = false>
__global__ __launch_bounds__(256) void linear_fwd_bf16_smallk_kernel(const float *__restrict__ A, int64_t lda,
                                                                     const float *__restrict__ W, int64_t ldw,
                                                                     const float *__restrict__ bias,
                                                                     float *__restrict__ C, int64_t ldc, int64_t M,
                                                                     int N, int K, const float *__restrict__ aux,
                                                                     int64_t ldaux, int ntiles,
                                                                     const float *__restrict__ rs = nullptr,
                                                                     const float *__restrict__ radd = nullptr,
                                                                     int64_t ldadd = 0, int nadd = 0) {
    constexpr int NP = BfProd<NPROD>::NP;
    __shared__ __attribute__((aligned(16))) __bf16 Ws[2][NP][BN][SK_LD];
    const int t = threadIdx.x;
    const int64_t m0 = (int64_t)blockIdx.x * 128;
    const int w = t >> 6, l = t & 63, i = l & 31, h = l >> 5;
    const int lr = t >> 4, lc = (t & 15) * 4;
    const int tile0 = blockIdx.y * ntiles;
    int tile1 = tile0 + ntiles;
    const int tiles_all = (N + BN - 1) / BN;
    if (tile1 > tiles_all) tile1 = tiles_all;

    // A fragments: lane (i, h) holds row 32w+i, k = 16*ks + 8h .. +7
    bf16x8 af[4][NP];
    {
        const int64_t m = m0 + 32 * w + i;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int k = ks * 16 + 8 * h;
            f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = {0.f, 0.f, 0.f, 0.f};
            if (FULL) {
                // unguarded: clamp the column into range and zero by select (no branch)
                const int ka = (k + 4 <= K) ? k : 0, kb = (k + 8 <= K) ? k + 4 : 0;
                const f32x4 t0 = *reinterpret_cast<const f32x4 *>(A + m * lda + ka);
                const f32x4 t1 = *reinterpret_cast<const f32x4 *>(A + m * lda + kb);
                const float s0 = (k + 4 <= K) ? 1.f : 0.f, s1 = (k + 8 <= K) ? 1.f : 0.f;
                v0 = t0 * s0;
                v1 = t1 * s1;
            } else if (m < M) {
                v0 = load4_guard(A + m * lda + k, K - k, false);
                v1 = load4_guard(A + m * lda + k + 4, K - (k + 4), false);
            }
            f32x8 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] = v0[e];
                v[4 + e] = v1[e];
            }
            bf_split8<NP>(v, af[ks]);
        }
    }
    // ROWADD operands in the C layout: rows (r&3) + 8*(r>>2) + 4*h of this wave's 32, column i of each tile half
    float rsv[16], sv[2][16];
    if (ROWADD) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t m = m0 + 32 * w + (r & 3) + 8 * (r >> 2) + 4 * h;
            rsv[r] = rs[m];
            sv[0][r] = radd[m * ldadd + i];
            sv[1][r] = radd[m * ldadd + 32 + i];
        }
    }
    f32x4 rw[4];
    const int lcc = (lc + 4 <= K) ? lc : 0;
    const float lcs = (lc + 4 <= K) ? 1.f : 0.f;
    // FULL: the bias of the NEXT tile is fetched together with its W tile (i.e. before this tile's stores) so that no
    // load younger than the stores has to be waited for; a null bias reads W instead and is zeroed by a select.
    const float *bp = (bias != nullptr) ? bias : W;
    const float bsel = (bias != nullptr) ? 1.f : 0.f;
    float bnx[2] = {0.f, 0.f};
    auto load_w = [&](int tile) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = tile * BN + lr + 16 * j;
            if (FULL)
                rw[j] = *reinterpret_cast<const f32x4 *>(W + (int64_t)n * ldw + lcc);  // (* lcs at the point of use)
            else
                rw[j] = (n < N) ? load4_guard(W + (int64_t)n * ldw + lc, K - lc, false) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if (FULL) {
            bnx[0] = bp[tile * BN + i];
            bnx[1] = bp[tile * BN + 32 + i];
        }
    };
    if (tile0 < tile1) load_w(tile0);
    for (int tile = tile0; tile < tile1; ++tile) {
        const int buf = (tile - tile0) & 1;
        const float bcur[2] = {bnx[0] * bsel, bnx[1] * bsel};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            bf16x4 pc[NP];
            bf_split4<NP>(FULL ? rw[j] * lcs : rw[j], pc);
#pragma unroll
            for (int q = 0; q < NP; ++q) *reinterpret_cast<bf16x4 *>(&Ws[buf][q][lr + 16 * j][lc]) = pc[q];
        }
        // LDS-only barrier (no release fence -> no vmcnt drain).  Tile `tile` is visible after it; everyone is also
        // past the reads of this buffer two tiles ago.
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (FULL) {
            load_w(tile + 1 < tile1 ? tile + 1 : tile);  // unconditional (the last one is a harmless re-read)
        } else if (tile + 1 < tile1) {
            load_w(tile + 1);
        }
        const int n0 = tile * BN;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            if (!FULL && n0 + nt * 32 >= N) continue;  // wave-uniform
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                bf16x8 b[NP];
#pragma unroll
                for (int q = 0; q < NP; ++q)
                    b[q] = *reinterpret_cast<const bf16x8 *>(&Ws[buf][q][nt * 32 + i][ks * 16 + 8 * h]);
#pragma unroll
                for (int pr = 0; pr < NPROD; ++pr)
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][BfProd<NPROD>::pa(pr)], b[BfProd<NPROD>::pb(pr)],
                                                                  acc, 0, 0, 0);
            }
            const int n = n0 + nt * 32 + i;
            if (FULL) {
                const float bv = bcur[nt];
                float *cp = C + (m0 + 32 * w + 4 * h) * ldc + n;
                if (ACT == RP_ACT_MASK) {
                    const float *ap = aux + (m0 + 32 * w + 4 * h) * ldaux + n;
                    float mk[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) mk[r] = ap[((r & 3) + 8 * (r >> 2)) * ldaux];
#pragma unroll
                    for (int r = 0; r < 16; ++r) cp[((r & 3) + 8 * (r >> 2)) * ldc] = (mk[r] > 0.f) ? acc[r] + bv : 0.f;
                } else {
                    const float fsel = (ROWADD && tile < nadd) ? 1.f : 0.f;  // (a select, not a branch)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float v = acc[r] + bv;
                        if (ROWADD) v = __builtin_fmaf(fsel * rsv[r], sv[nt][r], v);
                        cp[((r & 3) + 8 * (r >> 2)) * ldc] = (ACT == RP_ACT_RELU) ? fmaxf(v, 0.f) : v;
                    }
                }
            } else if (n < N) {
                const float bv = (bias != nullptr) ? bias[n] : 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int64_t m = m0 + 32 * w + (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (m >= M) continue;
                    float v = acc[r] + bv;
                    if (ACT == RP_ACT_RELU)
                        v = v > 0.f ? v : 0.f;
                    else if (ACT == RP_ACT_MASK)
                        v = (aux[m * ldaux + n] > 0.f) ? v : 0.f;
                    C[m * ldc + n] = v;
                }
            }
        }
    }
}

template <int NPROD>
static void launch_smallk(bool full, const float *a, int64_t lda, const float *w, int64_t ldw, const float *bias,
                          float *out, int64_t ldo, int64_t M, int N, int K, int act, const float *aux, int64_t ldaux,
                          hipStream_t s) {
    // split the column tiles over gridDim.y only as far as needed to fill the chip (each split re-reads the A tile)
    const int64_t mb = rp_cdiv(M, 128);
    const int tiles = (int)rp_cdiv(N, BN);
    int64_t ysplit = rp_cdiv(768, mb);
    if (ysplit > tiles) ysplit = tiles;
    if (ysplit < 1) ysplit = 1;
    const int per = (int)rp_cdiv(tiles, ysplit);
    dim3 grid((unsigned)mb, (unsigned)rp_cdiv(tiles, per));
#define SK(ACT, FULL)                                                                                                  \
    hipLaunchKernelGGL((linear_fwd_bf16_smallk_kernel<NPROD, ACT, FULL>), grid, dim3(256), 0, s, a, lda, w, ldw, bias, \
                       out, ldo, M, N, K, aux, ldaux, per)
    if (full) {
        if (act == RP_ACT_RELU) SK(RP_ACT_RELU, true);
        else if (act == RP_ACT_MASK) SK(RP_ACT_MASK, true);
        else SK(RP_ACT_NONE, true);
    } else {
        if (act == RP_ACT_RELU) SK(RP_ACT_RELU, false);
        else if (act == RP_ACT_MASK) SK(RP_ACT_MASK, false);
        else SK(RP_ACT_NONE, false);
    }
#undef SK
}

// interior region with the straight-line kernel, the right / bottom edges with the guarded one
template <int NPROD>
static void run_smallk(const float *a, int64_t lda, const float *w, int64_t ldw, const float *bias, float *out,
                       int64_t ldo, int64_t M, int N, int K, int act, const float *aux, int64_t ldaux, hipStream_t s) {
    const bool aligned = (lda % 4 == 0) && (ldw % 4 == 0) && rp_aligned16(a) && rp_aligned16(w) && (K % 4 == 0);
    const int64_t Mf = aligned ? (M / 128) * 128 : 0;
    const int Nf = aligned ? (N / BN) * BN : 0;
    if (Mf > 0 && Nf > 0)
        launch_smallk<NPROD>(true, a, lda, w, ldw, bias, out, ldo, Mf, Nf, K, act, aux, ldaux, s);
    else {
        launch_smallk<NPROD>(false, a, lda, w, ldw, bias, out, ldo, M, N, K, act, aux, ldaux, s);
        return;
    }
    if (N > Nf)  // right edge: all rows, the last (partial) column tile
        launch_smallk<NPROD>(false, a, lda, w + (int64_t)Nf * ldw, ldw, bias ? bias + Nf : nullptr, out + Nf, ldo, M,
                             N - Nf, K, act, aux ? aux + Nf : nullptr, ldaux, s);
    if (M > Mf)  // bottom edge: the last rows over the interior columns
        launch_smallk<NPROD>(false, a + Mf * lda, lda, w, ldw, bias, out + Mf * ldo, ldo, M - Mf, Nf, K, act,
                             aux ? aux + Mf * ldaux : nullptr, ldaux, s);
}

// TN:  dW[N,K] = dY[M,N]^T . X[M,K].  Output tile 64 (n) x 128 (k); 32 batch rows per stage, transposed while they
// are staged: thread (column c = t & 63, octet o = t >> 6) loads the 8 consecutive batch rows 8o..8o+7 of its dY
// column (and of two X columns), converts and writes them as ONE 16-byte LDS row segment, so the LDS images are
// [n][m] / [k][m] and the MFMA fragments (8 consecutive m per lane) are ds_read_b128.  Split-K over gridDim.z with a
// deterministic second-stage sum, as the f32 kernel; the bias gradient is summed in exact fp32 from the staged
// registers.
// (NA = 1: 168 registers, three workgroups per CU.  Deeper register prefetch (PF = 2, 3: 224 / 246 registers, two per CU)
//  measured 4 % and 16 % SLOWER at 65536 x 64 x 1677.)
// GATHER (rp_linear_wgrad_gather: the first layer of a model whose input is the embedding lookup): the first Kg = F * 64
// columns of X are not read from a materialised activation buffer but GATHERED — X[m, f*64 + j] = arena[keys[f*keyB + m]*64
// + j] with the arena-row keys the forward saved — so the forward never has to store its 436 MB of gathered rows; columns
// >= Kg (the dense features) come from the compact buffer X (leading dimension ldx, column k - Kg).  A k-tile of 128
// columns is two fields: lanes c < 32 read the row of field k0/64, the others the next one (two 256-byte segments per
// wave-instruction, as coalesced as the activation read they replace).  Keys are fetched one stage ahead of the rows
// they address.
// X_BF16 (round 4, the bf16-storage training mode): X is stored as bf16 [M, ldx] (ldx in elements): a lane reads its two
// columns as one dword and widens them — exact — so everything downstream is unchanged; half the bytes of the operand that
// dominates this launch's traffic.
template <int NPROD, int NA, int PF, bool VEC_X, bool GATHER = false, bool X_BF16 = false>
__global__ __launch_bounds__(256, NA == 1 ? 3 : 2) void linear_wgrad_bf16_kernel(const float *__restrict__ dY, int64_t lddy,
                                                                const float *__restrict__ X, int64_t ldx,
                                                                float *__restrict__ P, float *__restrict__ Pb,
                                                                int64_t M, int N, int K, int64_t rows_per_split,
                                                                const float *__restrict__ arena = nullptr,
                                                                const int32_t *__restrict__ keys = nullptr,
                                                                int64_t keyB = 0, int Kg = 0, int xcd_tiles = 0,
                                                                int xcd_ktiles = 0, int xcd_splits = 0) {
    // output tile (64*NA) n x 128 k; waves 2 (n) x 2 (k), each NA x 2 MFMA tiles (NA = 2 for wide layers: every
    // fragment read from LDS then feeds two MFMA tiles)
    constexpr int NP = BfProd<NPROD>::NP;
    constexpr int BNT = TN_BN * NA;
    __shared__ __attribute__((aligned(16))) __bf16 Yt[NP][BNT][BF_LD];
    __shared__ __attribute__((aligned(16))) __bf16 Xt[NP][TN_BK][BF_LD];
    __shared__ float bred[4][BNT];
    const int t = threadIdx.x;
    // XCD-local order (xcd_tiles > 0: a 1-D grid, id = xcd + 8 j): every (k, n) tile of batch split 8 g + xcd runs on XCD
    // `xcd`, one split after the other.  The tiles of a split read the SAME batch rows of dY and X and walk them in step, so
    // with all of them behind one L2 a row stage is fetched from HBM once instead of once per XCD — with workgroups dealt
    // round-robin over the eight XCDs the wide layer moved 5.6 GB for 0.72 GB of operands (rocprofv3 FETCH_SIZE, round 2).
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (xcd_tiles > 0) {
        const int id = blockIdx.x, xcd = id & 7, j = id >> 3;
        const int g = j / xcd_tiles, tile = j - g * xcd_tiles;
        bz = g * 8 + xcd;
        if (bz >= xcd_splits) return;  // (the grid is padded to whole groups of eight splits)
        by = tile / xcd_ktiles;
        bx = tile - by * xcd_ktiles;
    }
    const int k0 = bx * TN_BK;
    const int n0 = by * BNT;
    const int64_t mbeg = (int64_t)bz * rows_per_split;
    int64_t mend = mbeg + rows_per_split;
    if (mend > M) mend = M;
    const int w = t >> 6, l = t & 63, i = l & 31, h = l >> 5;
    const int nt = (w & 1) * 32 * NA, kt = (w >> 1) * 64;
    const int c = t & 63, o = t >> 6;
    const int kx = k0 + 2 * c;

    float ry[PF][NA][8];
    f32x2 rx[PF][8];
    const bool gtile = GATHER && (k0 + TN_BK <= Kg);                      // this k-tile is gathered (block-uniform)
    const int32_t *kcol = GATHER ? keys + (int64_t)(kx >> 6) * keyB : nullptr;  // the keys of this lane's field
    const int cin = kx & 63;
    auto load_tile = [&](int64_t mm, float (&dy_)[NA][8], f32x2 (&dx_)[8]) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int64_t m = mm + 8 * o + r;
            const bool ok = m < mend;
#pragma unroll
            for (int u = 0; u < NA; ++u) dy_[u][r] = (ok && n0 + c + 64 * u < N) ? dY[m * lddy + n0 + c + 64 * u] : 0.f;
            f32x2 v = {0.f, 0.f};
            if (GATHER && ok) {
                if (gtile) {
                    v = *reinterpret_cast<const f32x2 *>(arena + (int64_t)kcol[m] * 64 + cin);
                } else {  // the dense columns, from the compact buffer
                    const float *px = X + m * ldx + (kx - Kg);
                    if (kx < K) v.x = px[0];
                    if (kx + 1 < K) v.y = px[1];
                }
            } else if (X_BF16) {
                // branch-free (a load behind a branch turns every wait of the loop into vmcnt(0): the guarded k-tile of a
                // bf16 activation ran 16 serialized 2-byte loads per stage): row and column clamped into the buffer, the
                // dword always loaded, what lies outside masked afterwards (ldx is even and >= K)
                const int64_t mc = ok ? m : mbeg;
                const int kc = kx + 1 < (int)ldx ? kx : (int)ldx - 2;
                const uint32_t w2 = *reinterpret_cast<const uint32_t *>(reinterpret_cast<const uint16_t *>(X) + mc * ldx + kc);
                v.x = (ok && kx < K) ? __uint_as_float(w2 << 16) : 0.f;
                v.y = (ok && kx + 1 < K) ? __uint_as_float(w2 & 0xFFFF0000u) : 0.f;
            } else if (VEC_X) {
                // branch-free as above (VEC_X: ldx even, 8-byte aligned rows; ldx >= K): the pair is always loaded from a
                // clamped (row, column) inside the buffer and masked afterwards — the k-tile that holds column K - 1 (one of
                // 14 at K = 1677) otherwise waits for each of its row loads on its own
                const int64_t mc = ok ? m : mbeg;
                const int kc = kx + 1 < (int)ldx ? kx : (int)ldx - 2;
                const f32x2 w2 = *reinterpret_cast<const f32x2 *>(X + mc * ldx + kc);
                v.x = (ok && kx < K) ? w2.x : 0.f;
                v.y = (ok && kx + 1 < K) ? w2.y : 0.f;
            } else if (ok) {
                const float *px = X + m * ldx + kx;
                if (kx < K) v.x = px[0];
                if (kx + 1 < K) v.y = px[1];
            }
            dx_[r] = v;
        }
    };
    // interior tile: unguarded loads -> straight-line main loop -> counted vmcnt waits (see linear_fwd_bf16_kernel)
    const float *yb = dY + (mbeg + 8 * o) * lddy + n0 + c;
    const float *xb = X + (mbeg + 8 * o) * ldx + kx;
    const uint16_t *xb16 = reinterpret_cast<const uint16_t *>(X) + (mbeg + 8 * o) * ldx + kx;  // (X_BF16)
    const int32_t *kb = GATHER ? kcol + mbeg + 8 * o : nullptr;
    int32_t kreg[8];  // GATHER: the keys of the rows the NEXT load_tile_full fetches
    auto load_keys = [&](int64_t moff) {
#pragma unroll
        for (int r = 0; r < 8; ++r) kreg[r] = kb[moff + r];
    };
    auto load_tile_full = [&](int64_t moff, float (&dy_)[NA][8], f32x2 (&dx_)[8]) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int u = 0; u < NA; ++u) dy_[u][r] = yb[(moff + r) * lddy + 64 * u];
            if (GATHER) {
                dx_[r] = *reinterpret_cast<const f32x2 *>(arena + (int64_t)kreg[r] * 64 + cin);
            } else if (X_BF16) {
                const uint32_t w2 = *reinterpret_cast<const uint32_t *>(xb16 + (moff + r) * ldx);
                dx_[r] = f32x2{__uint_as_float(w2 << 16), __uint_as_float(w2 & 0xFFFF0000u)};
            } else {
                dx_[r] = *reinterpret_cast<const f32x2 *>(xb + (moff + r) * ldx);
            }
        }
    };

    f32x16 acc[NA][2];
#pragma unroll
    for (int u = 0; u < NA; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            acc[u][0][r] = 0.f;
            acc[u][1][r] = 0.f;
        }
    float bsum[NA];
#pragma unroll
    for (int u = 0; u < NA; ++u) bsum[u] = 0.f;
    const bool do_bias = (Pb != nullptr) && (bx == 0);

    auto stage = [&](const float (&sy)[NA][8], const f32x2 (&sx)[8]) {
        __syncthreads();
        f32x8 vx0, vx1;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            vx0[r] = sx[r].x;
            vx1[r] = sx[r].y;
        }
        bf16x8 pc[NP];
#pragma unroll
        for (int u = 0; u < NA; ++u) {
            f32x8 vy;
#pragma unroll
            for (int r = 0; r < 8; ++r) vy[r] = sy[u][r];
            if (do_bias)
                bsum[u] += ((sy[u][0] + sy[u][1]) + (sy[u][2] + sy[u][3])) + ((sy[u][4] + sy[u][5]) + (sy[u][6] + sy[u][7]));
            bf_split8<NP>(vy, pc);
#pragma unroll
            for (int q = 0; q < NP; ++q) *reinterpret_cast<bf16x8 *>(&Yt[q][c + 64 * u][8 * o]) = pc[q];
        }
        bf_split8<NP>(vx0, pc);
#pragma unroll
        for (int q = 0; q < NP; ++q) *reinterpret_cast<bf16x8 *>(&Xt[q][2 * c][8 * o]) = pc[q];
        bf_split8<NP>(vx1, pc);
#pragma unroll
        for (int q = 0; q < NP; ++q) *reinterpret_cast<bf16x8 *>(&Xt[q][2 * c + 1][8 * o]) = pc[q];
        __syncthreads();
    };
    auto compute = [&]() {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 a[NA][NP], b0[NP], b1[NP];
#pragma unroll
            for (int q = 0; q < NP; ++q) {
#pragma unroll
                for (int u = 0; u < NA; ++u)
                    a[u][q] = *reinterpret_cast<const bf16x8 *>(&Yt[q][nt + 32 * u + i][ks * 16 + 8 * h]);
                b0[q] = *reinterpret_cast<const bf16x8 *>(&Xt[q][kt + i][ks * 16 + 8 * h]);
                b1[q] = *reinterpret_cast<const bf16x8 *>(&Xt[q][kt + 32 + i][ks * 16 + 8 * h]);
            }
#pragma unroll
            for (int pr = 0; pr < NPROD; ++pr) {
                // (a bf16 activation IS its first piece: the products its two zero pieces enter are skipped)
                if (X_BF16 && BfProd<NPROD>::pb(pr) != 0) continue;
#pragma unroll
                for (int u = 0; u < NA; ++u) {
                    acc[u][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[u][BfProd<NPROD>::pa(pr)], b0[BfProd<NPROD>::pb(pr)],
                                                                        acc[u][0], 0, 0, 0);
                    acc[u][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[u][BfProd<NPROD>::pa(pr)], b1[BfProd<NPROD>::pb(pr)],
                                                                        acc[u][1], 0, 0, 0);
                }
            }
        }
    };

    const int64_t nst = (mend > mbeg) ? (mend - mbeg + TN_BM - 1) / TN_BM : 0;  // stages of 32 batch rows
    const int64_t nstf = (mend > mbeg) ? (mend - mbeg) / TN_BM : 0;             // ... that are complete
    int64_t sbeg = 0;
    if (GATHER && gtile && PF == 1 && (n0 + BNT <= N) && nstf >= 3) {
        // straight-line main loop as below; the keys of stage s + 2 are requested right after the rows of stage s + 1
        load_keys(0);
        load_tile_full(0, ry[0], rx[0]);
        load_keys(TN_BM);
        const int64_t nloop = nstf - 2;
        for (int64_t it = 0; it < nloop; ++it) {
            stage(ry[0], rx[0]);
            load_tile_full((it + 1) * TN_BM, ry[0], rx[0]);
            load_keys((it + 2) * TN_BM);
            compute();
        }
        stage(ry[0], rx[0]);
        load_tile_full((nloop + 1) * TN_BM, ry[0], rx[0]);
        compute();
        sbeg = nloop + 1;
    } else if (!GATHER && VEC_X && (n0 + BNT <= N) && (k0 + TN_BK <= K) && nstf >= 2 * PF) {
#pragma unroll
        for (int p = 0; p < PF; ++p) load_tile_full((int64_t)p * TN_BM, ry[p], rx[p]);
        const int64_t nloop = (nstf - PF) / PF;
        for (int64_t it = 0; it < nloop; ++it) {
#pragma unroll
            for (int p = 0; p < PF; ++p) {
                stage(ry[p], rx[p]);
                load_tile_full((it * PF + p + PF) * TN_BM, ry[p], rx[p]);
                compute();
            }
        }
        sbeg = nloop * PF;
    } else {
#pragma unroll
        for (int p = 0; p < PF; ++p)
            if (p < nst) load_tile(mbeg + (int64_t)p * TN_BM, ry[p], rx[p]);
    }
    for (int64_t s0 = sbeg; s0 < nst; s0 += PF) {
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            const int64_t st = s0 + p;
            if (st >= nst) break;
            stage(ry[p], rx[p]);
            if (st + PF < nst) load_tile(mbeg + (st + PF) * TN_BM, ry[p], rx[p]);
            compute();
        }
    }
    float *Pz = P + (int64_t)bz * N * K;
#pragma unroll
    for (int u = 0; u < NA; ++u)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int k = k0 + kt + kk * 32 + i;
            if (k >= K) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + nt + 32 * u + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (n < N) Pz[(int64_t)n * K + k] = acc[u][kk][r];
            }
        }
    if (do_bias) {
#pragma unroll
        for (int u = 0; u < NA; ++u) bred[o][c + 64 * u] = bsum[u];
        __syncthreads();
        for (int e = t; e < BNT; e += 256)
            if (n0 + e < N) Pb[(int64_t)bz * N + n0 + e] = (bred[0][e] + bred[1][e]) + (bred[2][e] + bred[3][e]);
    }
}

// second stage of the split-K wgrad: 16 output elements x 16 slices per workgroup; slice q sums partials q, q+16, ...
// and the 16 slice sums are combined in a fixed order (deterministic).  One thread per element with a serial loop
// over S is latency-bound: 512 dependent-address loads for a 64 x 64 layer.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float *__restrict__ P, const float *__restrict__ Pb,
                                                           int S, int N, int K, float *__restrict__ dw,
                                                           int64_t lddw, float *__restrict__ db, int accumulate) {
    __shared__ float red[16][17];
    const int c = threadIdx.x & 15, q = threadIdx.x >> 4;
    const int64_t e = (int64_t)blockIdx.x * 16 + c;
    const int64_t nk = (int64_t)N * K;
    const bool is_w = e < nk, is_b = !is_w && e < nk + N && db != nullptr;
    float s = 0.f;
    if (is_w) {
#pragma unroll 4
        for (int z = q; z < S; z += 16) s += P[(int64_t)z * nk + e];
    } else if (is_b) {
        const int n = (int)(e - nk);
#pragma unroll 4
        for (int z = q; z < S; z += 16) s += Pb[(int64_t)z * N + n];
    }
    red[q][c] = s;
    __syncthreads();
    if (q != 0) return;
#pragma unroll
    for (int j = 1; j < 16; ++j) s += red[j][c];
    if (is_w) {
        const int n = (int)(e / K), k = (int)(e - (int64_t)n * K);
        float *d = dw + (int64_t)n * lddw + k;
        *d = accumulate ? (*d + s) : s;
    } else if (is_b) {
        const int n = (int)(e - nk);
        db[n] = accumulate ? (db[n] + s) : s;
    }
}

// out rows C .. C_out-1 (if any) are written as zeros
// copy (optional): the same elements are also written UNtransposed with another row stride, copy[r * ldcopy + c] — the
// 16-byte-aligned staging copy of a weight whose rows are not (rp_copy_rows), made by the launch that reads the weight anyway
__global__ __launch_bounds__(256) void transpose_kernel(const float *__restrict__ in, int64_t ldin,
                                                        float *__restrict__ out, int64_t ldout, int R, int C, int C_out,
                                                        float *__restrict__ copy = nullptr, int64_t ldcopy = 0) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
#pragma unroll
    for (int j = 0; j < 32; j += 8) {
        const int r = r0 + ty + j, c = c0 + tx;
        const bool ok = r < R && c < C;
        const float v = ok ? in[(int64_t)r * ldin + c] : 0.f;
        tile[ty + j][tx] = v;
        if (copy != nullptr && ok) copy[(int64_t)r * ldcopy + c] = v;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 32; j += 8) {
        const int c = c0 + ty + j, r = r0 + tx;
        if (r < R && c < C_out) out[(int64_t)c * ldout + r] = tile[tx][ty + j];
    }
}

__global__ __launch_bounds__(256) void relu_bwd_kernel(const float *__restrict__ dy, int64_t lddy,
                                                       const float *__restrict__ y, int64_t ldy,
                                                       float *__restrict__ out, int64_t ldo, int64_t M, int N) {
    const int64_t total = M * N;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t m = e / N;
        const int n = (int)(e - m * N);
        out[m * ldo + n] = (y[m * ldy + n] > 0.f) ? dy[m * lddy + n] : 0.f;
    }
}

template <int NPROD>
static void launch_smallk_rowadd(const float *a, int64_t lda, const float *w, int64_t ldw, float *out, int64_t ldo,
                                 int64_t M, int N, int K, const float *rs, const float *radd, int64_t ldadd, int nadd,
                                 hipStream_t s) {
    const int64_t mb = M / 128;
    const int tiles = N / BN;
    int64_t ysplit = rp_cdiv(768, mb);
    if (ysplit > tiles) ysplit = tiles;
    if (ysplit < 1) ysplit = 1;
    const int per = (int)rp_cdiv(tiles, ysplit);
    dim3 grid((unsigned)mb, (unsigned)rp_cdiv(tiles, per));
    hipLaunchKernelGGL((linear_fwd_bf16_smallk_kernel<NPROD, RP_ACT_NONE, true, true>), grid, dim3(256), 0, s, a, lda, w, ldw,
                       nullptr, out, ldo, M, N, K, nullptr, 0, per, rs, radd, ldadd, nadd);
}

// process-wide matrix-core precision for the GEMM-shaped entry points (rp_linear_*, rp_cin_pair_*)
static int g_matmul_precision = RP_MATMUL_AUTO;

extern "C" int rp_set_matmul_precision(int mode) {
    RP_REQUIRE(mode == RP_MATMUL_FP32 || mode == RP_MATMUL_BF16 || mode == RP_MATMUL_BF16X3 || mode == RP_MATMUL_BF16X6 ||
                   mode == RP_MATMUL_AUTO, "set_matmul_precision: unknown mode %d", mode);
    g_matmul_precision = mode;
    return RP_OK;
}

int rp_matmul_products(double flops, double bytes) {
    switch (g_matmul_precision) {
        case RP_MATMUL_FP32: return 0;
        case RP_MATMUL_BF16: return 1;
        case RP_MATMUL_BF16X3: return 3;
        case RP_MATMUL_BF16X6: return 6;
        default: break;
    }
    // ridge of the part: 2.5 PFLOP/s dense bf16 over 8 TB/s = 312 products per byte
    return (bytes > 0 && flops * 3.0 / bytes > 312.0) ? 3 : 6;
}

static int g_wide_kernel = 1;  // RP_WIDE_KERNEL=0 in the environment: fall back to the 128 x 128 kernel (A/B measurements)

static int linear_mode(int64_t M, int N, int K) {  // the RP_MATMUL_* mode one GEMM launch runs in
    if (g_matmul_precision != RP_MATMUL_AUTO) return g_matmul_precision;
    return rp_matmul_products(2.0 * M * N * K, 4.0 * ((double)M * K + (double)N * K + (double)M * N)) == 3
               ? RP_MATMUL_BF16X3 : RP_MATMUL_BF16X6;
}

extern "C" int rp_get_matmul_precision(void) { return g_matmul_precision; }

namespace {
struct WideKernelEnv {
    WideKernelEnv() {
        const char *e = getenv("RP_WIDE_KERNEL");
        if (e && e[0] == '0') g_wide_kernel = 0;
    }
} g_wide_kernel_env;
}  // namespace

// out[M,N] = a[M,K] . w[N,K]^T  +  row_scale[m] * row_add[m, n % 64] for n < add_cols       (no bias, no activation)
// Restricted to what the DeepFM dgrad needs and the straight-line short-K kernel covers: K <= 64 and a multiple of 4,
// M % 128 == 0, N % 64 == 0, add_cols % 64 == 0, 16-byte aligned rows, a split-bf16 matmul mode.  Anything else:
// RP_ERR_UNSUPPORTED (the caller composes rp_linear_fwd and keeps the FM term in rp_embed_grad_reduce).
extern "C" int rp_linear_fwd_rowadd(const float *a, int64_t lda, const float *w, int64_t ldw, float *out, int64_t ldo,
                                    int64_t M, int N, int K, const float *row_scale, const float *row_add,
                                    int64_t ld_add, int add_cols, rp_stream_t stream) {
    RP_REQUIRE(a && w && out && row_scale && row_add, "linear_fwd_rowadd: null pointer");
    RP_REQUIRE(lda >= K && ldw >= K && ldo >= N && ld_add >= 64, "linear_fwd_rowadd: leading dimension too small");
    const int mode = linear_mode(M, N, K);
    const bool ok = mode != RP_MATMUL_FP32 && K >= 4 && K <= 64 && K % 4 == 0 && M >= 128 && M % 128 == 0 && N >= 64 &&
                    N % 64 == 0 && add_cols >= 0 && add_cols % 64 == 0 && add_cols <= N && lda % 4 == 0 && ldw % 4 == 0 &&
                    rp_aligned16(a) && rp_aligned16(w);
    if (!ok) return rp_fail(RP_ERR_UNSUPPORTED, "linear_fwd_rowadd: shape/alignment/mode outside the fused kernel");
    hipStream_t s = (hipStream_t)stream;
    if (mode == RP_MATMUL_BF16X6) launch_smallk_rowadd<6>(a, lda, w, ldw, out, ldo, M, N, K, row_scale, row_add, ld_add, add_cols / 64, s);
    else if (mode == RP_MATMUL_BF16X3) launch_smallk_rowadd<3>(a, lda, w, ldw, out, ldo, M, N, K, row_scale, row_add, ld_add, add_cols / 64, s);
    else launch_smallk_rowadd<1>(a, lda, w, ldw, out, ldo, M, N, K, row_scale, row_add, ld_add, add_cols / 64, s);
    RP_LAUNCH_CHECK("linear_fwd_rowadd");
    return RP_OK;
}

extern "C" int rp_linear_fwd(const float *a, int64_t lda, const float *w, int64_t ldw, const float *bias, float *out,
                             int64_t ldo, int64_t M, int N, int K, int act, const float *aux, int64_t ldaux,
                             rp_stream_t stream) {
    RP_REQUIRE(a && w && out, "linear_fwd: null pointer");
    RP_REQUIRE(M >= 0 && N >= 1 && K >= 1, "linear_fwd: bad M/N/K");
    RP_REQUIRE(lda >= K && ldw >= K && ldo >= N, "linear_fwd: leading dimension too small");
    RP_REQUIRE(act == RP_ACT_NONE || act == RP_ACT_RELU || (act == RP_ACT_MASK && aux && ldaux >= N) ||
                   act == RP_ACT_TANH || act == RP_ACT_SIGMOID || act == RP_ACT_LEAKY,
               "linear_fwd: bad act/aux");
    if (M == 0) return RP_OK;
    const bool va = (lda % 4 == 0) && rp_aligned16(a);
    const bool vw = (ldw % 4 == 0) && rp_aligned16(w);
    hipStream_t s = (hipStream_t)stream;
    const int mode = linear_mode(M, N, K);
    if (mode != RP_MATMUL_FP32) {
        if (K <= 64) {  // A-stationary walk over the output columns
            // (its epilogue is a template parameter: tanh / sigmoid / leaky ReLU follow as one elementwise launch in place)
            const bool post = act == RP_ACT_TANH || act == RP_ACT_SIGMOID || act == RP_ACT_LEAKY;
            const int act_k = post ? RP_ACT_NONE : act;
            if (mode == RP_MATMUL_BF16X6) run_smallk<6>(a, lda, w, ldw, bias, out, ldo, M, N, K, act_k, aux, ldaux, s);
            else if (mode == RP_MATMUL_BF16X3) run_smallk<3>(a, lda, w, ldw, bias, out, ldo, M, N, K, act_k, aux, ldaux, s);
            else run_smallk<1>(a, lda, w, ldw, bias, out, ldo, M, N, K, act_k, aux, ldaux, s);
            RP_LAUNCH_CHECK("linear_fwd (bf16 split, short K)");
            if (post) return rp_act_fwd(out, ldo, out, ldo, M, N, act, stream);
            return RP_OK;
        }
        // a wide output that is not a multiple of 128 columns (the MMOE expert + gate GEMM: 520): the whole 128-column
        // tiles go to the 128 x 128 kernel, the few columns left to a second launch (463 -> 377 us at M = 65536,
        // K = 649: otherwise either a fifth, almost empty 128-wide tile or nine LDS-bound 64-wide tiles)
        const int Nb = (N / 128) * 128;
        if (N >= 256 && Nb < N && rp_cdiv(M, 128) * (Nb / 128) >= 1024) {
            int rc = rp_linear_fwd(a, lda, w, ldw, bias, out, ldo, M, Nb, K, act, aux, ldaux, stream);
            if (rc != RP_OK) return rc;
            return rp_linear_fwd(a, lda, w + (int64_t)Nb * ldw, ldw, bias ? bias + Nb : nullptr, out + Nb, ldo, M, N - Nb, K,
                                 act, aux ? aux + Nb : nullptr, ldaux, stream);
        }
        // wide, matrix-core-bound layers in a two-piece mode: 256 x 128 tiles, 64 x 64 wave tiles, double-buffered LDS
        if ((mode == RP_MATMUL_BF16X3 || mode == RP_MATMUL_BF16) && g_wide_kernel && N >= 256 && N % 32 == 0 && M % 256 == 0 &&
            K >= 192 && va && vw && rp_cdiv(N, 128) * (M / 256) >= 512) {
            const int mblocks = (int)(M / 256), nblocks = (int)rp_cdiv(N, 128);
            const int groups = (int)rp_cdiv(mblocks, 8);
            dim3 gridw((unsigned)(groups * 8 * nblocks));
            if (mode == RP_MATMUL_BF16X3)
                hipLaunchKernelGGL((linear_fwd_bf16_wide_kernel<3>), gridw, dim3(512), 0, s, a, lda, w, ldw, bias, out, ldo, M, N, K,
                                   act, aux, ldaux, mblocks, nblocks);
            else
                hipLaunchKernelGGL((linear_fwd_bf16_wide_kernel<1>), gridw, dim3(512), 0, s, a, lda, w, ldw, bias, out, ldo, M, N, K,
                                   act, aux, ldaux, mblocks, nblocks);
            RP_LAUNCH_CHECK("linear_fwd (bf16 split, wide)");
            return RP_OK;
        }
        // tile shape: 128 x 128 (eight waves of 32 x 64) for wide outputs; 128 x 64 otherwise — with eight waves when that grid
        // would give the 256 CUs fewer than ~4 workgroups each
        const int64_t n128 = rp_cdiv(N, 128) * 128;
        const bool big = N >= 256 && (n128 - N) * 8 <= N && rp_cdiv(M, 128) * (n128 / 128) >= 1024;
        const bool small = !big && rp_cdiv(M, 128) * rp_cdiv(N, BN) < 1024 && M > 64;
        dim3 grid((unsigned)rp_cdiv(M, 128), (unsigned)rp_cdiv(N, big ? 128 : BN));
#define CALLB(NPROD, WM, WN, MI, NI, PF, VA, VW)                                                                      \
    hipLaunchKernelGGL((linear_fwd_bf16_kernel<NPROD, WM, WN, MI, NI, PF, VA, VW>), grid, dim3(64 * WM * WN), 0, s, a, lda, w, \
                       ldw, bias, out, ldo, M, N, K, act, aux, ldaux)
#define CALLV(NPROD, WM, WN, MI, NI, PF)                            \
    do {                                                            \
        if (va && vw) CALLB(NPROD, WM, WN, MI, NI, PF, true, true); \
        else if (va) CALLB(NPROD, WM, WN, MI, NI, PF, true, false); \
        else if (vw) CALLB(NPROD, WM, WN, MI, NI, PF, false, true); \
        else CALLB(NPROD, WM, WN, MI, NI, PF, false, false);        \
    } while (0)
#define CALLP(NPROD)                              \
    do {                                          \
        if (big) CALLV(NPROD, 4, 2, 1, 2, 2);     \
        else if (small) CALLV(NPROD, 4, 2, 1, 1, 2); \
        else CALLV(NPROD, 4, 1, 1, 2, 2);         \
    } while (0)
        if (mode == RP_MATMUL_BF16X6) CALLP(6);
        else if (mode == RP_MATMUL_BF16X3) CALLP(3);
        else CALLP(1);
#undef CALLP
#undef CALLV
#undef CALLB
        RP_LAUNCH_CHECK("linear_fwd (bf16 split)");
        return RP_OK;
    }
    dim3 grid((unsigned)rp_cdiv(M, BM), (unsigned)rp_cdiv(N, BN));
#define CALL(VA, VW)                                                                                              \
    hipLaunchKernelGGL((linear_fwd_kernel<VA, VW>), grid, dim3(256), 0, s, a, lda, w, ldw, bias, out, ldo, M, N, K, \
                       act, aux, ldaux)
    if (va && vw) CALL(true, true);
    else if (va) CALL(true, false);
    else if (vw) CALL(false, true);
    else CALL(false, false);
#undef CALL
    RP_LAUNCH_CHECK("linear_fwd");
    return RP_OK;
}

static bool wgrad_wide(int N) { return g_matmul_precision != RP_MATMUL_FP32 && N >= 128; }  // 128 x 128 output tiles
static bool wgrad_xcd_local() {  // RP_WGRAD_XCD=0: the plain 3-D grid (A/B switch of the XCD-local tile order)
    static const bool on = !(getenv("RP_WGRAD_XCD") && atoi(getenv("RP_WGRAD_XCD")) == 0);
    return on;
}

static void wgrad_plan(int64_t M, int N, int K, int *S, int64_t *rows) {
    const bool wide = wgrad_wide(N);
    const int64_t tiles = rp_cdiv(K, TN_BK) * rp_cdiv(N, wide ? 2 * TN_BN : TN_BN);
    // batch splits: enough workgroups to fill the chip (256 CUs x 3 resident 64-row-tile workgroups, x 2 of the 128-row
    // ones), in a count that fills its last round of workgroups — 1120 workgroups on 512 slots run as 3 rounds, the last
    // one 19 % full.  One full round beats two (measured at 65536 x 64 x 1677: 54 splits 0.163 ms, 109 splits 0.168,
    // 73 splits 0.184), so a larger count must fill its rounds at least 2 % better to be chosen.
    const int64_t slots = wide ? 512 : 768;
    int64_t s = rp_cdiv(2 * slots, tiles);
    if (s > 512) s = 512;  // a 64 x 64 layer has ONE output tile: all the parallelism must come from the batch split
    if (tiles * s > slots) {
        int64_t best = s;
        double best_eff = 0.0;
        for (int64_t c = rp_cdiv(3 * slots / 4, tiles); c <= rp_cdiv(2 * slots, tiles); ++c) {
            if (c < 1 || c > 512) continue;
            const double eff = (double)(tiles * c) / (double)(rp_cdiv(tiles * c, slots) * slots);
            if (eff > best_eff + 0.02) {
                best_eff = eff;
                best = c;
            }
        }
        s = best;
    }
    if (wide && wgrad_xcd_local()) {  // XCD-local order: whole groups of eight splits (one per XCD)
        s = ((s + 4) / 8) * 8;
        if (s < 8) s = 8;
    }
    static const int s_env = getenv("RP_WGRAD_S") ? atoi(getenv("RP_WGRAD_S")) : 0;  // (profiles/microbench/wgrad_one.py)
    if (s_env > 0) s = s_env;
    int64_t r = rp_cdiv(rp_cdiv(M, s), TN_BM) * TN_BM;
    if (r < TN_BM) r = TN_BM;
    *rows = r;
    *S = (int)rp_cdiv(M > 0 ? M : 1, r);
}

extern "C" int rp_linear_wgrad_workspace_bytes(int64_t M, int N, int K, size_t *bytes) {
    RP_REQUIRE(bytes && M >= 0 && N >= 1 && K >= 1, "linear_wgrad_workspace_bytes: bad argument");
    int S;
    int64_t rows;
    wgrad_plan(M, N, K, &S, &rows);
    *bytes = ((size_t)S * N * K + (size_t)S * N) * sizeof(float) + 256;
    return RP_OK;
}

extern "C" int rp_linear_wgrad(const float *dy, int64_t lddy, const float *x, int64_t ldx, float *dw, int64_t lddw,
                               float *db, int64_t M, int N, int K, int accumulate, void *workspace,
                               size_t workspace_bytes, rp_stream_t stream) {
    RP_REQUIRE(dy && x && dw && workspace, "linear_wgrad: null pointer");
    RP_REQUIRE(M >= 1 && N >= 1 && K >= 1, "linear_wgrad: bad M/N/K");
    RP_REQUIRE(lddy >= N && ldx >= K && lddw >= K, "linear_wgrad: leading dimension too small");
    size_t need = 0;
    rp_linear_wgrad_workspace_bytes(M, N, K, &need);
    RP_REQUIRE(workspace_bytes >= need, "linear_wgrad: workspace %zu < %zu bytes", workspace_bytes, need);
    int S;
    int64_t rows;
    wgrad_plan(M, N, K, &S, &rows);
    float *P = reinterpret_cast<float *>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
    float *Pb = P + (size_t)S * N * K;
    const bool vy = (lddy % 4 == 0) && rp_aligned16(dy);
    const bool vx = (ldx % 4 == 0) && rp_aligned16(x);
    dim3 grid((unsigned)rp_cdiv(K, TN_BK), (unsigned)rp_cdiv(N, TN_BN), (unsigned)S);
    hipStream_t s = (hipStream_t)stream;
    const int mode = linear_mode(M, N, K);
    if (mode != RP_MATMUL_FP32) {
        const bool vx2 = (ldx % 2 == 0) && (reinterpret_cast<uintptr_t>(x) % 8 == 0);
        const bool wide = wgrad_wide(N);
        dim3 gridb((unsigned)rp_cdiv(K, TN_BK), (unsigned)rp_cdiv(N, wide ? 2 * TN_BN : TN_BN), (unsigned)S);
        const bool xl = wide && wgrad_xcd_local();
        const int xt = xl ? (int)(gridb.x * gridb.y) : 0, xk = (int)gridb.x;
        if (xl) gridb = dim3((unsigned)(8 * rp_cdiv(S, 8) * xt), 1u, 1u);
#define CALLW(NPROD, NA, PF, VX)                                                                                        \
    hipLaunchKernelGGL((linear_wgrad_bf16_kernel<NPROD, NA, PF, VX>), gridb, dim3(256), 0, s, dy, lddy, x, ldx, P, Pb, M, \
                       N, K, rows, (const float *)nullptr, (const int32_t *)nullptr, (int64_t)0, 0, xt, xk, S)
#define CALLB(NPROD)                          \
    do {                                      \
        if (wide && vx2) CALLW(NPROD, 2, 2, true);   \
        else if (wide) CALLW(NPROD, 2, 2, false);    \
        else if (vx2) CALLW(NPROD, 1, 1, true);      \
        else CALLW(NPROD, 1, 1, false);              \
    } while (0)
        if (mode == RP_MATMUL_BF16X6) CALLB(6);
        else if (mode == RP_MATMUL_BF16X3) CALLB(3);
        else CALLB(1);
#undef CALLB
#undef CALLW
        RP_LAUNCH_CHECK("linear_wgrad partial (bf16 split)");
    } else {
#define CALL(VY, VX)                                                                                                \
    hipLaunchKernelGGL((linear_wgrad_partial_kernel<VY, VX>), grid, dim3(256), 0, s, dy, lddy, x, ldx, P, Pb, M, N, K, \
                       rows)
    if (vy && vx) CALL(true, true);
    else if (vy) CALL(true, false);
    else if (vx) CALL(false, true);
    else CALL(false, false);
#undef CALL
    RP_LAUNCH_CHECK("linear_wgrad partial");
    }
    const int64_t total = (int64_t)N * K + N;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)rp_cdiv(total, 16)), dim3(256), 0, s, P, Pb, S, N, K, dw,
                       lddw, db, accumulate);
    RP_LAUNCH_CHECK("linear_wgrad reduce");
    return RP_OK;
}

extern "C" int rp_linear_wgrad_xbf16(const float *dy, int64_t lddy, const void *x_bf16, int64_t ldx, float *dw, int64_t lddw,
                                     float *db, int64_t M, int N, int K, int accumulate, void *workspace,
                                     size_t workspace_bytes, rp_stream_t stream) {
    RP_REQUIRE(dy && x_bf16 && dw && workspace, "linear_wgrad_xbf16: null pointer");
    RP_REQUIRE(M >= 1 && N >= 1 && K >= 1, "linear_wgrad_xbf16: bad M/N/K");
    RP_REQUIRE(lddy >= N && ldx >= K && lddw >= K, "linear_wgrad_xbf16: leading dimension too small");
    const int mode = linear_mode(M, N, K);
    if (mode == RP_MATMUL_FP32 || wgrad_wide(N) || ldx % 2 != 0 || (reinterpret_cast<uintptr_t>(x_bf16) & 3u) != 0)
        return rp_fail(RP_ERR_UNSUPPORTED, "linear_wgrad_xbf16: bf16 matrix-core modes, N <= 128, ldx even, 4-byte aligned x only");
    size_t need = 0;
    rp_linear_wgrad_workspace_bytes(M, N, K, &need);
    RP_REQUIRE(workspace_bytes >= need, "linear_wgrad_xbf16: workspace %zu < %zu bytes", workspace_bytes, need);
    int S;
    int64_t rows;
    wgrad_plan(M, N, K, &S, &rows);
    float *P = reinterpret_cast<float *>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
    float *Pb = P + (size_t)S * N * K;
    dim3 gridb((unsigned)rp_cdiv(K, TN_BK), (unsigned)rp_cdiv(N, TN_BN), (unsigned)S);
    hipStream_t s = (hipStream_t)stream;
    const float *xf = reinterpret_cast<const float *>(x_bf16);
#define CALLX(NPROD)                                                                                                       \
    hipLaunchKernelGGL((linear_wgrad_bf16_kernel<NPROD, 1, 1, true, false, true>), gridb, dim3(256), 0, s, dy, lddy, xf, ldx, P, \
                       Pb, M, N, K, rows, (const float *)nullptr, (const int32_t *)nullptr, (int64_t)0, 0, 0, 0, S)
    if (mode == RP_MATMUL_BF16X6) CALLX(6);
    else if (mode == RP_MATMUL_BF16X3) CALLX(3);
    else CALLX(1);
#undef CALLX
    RP_LAUNCH_CHECK("linear_wgrad_xbf16 partial");
    const int64_t total = (int64_t)N * K + N;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)rp_cdiv(total, 16)), dim3(256), 0, s, P, Pb, S, N, K, dw, lddw, db,
                       accumulate);
    RP_LAUNCH_CHECK("linear_wgrad_xbf16 reduce");
    return RP_OK;
}

// dW[N, K] = dY[M, N]^T . X[M, K] where the first Kg = F * 64 columns of X are the embedding rows of the batch, gathered
// from the arena through the keys the forward saved (keys[f * M + m] = arena row of field f of sample m) instead of being
// read from a stored activation buffer, and the remaining K - Kg <= 64 columns (the dense features) come from xd [M, ldxd].
// N = 64 (the fused first layer), F even, bf16 matrix-core modes only (rp_linear_wgrad_gather_fits).
extern "C" int rp_linear_wgrad_gather_fits(int64_t M, int N, int K, int Kg) {
    return (g_matmul_precision != RP_MATMUL_FP32 && N == TN_BN && Kg >= TN_BK && Kg % TN_BK == 0 && K >= Kg &&
            K - Kg <= 64 && M >= 1) ? 1 : 0;
}

extern "C" int rp_linear_wgrad_gather(const float *dy, int64_t lddy, const float *arena, const int32_t *keys, int Kg,
                                      const float *xd, int64_t ldxd, float *dw, int64_t lddw, float *db, int64_t M, int N,
                                      int K, int accumulate, void *workspace, size_t workspace_bytes, rp_stream_t stream) {
    RP_REQUIRE(dy && arena && keys && dw && workspace, "linear_wgrad_gather: null pointer");
    RP_REQUIRE(K == Kg || xd, "linear_wgrad_gather: the dense columns need xd");
    if (!rp_linear_wgrad_gather_fits(M, N, K, Kg) || !rp_aligned16(arena))
        return rp_fail(RP_ERR_UNSUPPORTED, "linear_wgrad_gather: needs N = 64, F*64 a multiple of 128, <= 64 dense columns, a "
                                           "bf16 matrix-core mode");
    RP_REQUIRE(lddy >= N && lddw >= K && (K == Kg || ldxd >= K - Kg), "linear_wgrad_gather: leading dimension too small");
    size_t need = 0;
    rp_linear_wgrad_workspace_bytes(M, N, K, &need);
    RP_REQUIRE(workspace_bytes >= need, "linear_wgrad_gather: workspace %zu < %zu bytes", workspace_bytes, need);
    int S;
    int64_t rows;
    wgrad_plan(M, N, K, &S, &rows);
    float *P = reinterpret_cast<float *>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
    float *Pb = P + (size_t)S * N * K;
    hipStream_t s = (hipStream_t)stream;
    const int mode = linear_mode(M, N, K);
    dim3 gridb((unsigned)rp_cdiv(K, TN_BK), 1u, (unsigned)S);
#define CALLG(NPROD)                                                                                                        \
    hipLaunchKernelGGL((linear_wgrad_bf16_kernel<NPROD, 1, 1, true, true>), gridb, dim3(256), 0, s, dy, lddy, xd, ldxd, P, Pb, \
                       M, N, K, rows, arena, keys, M, Kg)
    if (mode == RP_MATMUL_BF16X6) CALLG(6);
    else if (mode == RP_MATMUL_BF16X3) CALLG(3);
    else CALLG(1);
#undef CALLG
    RP_LAUNCH_CHECK("linear_wgrad_gather partial");
    const int64_t total = (int64_t)N * K + N;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)rp_cdiv(total, 16)), dim3(256), 0, s, P, Pb, S, N, K, dw, lddw, db,
                       accumulate);
    RP_LAUNCH_CHECK("linear_wgrad_gather reduce");
    return RP_OK;
}

extern "C" int rp_transpose(const float *in, int64_t ldin, float *out, int64_t ldout, int R, int C, int C_out,
                            rp_stream_t stream) {
    RP_REQUIRE(in && out && R >= 1 && C >= 1 && C_out >= C && ldin >= C && ldout >= R, "transpose: bad argument");
    dim3 grid((unsigned)rp_cdiv(C_out, 32), (unsigned)rp_cdiv(R, 32));
    hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, (hipStream_t)stream, in, ldin, out, ldout, R, C, C_out);
    RP_LAUNCH_CHECK("transpose");
    return RP_OK;
}

// out[r, 0:C] = in[r, 0:C] with another row stride (the [N, ceil4(K)] staging copy of a weight whose rows are not 16-byte
// aligned: functional._rows16) — a library launch, so that a recorded launch plan of a training step holds it
__global__ __launch_bounds__(256) void copy_rows_kernel(const float *__restrict__ in, int64_t ldin, float *__restrict__ out,
                                                        int64_t ldout, int R, int C) {
    const int64_t total = (int64_t)R * C;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t r = e / C;
        const int c = (int)(e - r * C);
        out[r * ldout + c] = in[r * ldin + c];
    }
}

extern "C" int rp_transpose_copy(const float *in, int64_t ldin, float *out, int64_t ldout, int R, int C, int C_out,
                                 float *copy, int64_t ldcopy, rp_stream_t stream) {
    RP_REQUIRE(in && out && copy && R >= 1 && C >= 1 && C_out >= C && ldin >= C && ldout >= R && ldcopy >= C,
               "transpose_copy: bad argument");
    const dim3 grid((unsigned)rp_cdiv(C_out, 32), (unsigned)rp_cdiv(R, 32));
    hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, (hipStream_t)stream, in, ldin, out, ldout, R, C, C_out, copy, ldcopy);
    RP_LAUNCH_CHECK("transpose_copy");
    return RP_OK;
}

extern "C" int rp_copy_rows(const float *in, int64_t ldin, float *out, int64_t ldout, int R, int C, rp_stream_t stream) {
    RP_REQUIRE(in && out && R >= 1 && C >= 1 && ldin >= C && ldout >= C, "copy_rows: bad argument");
    int64_t blocks = rp_cdiv((int64_t)R * C, 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(copy_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, in, ldin, out, ldout, R, C);
    RP_LAUNCH_CHECK("copy_rows");
    return RP_OK;
}

// out[r, 0:C] += in[r, 0:C] with independent row strides (a second consumer's gradient of a column block of x added into the
// first consumer's: autograd would sum two full-width tensors with an ATen launch and a zero-filled slice gradient)
__global__ __launch_bounds__(256) void add_rows_kernel(const float *__restrict__ in, int64_t ldin, float *__restrict__ out,
                                                       int64_t ldout, int R, int C) {
    const int64_t total = (int64_t)R * C;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t r = e / C;
        const int c = (int)(e - r * C);
        out[r * ldout + c] += in[r * ldin + c];
    }
}

extern "C" int rp_add_rows(const float *in, int64_t ldin, float *out, int64_t ldout, int R, int C, rp_stream_t stream) {
    RP_REQUIRE(in && out && R >= 1 && C >= 1 && ldin >= C && ldout >= C, "add_rows: bad argument");
    int64_t blocks = rp_cdiv((int64_t)R * C, 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(add_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, in, ldin, out, ldout, R, C);
    RP_LAUNCH_CHECK("add_rows");
    return RP_OK;
}

__global__ __launch_bounds__(256) void act_fwd_kernel(const float *__restrict__ x, int64_t ldx, float *__restrict__ y,
                                                      int64_t ldy, int64_t M, int N, int act) {
    const int64_t total = M * N;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t r = e / N;
        const int c = (int)(e - r * N);
        y[r * ldy + c] = rp_act_apply(act, x[r * ldx + c]);
    }
}

__global__ __launch_bounds__(256) void act_bwd_kernel(const float *__restrict__ dy, int64_t lddy, const float *__restrict__ ya,
                                                      int64_t ldact, float *__restrict__ out, int64_t ldo, int64_t M, int N,
                                                      int act) {
    const int64_t total = M * N;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t r = e / N;
        const int c = (int)(e - r * N);
        out[r * ldo + c] = rp_act_grad(act, dy[r * lddy + c], ya[r * ldact + c]);
    }
}

static bool act_kind_ok(int act) { return act == RP_ACT_RELU || act == RP_ACT_TANH || act == RP_ACT_SIGMOID || act == RP_ACT_LEAKY; }

extern "C" int rp_act_fwd(const float *x, int64_t ldx, float *y, int64_t ldy, int64_t M, int N, int act, rp_stream_t stream) {
    RP_REQUIRE(x && y && M >= 0 && N >= 1 && ldx >= N && ldy >= N && act_kind_ok(act), "act_fwd: bad argument");
    if (M == 0) return RP_OK;
    int64_t blocks = rp_cdiv(M * N, 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(act_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, ldx, y, ldy, M, N, act);
    RP_LAUNCH_CHECK("act_fwd");
    return RP_OK;
}

extern "C" int rp_act_bwd(const float *dy, int64_t lddy, const float *act_out, int64_t ldact, float *out, int64_t ldo,
                          int64_t M, int N, int act, rp_stream_t stream) {
    RP_REQUIRE(dy && act_out && out && M >= 0 && N >= 1 && act_kind_ok(act), "act_bwd: bad argument");
    if (M == 0) return RP_OK;
    int64_t blocks = rp_cdiv(M * N, 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(act_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, dy, lddy, act_out, ldact, out, ldo,
                       M, N, act);
    RP_LAUNCH_CHECK("act_bwd");
    return RP_OK;
}

extern "C" int rp_relu_bwd(const float *dy, int64_t lddy, const float *act_out, int64_t ldact, float *out,
                           int64_t ldo, int64_t M, int N, rp_stream_t stream) {
    RP_REQUIRE(dy && act_out && out && M >= 0 && N >= 1, "relu_bwd: bad argument");
    if (M == 0) return RP_OK;
    int64_t blocks = rp_cdiv(M * N, 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(relu_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, dy, lddy, act_out,
                       ldact, out, ldo, M, N);
    RP_LAUNCH_CHECK("relu_bwd");
    return RP_OK;
}
