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
            if (act == RP_ACT_RELU)
                v = v > 0.f ? v : 0.f;
            else if (act == RP_ACT_MASK)
                v = (aux[m * ldaux + n] > 0.f) ? v : 0.f;
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

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float *__restrict__ P, const float *__restrict__ Pb,
                                                           int S, int N, int K, float *__restrict__ dw,
                                                           int64_t lddw, float *__restrict__ db, int accumulate) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t nk = (int64_t)N * K;
    if (e < nk) {
        float s = 0.f;
        for (int z = 0; z < S; ++z) s += P[(int64_t)z * nk + e];
        const int n = (int)(e / K), k = (int)(e - (int64_t)n * K);
        float *d = dw + (int64_t)n * lddw + k;
        *d = accumulate ? (*d + s) : s;
    } else if (e < nk + N && db != nullptr) {
        const int n = (int)(e - nk);
        float s = 0.f;
        for (int z = 0; z < S; ++z) s += Pb[(int64_t)z * N + n];
        db[n] = accumulate ? (db[n] + s) : s;
    }
}

__global__ __launch_bounds__(256) void transpose_kernel(const float *__restrict__ in, int64_t ldin,
                                                        float *__restrict__ out, int64_t ldout, int R, int C) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
#pragma unroll
    for (int j = 0; j < 32; j += 8) {
        const int r = r0 + ty + j, c = c0 + tx;
        if (r < R && c < C) tile[ty + j][tx] = in[(int64_t)r * ldin + c];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 32; j += 8) {
        const int c = c0 + ty + j, r = r0 + tx;
        if (r < R && c < C) out[(int64_t)c * ldout + r] = tile[tx][ty + j];
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

extern "C" int rp_linear_fwd(const float *a, int64_t lda, const float *w, int64_t ldw, const float *bias, float *out,
                             int64_t ldo, int64_t M, int N, int K, int act, const float *aux, int64_t ldaux,
                             rp_stream_t stream) {
    RP_REQUIRE(a && w && out, "linear_fwd: null pointer");
    RP_REQUIRE(M >= 0 && N >= 1 && K >= 1, "linear_fwd: bad M/N/K");
    RP_REQUIRE(lda >= K && ldw >= K && ldo >= N, "linear_fwd: leading dimension too small");
    RP_REQUIRE(act == RP_ACT_NONE || act == RP_ACT_RELU || (act == RP_ACT_MASK && aux && ldaux >= N),
               "linear_fwd: bad act/aux");
    if (M == 0) return RP_OK;
    const bool va = (lda % 4 == 0) && rp_aligned16(a);
    const bool vw = (ldw % 4 == 0) && rp_aligned16(w);
    dim3 grid((unsigned)rp_cdiv(M, BM), (unsigned)rp_cdiv(N, BN));
    hipStream_t s = (hipStream_t)stream;
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

static void wgrad_plan(int64_t M, int N, int K, int *S, int64_t *rows) {
    const int64_t tiles = rp_cdiv(K, TN_BK) * rp_cdiv(N, TN_BN);
    int64_t s = rp_cdiv(1024, tiles);
    if (s > 128) s = 128;
    if (s < 1) s = 1;
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
#define CALL(VY, VX)                                                                                                \
    hipLaunchKernelGGL((linear_wgrad_partial_kernel<VY, VX>), grid, dim3(256), 0, s, dy, lddy, x, ldx, P, Pb, M, N, K, \
                       rows)
    if (vy && vx) CALL(true, true);
    else if (vy) CALL(true, false);
    else if (vx) CALL(false, true);
    else CALL(false, false);
#undef CALL
    RP_LAUNCH_CHECK("linear_wgrad partial");
    const int64_t total = (int64_t)N * K + N;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)rp_cdiv(total, 256)), dim3(256), 0, s, P, Pb, S, N, K, dw,
                       lddw, db, accumulate);
    RP_LAUNCH_CHECK("linear_wgrad reduce");
    return RP_OK;
}

extern "C" int rp_transpose(const float *in, int64_t ldin, float *out, int64_t ldout, int R, int C,
                            rp_stream_t stream) {
    RP_REQUIRE(in && out && R >= 1 && C >= 1 && ldin >= C && ldout >= R, "transpose: bad argument");
    dim3 grid((unsigned)rp_cdiv(C, 32), (unsigned)rp_cdiv(R, 32));
    hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, (hipStream_t)stream, in, ldin, out, ldout, R, C);
    RP_LAUNCH_CHECK("transpose");
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
