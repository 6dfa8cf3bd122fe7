// Stand-alone FM pooling on a [B,F,D] tensor and BatchNorm1d (training statistics) for the MMOE towers, gfx950.
//
// rp_fm_pool_*   reference: layers/interaction.py:36-44 (InnerProductLayer).  In DeepFM/FM the second-order term
//                comes out of the gather kernel; these entry points serve a caller that already holds [B,F,D]
//                (product_sum_pooling -> [B,1], Bi_interaction_pooling -> [B,D]).  HBM-bound: one read of the tensor.
// rp_batchnorm_* reference: multi_task/mmoe.py:54 (nn.BatchNorm1d between the tower Linears), training mode:
//                per-column mean / biased variance over the batch, y = (x-mean)/sqrt(var+eps)*gamma+beta.
//                Column statistics are two-stage deterministic reductions (row-chunk partials, then a fixed-order
//                sum); variance is taken around the mean in a second pass (no E[x^2]-E[x]^2 cancellation).
#include "common.h"

// ------------------------------------------------------------------------------------------------ FM pooling
// TPR lanes per sample, each lane owns 4 consecutive d (float4) — same decomposition as the gather.
template <int TPR>
__global__ __launch_bounds__(256) void fm_pool_fwd_kernel(const float *__restrict__ x, int64_t ldb, int F, int D,
                                                          float *__restrict__ out_sum, float *__restrict__ out_bi,
                                                          int64_t B) {
    constexpr int SPB = 256 / TPR;
    const int t = threadIdx.x % TPR;
    const int64_t b = (int64_t)blockIdx.x * SPB + threadIdx.x / TPR;
    if (b >= B) return;
    float tot = 0.f;
    for (int c = t * 4; c < D; c += TPR * 4) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f}, q = {0.f, 0.f, 0.f, 0.f};
        for (int f = 0; f < F; ++f) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(x + b * ldb + (int64_t)f * D + c);
            s += v;
            q += v * v;
        }
        const f32x4 bi = (s * s - q) * 0.5f;
        if (out_bi != nullptr) *reinterpret_cast<f32x4 *>(out_bi + b * D + c) = bi;
        tot += (bi.x + bi.y) + (bi.z + bi.w);
    }
    if (out_sum != nullptr) {
#pragma unroll
        for (int o = TPR / 2; o > 0; o >>= 1) tot += __shfl_xor(tot, o, TPR);
        if (t == 0) out_sum[b] = tot;
    }
}

// dx[b,f,:] = g[b,:] * (S[b,:] - x[b,f,:]),  g = g_sum[b] broadcast over d and/or g_bi[b,:]
template <int TPR>
__global__ __launch_bounds__(256) void fm_pool_bwd_kernel(const float *__restrict__ x, int64_t ldb, int F, int D,
                                                          const float *__restrict__ g_sum,
                                                          const float *__restrict__ g_bi, float *__restrict__ dx,
                                                          int64_t lddx, int64_t B) {
    constexpr int SPB = 256 / TPR;
    const int t = threadIdx.x % TPR;
    const int64_t b = (int64_t)blockIdx.x * SPB + threadIdx.x / TPR;
    if (b >= B) return;
    const float gs = (g_sum != nullptr) ? g_sum[b] : 0.f;
    for (int c = t * 4; c < D; c += TPR * 4) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        for (int f = 0; f < F; ++f) s += *reinterpret_cast<const f32x4 *>(x + b * ldb + (int64_t)f * D + c);
        f32x4 g = {gs, gs, gs, gs};
        if (g_bi != nullptr) g += *reinterpret_cast<const f32x4 *>(g_bi + b * D + c);
        for (int f = 0; f < F; ++f) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(x + b * ldb + (int64_t)f * D + c);
            *reinterpret_cast<f32x4 *>(dx + b * lddx + (int64_t)f * D + c) = g * (s - v);
        }
    }
}

static int fm_tpr(int D) {
    int need = D / 4, tpr = 1;
    while (tpr < need && tpr < 64) tpr <<= 1;
    return tpr;
}

#define FM_DISPATCH(tpr, CALL)    \
    switch (tpr) {                \
        case 1: CALL(1); break;   \
        case 2: CALL(2); break;   \
        case 4: CALL(4); break;   \
        case 8: CALL(8); break;   \
        case 16: CALL(16); break; \
        case 32: CALL(32); break; \
        default: CALL(64); break; \
    }

extern "C" int rp_fm_pool_fwd(const float *x, int64_t ldb, int F, int D, float *out_sum, float *out_bi, int64_t B,
                              rp_stream_t stream) {
    RP_REQUIRE(x && (out_sum || out_bi) && F >= 1 && B >= 0 && ldb >= (int64_t)F * D, "fm_pool_fwd: bad argument");
    RP_REQUIRE(D >= 4 && D % 4 == 0 && ldb % 4 == 0 && rp_aligned16(x) && (!out_bi || rp_aligned16(out_bi)),
               "fm_pool_fwd: D and the sample stride must be multiples of 4 floats, pointers 16-byte aligned");
    if (B == 0) return RP_OK;
    const int tpr = fm_tpr(D);
    const unsigned grid = (unsigned)rp_cdiv(B, 256 / tpr);
    hipStream_t s = (hipStream_t)stream;
#define CALL(T) hipLaunchKernelGGL((fm_pool_fwd_kernel<T>), dim3(grid), dim3(256), 0, s, x, ldb, F, D, out_sum, out_bi, B)
    FM_DISPATCH(tpr, CALL)
#undef CALL
    RP_LAUNCH_CHECK("fm_pool_fwd");
    return RP_OK;
}

extern "C" int rp_fm_pool_bwd(const float *x, int64_t ldb, int F, int D, const float *g_sum, const float *g_bi,
                              float *dx, int64_t lddx, int64_t B, rp_stream_t stream) {
    RP_REQUIRE(x && dx && (g_sum || g_bi) && F >= 1 && B >= 0 && ldb >= (int64_t)F * D && lddx >= (int64_t)F * D,
               "fm_pool_bwd: bad argument");
    RP_REQUIRE(D >= 4 && D % 4 == 0 && ldb % 4 == 0 && lddx % 4 == 0 && rp_aligned16(x) && rp_aligned16(dx) &&
                   (!g_bi || rp_aligned16(g_bi)), "fm_pool_bwd: alignment");
    if (B == 0) return RP_OK;
    const int tpr = fm_tpr(D);
    const unsigned grid = (unsigned)rp_cdiv(B, 256 / tpr);
    hipStream_t s = (hipStream_t)stream;
#define CALL(T) \
    hipLaunchKernelGGL((fm_pool_bwd_kernel<T>), dim3(grid), dim3(256), 0, s, x, ldb, F, D, g_sum, g_bi, dx, lddx, B)
    FM_DISPATCH(tpr, CALL)
#undef CALL
    RP_LAUNCH_CHECK("fm_pool_bwd");
    return RP_OK;
}

// ------------------------------------------------------------------------------------------------ BatchNorm1d
// Thread layout shared by every kernel below: a 256-thread block is R rows x NC columns (NC = 2^ncl <= 256 covers
// min(N,256) columns, R = 256/NC), so a wave reads consecutive floats of one or more rows (coalesced) and a thread
// keeps ONE column: its mean/rstd/gamma/beta live in registers and no index division is needed.
struct BnPlan {
    int ncl;     // log2(NC)
    int rows;    // rows per partial block (multiple of R)
    int nblk;    // partial blocks over M
    int ncolblk; // column groups of NC
};

static BnPlan bn_plan(int64_t M, int N) {
    BnPlan p;
    p.ncl = 0;
    while ((1 << p.ncl) < N && p.ncl < 8) ++p.ncl;
    const int R = 256 >> p.ncl;
    int64_t rows = rp_cdiv(M, 512);  // <= 512 partial blocks: the finish kernel sums them in a fixed order
    if (rows < 4 * R) rows = 4 * R;
    rows = rp_cdiv(rows, R) * R;
    p.rows = (int)rows;
    p.nblk = (int)rp_cdiv(M, rows);
    p.ncolblk = (int)rp_cdiv(N, 1 << p.ncl);
    return p;
}

// partial[blk][0][n], partial[blk][1][n] over the block's rows.  MODE 0: {sum x, -}; 1: {sum (x-mean)^2, -};
// 2: {sum dy, sum dy*xhat}
template <int MODE>
__global__ __launch_bounds__(256) void bn_partial_kernel(const float *__restrict__ x, int64_t ldx,
                                                         const float *__restrict__ dy, int64_t lddy,
                                                         const float *__restrict__ mean,
                                                         const float *__restrict__ rstd, int64_t M, int N, int ncl,
                                                         int rows, float *__restrict__ partial) {
    __shared__ float red[2][256];
    const int NC = 1 << ncl, R = 256 >> ncl;
    const int c = threadIdx.x & (NC - 1), r = threadIdx.x >> ncl;
    const int n = blockIdx.y * NC + c;
    const int64_t m0 = (int64_t)blockIdx.x * rows;
    int64_t m1 = m0 + rows;
    if (m1 > M) m1 = M;
    float a = 0.f, b = 0.f;
    if (n < N) {
        const float mu = (MODE >= 1) ? mean[n] : 0.f;
        const float rs = (MODE == 2) ? rstd[n] : 0.f;
        const float *xp = x + n;
        const float *gp = (MODE == 2) ? dy + n : nullptr;
#pragma unroll 4
        for (int64_t m = m0 + r; m < m1; m += R) {
            const float xv = xp[m * ldx];
            if (MODE == 0) {
                a += xv;
            } else if (MODE == 1) {
                const float dlt = xv - mu;
                a += dlt * dlt;
            } else {
                const float g = gp[m * lddy];
                a += g;
                b += g * ((xv - mu) * rs);
            }
        }
    }
    red[0][threadIdx.x] = a;
    red[1][threadIdx.x] = b;
    __syncthreads();
    if (r == 0 && n < N) {
        for (int rr = 1; rr < R; ++rr) {  // fixed order -> deterministic
            a += red[0][rr * NC + c];
            b += red[1][rr * NC + c];
        }
        partial[((int64_t)blockIdx.x * 2 + 0) * N + n] = a;
        if (MODE == 2) partial[((int64_t)blockIdx.x * 2 + 1) * N + n] = b;
    }
}

// 16 columns x 16 slices per block; slice s sums partial blocks s, s+16, ...; slices combine in order.
// mode 0: out0 = a*scale;  1: out0 = var = a*scale, out1 = rsqrt(var+eps);
// mode 2: out0 = a (dbeta), out1 = b (dgamma), out2 = a*scale (mean dy), out3 = b*scale (mean dy*xhat)
__global__ __launch_bounds__(256) void bn_finish_kernel(const float *__restrict__ partial, int nblk, int N, float scale,
                                                        float eps, int mode, float *__restrict__ out0,
                                                        float *__restrict__ out1, float *__restrict__ out2,
                                                        float *__restrict__ out3) {
    __shared__ float red[2][256];
    const int c = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int n = blockIdx.x * 16 + c;
    float a = 0.f, b = 0.f;
    if (n < N) {
#pragma unroll 8
        for (int k = sl; k < nblk; k += 16) {
            a += partial[((int64_t)k * 2 + 0) * N + n];
            if (mode == 2) b += partial[((int64_t)k * 2 + 1) * N + n];
        }
    }
    red[0][threadIdx.x] = a;
    red[1][threadIdx.x] = b;
    __syncthreads();
    if (sl != 0 || n >= N) return;
    for (int q = 1; q < 16; ++q) {
        a += red[0][q * 16 + c];
        b += red[1][q * 16 + c];
    }
    if (mode == 0) {
        out0[n] = a * scale;
    } else if (mode == 1) {
        const float var = a * scale;
        out0[n] = var;
        out1[n] = 1.f / sqrtf(var + eps);
    } else {
        out0[n] = a;
        out1[n] = b;
        out2[n] = a * scale;
        out3[n] = b * scale;
    }
}

// MODE 0: y = (x-mean)*rstd*gamma+beta
// MODE 1: dx = gamma*rstd * (dy - mean(dy) - xhat*mean(dy*xhat))          (training backward)
// MODE 2: dx = dy*gamma*rstd                                               (statistics are constants)
template <int MODE>
__global__ __launch_bounds__(256) void bn_elem_kernel(const float *__restrict__ x, int64_t ldx,
                                                      const float *__restrict__ dy, int64_t lddy,
                                                      const float *__restrict__ mean, const float *__restrict__ rstd,
                                                      const float *__restrict__ gamma, const float *__restrict__ beta,
                                                      const float *__restrict__ mdy, const float *__restrict__ mdyx,
                                                      float *__restrict__ out, int64_t ldo, int64_t M, int N, int ncl) {
    const int NC = 1 << ncl, R = 256 >> ncl;
    const int c = threadIdx.x & (NC - 1), r = threadIdx.x >> ncl;
    const int n = blockIdx.y * NC + c;
    if (n >= N) return;
    const float g = gamma != nullptr ? gamma[n] : 1.f;
    const float rs = rstd[n];
    const float mu = (MODE != 2) ? mean[n] : 0.f;
    const float k0 = rs * g;
    float k1 = 0.f, k2 = 0.f;
    if (MODE == 0) k1 = beta != nullptr ? beta[n] : 0.f;
    if (MODE == 1) {
        k1 = mdy[n];
        k2 = mdyx[n];
    }
    const int64_t step = (int64_t)gridDim.x * R;
#pragma unroll 4
    for (int64_t m = (int64_t)blockIdx.x * R + r; m < M; m += step) {
        if (MODE == 0) {
            out[m * ldo + n] = (x[m * ldx + n] - mu) * k0 + k1;
        } else if (MODE == 1) {
            const float xh = (x[m * ldx + n] - mu) * rs;
            out[m * ldo + n] = k0 * (dy[m * lddy + n] - k1 - xh * k2);
        } else {
            out[m * ldo + n] = dy[m * lddy + n] * k0;
        }
    }
}

static dim3 bn_elem_grid(int64_t M, const BnPlan &p) {
    const int R = 256 >> p.ncl;
    int64_t gx = rp_cdiv(M, (int64_t)R * 8);
    if (gx > 4096) gx = 4096;
    if (gx < 1) gx = 1;
    return dim3((unsigned)gx, (unsigned)p.ncolblk);
}

extern "C" int rp_batchnorm_workspace_bytes(int64_t M, int N, size_t *bytes) {
    RP_REQUIRE(bytes && M >= 1 && N >= 1, "batchnorm_workspace_bytes: bad argument");
    const BnPlan p = bn_plan(M, N);
    *bytes = ((size_t)p.nblk * 2 + 2) * N * sizeof(float) + 256;
    return RP_OK;
}

// training forward: mean[N], var[N] (biased), rstd[N] out; y = (x-mean)*rstd*gamma+beta
extern "C" int rp_batchnorm_train_fwd(const float *x, int64_t ldx, const float *gamma, const float *beta, float eps,
                                      float *y, int64_t ldy, float *mean, float *var, float *rstd, int64_t M, int N,
                                      void *workspace, size_t workspace_bytes, rp_stream_t stream) {
    RP_REQUIRE(x && y && mean && var && rstd && workspace && M >= 1 && N >= 1 && ldx >= N && ldy >= N,
               "batchnorm_train_fwd: bad argument");
    size_t need = 0;
    rp_batchnorm_workspace_bytes(M, N, &need);
    RP_REQUIRE(workspace_bytes >= need, "batchnorm_train_fwd: workspace %zu < %zu", workspace_bytes, need);
    float *P = reinterpret_cast<float *>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
    const BnPlan p = bn_plan(M, N);
    dim3 pg((unsigned)p.nblk, (unsigned)p.ncolblk), fg((unsigned)rp_cdiv(N, 16));
    hipStream_t s = (hipStream_t)stream;
    const float inv = 1.f / (float)M;
    hipLaunchKernelGGL((bn_partial_kernel<0>), pg, dim3(256), 0, s, x, ldx, nullptr, 0, nullptr, nullptr, M, N, p.ncl,
                       p.rows, P);
    RP_LAUNCH_CHECK("batchnorm mean partial");
    hipLaunchKernelGGL(bn_finish_kernel, fg, dim3(256), 0, s, P, p.nblk, N, inv, eps, 0, mean, nullptr, nullptr, nullptr);
    RP_LAUNCH_CHECK("batchnorm mean");
    hipLaunchKernelGGL((bn_partial_kernel<1>), pg, dim3(256), 0, s, x, ldx, nullptr, 0, mean, nullptr, M, N, p.ncl,
                       p.rows, P);
    RP_LAUNCH_CHECK("batchnorm var partial");
    hipLaunchKernelGGL(bn_finish_kernel, fg, dim3(256), 0, s, P, p.nblk, N, inv, eps, 1, var, rstd, nullptr, nullptr);
    RP_LAUNCH_CHECK("batchnorm var");
    hipLaunchKernelGGL((bn_elem_kernel<0>), bn_elem_grid(M, p), dim3(256), 0, s, x, ldx, nullptr, 0, mean, rstd, gamma,
                       beta, nullptr, nullptr, y, ldy, M, N, p.ncl);
    RP_LAUNCH_CHECK("batchnorm apply");
    return RP_OK;
}

// training backward: dgamma[N] = sum dy*xhat, dbeta[N] = sum dy, dx as above
extern "C" int rp_batchnorm_train_bwd(const float *x, int64_t ldx, const float *dy, int64_t lddy, const float *mean,
                                      const float *rstd, const float *gamma, float *dx, int64_t lddx, float *dgamma,
                                      float *dbeta, int64_t M, int N, void *workspace, size_t workspace_bytes,
                                      rp_stream_t stream) {
    RP_REQUIRE(x && dy && mean && rstd && dx && dgamma && dbeta && workspace && M >= 1 && N >= 1,
               "batchnorm_train_bwd: bad argument");
    RP_REQUIRE(ldx >= N && lddy >= N && lddx >= N, "batchnorm_train_bwd: bad ld");
    size_t need = 0;
    rp_batchnorm_workspace_bytes(M, N, &need);
    RP_REQUIRE(workspace_bytes >= need, "batchnorm_train_bwd: workspace %zu < %zu", workspace_bytes, need);
    float *P = reinterpret_cast<float *>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
    const BnPlan p = bn_plan(M, N);
    float *mdy = P + (size_t)p.nblk * 2 * N, *mdyx = mdy + N;  // the two mean vectors the dx formula needs
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL((bn_partial_kernel<2>), dim3((unsigned)p.nblk, (unsigned)p.ncolblk), dim3(256), 0, s, x, ldx, dy,
                       lddy, mean, rstd, M, N, p.ncl, p.rows, P);
    RP_LAUNCH_CHECK("batchnorm bwd partial");
    hipLaunchKernelGGL(bn_finish_kernel, dim3((unsigned)rp_cdiv(N, 16)), dim3(256), 0, s, P, p.nblk, N, 1.f / (float)M,
                       0.f, 2, dbeta, dgamma, mdy, mdyx);
    RP_LAUNCH_CHECK("batchnorm bwd sums");
    hipLaunchKernelGGL((bn_elem_kernel<1>), bn_elem_grid(M, p), dim3(256), 0, s, x, ldx, dy, lddy, mean, rstd, gamma,
                       nullptr, mdy, mdyx, dx, lddx, M, N, p.ncl);
    RP_LAUNCH_CHECK("batchnorm bwd apply");
    return RP_OK;
}

// inference / generic affine normalisation with given per-column mean and rstd (running statistics in eval mode)
extern "C" int rp_batchnorm_apply(const float *x, int64_t ldx, const float *mean, const float *rstd,
                                  const float *gamma, const float *beta, float *y, int64_t ldy, int64_t M, int N,
                                  rp_stream_t stream) {
    RP_REQUIRE(x && mean && rstd && y && M >= 0 && N >= 1 && ldx >= N && ldy >= N, "batchnorm_apply: bad argument");
    if (M == 0) return RP_OK;
    const BnPlan p = bn_plan(M, N);
    hipLaunchKernelGGL((bn_elem_kernel<0>), bn_elem_grid(M, p), dim3(256), 0, (hipStream_t)stream, x, ldx, nullptr, 0,
                       mean, rstd, gamma, beta, nullptr, nullptr, y, ldy, M, N, p.ncl);
    RP_LAUNCH_CHECK("batchnorm_apply");
    return RP_OK;
}

// dx = dy * gamma * rstd (eval-mode backward: statistics are constants)
extern "C" int rp_batchnorm_apply_bwd(const float *dy, int64_t lddy, const float *rstd, const float *gamma, float *dx,
                                      int64_t lddx, int64_t M, int N, rp_stream_t stream) {
    RP_REQUIRE(dy && rstd && dx && M >= 0 && N >= 1 && lddy >= N && lddx >= N, "batchnorm_apply_bwd: bad argument");
    if (M == 0) return RP_OK;
    const BnPlan p = bn_plan(M, N);
    hipLaunchKernelGGL((bn_elem_kernel<2>), bn_elem_grid(M, p), dim3(256), 0, (hipStream_t)stream, nullptr, 0, dy, lddy,
                       nullptr, rstd, gamma, nullptr, nullptr, nullptr, dx, lddx, M, N, p.ncl);
    RP_LAUNCH_CHECK("batchnorm_apply_bwd");
    return RP_OK;
}


// ---- building blocks of a BatchNorm whose statistics span several ranks (rec_pangu_amd/sharded.py SyncBatchNorm1d):
// the local column sums come out of the same two-stage deterministic reductions, the all-reduce between the stages is
// the caller's (RCCL), and the element-wise passes take the GLOBAL statistics.
//   rp_batchnorm_colsum     out[n] = sum_m x[m,n]                    (center == NULL)
//                           out[n] = sum_m (x[m,n] - center[n])^2     (center given: variance around the global mean)
//   rp_batchnorm_bwd_sums   dbeta[n] = sum_m dy, dgamma[n] = sum_m dy * xhat   (xhat from the given mean / rstd)
//   rp_batchnorm_bwd_apply  dx = gamma * rstd * (dy - mean_dy - xhat * mean_dyx)  with the given (global) means

// nn.BatchNorm1d's running statistics of one training forward (torch/nn/modules/batchnorm.py: exponential with `momentum`, or
// the cumulative average 1 / num_batches_tracked when momentum < 0 stands for None), as ONE launch instead of six ATen ones:
//   num_batches_tracked += 1;  running_mean = (1 - m) running_mean + m mean;  running_var = (1 - m) running_var + m var M/(M-1)
__global__ __launch_bounds__(256) void bn_running_kernel(const float *__restrict__ mean, const float *__restrict__ var,
                                                         float *__restrict__ rmean, float *__restrict__ rvar,
                                                         long long *__restrict__ tracked, float momentum, float unbias, int N) {
    const long long seen = tracked != nullptr ? tracked[0] + 1 : 1;
    const float m = momentum >= 0.f ? momentum : 1.f / (float)seen;
    for (int i = threadIdx.x; i < N; i += 256) {
        rmean[i] = __builtin_fmaf(m, mean[i], rmean[i] * (1.f - m));
        rvar[i] = __builtin_fmaf(m, var[i] * unbias, rvar[i] * (1.f - m));
    }
    __syncthreads();  // (every thread has read the counter)
    if (threadIdx.x == 0 && tracked != nullptr) tracked[0] = seen;
}

extern "C" int rp_batchnorm_update_running(const float *mean, const float *var, float *running_mean, float *running_var,
                                           int64_t *num_batches_tracked, float momentum, int64_t M, int N, rp_stream_t stream) {
    RP_REQUIRE(mean && var && running_mean && running_var && N >= 1 && M >= 1, "batchnorm_update_running: bad argument");
    const float unbias = (float)M / (float)(M > 1 ? M - 1 : 1);
    hipLaunchKernelGGL(bn_running_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, mean, var, running_mean, running_var,
                       reinterpret_cast<long long *>(num_batches_tracked), momentum, unbias, N);
    RP_LAUNCH_CHECK("batchnorm_update_running");
    return RP_OK;
}

extern "C" int rp_batchnorm_colsum(const float *x, int64_t ldx, const float *center, float *out, int64_t M, int N,
                                   void *workspace, size_t workspace_bytes, rp_stream_t stream) {
    RP_REQUIRE(x && out && workspace && M >= 1 && N >= 1 && ldx >= N, "batchnorm_colsum: bad argument");
    size_t need = 0;
    rp_batchnorm_workspace_bytes(M, N, &need);
    RP_REQUIRE(workspace_bytes >= need, "batchnorm_colsum: workspace %zu < %zu", workspace_bytes, need);
    float *P = reinterpret_cast<float *>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
    const BnPlan p = bn_plan(M, N);
    dim3 pg((unsigned)p.nblk, (unsigned)p.ncolblk), fg((unsigned)rp_cdiv(N, 16));
    hipStream_t s = (hipStream_t)stream;
    if (center == nullptr)
        hipLaunchKernelGGL((bn_partial_kernel<0>), pg, dim3(256), 0, s, x, ldx, nullptr, 0, nullptr, nullptr, M, N, p.ncl,
                           p.rows, P);
    else
        hipLaunchKernelGGL((bn_partial_kernel<1>), pg, dim3(256), 0, s, x, ldx, nullptr, 0, center, nullptr, M, N, p.ncl,
                           p.rows, P);
    RP_LAUNCH_CHECK("batchnorm colsum partial");
    hipLaunchKernelGGL(bn_finish_kernel, fg, dim3(256), 0, s, P, p.nblk, N, 1.f, 0.f, 0, out, nullptr, nullptr, nullptr);
    RP_LAUNCH_CHECK("batchnorm colsum");
    return RP_OK;
}

extern "C" int rp_batchnorm_bwd_sums(const float *x, int64_t ldx, const float *dy, int64_t lddy, const float *mean,
                                     const float *rstd, float *dgamma, float *dbeta, int64_t M, int N, void *workspace,
                                     size_t workspace_bytes, rp_stream_t stream) {
    RP_REQUIRE(x && dy && mean && rstd && dgamma && dbeta && workspace && M >= 1 && N >= 1 && ldx >= N && lddy >= N,
               "batchnorm_bwd_sums: bad argument");
    size_t need = 0;
    rp_batchnorm_workspace_bytes(M, N, &need);
    RP_REQUIRE(workspace_bytes >= need, "batchnorm_bwd_sums: workspace %zu < %zu", workspace_bytes, need);
    float *P = reinterpret_cast<float *>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
    const BnPlan p = bn_plan(M, N);
    float *mdy = P + (size_t)p.nblk * 2 * N, *mdyx = mdy + N;  // (the local means: not used by the caller)
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL((bn_partial_kernel<2>), dim3((unsigned)p.nblk, (unsigned)p.ncolblk), dim3(256), 0, s, x, ldx, dy,
                       lddy, mean, rstd, M, N, p.ncl, p.rows, P);
    RP_LAUNCH_CHECK("batchnorm bwd partial");
    hipLaunchKernelGGL(bn_finish_kernel, dim3((unsigned)rp_cdiv(N, 16)), dim3(256), 0, s, P, p.nblk, N, 1.f / (float)M,
                       0.f, 2, dbeta, dgamma, mdy, mdyx);
    RP_LAUNCH_CHECK("batchnorm bwd sums");
    return RP_OK;
}

extern "C" int rp_batchnorm_bwd_apply(const float *x, int64_t ldx, const float *dy, int64_t lddy, const float *mean,
                                      const float *rstd, const float *gamma, const float *mean_dy,
                                      const float *mean_dyx, float *dx, int64_t lddx, int64_t M, int N,
                                      rp_stream_t stream) {
    RP_REQUIRE(x && dy && mean && rstd && mean_dy && mean_dyx && dx && M >= 1 && N >= 1 && ldx >= N && lddy >= N &&
               lddx >= N, "batchnorm_bwd_apply: bad argument");
    const BnPlan p = bn_plan(M, N);
    hipLaunchKernelGGL((bn_elem_kernel<1>), bn_elem_grid(M, p), dim3(256), 0, (hipStream_t)stream, x, ldx, dy, lddy, mean,
                       rstd, gamma, nullptr, mean_dy, mean_dyx, dx, lddx, M, N, p.ncl);
    RP_LAUNCH_CHECK("batchnorm bwd apply");
    return RP_OK;
}

// dst[0:n] += src[0:n]  (partial results of launches that cannot accumulate themselves: the map chunks of a CIN layer)
__global__ __launch_bounds__(256) void accumulate_kernel(float *__restrict__ dst, const float *__restrict__ src, int64_t n4,
                                                         int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        f32x4 a = reinterpret_cast<f32x4 *>(dst)[i];
        const f32x4 b = reinterpret_cast<const f32x4 *>(src)[i];
        a += b;
        reinterpret_cast<f32x4 *>(dst)[i] = a;
    }
    for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] += src[i];
}

extern "C" int rp_accumulate(float *dst, const float *src, int64_t n, rp_stream_t stream) {
    RP_REQUIRE(dst && src && n >= 0, "accumulate: bad argument");
    if (n == 0) return RP_OK;
    const int64_t n4 = (rp_aligned16(dst) && rp_aligned16(src)) ? n / 4 : 0;
    int64_t nb = rp_cdiv(n4 > 0 ? n4 : n, 256);
    if (nb > 16384) nb = 16384;
    hipLaunchKernelGGL(accumulate_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, dst, src, n4, n);
    RP_LAUNCH_CHECK("accumulate");
    return RP_OK;
}

// ------------------------------------------------------------------------------------------------ Dice gate
// reference: layers/activation.py:10-34   p = sigmoid(BatchNorm1d(x, affine=False)),  y = p x + (1 - p) alpha x.
// The normalisation is rp_batchnorm_* (training: batch statistics; eval: running statistics); these two launches are the
// gate around it, on the normalised tensor xhat:
//   fwd   y  = x (alpha + s (1 - alpha)),  s = sigmoid(xhat)
//   bwd   dx_direct = dy (alpha + s (1 - alpha));  dxhat = dy x (1 - alpha) s (1 - s);  dal = dy x (1 - s)  (column-summed
//         by the caller with rp_batchnorm_colsum: one deterministic reduction for every gradient of a [N] parameter)
// Elementwise, HBM-bound: grid-stride over rows, one lane per column quad when N % 4 == 0 is not required (scalar lanes).
__global__ __launch_bounds__(256) void dice_gate_fwd_kernel(const float *__restrict__ x, int64_t ldx,
                                                            const float *__restrict__ xhat, int64_t ldh,
                                                            const float *__restrict__ alpha, float *__restrict__ y,
                                                            int64_t ldy, int64_t M, int N) {
    const int64_t total = M * (int64_t)N;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t m = i / N;
        const int n = (int)(i - m * N);
        const float xv = x[m * ldx + n], a = alpha[n];
        const float s = 1.f / (1.f + __expf(-xhat[m * ldh + n]));
        y[m * ldy + n] = xv * (a + s * (1.f - a));
    }
}

__global__ __launch_bounds__(256) void dice_gate_bwd_kernel(const float *__restrict__ x, int64_t ldx,
                                                            const float *__restrict__ xhat, int64_t ldh,
                                                            const float *__restrict__ alpha,
                                                            const float *__restrict__ dy, int64_t lddy,
                                                            float *__restrict__ dx_direct, float *__restrict__ dxhat,
                                                            float *__restrict__ dal, int64_t M, int N) {
    const int64_t total = M * (int64_t)N;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t m = i / N;
        const int n = (int)(i - m * N);
        const float xv = x[m * ldx + n], a = alpha[n], g = dy[m * lddy + n];
        const float s = 1.f / (1.f + __expf(-xhat[m * ldh + n]));
        dx_direct[i] = g * (a + s * (1.f - a));
        dxhat[i] = g * xv * (1.f - a) * s * (1.f - s);
        dal[i] = g * xv * (1.f - s);
    }
}

extern "C" int rp_dice_gate_fwd(const float *x, int64_t ldx, const float *xhat, int64_t ldh, const float *alpha, float *y,
                                int64_t ldy, int64_t M, int N, rp_stream_t stream) {
    RP_REQUIRE(x && xhat && alpha && y, "dice_gate_fwd: null pointer");
    RP_REQUIRE(M >= 0 && N >= 1 && ldx >= N && ldh >= N && ldy >= N, "dice_gate_fwd: bad shape");
    if (M == 0) return RP_OK;
    int64_t nb = rp_cdiv(M * (int64_t)N, 256);
    if (nb > 16384) nb = 16384;
    hipLaunchKernelGGL(dice_gate_fwd_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, x, ldx, xhat, ldh, alpha, y,
                       ldy, M, N);
    RP_LAUNCH_CHECK("dice_gate_fwd");
    return RP_OK;
}

extern "C" int rp_dice_gate_bwd(const float *x, int64_t ldx, const float *xhat, int64_t ldh, const float *alpha,
                                const float *dy, int64_t lddy, float *dx_direct, float *dxhat, float *dal, int64_t M, int N,
                                rp_stream_t stream) {
    RP_REQUIRE(x && xhat && alpha && dy && dx_direct && dxhat && dal, "dice_gate_bwd: null pointer");
    RP_REQUIRE(M >= 0 && N >= 1 && ldx >= N && ldh >= N && lddy >= N, "dice_gate_bwd: bad shape");
    if (M == 0) return RP_OK;
    int64_t nb = rp_cdiv(M * (int64_t)N, 256);
    if (nb > 16384) nb = 16384;
    hipLaunchKernelGGL(dice_gate_bwd_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, x, ldx, xhat, ldh, alpha, dy,
                       lddy, dx_direct, dxhat, dal, M, N);
    RP_LAUNCH_CHECK("dice_gate_bwd");
    return RP_OK;
}
