// AutoInt field self-attention layer (MultiHeadSelfAttention, align_to="output"), forward and backward, gfx950.
//   reference: layers/attention.py:63-101.  Per sample, T field tokens of width Din:
//     Q,K,V = X Wq^T, X Wk^T, X Wv^T            [T, HA], HA = H*a, bias-free
//     heads split by a RAW view of the flat [T*HA] buffer: head g, token t', dim j = flat[g*T*a + t'*a + j]
//     P = softmax_rows(Qg Kg^T [/ scale]);  Og = P Vg;  O = flat view back to [T, HA]
//     out = relu(O + R),  R = X Wres^T if Din != HA else X
// The whole layer for one sample lives in LDS (X 6.6 KB + ~10 KB of intermediates at T=26, Din=64, HA=8):
// one wave per sample, NW (<= 4, as LDS allows) waves per workgroup sharing the weights in LDS; the only HBM traffic is X in,
// out back (and dX / per-block weight-gradient partials in the backward, which recomputes the forward).
// The arithmetic is tiny (64 K MAC/sample) and runs on the VALU: matrix cores cannot be fed by 26x8 tiles.
// Launch/latency-bound in the reference (5 bmm/softmax launches + 4 Linear launches per layer); here 1.
#include "common.h"

struct AttnDims {
    int T, Din, H, a, HA;
    int has_res;   // Wres present
    float inv_scale;  // 1/scale or 1
};

struct AttnLds {  // per-wave float offsets
    int X, Q, K, V, R, P, O, dZ, dP, dQ, dK, dV, total;
};

__host__ __device__ static inline AttnLds attn_layout(const AttnDims &d, bool bwd) {
    AttnLds L;
    int o = 0;
    const int TH = d.T * d.HA, PP = d.H * d.T * d.T;
    L.X = o; o += d.T * (d.Din | 1);  // odd row stride: lanes reading different rows hit different banks
    L.Q = o; o += TH;
    L.K = o; o += TH;
    L.V = o; o += TH;
    L.R = o; o += TH;
    L.P = o; o += PP;
    L.O = o; o += TH;
    L.dZ = L.dP = L.dQ = L.dK = L.dV = 0;
    if (bwd) {
        L.dZ = o; o += TH;
        L.dP = o; o += PP;
        L.dQ = o; o += TH;
        L.dK = o; o += TH;
        L.dV = o; o += TH;
    }
    L.total = (o + 3) & ~3;
    return L;
}

// forward of one sample into the wave's LDS region; every lane of the block calls this (barriers inside)
__device__ __forceinline__ void attn_forward_lds(const AttnDims &d, const AttnLds &L, float *S, const float *Ws,
                                                 const float *__restrict__ xg, bool valid, int lane) {
    const int T = d.T, Din = d.Din, HA = d.HA, a = d.a, H = d.H, TH = T * HA, Dx = Din | 1;
    for (int i = lane; i < T * Din; i += 64) {
        const int t = i / Din, k = i - t * Din;
        S[L.X + t * Dx + k] = valid ? xg[i] : 0.f;
    }
    __syncthreads();
    const int nproj = d.has_res ? 4 : 3;
    for (int o = lane; o < nproj * TH; o += 64) {
        const int p = o / TH, r = o - p * TH, t = r / HA, c = r - t * HA;
        const float *w = Ws + (p * HA + c) * Dx;
        const float *x = S + L.X + t * Dx;
        float acc = 0.f;
        for (int k = 0; k < Din; ++k) acc += x[k] * w[k];
        S[(p == 0 ? L.Q : p == 1 ? L.K : p == 2 ? L.V : L.R) + r] = acc;
    }
    if (!d.has_res)
        for (int i = lane; i < TH; i += 64) {  // Din == HA: residual is X itself
            const int t = i / HA, c = i - t * HA;
            S[L.R + i] = S[L.X + t * Dx + c];
        }
    __syncthreads();
    for (int o = lane; o < H * T * T; o += 64) {
        const int g = o / (T * T), r = o - g * T * T, tq = r / T, tk = r - tq * T;
        const float *q = S + L.Q + g * T * a + tq * a;
        const float *k = S + L.K + g * T * a + tk * a;
        float acc = 0.f;
        for (int j = 0; j < a; ++j) acc += q[j] * k[j];
        S[L.P + o] = acc * d.inv_scale;
    }
    __syncthreads();
    for (int row = lane; row < H * T; row += 64) {
        float *p = S + L.P + row * T;
        float mx = -INFINITY;
        for (int u = 0; u < T; ++u) mx = fmaxf(mx, p[u]);
        float den = 0.f;
        for (int u = 0; u < T; ++u) {
            const float e = expf(p[u] - mx);
            p[u] = e;
            den += e;
        }
        const float inv = 1.f / den;
        for (int u = 0; u < T; ++u) p[u] *= inv;
    }
    __syncthreads();
    for (int o = lane; o < TH; o += 64) {  // o is the flat index g*T*a + t'*a + j
        const int g = o / (T * a), r = o - g * T * a, tq = r / a, j = r - tq * a;
        const float *p = S + L.P + (g * T + tq) * T;
        const float *v = S + L.V + g * T * a + j;
        float acc = 0.f;
        for (int u = 0; u < T; ++u) acc += p[u] * v[u * a];
        S[L.O + o] = acc;
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void attn_fwd_kernel(AttnDims d, const float *__restrict__ x, int64_t ldx,
                                                       const float *__restrict__ W, float *__restrict__ out,
                                                       int64_t B, int NW) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const AttnLds L = attn_layout(d, false);
    const int nproj = d.has_res ? 4 : 3;
    const int wsz = nproj * d.HA * d.Din, Dx = d.Din | 1, wlds = (nproj * d.HA * Dx + 3) & ~3;
    float *Ws = smem;  // [nproj*HA][Dx]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float *S = smem + wlds + wave * L.total;
    for (int i = threadIdx.x; i < wsz; i += blockDim.x) {
        const int row = i / d.Din, k = i - row * d.Din;
        Ws[row * Dx + k] = W[i];
    }
    __syncthreads();
    const int TH = d.T * d.HA;
    for (int64_t base = (int64_t)blockIdx.x * NW; base < B; base += (int64_t)gridDim.x * NW) {
        const int64_t b = base + wave;
        const bool valid = (wave < NW) && (b < B);
        attn_forward_lds(d, L, S, Ws, x + (valid ? b : 0) * ldx, valid, lane);
        if (valid)
            for (int i = lane; i < TH; i += 64) out[b * TH + i] = fmaxf(S[L.O + i] + S[L.R + i], 0.f);
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void attn_bwd_kernel(AttnDims d, const float *__restrict__ x, int64_t ldx,
                                                       const float *__restrict__ W, const float *__restrict__ gout,
                                                       float *__restrict__ dx, int64_t lddx,
                                                       float *__restrict__ partial, int64_t B, int NW) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const AttnLds L = attn_layout(d, true);
    const int T = d.T, Din = d.Din, HA = d.HA, a = d.a, H = d.H, TH = T * HA;
    const int nproj = d.has_res ? 4 : 3;
    const int wsz = nproj * HA * Din, wpad = (wsz + 3) & ~3, Dx = Din | 1, wlds = (nproj * HA * Dx + 3) & ~3;
    float *Ws = smem;          // [nproj*HA][Dx]
    float *Acc = smem + wlds;  // [NW][wsz] per-wave weight-gradient accumulators (dense)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float *S = smem + wlds + NW * wpad + wave * L.total;
    float *A = Acc + wave * wpad;
    for (int i = threadIdx.x; i < wsz; i += blockDim.x) {
        const int row = i / Din, k = i - row * Din;
        Ws[row * Dx + k] = W[i];
    }
    for (int i = threadIdx.x; i < NW * wpad; i += blockDim.x) Acc[i] = 0.f;
    __syncthreads();
    for (int64_t base = (int64_t)blockIdx.x * NW; base < B; base += (int64_t)gridDim.x * NW) {
        const int64_t b = base + wave;
        const bool valid = (wave < NW) && (b < B);
        attn_forward_lds(d, L, S, Ws, x + (valid ? b : 0) * ldx, valid, lane);
        // dZ = gout * (out > 0);  dO (flat) = dZ;  dR = dZ
        for (int i = lane; i < TH; i += 64) {
            const float z = S[L.O + i] + S[L.R + i];
            S[L.dZ + i] = (valid && z > 0.f) ? gout[b * TH + i] : 0.f;
        }
        __syncthreads();
        // softmax backward per (head, query) row: dP = dO V^T, dS = P * (dP - sum(P*dP)) (in place in dP)
        for (int row = lane; row < H * T; row += 64) {
            const int g = row / T, tq = row - g * T;
            const float *p = S + L.P + row * T;
            float *dp = S + L.dP + row * T;
            const float *dO = S + L.dZ + g * T * a + tq * a;
            float dot = 0.f;
            for (int u = 0; u < T; ++u) {
                const float *v = S + L.V + g * T * a + u * a;
                float acc = 0.f;
                for (int j = 0; j < a; ++j) acc += dO[j] * v[j];
                dp[u] = acc;
                dot += p[u] * acc;
            }
            for (int u = 0; u < T; ++u) dp[u] = p[u] * (dp[u] - dot) * d.inv_scale;
        }
        __syncthreads();
        // dV, dQ, dK in the flat head layout (same flat index as Q/K/V)
        for (int o = lane; o < TH; o += 64) {
            const int g = o / (T * a), r = o - g * T * a, tt = r / a, j = r - tt * a;
            float dv = 0.f, dq = 0.f, dk = 0.f;
            for (int u = 0; u < T; ++u) {
                // dV[g][tt][j] = sum_tq P[g][tq=u][tt] * dO[g][u][j]
                dv += S[L.P + (g * T + u) * T + tt] * S[L.dZ + g * T * a + u * a + j];
                // dQ[g][tt][j] = sum_u dS[g][tt][u] * K[g][u][j]
                dq += S[L.dP + (g * T + tt) * T + u] * S[L.K + g * T * a + u * a + j];
                // dK[g][tt][j] = sum_tq dS[g][tq=u][tt] * Q[g][u][j]
                dk += S[L.dP + (g * T + u) * T + tt] * S[L.Q + g * T * a + u * a + j];
            }
            S[L.dV + o] = dv;
            S[L.dQ + o] = dq;
            S[L.dK + o] = dk;
        }
        __syncthreads();
        // dX[t][k] = sum_c dQ[t][c] Wq[c][k] + dK Wk + dV Wv + dZ (Wres | identity)
        if (valid && dx != nullptr) {
            for (int o = lane; o < T * Din; o += 64) {
                const int t = o / Din, k = o - t * Din;
                float acc = 0.f;
                for (int c = 0; c < HA; ++c) {
                    acc += S[L.dQ + t * HA + c] * Ws[(0 * HA + c) * Dx + k];
                    acc += S[L.dK + t * HA + c] * Ws[(1 * HA + c) * Dx + k];
                    acc += S[L.dV + t * HA + c] * Ws[(2 * HA + c) * Dx + k];
                    if (d.has_res) acc += S[L.dZ + t * HA + c] * Ws[(3 * HA + c) * Dx + k];
                }
                if (!d.has_res) acc += S[L.dZ + o];  // Din == HA
                dx[b * lddx + o] = acc;
            }
        }
        // weight gradients: dW_p[c][k] += sum_t dP_p[t][c] * X[t][k]   (per-wave accumulators, no atomics)
        for (int o = lane; o < wsz; o += 64) {
            const int p = o / (HA * Din), r = o - p * HA * Din, c = r / Din, k = r - c * Din;
            const float *gsrc = S + (p == 0 ? L.dQ : p == 1 ? L.dK : p == 2 ? L.dV : L.dZ);
            float acc = 0.f;
            for (int t = 0; t < T; ++t) acc += gsrc[t * HA + c] * S[L.X + t * Dx + k];
            A[o] += acc;  // invalid samples contribute exact zeros (X = 0, dZ = 0)
        }
        __syncthreads();
    }
    // fixed-order sum over the block's waves -> per-block partial
    for (int i = threadIdx.x; i < wsz; i += blockDim.x) {
        float s = 0.f;
        for (int w = 0; w < NW; ++w) s += Acc[w * wpad + i];
        partial[(int64_t)blockIdx.x * wsz + i] = s;
    }
}

__global__ __launch_bounds__(256) void attn_partial_sum_kernel(const float *__restrict__ partial, int nblk, int64_t n,
                                                               float *__restrict__ out) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    float s = 0.f;
    for (int k = 0; k < nblk; ++k) s += partial[(int64_t)k * n + e];
    out[e] = s;
}

#define ATTN_LDS_BUDGET (150 * 1024)

static int attn_plan(const AttnDims &d, bool bwd, int *NW, size_t *lds) {
    const AttnLds L = attn_layout(d, bwd);
    const int nproj = d.has_res ? 4 : 3;
    const size_t wpad = ((size_t)nproj * d.HA * d.Din + 3) & ~(size_t)3;
    const size_t wlds = ((size_t)nproj * d.HA * (d.Din | 1) + 3) & ~(size_t)3;
    for (int nw = 4; nw >= 1; --nw) {
        const size_t need = (wlds + (bwd ? nw * wpad : 0) + (size_t)nw * L.total) * sizeof(float);
        if (need <= ATTN_LDS_BUDGET) {
            *NW = nw;
            *lds = need;
            return RP_OK;
        }
    }
    return rp_fail(RP_ERR_UNSUPPORTED, "field attention T=%d Din=%d H=%d a=%d does not fit in LDS", d.T, d.Din, d.H, d.a);
}

static int attn_dims(AttnDims *d, int T, int Din, int H, int a, int has_res, float scale) {
    RP_REQUIRE(T >= 1 && Din >= 1 && H >= 1 && a >= 1, "attention: bad dims");
    RP_REQUIRE(has_res || Din == H * a, "attention: no W_res needs Din == H*a");
    d->T = T; d->Din = Din; d->H = H; d->a = a; d->HA = H * a; d->has_res = has_res ? 1 : 0;
    d->inv_scale = (scale > 0.f) ? 1.f / scale : 1.f;
    return RP_OK;
}

static unsigned attn_grid(int64_t B, int NW) {
    int64_t nb = rp_cdiv(B, NW);
    return (unsigned)(nb > 2048 ? 2048 : (nb < 1 ? 1 : nb));
}

// 1 if the one-launch layer fits the 160 KB LDS for these dimensions (forward AND backward), else 0
extern "C" int rp_field_attention_fits(int T, int Din, int H, int a, int has_res) {
    AttnDims d;
    if (T < 1 || Din < 1 || H < 1 || a < 1 || (!has_res && Din != H * a)) return 0;
    d.T = T; d.Din = Din; d.H = H; d.a = a; d.HA = H * a; d.has_res = has_res ? 1 : 0; d.inv_scale = 1.f;
    const AttnLds L = attn_layout(d, true);
    const size_t wpad = ((size_t)(has_res ? 4 : 3) * d.HA * Din + 3) & ~(size_t)3;
    const size_t wlds = ((size_t)(has_res ? 4 : 3) * d.HA * (Din | 1) + 3) & ~(size_t)3;
    return ((wlds + wpad + (size_t)L.total) * sizeof(float) <= ATTN_LDS_BUDGET) ? 1 : 0;
}

extern "C" int rp_field_attention_fwd(const float *x, int64_t ldx, const float *W, int T, int Din, int H, int a,
                                      int has_res, float scale, float *out, int64_t B, rp_stream_t stream) {
    RP_REQUIRE(x && W && out && B >= 0 && ldx >= (int64_t)T * Din, "field_attention_fwd: bad argument");
    AttnDims d;
    int rc = attn_dims(&d, T, Din, H, a, has_res, scale);
    if (rc != RP_OK) return rc;
    int NW;
    size_t lds;
    rc = attn_plan(d, false, &NW, &lds);
    if (rc != RP_OK) return rc;
    if (B == 0) return RP_OK;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(attn_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                        (int)lds);
    hipLaunchKernelGGL(attn_fwd_kernel, dim3(attn_grid(B, NW)), dim3(64 * NW), lds, (hipStream_t)stream, d, x, ldx, W, out,
                       B, NW);
    RP_LAUNCH_CHECK("field_attention_fwd");
    return RP_OK;
}

extern "C" int rp_field_attention_bwd_workspace_bytes(int64_t B, int T, int Din, int H, int a, int has_res,
                                                      size_t *bytes) {
    RP_REQUIRE(bytes, "field_attention_bwd_workspace_bytes: null");
    AttnDims d;
    int rc = attn_dims(&d, T, Din, H, a, has_res, 0.f);
    if (rc != RP_OK) return rc;
    int NW;
    size_t lds;
    rc = attn_plan(d, true, &NW, &lds);
    if (rc != RP_OK) return rc;
    *bytes = (size_t)attn_grid(B, NW) * (has_res ? 4 : 3) * d.HA * Din * sizeof(float) + 256;
    return RP_OK;
}

extern "C" int rp_field_attention_bwd(const float *x, int64_t ldx, const float *W, int T, int Din, int H, int a,
                                      int has_res, float scale, const float *gout, float *dx, int64_t lddx,
                                      float *dW, int64_t B, void *workspace, size_t workspace_bytes,
                                      rp_stream_t stream) {
    RP_REQUIRE(x && W && gout && dW && workspace && B >= 1, "field_attention_bwd: bad argument");
    RP_REQUIRE(ldx >= (int64_t)T * Din && (dx == nullptr || lddx >= (int64_t)T * Din), "field_attention_bwd: bad ld");
    AttnDims d;
    int rc = attn_dims(&d, T, Din, H, a, has_res, scale);
    if (rc != RP_OK) return rc;
    int NW;
    size_t lds, need;
    rc = attn_plan(d, true, &NW, &lds);
    if (rc != RP_OK) return rc;
    rp_field_attention_bwd_workspace_bytes(B, T, Din, H, a, has_res, &need);
    RP_REQUIRE(workspace_bytes >= need, "field_attention_bwd: workspace %zu < %zu", workspace_bytes, need);
    float *P = reinterpret_cast<float *>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
    const unsigned nb = attn_grid(B, NW);
    hipStream_t s = (hipStream_t)stream;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(attn_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                        (int)lds);
    hipLaunchKernelGGL(attn_bwd_kernel, dim3(nb), dim3(64 * NW), lds, s, d, x, ldx, W, gout, dx, lddx, P, B, NW);
    RP_LAUNCH_CHECK("field_attention_bwd");
    const int64_t n = (int64_t)(has_res ? 4 : 3) * d.HA * Din;
    hipLaunchKernelGGL(attn_partial_sum_kernel, dim3((unsigned)rp_cdiv(n, 256)), dim3(256), 0, s, P, (int)nb, n, dW);
    RP_LAUNCH_CHECK("field_attention_bwd reduce");
    return RP_OK;
}
