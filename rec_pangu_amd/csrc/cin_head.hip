// xDeepFM's CIN, the weight-space arithmetic around the pair kernels and the collapsed last layer (round 6) — what made the
// xDeepFM step hold ~50 ATen launches (profiles/microbench/probes/aten_sources.py): the per-step preparation of the pair kernels'
// weights (triangular fold + three bf16 pieces in two layouts: ~36 elementwise / index launches) and the head
//     V = c_last . W_L,  vb = c_last . b_L,  logit = cin_last(X_0, X_{L-1}, V) + D vb + fc.bias        (interaction.py:157-171:
//     the last layer only feeds sum-pooling and fc, both linear, and the CIN has no activation — models/layers/interaction.py)
// with its backward  dW_L = c_last^T (x) dV,  db_L = D sg c_last,  dc_last = W_L . dV + D sg b_L,  dfc.bias = sg = sum_b g[b].
// Everything here is a few hundred KB of weights: one small launch each, fixed summation orders (deterministic).
#include "common.h"

// ---- pair weights: Ws[o, p = (h <= m)] = W[o,h,m] + W[o,m,h] (W[o,h,h] on the diagonal) as three bf16 pieces (round to nearest
//      even, piece k = bf16 of what the pieces before it left) in rp_cin_pair_fwd's layout wsp [3][128][KP] and / or
//      rp_cin_pair_bwd_x's wst [3][KPT][128]; rows o >= O and pairs >= H(H+1)/2 are zero --------------------------------------
__global__ __launch_bounds__(256) void cin_pair_pieces_kernel(const float *__restrict__ W, int O, int H, int npair, int KP, int KPT,
                                                              __bf16 *__restrict__ wsp, __bf16 *__restrict__ wst) {
    const int o = threadIdx.x & 127;
    const int p = (int)blockIdx.x * 2 + (threadIdx.x >> 7);
    const int pmax = KP > KPT ? KP : KPT;
    if (p >= pmax) return;
    float v = 0.f;
    if (o < O && p < npair) {
        int h = 0, rem = p;
        while (rem >= H - h) {
            rem -= H - h;
            ++h;
        }
        const int m = h + rem;
        const float *Wo = W + (int64_t)o * H * H;
        v = Wo[h * H + m];
        if (m != h) v += Wo[m * H + h];
    }
    const __bf16 hi = (__bf16)v;
    const float r1 = v - (float)hi;
    const __bf16 mid = (__bf16)r1;
    const __bf16 lo = (__bf16)(r1 - (float)mid);
    const __bf16 pc[3] = {hi, mid, lo};
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        if (wsp != nullptr && p < KP) wsp[((int64_t)q * 128 + o) * KP + p] = pc[q];
        if (wst != nullptr && p < KPT) wst[((int64_t)q * KPT + p) * 128 + o] = pc[q];
    }
}

extern "C" int rp_cin_pair_pieces(const float *W, int O, int H, void *wsp, void *wst, rp_stream_t stream) {
    RP_REQUIRE(W && (wsp || wst), "cin_pair_pieces: null pointer");
    RP_REQUIRE(O >= 1 && O <= 128 && H >= 1 && H <= 32, "cin_pair_pieces: O = %d (1..128), H = %d (1..32)", O, H);
    const int npair = H * (H + 1) / 2;
    const int KP = (int)rp_cdiv(npair, 32) * 32, KPT = (int)rp_cdiv(npair, 128) * 128;
    const int pmax = KP > KPT ? KP : KPT;
    hipLaunchKernelGGL(cin_pair_pieces_kernel, dim3((unsigned)rp_cdiv(pmax, 2)), dim3(256), 0, (hipStream_t)stream, W, O, H, npair,
                       KP, KPT, reinterpret_cast<__bf16 *>(wsp), reinterpret_cast<__bf16 *>(wst));
    RP_LAUNCH_CHECK("cin_pair_pieces");
    return RP_OK;
}

// ---- head, forward: vt[m][h] = sum_o c[o] W_L[o][h M + m] (h < H; 0 up to 32: rp_cin_last_fwd's V^T layout), vb = sum_o c[o] b_L[o]
__global__ __launch_bounds__(256) void cin_head_params_fwd_kernel(const float *__restrict__ WL, const float *__restrict__ bL,
                                                                  const float *__restrict__ c, int O, int H, int M,
                                                                  float *__restrict__ vt, float *__restrict__ vb) {
    const int e = (int)blockIdx.x * 256 + threadIdx.x;
    if (e < M * 32) {
        const int m = e >> 5, h = e & 31;
        float s = 0.f;
        if (h < H)
            for (int o = 0; o < O; ++o) s = __builtin_fmaf(c[o], WL[(int64_t)o * H * M + h * M + m], s);
        vt[e] = s;
    }
    if (e == 0) {
        float s = 0.f;
        if (bL != nullptr)
            for (int o = 0; o < O; ++o) s = __builtin_fmaf(c[o], bL[o], s);
        *vb = s;
    }
}

extern "C" int rp_cin_head_params_fwd(const float *WL, const float *bL, const float *c, int O, int H, int M, float *vt, float *vb,
                                      rp_stream_t stream) {
    RP_REQUIRE(WL && c && vt && vb, "cin_head_params_fwd: null pointer");
    RP_REQUIRE(O >= 1 && H >= 1 && H <= 32 && M >= 1, "cin_head_params_fwd: bad O / H / M");
    hipLaunchKernelGGL(cin_head_params_fwd_kernel, dim3((unsigned)rp_cdiv((int64_t)M * 32, 256)), dim3(256), 0, (hipStream_t)stream, WL,
                       bL, c, O, H, M, vt, vb);
    RP_LAUNCH_CHECK("cin_head_params_fwd");
    return RP_OK;
}

// ---- out[b] += scale * a[0] + (b0 ? b0[0] : 0): the head's two scalars (D vb + fc.bias) onto the [B] logit ------------------
__global__ __launch_bounds__(256) void add_scalars_kernel(float *__restrict__ out, int64_t n, const float *__restrict__ a, float scale,
                                                          const float *__restrict__ b0) {
    const float add = scale * a[0] + (b0 != nullptr ? b0[0] : 0.f);
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] += add;
}

extern "C" int rp_add_scalars(float *out, int64_t n, const float *a, float scale, const float *b0, rp_stream_t stream) {
    RP_REQUIRE(n >= 0 && (n == 0 || (out && a)), "add_scalars: null pointer");
    if (n == 0) return RP_OK;
    hipLaunchKernelGGL(add_scalars_kernel, dim3((unsigned)rp_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, out, n, a, scale, b0);
    RP_LAUNCH_CHECK("add_scalars");
    return RP_OK;
}

// ---- out[0] = sum_i x[i] in a fixed order (one workgroup: 1024 partial sums over strided elements, then a tree) -------------
__global__ __launch_bounds__(1024) void sum_all_kernel(const float *__restrict__ x, int64_t n, float *__restrict__ out) {
    __shared__ float part[1024];
    float s = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += 1024) s += x[i];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int d = 512; d > 0; d >>= 1) {
        if ((int)threadIdx.x < d) part[threadIdx.x] += part[threadIdx.x + d];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = part[0];
}

extern "C" int rp_sum_all(const float *x, int64_t n, float *out, rp_stream_t stream) {
    RP_REQUIRE(out && n >= 0 && (n == 0 || x), "sum_all: null pointer");
    hipLaunchKernelGGL(sum_all_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, x, n, out);
    RP_LAUNCH_CHECK("sum_all");
    return RP_OK;
}

// ---- head, backward: one workgroup per output channel o --------------------------------------------------------------------
//      dWL[o][j] = c[o] dV[j],   dbL[o] = Dscale sg c[o],   dc[o] = sum_j WL[o][j] dV[j] + Dscale sg bL[o]        (j = h M + m)
__global__ __launch_bounds__(256) void cin_head_params_bwd_kernel(const float *__restrict__ WL, const float *__restrict__ bL,
                                                                  const float *__restrict__ c, const float *__restrict__ dV,
                                                                  const float *__restrict__ sg, float Dscale, int HM,
                                                                  float *__restrict__ dWL, float *__restrict__ dbL,
                                                                  float *__restrict__ dc) {
    __shared__ float part[256];
    const int o = blockIdx.x;
    const float co = c[o];
    const float *w = WL + (int64_t)o * HM;
    float s = 0.f;
    for (int j = threadIdx.x; j < HM; j += 256) {
        const float dv = dV[j];
        dWL[(int64_t)o * HM + j] = co * dv;
        s = __builtin_fmaf(w[j], dv, s);
    }
    part[threadIdx.x] = s;
    __syncthreads();
    for (int d = 128; d > 0; d >>= 1) {
        if ((int)threadIdx.x < d) part[threadIdx.x] += part[threadIdx.x + d];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float g = Dscale * sg[0];
        dc[o] = part[0] + (bL != nullptr ? g * bL[o] : 0.f);
        if (dbL != nullptr) dbL[o] = g * co;
    }
}

extern "C" int rp_cin_head_params_bwd(const float *WL, const float *bL, const float *c, const float *dV, const float *sg, float Dscale,
                                      int O, int H, int M, float *dWL, float *dbL, float *dc, rp_stream_t stream) {
    RP_REQUIRE(WL && c && dV && sg && dWL && dc, "cin_head_params_bwd: null pointer");
    RP_REQUIRE(O >= 1 && H >= 1 && M >= 1, "cin_head_params_bwd: bad O / H / M");
    hipLaunchKernelGGL(cin_head_params_bwd_kernel, dim3((unsigned)O), dim3(256), 0, (hipStream_t)stream, WL, bL, c, dV, sg, Dscale,
                       H * M, dWL, dbL, dc);
    RP_LAUNCH_CHECK("cin_head_params_bwd");
    return RP_OK;
}
