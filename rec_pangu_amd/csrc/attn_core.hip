// AutoInt field self-attention, split form: the projections are GEMMs, only the T x T core runs here.  gfx950.
//   reference: layers/attention.py:63-101 (MultiHeadSelfAttention, align_to="output")
//
// The one-launch layer of attn.hip keeps a whole sample (X, weights, Q/K/V/P) in LDS and does the [T,Din]x[Din,4HA]
// projections — 80 % of the layer's arithmetic — with scalar LDS reads: LDS-instruction-bound, ~1 % of the HBM
// roofline at B=65536.  Here the projections  QKVR = X . [Wq|Wk|Wv|Wres]^T  run as ONE row-major GEMM over the
// B*T token rows on the matrix core (rp_linear_fwd, short-K kernel), and this file is what is left per sample:
//     scores = Qg Kg^T (/scale), softmax over keys, Og = P Vg, out = relu(O + R)          (T x T x a, tiny)
// with the reference's RAW-view head split: head g, token t, dim j of a projection is element g*T*a + t*a + j of the
// sample's flat [T*HA] buffer, i.e. row (g*T+t)/H, columns ((g*T+t)%H)*a + j of its [T, HA] matrix.
//
// One wave per sample (one 64-thread workgroup), lane = one (head, query) row:
//   forward : K, V staged flat in LDS (broadcast reads), q and the output accumulator in registers, online softmax
//             (running max / sum), row statistics (m, l) saved for the backward.
//   backward: phase 1, lane = query row i : D_i = sum_j P_ij dP_ij, then dQ_i = sum_j dS_ij K_j
//             phase 2, lane = key row j   : dK_j = sum_i dS_ij Q_i, dV_j = sum_i P_ij dO_i
//             with P_ij = exp(s_ij - m_i) / l_i recomputed from the saved statistics, dS_ij = P_ij (dP_ij - D_i)/scale;
//             no atomics, no cross-lane reductions.  The ReLU mask and the residual gradient are applied here; the
//             gradients of X and of the weights are two more GEMMs on dQKVR (rp_linear_fwd / rp_linear_wgrad).
// HBM per sample: QKVR in (T*4HA*4 B) + out; the arithmetic (2*H*T*T*a MAC) is ~1 us of VALU per wave.
#include "common.h"

#define AC_MAXA 16  // head width a <= 16 (registers: q[a], acc[a])

struct AcDims {
    int T, H, a, HA, nproj;
    float inv_scale;
};

// flat element f of a projection p (0 = Q, 1 = K, 2 = V, 3 = R) of sample b lives at
//   qkvr[(b*T + f / HA) * ld + p*HA + f % HA]
__device__ __forceinline__ int64_t ac_addr(const AcDims &d, int64_t b, int64_t ld, int p, int f) {
    const int t = f / d.HA, c = f - t * d.HA;
    return (b * d.T + t) * ld + p * d.HA + c;
}

__global__ __launch_bounds__(64) void attn_core_fwd_kernel(AcDims d, const float *__restrict__ qkvr, int64_t ldq,
                                                           const float *__restrict__ xres, int64_t ldr,
                                                           float *__restrict__ out, float *__restrict__ stats,
                                                           int64_t B, int spw) {
    extern __shared__ float sm[];  // per sample slot: K flat [T*HA] | V flat [T*HA]
    const int TH = d.T * d.HA, HT = d.H * d.T, a = d.a;
    const int lane = threadIdx.x;
    // spw samples share the wave when a sample has <= 32 (head, query) rows: AutoInt's T = 26, H = 1 uses 52 lanes
    for (int64_t b0 = (int64_t)blockIdx.x * spw; b0 < B; b0 += (int64_t)gridDim.x * spw) {
        __syncthreads();  // previous samples' readers are done
        for (int f = lane; f < spw * TH; f += 64) {
            const int sl = f / TH, ff = f - sl * TH;
            if (b0 + sl < B) {
                sm[sl * 2 * TH + ff] = qkvr[ac_addr(d, b0 + sl, ldq, 1, ff)];
                sm[sl * 2 * TH + TH + ff] = qkvr[ac_addr(d, b0 + sl, ldq, 2, ff)];
            }
        }
        __syncthreads();
        for (int idx = lane; idx < spw * HT; idx += 64) {
            const int sl = idx / HT, r = idx - sl * HT;
            const int64_t b = b0 + sl;
            if (b >= B) continue;
            const float *Ks = sm + sl * 2 * TH, *Vs = Ks + TH;
            const int g = r / d.T;
            const int f0 = r * a;  // (g*T + i) * a
            float q[AC_MAXA], acc[AC_MAXA];
#pragma unroll
            for (int c = 0; c < AC_MAXA; ++c) {
                q[c] = (c < a) ? qkvr[ac_addr(d, b, ldq, 0, f0 + c)] : 0.f;
                acc[c] = 0.f;
            }
            float m = -INFINITY, l = 0.f;
            const float *kg = Ks + g * d.T * a, *vg = Vs + g * d.T * a;
            for (int j = 0; j < d.T; ++j) {
                float s = 0.f;
#pragma unroll
                for (int c = 0; c < AC_MAXA; ++c)
                    if (c < a) s += q[c] * kg[j * a + c];
                s *= d.inv_scale;
                const float mn = fmaxf(m, s);
                const float sc = expf(m - mn), p = expf(s - mn);
                l = l * sc + p;
#pragma unroll
                for (int c = 0; c < AC_MAXA; ++c)
                    if (c < a) acc[c] = acc[c] * sc + p * vg[j * a + c];
                m = mn;
            }
            const float il = 1.f / l;
            if (stats != nullptr) {
                stats[(b * HT + r) * 2 + 0] = m;
                stats[(b * HT + r) * 2 + 1] = l;
            }
#pragma unroll
            for (int c = 0; c < AC_MAXA; ++c) {
                if (c < a) {
                    const int f = f0 + c, t = f / d.HA, cc = f - t * d.HA;
                    const float res = (d.nproj == 4) ? qkvr[ac_addr(d, b, ldq, 3, f)] : xres[(b * d.T + t) * ldr + cc];
                    const float v = acc[c] * il + res;
                    out[(b * d.T + t) * d.HA + cc] = v > 0.f ? v : 0.f;
                }
            }
        }
    }
}

__global__ __launch_bounds__(64) void attn_core_bwd_kernel(AcDims d, const float *__restrict__ qkvr, int64_t ldq,
                                                           const float *__restrict__ out,
                                                           const float *__restrict__ dout,
                                                           const float *__restrict__ stats,
                                                           float *__restrict__ dqkvr, int64_t lddq,
                                                           float *__restrict__ dxres, int64_t lddr, int64_t B,
                                                           int spw) {
    extern __shared__ float sm[];  // per sample slot: Q | K | V | dO flat [T*HA] each, then m | 1/l | D [H*T] each
    const int TH = d.T * d.HA, HT = d.H * d.T, a = d.a;
    const int slot = 4 * TH + 3 * HT;
    const int lane = threadIdx.x;
    for (int64_t b0 = (int64_t)blockIdx.x * spw; b0 < B; b0 += (int64_t)gridDim.x * spw) {
        __syncthreads();
        for (int f2 = lane; f2 < spw * TH; f2 += 64) {
            const int sl = f2 / TH, f = f2 - sl * TH;
            const int64_t b = b0 + sl;
            if (b >= B) continue;
            float *Qs = sm + sl * slot, *Ks = Qs + TH, *Vs = Qs + 2 * TH, *Os = Qs + 3 * TH;
            Qs[f] = qkvr[ac_addr(d, b, ldq, 0, f)];
            Ks[f] = qkvr[ac_addr(d, b, ldq, 1, f)];
            Vs[f] = qkvr[ac_addr(d, b, ldq, 2, f)];
            const int t = f / d.HA, cc = f - t * d.HA;
            const int64_t o = (b * d.T + t) * d.HA + cc;
            const float dz = (out[o] > 0.f) ? dout[o] : 0.f;  // ReLU backward
            Os[f] = dz;                                        // dO = dZ (raw view back)
            if (d.nproj == 4) dqkvr[ac_addr(d, b, lddq, 3, f)] = dz;  // dR
            else dxres[(b * d.T + t) * lddr + cc] = dz;               // residual = X itself
        }
        for (int idx = lane; idx < spw * HT; idx += 64) {
            const int sl = idx / HT, r = idx - sl * HT;
            if (b0 + sl < B) {
                float *Ms = sm + sl * slot + 4 * TH;
                Ms[r] = stats[((b0 + sl) * HT + r) * 2 + 0];
                Ms[HT + r] = 1.f / stats[((b0 + sl) * HT + r) * 2 + 1];
            }
        }
        __syncthreads();
        // phase 1: query rows
        for (int idx = lane; idx < spw * HT; idx += 64) {
            const int sl = idx / HT, r = idx - sl * HT;
            const int64_t b = b0 + sl;
            if (b >= B) continue;
            float *Qs = sm + sl * slot, *Ks = Qs + TH, *Vs = Qs + 2 * TH, *Os = Qs + 3 * TH;
            float *Ms = Qs + 4 * TH, *Ls = Ms + HT, *Ds = Ls + HT;
            const int g = r / d.T;
            const int f0 = r * a;
            float q[AC_MAXA], go[AC_MAXA], dq[AC_MAXA];
#pragma unroll
            for (int c = 0; c < AC_MAXA; ++c) {
                q[c] = (c < a) ? Qs[f0 + c] : 0.f;
                go[c] = (c < a) ? Os[f0 + c] : 0.f;
                dq[c] = 0.f;
            }
            const float m = Ms[r], il = Ls[r];
            const float *kg = Ks + g * d.T * a, *vg = Vs + g * d.T * a;
            float Dsum = 0.f;
            for (int j = 0; j < d.T; ++j) {
                float s = 0.f, dp = 0.f;
#pragma unroll
                for (int c = 0; c < AC_MAXA; ++c)
                    if (c < a) {
                        s += q[c] * kg[j * a + c];
                        dp += go[c] * vg[j * a + c];
                    }
                Dsum += expf(s * d.inv_scale - m) * il * dp;
            }
            Ds[r] = Dsum;
            for (int j = 0; j < d.T; ++j) {
                float s = 0.f, dp = 0.f;
#pragma unroll
                for (int c = 0; c < AC_MAXA; ++c)
                    if (c < a) {
                        s += q[c] * kg[j * a + c];
                        dp += go[c] * vg[j * a + c];
                    }
                const float ds = expf(s * d.inv_scale - m) * il * (dp - Dsum) * d.inv_scale;
#pragma unroll
                for (int c = 0; c < AC_MAXA; ++c)
                    if (c < a) dq[c] += ds * kg[j * a + c];
            }
#pragma unroll
            for (int c = 0; c < AC_MAXA; ++c)
                if (c < a) dqkvr[ac_addr(d, b, lddq, 0, f0 + c)] = dq[c];
        }
        __syncthreads();  // D_i of every row visible
        // phase 2: key rows
        for (int idx = lane; idx < spw * HT; idx += 64) {
            const int sl = idx / HT, r = idx - sl * HT;
            const int64_t b = b0 + sl;
            if (b >= B) continue;
            float *Qs = sm + sl * slot, *Ks = Qs + TH, *Vs = Qs + 2 * TH, *Os = Qs + 3 * TH;
            float *Ms = Qs + 4 * TH, *Ls = Ms + HT, *Ds = Ls + HT;
            const int g = r / d.T;
            const int f0 = r * a;
            float k[AC_MAXA], v[AC_MAXA], dk[AC_MAXA], dv[AC_MAXA];
#pragma unroll
            for (int c = 0; c < AC_MAXA; ++c) {
                k[c] = (c < a) ? Ks[f0 + c] : 0.f;
                v[c] = (c < a) ? Vs[f0 + c] : 0.f;
                dk[c] = 0.f;
                dv[c] = 0.f;
            }
            const float *qg = Qs + g * d.T * a, *og = Os + g * d.T * a;
            for (int i = 0; i < d.T; ++i) {
                float s = 0.f, dp = 0.f;
#pragma unroll
                for (int c = 0; c < AC_MAXA; ++c)
                    if (c < a) {
                        s += qg[i * a + c] * k[c];
                        dp += og[i * a + c] * v[c];
                    }
                const int ri = g * d.T + i;
                const float p = expf(s * d.inv_scale - Ms[ri]) * Ls[ri];
                const float ds = p * (dp - Ds[ri]) * d.inv_scale;
#pragma unroll
                for (int c = 0; c < AC_MAXA; ++c)
                    if (c < a) {
                        dk[c] += ds * qg[i * a + c];
                        dv[c] += p * og[i * a + c];
                    }
            }
#pragma unroll
            for (int c = 0; c < AC_MAXA; ++c)
                if (c < a) {
                    dqkvr[ac_addr(d, b, lddq, 1, f0 + c)] = dk[c];
                    dqkvr[ac_addr(d, b, lddq, 2, f0 + c)] = dv[c];
                }
        }
    }
}

static int ac_check(int T, int H, int a, int nproj, size_t lds_bytes) {
    if (T < 1 || H < 1 || a < 1 || a > AC_MAXA || (nproj != 3 && nproj != 4))
        return rp_fail(RP_ERR_UNSUPPORTED, "attention_core: T=%d H=%d a=%d (a <= %d) nproj=%d unsupported", T, H, a,
                       AC_MAXA, nproj);
    if (lds_bytes > 64 * 1024)
        return rp_fail(RP_ERR_UNSUPPORTED, "attention_core: a sample's Q/K/V do not fit in LDS (%zu bytes)", lds_bytes);
    return RP_OK;
}

extern "C" int rp_attention_core_fits(int T, int H, int a) {
    return (T >= 1 && H >= 1 && a >= 1 && a <= AC_MAXA && (size_t)(4 * T * H * a + 3 * H * T) * sizeof(float) <= 64 * 1024)
               ? 1
               : 0;
}

static int ac_spw(int T, int H) {  // samples per wave
    const int HT = T * H;
    return HT >= 64 ? 1 : 64 / HT;
}
static unsigned ac_grid(int64_t B, int spw) {
    const int64_t nb = rp_cdiv(B, spw);
    return (unsigned)(nb < 262144 ? nb : 262144);
}

extern "C" int rp_attention_core_fwd(const float *qkvr, int64_t ldq, int nproj, const float *xres, int64_t ldr, int T,
                                     int H, int a, float scale, float *out, float *stats, int64_t B,
                                     rp_stream_t stream) {
    RP_REQUIRE(qkvr && out && B >= 0, "attention_core_fwd: null pointer");
    const int HA = H * a;
    int spw = ac_spw(T, H);
    while (spw > 1 && (size_t)spw * 2 * T * HA * sizeof(float) > 64 * 1024) --spw;
    const size_t lds = (size_t)spw * 2 * T * HA * sizeof(float);
    int rc = ac_check(T, H, a, nproj, lds);
    if (rc != RP_OK) return rc;
    RP_REQUIRE(ldq >= (int64_t)nproj * HA, "attention_core_fwd: ldq too small");
    RP_REQUIRE(nproj == 4 || (xres != nullptr && ldr >= HA), "attention_core_fwd: the residual needs xres when there is no W_res");
    if (B == 0) return RP_OK;
    AcDims d{T, H, a, HA, nproj, scale > 0.f ? 1.f / scale : 1.f};
    hipLaunchKernelGGL(attn_core_fwd_kernel, dim3(ac_grid(B, spw)), dim3(64), lds, (hipStream_t)stream, d, qkvr, ldq, xres,
                       ldr, out, stats, B, spw);
    RP_LAUNCH_CHECK("attention_core_fwd");
    return RP_OK;
}

extern "C" int rp_attention_core_bwd(const float *qkvr, int64_t ldq, int nproj, const float *out, const float *dout,
                                     const float *stats, int T, int H, int a, float scale, float *dqkvr, int64_t lddq,
                                     float *dxres, int64_t lddr, int64_t B, rp_stream_t stream) {
    RP_REQUIRE(qkvr && out && dout && stats && dqkvr && B >= 0, "attention_core_bwd: null pointer");
    const int HA = H * a;
    int spw = ac_spw(T, H);
    while (spw > 1 && (size_t)spw * (4 * T * HA + 3 * H * T) * sizeof(float) > 64 * 1024) --spw;
    const size_t lds = (size_t)spw * (4 * T * HA + 3 * H * T) * sizeof(float);
    int rc = ac_check(T, H, a, nproj, lds);
    if (rc != RP_OK) return rc;
    RP_REQUIRE(ldq >= (int64_t)nproj * HA && lddq >= (int64_t)nproj * HA, "attention_core_bwd: ld too small");
    RP_REQUIRE(nproj == 4 || (dxres != nullptr && lddr >= HA), "attention_core_bwd: dxres needed when there is no W_res");
    if (B == 0) return RP_OK;
    AcDims d{T, H, a, HA, nproj, scale > 0.f ? 1.f / scale : 1.f};
    hipLaunchKernelGGL(attn_core_bwd_kernel, dim3(ac_grid(B, spw)), dim3(64), lds, (hipStream_t)stream, d, qkvr, ldq, out,
                       dout, stats, dqkvr, lddq, dxres, lddr, B, spw);
    RP_LAUNCH_CHECK("attention_core_bwd");
    return RP_OK;
}
