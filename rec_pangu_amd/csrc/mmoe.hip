// MMOE gate softmax + gate-weighted expert combine, forward and backward, gfx950.
//   reference: multi_task/mmoe.py:92-104 (softmax over experts per task, expand, multiply, sum over experts)
// Input is the ONE fused GEMM output z[B, ldz] = hidden . [experts | gates] + [experts_bias | gates_bias]:
//   columns [0, K*E)            expert outputs, laid out k*E + e (the reference's [hidden, K, E] parameter)
//   columns [K*E, K*E + T*E)    gate logits, t*E + e
// forward : gate[b,t,:] = softmax_e(z[b, K*E + t*E + :]);  out[t, b, k] = sum_e z[b, k*E+e] * gate[b,t,e]
// backward: dz[b,k*E+e] = sum_t dout[t,b,k] * gate[b,t,e]
//           dgate[b,t,e] = sum_k dout[t,b,k] * z[b,k*E+e];   dz[b,K*E+t*E+e] = gate*(dgate - sum_e' gate*dgate)
// Elementwise / small-reduction work: HBM-bound on reading z once (K*E*4 bytes per sample) — the reference
// materialises the [B,K,E] product T times.  One 256-thread workgroup per sample; T*E <= 32.
#include "common.h"

#define MM_MAX_TE 32

__global__ __launch_bounds__(256) void mmoe_combine_fwd_kernel(const float *__restrict__ z, int64_t ldz, int K, int E,
                                                               int T, float *__restrict__ out,
                                                               float *__restrict__ gate, int64_t B) {
    __shared__ float g[MM_MAX_TE];
    const int t = threadIdx.x;
    const int KE = K * E, TE = T * E;
    for (int64_t b = blockIdx.x; b < B; b += gridDim.x) {
        const float *zr = z + b * ldz;
        __syncthreads();
        if (t < T) {  // one thread per task computes that task's softmax over E experts
            float mx = -INFINITY;
            for (int e = 0; e < E; ++e) mx = fmaxf(mx, zr[KE + t * E + e]);
            float den = 0.f;
            for (int e = 0; e < E; ++e) den += expf(zr[KE + t * E + e] - mx);
            for (int e = 0; e < E; ++e) {
                const float v = expf(zr[KE + t * E + e] - mx) / den;
                g[t * E + e] = v;
                gate[b * TE + t * E + e] = v;
            }
        }
        __syncthreads();
        for (int k = t; k < K; k += 256) {
            for (int tt = 0; tt < T; ++tt) {
                float acc = 0.f;
                for (int e = 0; e < E; ++e) acc += zr[k * E + e] * g[tt * E + e];
                out[((int64_t)tt * B + b) * K + k] = acc;
            }
        }
    }
}

__global__ __launch_bounds__(256) void mmoe_combine_bwd_kernel(const float *__restrict__ z, int64_t ldz, int K, int E,
                                                               int T, const float *__restrict__ gate,
                                                               const float *__restrict__ dout,
                                                               float *__restrict__ dz, int64_t lddz, int64_t B) {
    __shared__ float g[64];       // [t][e] with static stride 8
    __shared__ float red[4][64];
    const int t = threadIdx.x;
    const int KE = K * E, TE = T * E;
    for (int64_t b = blockIdx.x; b < B; b += gridDim.x) {
        const float *zr = z + b * ldz;
        float *dzr = dz + b * lddz;
        __syncthreads();
        if (t < 64) {
            const int tt = t >> 3, e = t & 7;
            g[t] = (tt < T && e < E) ? gate[b * TE + tt * E + e] : 0.f;
        }
        __syncthreads();
        float dg[8][8];  // static indices only -> registers
#pragma unroll
        for (int tt = 0; tt < 8; ++tt)
#pragma unroll
            for (int e = 0; e < 8; ++e) dg[tt][e] = 0.f;
        for (int k = t; k < K; k += 256) {
            float dk[8];
#pragma unroll
            for (int tt = 0; tt < 8; ++tt) dk[tt] = (tt < T) ? dout[((int64_t)tt * B + b) * K + k] : 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                if (e < E) {
                    const float ze = zr[k * E + e];
                    float acc = 0.f;
#pragma unroll
                    for (int tt = 0; tt < 8; ++tt) {
                        acc += dk[tt] * g[tt * 8 + e];  // dk = 0 / g = 0 beyond T
                        dg[tt][e] += dk[tt] * ze;
                    }
                    dzr[k * E + e] = acc;
                }
            }
        }
#pragma unroll
        for (int tt = 0; tt < 8; ++tt)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                if (tt < T && e < E) {  // uniform: only the T*E live accumulators are reduced
                    float v = dg[tt][e];
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
                    if ((t & 63) == 0) red[t >> 6][tt * 8 + e] = v;
                }
            }
        __syncthreads();
        if (t < T) {
            float dot = 0.f;
            for (int e = 0; e < E; ++e) {
                const int i = t * 8 + e;
                dot += g[i] * ((red[0][i] + red[1][i]) + (red[2][i] + red[3][i]));
            }
            for (int e = 0; e < E; ++e) {
                const int i = t * 8 + e;
                const float d = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
                dzr[KE + t * E + e] = g[i] * (d - dot);
            }
        }
    }
}

static unsigned mm_blocks(int64_t B) { return (unsigned)(B < 65536 * 2 ? B : 65536 * 2); }

extern "C" int rp_mmoe_combine_fwd(const float *z, int64_t ldz, int K, int E, int T, float *out, float *gate,
                                   int64_t B, rp_stream_t stream) {
    RP_REQUIRE(z && out && gate && B >= 0, "mmoe_combine_fwd: null pointer");
    RP_REQUIRE(K >= 1 && E >= 1 && T >= 1 && ldz >= (int64_t)K * E + (int64_t)T * E, "mmoe_combine_fwd: bad K/E/T/ldz");
    if (T * E > MM_MAX_TE || T > 8 || E > 8)
        return rp_fail(RP_ERR_UNSUPPORTED, "mmoe_combine: T=%d E=%d unsupported (T,E <= 8, T*E <= %d)", T, E, MM_MAX_TE);
    if (B == 0) return RP_OK;
    hipLaunchKernelGGL(mmoe_combine_fwd_kernel, dim3(mm_blocks(B)), dim3(256), 0, (hipStream_t)stream, z, ldz, K, E, T,
                       out, gate, B);
    RP_LAUNCH_CHECK("mmoe_combine_fwd");
    return RP_OK;
}

extern "C" int rp_mmoe_combine_bwd(const float *z, int64_t ldz, int K, int E, int T, const float *gate,
                                   const float *dout, float *dz, int64_t lddz, int64_t B, rp_stream_t stream) {
    RP_REQUIRE(z && gate && dout && dz && B >= 0, "mmoe_combine_bwd: null pointer");
    RP_REQUIRE(K >= 1 && E >= 1 && T >= 1 && ldz >= (int64_t)K * E + (int64_t)T * E && lddz >= (int64_t)K * E + (int64_t)T * E,
               "mmoe_combine_bwd: bad K/E/T/ld");
    if (T * E > MM_MAX_TE || T > 8 || E > 8)
        return rp_fail(RP_ERR_UNSUPPORTED, "mmoe_combine: T=%d E=%d unsupported (T,E <= 8, T*E <= %d)", T, E, MM_MAX_TE);
    if (B == 0) return RP_OK;
    hipLaunchKernelGGL(mmoe_combine_bwd_kernel, dim3(mm_blocks(B)), dim3(256), 0, (hipStream_t)stream, z, ldz, K, E, T,
                       gate, dout, dz, lddz, B);
    RP_LAUNCH_CHECK("mmoe_combine_bwd");
    return RP_OK;
}
