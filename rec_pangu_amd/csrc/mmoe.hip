// MMOE gate softmax + gate-weighted expert combine, forward and backward, gfx950.
//   reference: multi_task/mmoe.py:92-104 (softmax over experts per task, expand, multiply, sum over experts)
// Input is the ONE fused GEMM output z[B, ldz] = hidden . [experts | gates] + [experts_bias | gates_bias]:
//   columns [0, K*E)            expert outputs, laid out k*E + e (the reference's [hidden, K, E] parameter)
//   columns [K*E, K*E + T*E)    gate logits, t*E + e
// forward : gate[b,t,:] = softmax_e(z[b, K*E + t*E + :]);  out[t, b, k] = sum_e z[b, k*E+e] * gate[b,t,e]
// backward: dz[b,k*E+e] = sum_t dout[t,b,k] * gate[b,t,e]
//           dgate[b,t,e] = sum_k dout[t,b,k] * z[b,k*E+e];   dz[b,K*E+t*E+e] = gate*(dgate - sum_e' gate*dgate)
// Elementwise / small-reduction work: HBM-bound on reading z once (K*E*4 bytes per sample) — the reference
// materialises the [B,K,E] product T times.  One 64-lane wave per sample; T, E <= 8 and T*E <= 32.
#include "common.h"

#define MM_MAX_TE 32

// One 64-lane wave per sample (no LDS, no barriers): lane l owns k = l, l+64, ... and reads/writes the E consecutive
// floats z[b, k*E .. k*E+E) as one vector when VEC (E even, rows 8/16-byte aligned).  Gates live in registers,
// broadcast across the wave with shuffles; T <= 8 is a runtime bound on statically unrolled loops.
template <int E, bool VEC>
__device__ __forceinline__ void mm_load(const float *p, float (&v)[E]) {
    if (VEC) {
        constexpr int A = (E % 4 == 0) ? 16 : 8;
        __builtin_memcpy(v, __builtin_assume_aligned(p, A), sizeof(float) * E);
    } else {
#pragma unroll
        for (int e = 0; e < E; ++e) v[e] = p[e];
    }
}

template <int E, bool VEC>
__device__ __forceinline__ void mm_store(float *p, const float (&v)[E]) {
    if (VEC) {
        constexpr int A = (E % 4 == 0) ? 16 : 8;
        __builtin_memcpy(__builtin_assume_aligned(p, A), v, sizeof(float) * E);
    } else {
#pragma unroll
        for (int e = 0; e < E; ++e) p[e] = v[e];
    }
}

template <int E, bool VEC>
__global__ __launch_bounds__(256) void mmoe_combine_fwd_kernel(const float *__restrict__ z, int64_t ldz, int K, int T,
                                                               float *__restrict__ out, float *__restrict__ gate,
                                                               int64_t B) {
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int KE = K * E, TE = T * E;
    for (int64_t b = (int64_t)blockIdx.x * 4 + wid; b < B; b += (int64_t)gridDim.x * 4) {
        const float *zr = z + b * ldz;
        float ge[E];  // lane t < T: softmax of task t
#pragma unroll
        for (int e = 0; e < E; ++e) ge[e] = 0.f;
        if (lane < T) {
            float lg[E];
#pragma unroll
            for (int e = 0; e < E; ++e) lg[e] = zr[KE + lane * E + e];
            float mx = lg[0];
#pragma unroll
            for (int e = 1; e < E; ++e) mx = fmaxf(mx, lg[e]);
            float den = 0.f;
#pragma unroll
            for (int e = 0; e < E; ++e) den += expf(lg[e] - mx);
#pragma unroll
            for (int e = 0; e < E; ++e) {
                ge[e] = expf(lg[e] - mx) / den;
                gate[b * TE + lane * E + e] = ge[e];
            }
        }
        float g[8][E];
#pragma unroll
        for (int tt = 0; tt < 8; ++tt)
#pragma unroll
            for (int e = 0; e < E; ++e) g[tt][e] = __shfl(ge[e], tt, 64);  // zero beyond T
        for (int k = lane; k < K; k += 64) {
            float ze[E];
            mm_load<E, VEC>(zr + (int64_t)k * E, ze);
#pragma unroll
            for (int tt = 0; tt < 8; ++tt) {
                if (tt < T) {
                    float acc = 0.f;
#pragma unroll
                    for (int e = 0; e < E; ++e) acc += ze[e] * g[tt][e];
                    out[((int64_t)tt * B + b) * K + k] = acc;
                }
            }
        }
    }
}

template <int E, bool VEC>
__global__ __launch_bounds__(256) void mmoe_combine_bwd_kernel(const float *__restrict__ z, int64_t ldz, int K, int T,
                                                               const float *__restrict__ gate,
                                                               const float *__restrict__ dout,
                                                               float *__restrict__ dz, int64_t lddz, int64_t B) {
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int KE = K * E, TE = T * E;
    for (int64_t b = (int64_t)blockIdx.x * 4 + wid; b < B; b += (int64_t)gridDim.x * 4) {
        const float *zr = z + b * ldz;
        float *dzr = dz + b * lddz;
        const float *gb = gate + b * TE;  // wave-uniform address -> scalar loads
        float g[8][E], dg[8][E];
#pragma unroll
        for (int tt = 0; tt < 8; ++tt)
#pragma unroll
            for (int e = 0; e < E; ++e) {
                g[tt][e] = (tt < T) ? gb[tt * E + e] : 0.f;
                dg[tt][e] = 0.f;
            }
        for (int k = lane; k < K; k += 64) {
            float ze[E], acc[E];
            mm_load<E, VEC>(zr + (int64_t)k * E, ze);
#pragma unroll
            for (int e = 0; e < E; ++e) acc[e] = 0.f;
#pragma unroll
            for (int tt = 0; tt < 8; ++tt) {
                if (tt < T) {
                    const float dk = dout[((int64_t)tt * B + b) * K + k];
#pragma unroll
                    for (int e = 0; e < E; ++e) {
                        acc[e] += dk * g[tt][e];
                        dg[tt][e] += dk * ze[e];
                    }
                }
            }
            mm_store<E, VEC>(dzr + (int64_t)k * E, acc);
        }
#pragma unroll
        for (int tt = 0; tt < 8; ++tt) {
            if (tt < T) {  // wave-uniform: only the T*E live accumulators are reduced
                float d[E], dot = 0.f;
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    float v = dg[tt][e];
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
                    d[e] = v;
                    dot += g[tt][e] * v;
                }
                if (lane == 0) {
#pragma unroll
                    for (int e = 0; e < E; ++e) dzr[KE + tt * E + e] = g[tt][e] * (d[e] - dot);
                }
            }
        }
    }
}

static unsigned mm_blocks(int64_t B) {
    const int64_t nb = rp_cdiv(B, 4);
    return (unsigned)(nb < 65536 ? nb : 65536);
}

static bool mm_vec(int E, const void *p0, int64_t ld0, const void *p1, int64_t ld1) {
    if (E % 2 != 0) return false;
    const uintptr_t a = (E % 4 == 0) ? 16 : 8;
    return (reinterpret_cast<uintptr_t>(p0) % a) == 0 && (ld0 * 4) % (int64_t)a == 0 &&
           (p1 == nullptr || ((reinterpret_cast<uintptr_t>(p1) % a) == 0 && (ld1 * 4) % (int64_t)a == 0));
}

#define MM_DISPATCH_E(E, CALL)   \
    switch (E) {                 \
        case 1: CALL(1); break;  \
        case 2: CALL(2); break;  \
        case 3: CALL(3); break;  \
        case 4: CALL(4); break;  \
        case 5: CALL(5); break;  \
        case 6: CALL(6); break;  \
        case 7: CALL(7); break;  \
        default: CALL(8); break; \
    }

extern "C" int rp_mmoe_combine_fwd(const float *z, int64_t ldz, int K, int E, int T, float *out, float *gate,
                                   int64_t B, rp_stream_t stream) {
    RP_REQUIRE(z && out && gate && B >= 0, "mmoe_combine_fwd: null pointer");
    RP_REQUIRE(K >= 1 && E >= 1 && T >= 1 && ldz >= (int64_t)K * E + (int64_t)T * E, "mmoe_combine_fwd: bad K/E/T/ldz");
    if (T * E > MM_MAX_TE || T > 8 || E > 8)
        return rp_fail(RP_ERR_UNSUPPORTED, "mmoe_combine: T=%d E=%d unsupported (T,E <= 8, T*E <= %d)", T, E, MM_MAX_TE);
    if (B == 0) return RP_OK;
    const bool vec = mm_vec(E, z, ldz, nullptr, 0);
#define CALL(EE)                                                                                                         \
    if (vec)                                                                                                             \
        hipLaunchKernelGGL((mmoe_combine_fwd_kernel<EE, (EE % 2 == 0)>), dim3(mm_blocks(B)), dim3(256), 0,              \
                           (hipStream_t)stream, z, ldz, K, T, out, gate, B);                                            \
    else                                                                                                                 \
        hipLaunchKernelGGL((mmoe_combine_fwd_kernel<EE, false>), dim3(mm_blocks(B)), dim3(256), 0, (hipStream_t)stream, \
                           z, ldz, K, T, out, gate, B)
    MM_DISPATCH_E(E, CALL)
#undef CALL
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
    const bool vec = mm_vec(E, z, ldz, dz, lddz);
#define CALL(EE)                                                                                                         \
    if (vec)                                                                                                             \
        hipLaunchKernelGGL((mmoe_combine_bwd_kernel<EE, (EE % 2 == 0)>), dim3(mm_blocks(B)), dim3(256), 0,              \
                           (hipStream_t)stream, z, ldz, K, T, gate, dout, dz, lddz, B);                                 \
    else                                                                                                                 \
        hipLaunchKernelGGL((mmoe_combine_bwd_kernel<EE, false>), dim3(mm_blocks(B)), dim3(256), 0, (hipStream_t)stream, \
                           z, ldz, K, T, gate, dout, dz, lddz, B)
    MM_DISPATCH_E(E, CALL)
#undef CALL
    RP_LAUNCH_CHECK("mmoe_combine_bwd");
    return RP_OK;
}
