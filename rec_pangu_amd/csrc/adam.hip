// Fused multi-tensor Adam, gfx950.  Pure HBM stream: reads p,g,m,v and writes p,m,v (+g when the
// fused zero_grad is on) = 28 (32) bytes per parameter; the reference runs DENSE Adam over every
// embedding row every step (nn.Embedding is never sparse=True, trainer.py:75), so at
// Criteo shape this kernel moves ~60 GB per step and bounds the train step.
// Operation order follows torch.optim.Adam's single-tensor path:
//   m += (g - m) * (1 - b1);  v = v * b2 + (1 - b2) * g * g;
//   p += -(lr / (1 - b1^t)) * (m / (sqrt(v) / sqrt(1 - b2^t) + eps))
// blockIdx.y selects the tensor, blockIdx.x grid-strides over its float4 chunks.
#include "common.h"
#include <cmath>

struct AdamPtrs {
    float *p[RP_MAX_FIELDS];
    float *g[RP_MAX_FIELDS];
    float *m[RP_MAX_FIELDS];
    float *v[RP_MAX_FIELDS];
    int64_t n[RP_MAX_FIELDS];
};

template <typename T>
__device__ __forceinline__ void adam1(T &p, const T g, T &m, T &v, float one_m_b1, float b2, float one_m_b2,
                                      float step_size, float bc2_sqrt, float eps) {
    m = m + (g - m) * one_m_b1;
    v = v * b2 + one_m_b2 * g * g;
    const T denom = __builtin_elementwise_sqrt(v) / bc2_sqrt + eps;
    p = p - step_size * (m / denom);
}

template <bool ZERO_G>
__global__ __launch_bounds__(256) void adam_kernel(AdamPtrs a, float one_m_b1, float b2, float one_m_b2,
                                                   float step_size, float bc2_sqrt, float eps) {
    const int ti = blockIdx.y;
    float *__restrict__ P = a.p[ti];
    float *__restrict__ G = a.g[ti];
    float *__restrict__ Mo = a.m[ti];
    float *__restrict__ Vo = a.v[ti];
    const int64_t n = a.n[ti];
    const bool vec = (((uintptr_t)P | (uintptr_t)G | (uintptr_t)Mo | (uintptr_t)Vo) & 15u) == 0;
    const int64_t n4 = vec ? n / 4 : 0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n4; e += stride) {
        f32x4 p = reinterpret_cast<f32x4 *>(P)[e];
        const f32x4 g = reinterpret_cast<const f32x4 *>(G)[e];
        f32x4 m = reinterpret_cast<f32x4 *>(Mo)[e], v = reinterpret_cast<f32x4 *>(Vo)[e];
        adam1<f32x4>(p, g, m, v, one_m_b1, b2, one_m_b2, step_size, bc2_sqrt, eps);
        reinterpret_cast<f32x4 *>(P)[e] = p;
        reinterpret_cast<f32x4 *>(Mo)[e] = m;
        reinterpret_cast<f32x4 *>(Vo)[e] = v;
        if (ZERO_G) reinterpret_cast<f32x4 *>(G)[e] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (int64_t e = n4 * 4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += stride) {
        float p = P[e], m = Mo[e], v = Vo[e];
        const float g = G[e];
        adam1<float>(p, g, m, v, one_m_b1, b2, one_m_b2, step_size, bc2_sqrt, eps);
        P[e] = p;
        Mo[e] = m;
        Vo[e] = v;
        if (ZERO_G) G[e] = 0.f;
    }
}

extern "C" int rp_adam_step(float *const *p_ptrs, float *const *g_ptrs, float *const *m_ptrs, float *const *v_ptrs,
                            const int64_t *sizes, int n_tensors, float lr, float beta1, float beta2, float eps,
                            int64_t step, int zero_grad, rp_stream_t stream) {
    RP_REQUIRE(p_ptrs && g_ptrs && m_ptrs && v_ptrs && sizes, "adam_step: null pointer");
    RP_REQUIRE(n_tensors >= 1 && n_tensors <= RP_MAX_FIELDS, "adam_step: n_tensors=%d outside [1,%d]", n_tensors,
               RP_MAX_FIELDS);
    RP_REQUIRE(step >= 1, "adam_step: step must be >= 1");
    AdamPtrs a;
    int64_t maxn = 0;
    for (int i = 0; i < n_tensors; ++i) {
        RP_REQUIRE(p_ptrs[i] && g_ptrs[i] && m_ptrs[i] && v_ptrs[i] && sizes[i] >= 0, "adam_step: tensor %d invalid", i);
        a.p[i] = p_ptrs[i];
        a.g[i] = g_ptrs[i];
        a.m[i] = m_ptrs[i];
        a.v[i] = v_ptrs[i];
        a.n[i] = sizes[i];
        if (sizes[i] > maxn) maxn = sizes[i];
    }
    if (maxn == 0) return RP_OK;
    // scalar prep in double, as python floats are in torch.optim.Adam
    const double bc1 = 1.0 - std::pow((double)beta1, (double)step);
    const double bc2 = 1.0 - std::pow((double)beta2, (double)step);
    const float step_size = (float)((double)lr / bc1);
    const float bc2_sqrt = (float)std::sqrt(bc2);
    int64_t bx = rp_cdiv(rp_cdiv(maxn, 4), 256);
    if (bx > 8192) bx = 8192;
    if (bx < 1) bx = 1;
    dim3 grid((unsigned)bx, (unsigned)n_tensors);
    hipStream_t s = (hipStream_t)stream;
    const float one_m_b1 = (float)(1.0 - (double)beta1), one_m_b2 = (float)(1.0 - (double)beta2);
    if (zero_grad)
        hipLaunchKernelGGL((adam_kernel<true>), grid, dim3(256), 0, s, a, one_m_b1, beta2, one_m_b2, step_size,
                           bc2_sqrt, eps);
    else
        hipLaunchKernelGGL((adam_kernel<false>), grid, dim3(256), 0, s, a, one_m_b1, beta2, one_m_b2, step_size,
                           bc2_sqrt, eps);
    RP_LAUNCH_CHECK("adam_step");
    return RP_OK;
}
