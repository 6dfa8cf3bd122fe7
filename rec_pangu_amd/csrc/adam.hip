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

#ifndef RP_REPLAY_UNROLL
#define RP_REPLAY_UNROLL 2  // zero-gradient steps per iteration of the replay loop
#endif

struct AdamPtrs {
    float *p[RP_MAX_FIELDS];
    float *g[RP_MAX_FIELDS];
    float *m[RP_MAX_FIELDS];
    float *v[RP_MAX_FIELDS];
    int64_t n[RP_MAX_FIELDS];
};

// Every multiply-add is spelled out as an explicit fma so that the dense kernel and the lazy replay (which
// must reproduce it bit for bit) cannot be contracted differently by the compiler.
__device__ __forceinline__ float rp_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ f32x4 rp_fma(f32x4 a, f32x4 b, f32x4 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ float rp_splat(float x, float) { return x; }
__device__ __forceinline__ f32x4 rp_splat(float x, f32x4) { return f32x4{x, x, x, x}; }
typedef float f32x2a __attribute__((ext_vector_type(2)));  // one v_pk_*_f32 operand: two floats per lane, one issue slot
__device__ __forceinline__ f32x2a rp_fma(f32x2a a, f32x2a b, f32x2a c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2a rp_splat(float x, f32x2a) { return f32x2a{x, x}; }

// sqrt / reciprocal on the hardware approximations (v_sqrt_f32, v_rcp_f32: 1 ulp): the IEEE-rounded forms expand to
// ~10 instructions each, and the lazy replay of skipped steps is bound by exactly this arithmetic (one sqrt and one
// reciprocal per element per skipped step).  The dense kernel and the replay share this function, so they stay
// bit-identical to each other; against torch.optim.Adam the difference is ~3e-7 relative on the UPDATE term.
__device__ __forceinline__ float rp_sqrt_fast(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ f32x4 rp_sqrt_fast(f32x4 x) {
    return f32x4{__builtin_amdgcn_sqrtf(x.x), __builtin_amdgcn_sqrtf(x.y), __builtin_amdgcn_sqrtf(x.z),
                 __builtin_amdgcn_sqrtf(x.w)};
}
__device__ __forceinline__ float rp_rcp_fast(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ f32x2a rp_rcp_fast(f32x2a x) { return f32x2a{__builtin_amdgcn_rcpf(x.x), __builtin_amdgcn_rcpf(x.y)}; }
__device__ __forceinline__ f32x4 rp_rcp_fast(f32x4 x) {
    return f32x4{__builtin_amdgcn_rcpf(x.x), __builtin_amdgcn_rcpf(x.y), __builtin_amdgcn_rcpf(x.z),
                 __builtin_amdgcn_rcpf(x.w)};
}

// The second moment is kept as its SQUARE ROOT  s = sqrt(v)  (the `v` arrays of this file hold s).  Adam only ever
// uses sqrt(v), and for a zero gradient  v <- b2 v  is  s <- sqrt(b2) s : one multiply instead of a multiply and a
// transcendental — the lazy replay of skipped (zero-gradient) steps is bound by exactly this arithmetic.  With a
// gradient  s <- sqrt(b2 s^2 + (1 - b2) g^2).  The choice is made PER ELEMENT on g == 0, by the one function every
// kernel of this file calls, so "dense every step" and "lazy replay later" execute the same operations on the same
// values: bit-identical.  Against torch.optim.Adam (which stores v) each step differs by <= 1 ulp in s.
// Exported optimizer state squares s again (rec_pangu_amd/optim.py).
// The update itself is evaluated as  p += m * rcp(s * A_t + B_t)  with the two per-step scalars
//     A_t = (1 / sqrt(1 - b2^t)) / (-lr_t / (1 - b1^t)),   B_t = eps / (-lr_t / (1 - b1^t))      (computed in double)
// i.e. the step size is folded into the denominator: p - step_size * m / (sqrt(v) / sqrt(1 - b2^t) + eps) in one fma, one
// reciprocal and one more fma — one multiply less per element and step than the textbook order, which matters where
// it is executed 2 G times per training step (the replay).  lr_t = 0: (A, B) = (0, -inf) -> rcp = -0 -> p unchanged.
// sqrt_b2 = (float)sqrt((double)b2)
__device__ __forceinline__ float rp_sel_zero(float g, float a, float b) { return g == 0.f ? a : b; }
__device__ __forceinline__ f32x4 rp_sel_zero(f32x4 g, f32x4 a, f32x4 b) {
    return f32x4{g.x == 0.f ? a.x : b.x, g.y == 0.f ? a.y : b.y, g.z == 0.f ? a.z : b.z, g.w == 0.f ? a.w : b.w};
}

template <typename T>
__device__ __forceinline__ void adam1(T &p, const T g, T &m, T &s, float one_m_b1, float b2, float sqrt_b2,
                                      float one_m_b2, float sa, float sb) {
    const T c1 = rp_splat(one_m_b1, p), c2 = rp_splat(one_m_b2, p);
    m = rp_fma(g - m, c1, m);                      // m + (g - m)(1 - b1)
    const T vb = (s * s) * b2;                     // b2 v, v = s^2
    const T sg = rp_sqrt_fast(rp_fma(c2 * g, g, vb));  // sqrt(b2 v + ((1 - b2) g) g)
    s = rp_sel_zero(g, s * sqrt_b2, sg);           // zero gradient: sqrt(b2) s, exactly what the replay computes
    const T denom = rp_fma(s, rp_splat(sa, p), rp_splat(sb, p));  // (sqrt(v) / sqrt(1 - b2^t) + eps) / (-step_size)
    p = rp_fma(m, rp_rcp_fast(denom), p);                         // p - step_size * m / (sqrt(v)/... + eps)
}

// a zero-gradient step (the lazy replay): adam1(p, 0, m, s, ...) with the g == 0 branches folded by hand —
// (0 - m)(1 - b1) + m rounds exactly as the general form does, and s takes the sqrt(b2) s branch
template <typename T>
__device__ __forceinline__ void adam1_zero_grad(T &p, T &m, T &s, float one_m_b1, float sqrt_b2, float sa, float sb) {
    const T c1 = rp_splat(one_m_b1, p);
    m = rp_fma(-m, c1, m);
    s = s * sqrt_b2;
    const T denom = rp_fma(s, rp_splat(sa, p), rp_splat(sb, p));
    p = rp_fma(m, rp_rcp_fast(denom), p);
}

// {A_t, B_t} of step t (see adam1): shared by the dense launch and by the step table of the lazy execution
static void adam_scalars(double lr, double beta1, double beta2, double eps, int64_t step, float *sa, float *sb) {
    const double bc1 = 1.0 - std::pow(beta1, (double)step);
    const double bc2 = 1.0 - std::pow(beta2, (double)step);
    const double ns = -lr / bc1;  // minus the step size
    if (ns == 0.0) {                      // lr = 0: s * 0 - inf = -inf, rcp = -0, p + m * (-0) = p
        *sa = 0.f;
        *sb = -INFINITY;
        return;
    }
    *sa = (float)((1.0 / std::sqrt(bc2)) / ns);
    *sb = (float)(eps / ns);
}

// t_dev != NULL (hipGraph replays: the launch arguments are frozen at capture): the step scalars are read from the
// device table sc[] at row *t_dev + 1, *t_dev = number of completed steps
template <bool ZERO_G>
__global__ __launch_bounds__(256) void adam_kernel(AdamPtrs a, float one_m_b1, float b2, float sqrt_b2, float one_m_b2,
                                                   float sa, float sb, const float2 *__restrict__ sc,
                                                   const int32_t *__restrict__ t_dev) {
    if (t_dev != nullptr) {
        const float2 x = sc[*t_dev + 1];
        sa = x.x;
        sb = x.y;
    }
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
        adam1<f32x4>(p, g, m, v, one_m_b1, b2, sqrt_b2, one_m_b2, sa, sb);
        reinterpret_cast<f32x4 *>(P)[e] = p;
        reinterpret_cast<f32x4 *>(Mo)[e] = m;
        reinterpret_cast<f32x4 *>(Vo)[e] = v;
        if (ZERO_G) reinterpret_cast<f32x4 *>(G)[e] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (int64_t e = n4 * 4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += stride) {
        float p = P[e], m = Mo[e], v = Vo[e];
        const float g = G[e];
        adam1<float>(p, g, m, v, one_m_b1, b2, sqrt_b2, one_m_b2, sa, sb);
        P[e] = p;
        Mo[e] = m;
        Vo[e] = v;
        if (ZERO_G) G[e] = 0.f;
    }
}

extern "C" int rp_adam_step(float *const *p_ptrs, float *const *g_ptrs, float *const *m_ptrs, float *const *v_ptrs,
                            const int64_t *sizes, int n_tensors, double lr, double beta1, double beta2, double eps,
                            int64_t step, int zero_grad, const float *step_scalars, const int32_t *t_dev,
                            rp_stream_t stream) {
    RP_REQUIRE(t_dev == nullptr || step_scalars != nullptr, "adam_step: a device step counter needs the scalar table");
    RP_REQUIRE(p_ptrs && g_ptrs && m_ptrs && v_ptrs && sizes, "adam_step: null pointer");
    RP_REQUIRE(n_tensors >= 1 && n_tensors <= RP_MAX_FIELDS, "adam_step: n_tensors=%d outside [1,%d]", n_tensors,
               RP_MAX_FIELDS);
    RP_REQUIRE(step >= 1 || t_dev != nullptr, "adam_step: step must be >= 1");
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
    float sa = 0.f, sb = 0.f;  // the two per-step scalars, computed in double as python floats are in torch.optim.Adam
    if (t_dev == nullptr) adam_scalars(lr, beta1, beta2, eps, step, &sa, &sb);
    int64_t bx = rp_cdiv(rp_cdiv(maxn, 4), 256);
    if (bx > 8192) bx = 8192;
    if (bx < 1) bx = 1;
    dim3 grid((unsigned)bx, (unsigned)n_tensors);
    hipStream_t s = (hipStream_t)stream;
    const float one_m_b1 = (float)(1.0 - beta1), one_m_b2 = (float)(1.0 - beta2), b2f = (float)beta2;
    const float sqrt_b2 = (float)std::sqrt(beta2);
    const float2 *sc = reinterpret_cast<const float2 *>(step_scalars);
    if (zero_grad)
        hipLaunchKernelGGL((adam_kernel<true>), grid, dim3(256), 0, s, a, one_m_b1, b2f, sqrt_b2, one_m_b2, sa, sb, sc, t_dev);
    else
        hipLaunchKernelGGL((adam_kernel<false>), grid, dim3(256), 0, s, a, one_m_b1, b2f, sqrt_b2, one_m_b2, sa, sb, sc, t_dev);
    RP_LAUNCH_CHECK("adam_step");
    return RP_OK;
}

// ------------------------------------------------------------------------------------------------
// Exact LAZY dense Adam for arena rows.
//
// Dense Adam moves every row every step, but a row nobody looked up has g = 0 and its update depends
// only on its own (p, m, v) and the step number.  So the zero-gradient steps of a row can be replayed
// later, in registers, when the row is next needed (looked up by a forward, or flushed for a checkpoint):
//   last[row] = step at which the stored (p, m, v) of the row are current (0 = never updated: m = v = 0,
//   where a zero-gradient step is exactly the identity).
// The replay runs the SAME adam1() as the dense kernel with g = 0 and the per-step scalars of each
// skipped step (sc[j] = {lr_j/(1-b1^j), sqrt(1-b2^j)}), so the result is bit-identical to having run the
// dense kernel every step — at ~1/10 of the HBM traffic (only touched rows move).
// One D/4-lane group (D lanes when D is not a multiple of 4) per sorted key position; only run heads work
// (unique rows, no write race).
// ------------------------------------------------------------------------------------------------
struct LazyCfg {
    float one_m_b1, b2, sqrt_b2, one_m_b2, eps;
};

// bf16 LOOKUP COPY of the tables (round 4, the bf16-storage training mode): wherever the deferred kernels write a parameter
// row they also write its round-to-nearest-even bf16 image to `shadow` (same [rows, D] layout, 2 bytes per element) — the
// copy the forward gathers from (half the row bytes); the fp32 master and the moments stay what the optimizer works on.
__device__ __forceinline__ uint32_t rp_bf16_rn(float v) {
    uint32_t u = __float_as_uint(v);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return u >> 16;
}
__device__ __forceinline__ void rp_store_bf16(uint16_t *dst, float v) { *dst = (uint16_t)rp_bf16_rn(v); }
__device__ __forceinline__ void rp_store_bf16(uint16_t *dst, f32x4 v) {
    uint2 w;
    w.x = rp_bf16_rn(v[0]) | (rp_bf16_rn(v[1]) << 16);
    w.y = rp_bf16_rn(v[2]) | (rp_bf16_rn(v[3]) << 16);
    *reinterpret_cast<uint2 *>(dst) = w;
}

// ------------------------------------------------------------------------------------------------
// CLOSED-FORM replay of k zero-gradient steps (tolerance mode; the serial replay above stays the bit-exact one).
//
// k zero-gradient steps l+1 .. l+k of one element starting from (p, m, s):
//     m_i = m b1^i,   s_i = s r^i  (r = sqrt(b2)),   p_k = p + m * sum_{i=1..k} w_i / (s a_i + eps)
//     w_i = ns_{l+i} b1^i   (ns_j = -lr_j / (1 - b1^j)),      a_i = r^i / sqrt(1 - b2^{l+i})
// The a_i of the terms that carry weight (b1^i: a window of ~1/(1-b1) steps) differ by a few per cent once the
// bias correction has flattened out, so expand every term around the w-weighted mean abar of the a_i:
//     1 / (s a_i + eps) = q / (1 + y (a_i - abar)),   q = 1 / (s abar + eps),   y = s q  in [0, 1/abar)
//     sum_i w_i / (s a_i + eps) = q * [N0 + y^2 N2 - y^3 N3 + y^4 N4 - ...],   N_n = sum_i w_i (a_i - abar)^n  (N1 = 0)
// |y (a_i - abar)| <= |a_i/abar - 1| for EVERY s >= 0 (s -> 0 and s -> inf included): the series converges uniformly in
// s, with ratio ~2 % when l >= 256 (b2 = 0.999).  Truncated after N4 the relative error of the summed update is
// <= 9e-8 for l >= 256 and <= 7e-9 for l >= 512 (profiles/microbench/probes/closed_form_replay.py, double precision study) — fp32 rounding
// level.  Steps up to CF_FROM (= 256) are therefore replayed serially, everything after in one evaluation:
// one v_rcp_f32 and ~10 fp32 operations per element for ANY k.  The per-k coefficients {abar, N0, N2, -N3, N4, b1^k,
// r^k} come from a device table that rp_lazy_adam_cf_table rebuilds (in double) for the current end step.
// ------------------------------------------------------------------------------------------------
// TABLE LAYOUT (one buffer, two index meanings — so that the per-step rebuild is O(1) in the step count):
//   entry[l].{abar, n0, n2, n3m, n4}  describe a replay that STARTS at stamp l (steps l+1 .. t_end).  Only the first
//       J ~ 372 terms carry weight (b1^J is below double rounding), so an entry is final once t_end >= l + J: per step only
//       the J youngest stamps are rebuilt (rp_lazy_adam_cf_table), everything older stays as it was written;
//   entry[k].{pm, ps} = {b1^k, r^k}   depend on the NUMBER of skipped steps only: written once per buffer.
struct CfEntry {
    float abar, n0, n2, n3m, n4, pm, ps, pad;  // n3m = -N3;  pm = b1^k, ps = r^k
};

// the coefficients of the replay of steps l+1 .. l+k
__device__ __forceinline__ CfEntry cf_lookup(const CfEntry *__restrict__ cf, int l, int k) {
    CfEntry e = cf[l];
    e.pm = cf[k].pm;
    e.ps = cf[k].ps;
    return e;
}

template <typename T>
__device__ __forceinline__ void adam_zero_grad_closed(T &p, T &m, T &s, const CfEntry &e, float eps) {
    const T q = rp_rcp_fast(rp_fma(s, rp_splat(e.abar, p), rp_splat(eps, p)));
    const T y = s * q;
    T poly = rp_fma(y, rp_splat(e.n4, p), rp_splat(e.n3m, p));
    poly = rp_fma(y, poly, rp_splat(e.n2, p));
    poly = rp_fma(y * y, poly, rp_splat(e.n0, p));
    p = rp_fma(m * q, poly, p);
    m = m * e.pm;
    s = s * e.ps;
}

// entry l (t_from <= l < t_end) describes the steps l+1 .. t_end;  nsd[j] = {ns_j, 1/sqrt(1 - b2^j)} (double).
// One WAVE per entry: lane i takes the terms i+1, i+65, ... (at most J ~ 350 carry weight), wave reductions in double in
// a fixed order — the launch sits on the step's critical path (it needs this step's lr), a serial loop per entry cost 25 us.
// The launch covers the stamps [l_lo, t_end): l_lo = t_from for a full build, = (last build's t_end) - J for the per-step
// rebuild (stamps older than that were final already); with t_dev (graph replays: consecutive steps) the window is
// [*t_dev - J - 4, *t_dev) whatever the launch arguments say.
__device__ __forceinline__ double wave_sum_f64(double x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
    return x;
}

__global__ __launch_bounds__(256) void lazy_cf_table_kernel(const double2 *__restrict__ nsd, int t_end, int t_from,
                                                            int l_lo, int J, double b1e, double r,
                                                            CfEntry *__restrict__ out, const int32_t *__restrict__ t_dev) {
    if (t_dev != nullptr) {
        t_end = *t_dev;
        l_lo = t_end - J - 4;
    }
    if (l_lo < t_from) l_lo = t_from;
    const int lane = threadIdx.x & 63;
    const int l = l_lo + blockIdx.x * 4 + (threadIdx.x >> 6);
    if (l >= t_end) return;  // wave-uniform
    const int k = t_end - l;
    const int nterm = k < J ? k : J;  // b1^J is below double rounding: later terms carry no weight
    const double lb = log(b1e), lr_ = log(r);
    double w[8], a[8];  // J <= 512 terms: 8 per lane
    double sw = 0.0, swa = 0.0;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int i = lane + 64 * u + 1;
        w[u] = 0.0;
        a[u] = 0.0;
        if (i <= nterm) {
            const double2 x = nsd[l + i];
            w[u] = x.x * exp(lb * (double)i);
            a[u] = x.y * exp(lr_ * (double)i);
            sw += w[u];
            swa += w[u] * a[u];
        }
    }
    sw = wave_sum_f64(sw);
    swa = wave_sum_f64(swa);
    const double abar = (sw != 0.0) ? swa / sw : 1.0;  // every lr_j = 0: no movement, any abar will do
    double n2 = 0.0, n3 = 0.0, n4 = 0.0;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const double da = a[u] - abar, da2 = da * da;  // (terms beyond nterm have w = 0)
        n2 += w[u] * da2;
        n3 += w[u] * da2 * da;
        n4 += w[u] * da2 * da2;
    }
    n2 = wave_sum_f64(n2);
    n3 = wave_sum_f64(n3);
    n4 = wave_sum_f64(n4);
    if (lane == 0) {  // (pm / ps of this slot belong to k = l: lazy_cf_pow_kernel)
        out[l].abar = (float)abar;
        out[l].n0 = (float)sw;
        out[l].n2 = (float)n2;
        out[l].n3m = (float)(-n3);
        out[l].n4 = (float)n4;
    }
}

// entry[k].{pm, ps} = {b1e^k, r^k} for every k the buffer holds (once per buffer)
__global__ __launch_bounds__(256) void lazy_cf_pow_kernel(CfEntry *__restrict__ out, int capacity, double b1e, double r) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= capacity) return;
    out[k].pm = (float)exp(log(b1e) * (double)k);
    out[k].ps = (float)exp(log(r) * (double)k);
    out[k].pad = 0.f;
}

// REAL: the launch applies a real step (a compile-time copy of `real_step`: the replay and the step are then two kernels
// with two names in a profile, and each drops the other's branches)
template <int TPR, typename T, bool REAL>
__global__ __launch_bounds__(256) void lazy_adam_rows_kernel(const int32_t *__restrict__ sk, int64_t n, int D,
                                                             float *__restrict__ P, float *__restrict__ G,
                                                             float *__restrict__ Mo, float *__restrict__ Vo,
                                                             int32_t *__restrict__ last,
                                                             const float2 *__restrict__ sc, int t_target,
                                                             int real_step, int zero_grad, LazyCfg c,
                                                             const CfEntry *__restrict__ cf, int cf_from,
                                                             const int32_t *__restrict__ t_dev) {
    real_step = REAL ? 1 : 0;
    if (t_dev != nullptr) t_target = *t_dev + (REAL ? 1 : 0);  // *t_dev = completed steps (graph replays)
    constexpr int GPB = 256 / TPR;
    const int t = threadIdx.x % TPR;
    const int64_t i = (int64_t)blockIdx.x * GPB + threadIdx.x / TPR;
    if (i >= n) return;
    const int32_t row = sk[i];
    if (i > 0 && sk[i - 1] == row) return;  // not a run head
    const int l0 = last[row];
    const int t_catch = real_step ? t_target - 1 : t_target;
    if (!real_step && (l0 == 0 || l0 >= t_catch)) return;  // never updated (identity) or already current
    constexpr int VW = sizeof(T) / sizeof(float);  // f32x4 rows (D % 4 == 0, 16-byte aligned arenas) or scalars
    for (int cidx = t * VW; cidx < D; cidx += TPR * VW) {
        const int64_t off = (int64_t)row * D + cidx;
        T p = *reinterpret_cast<T *>(P + off);
        T m = *reinterpret_cast<T *>(Mo + off);
        T v = *reinterpret_cast<T *>(Vo + off);
        if (l0 > 0) {
            const int t_serial = (cf && cf_from < t_catch) ? cf_from : t_catch;  // (exact mode: everything)
#pragma unroll 4
            for (int j = l0 + 1; j <= t_serial; ++j) {
                const float2 s = sc[j];
                adam1_zero_grad<T>(p, m, v, c.one_m_b1, c.sqrt_b2, s.x, s.y);
            }
            const int lc = l0 > t_serial ? l0 : t_serial;
            if (lc < t_catch) adam_zero_grad_closed<T>(p, m, v, cf_lookup(cf, lc, t_catch - lc), c.eps);
        }
        if (real_step) {
            const T g = *reinterpret_cast<const T *>(G + off);
            const float2 s = sc[t_target];
            adam1<T>(p, g, m, v, c.one_m_b1, c.b2, c.sqrt_b2, c.one_m_b2, s.x, s.y);
            if (zero_grad) *reinterpret_cast<T *>(G + off) = rp_splat(0.f, p);
        }
        *reinterpret_cast<T *>(P + off) = p;
        *reinterpret_cast<T *>(Mo + off) = m;
        *reinterpret_cast<T *>(Vo + off) = v;
    }
    // a never-updated row stays at last = 0 until its first real step: nothing to replay for it
    if (t == 0 && (real_step || l0 > 0)) last[row] = t_target;
}

template <int TPR, typename T>
__global__ __launch_bounds__(256) void lazy_adam_flush_kernel(int64_t R, int D, float *__restrict__ P,
                                                              float *__restrict__ Mo, float *__restrict__ Vo,
                                                              int32_t *__restrict__ last,
                                                              const float2 *__restrict__ sc, int t_target, LazyCfg c,
                                                              const CfEntry *__restrict__ cf, int cf_from) {
    constexpr int GPB = 256 / TPR;
    constexpr int VW = sizeof(T) / sizeof(float);
    const int t = threadIdx.x % TPR;
    const int64_t stride = (int64_t)gridDim.x * GPB;
    // two rows per iteration: both rows' loads are issued before either replay starts (one row at a time left the
    // memory pipe idle during the dependent load -> replay -> store chain)
    for (int64_t row0 = (int64_t)blockIdx.x * GPB + threadIdx.x / TPR; row0 < R; row0 += 2 * stride) {
        int64_t rows[2] = {row0, row0 + stride};
        int l0[2];
        bool need[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            l0[u] = (rows[u] < R) ? last[rows[u]] : 0;
            need[u] = l0[u] > 0 && l0[u] < t_target;
        }
        if (!need[0] && !need[1]) continue;
        for (int cidx = t * VW; cidx < D; cidx += TPR * VW) {
            T p[2], m[2], v[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (need[u]) {
                    const int64_t off = rows[u] * D + cidx;
                    p[u] = *reinterpret_cast<T *>(P + off);
                    m[u] = *reinterpret_cast<T *>(Mo + off);
                    v[u] = *reinterpret_cast<T *>(Vo + off);
                }
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (need[u]) {
                    const int t_serial = (cf && cf_from < t_target) ? cf_from : t_target;
#pragma unroll 4
                    for (int j = l0[u] + 1; j <= t_serial; ++j) {
                        const float2 s = sc[j];
                        adam1_zero_grad<T>(p[u], m[u], v[u], c.one_m_b1, c.sqrt_b2, s.x, s.y);
                    }
                    const int lc = l0[u] > t_serial ? l0[u] : t_serial;
                    if (lc < t_target) adam_zero_grad_closed<T>(p[u], m[u], v[u], cf_lookup(cf, lc, t_target - lc), c.eps);
                    const int64_t off = rows[u] * D + cidx;
                    *reinterpret_cast<T *>(P + off) = p[u];
                    *reinterpret_cast<T *>(Mo + off) = m[u];
                    *reinterpret_cast<T *>(Vo + off) = v[u];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u)
            if (t == 0 && need[u]) last[rows[u]] = t_target;
    }
}

// ------------------------------------------------------------------------------------------------
// Replay with ONE ROW PER WAVE (rows of more than 16 floats).
//
// The replay of a row is a serial chain of (t - last[row]) zero-gradient steps, and that count differs from row to
// row (a geometric distribution around table_rows / batch).  With several rows side by side in a wave (16 lanes x
// float4 per row, the layout of lazy_adam_rows_kernel) every row waits for the longest chain of the four: ~2.1x the
// mean.  Here the 64 lanes of a wave hold the (up to 64 * EPL) floats of ONE row, so the whole chain is wave-uniform:
// the step count and the per-step scalars live in SGPRs (s_load of sc[j], scalar loop), no lane idles, and a wave
// walks over the rows that need work among 64 consecutive candidates (a ballot over "is a run head / is behind"),
// fetching the next row's p/m/v while the current one replays.  Same adam1_zero_grad() as everywhere else:
// bit-identical to the dense kernel.
// ------------------------------------------------------------------------------------------------
template <int EPL>
__device__ __forceinline__ void lazy_load_row(const float *__restrict__ P, const float *__restrict__ Mo,
                                              const float *__restrict__ Vo, int64_t off, int lane, int D,
                                              float (&p)[EPL], float (&m)[EPL], float (&v)[EPL]) {
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        // unguarded (a lane beyond the row re-reads its last column and never stores): a load behind a lane mask
        // could not be counted by the in-order vmcnt waits either
        const int cidx = (lane + 64 * e < D) ? lane + 64 * e : D - 1;
        p[e] = P[off + cidx];
        m[e] = Mo[off + cidx];
        v[e] = Vo[off + cidx];
    }
}

// `need`: this lane's candidate row must be replayed from step l0 (exclusive) to t_target (inclusive)
template <int EPL, bool FULL>  // FULL: D == 64 * EPL, every lane stores (no lane mask -> no branch around the stores)
__device__ __forceinline__ void lazy_replay_candidates(bool need, int row, int l0, int D, float *__restrict__ P,
                                                       float *__restrict__ Mo, float *__restrict__ Vo,
                                                       int32_t *__restrict__ last, const float2 *__restrict__ sc,
                                                       int t_target, const LazyCfg &c) {
    const int lane = threadIdx.x & 63;
    unsigned long long mask = __ballot(need);
    if (mask == 0) return;
    int src = __builtin_ctzll(mask);
    mask &= mask - 1;
    int r = __builtin_amdgcn_readlane(row, src);
    int l = __builtin_amdgcn_readlane(l0, src);
    float p[EPL], m[EPL], v[EPL];
    lazy_load_row<EPL>(P, Mo, Vo, (int64_t)r * D, lane, D, p, m, v);
    for (;;) {
        const bool more = mask != 0;  // wave-uniform
        // the next row's loads are issued UNCONDITIONALLY (the last row is simply read once more): gfx950 counts loads
        // on one in-order counter and a load behind a branch would turn the wait for THIS row's registers into
        // vmcnt(0), i.e. into a wait for the prefetch
        if (more) src = __builtin_ctzll(mask);
        mask &= mask - 1;
        const int rn = __builtin_amdgcn_readlane(row, src);
        const int ln = __builtin_amdgcn_readlane(l0, src);
        float pn[EPL], mn[EPL], vn[EPL];
        lazy_load_row<EPL>(P, Mo, Vo, (int64_t)rn * D, lane, D, pn, mn, vn);  // in flight during the replay below
#pragma unroll RP_REPLAY_UNROLL
        for (int j = l + 1; j <= t_target; ++j) {
            const float2 s = sc[j];  // uniform address: scalar load
#pragma unroll
            for (int e = 0; e < EPL; ++e) adam1_zero_grad<float>(p[e], m[e], v[e], c.one_m_b1, c.sqrt_b2, s.x, s.y);
        }
        const int64_t off = (int64_t)r * D;
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            const int cidx = lane + 64 * e;
            if (FULL || cidx < D) {
                P[off + cidx] = p[e];
                Mo[off + cidx] = m[e];
                Vo[off + cidx] = v[e];
            }
        }
        last[r] = t_target;  // every lane, same address, same value: one dword write and no branch
        if (!more) break;
        r = rn;
        l = ln;
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            p[e] = pn[e];
            m[e] = mn[e];
            v[e] = vn[e];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// TWO ROWS PER WAVE-ITERATION on packed fp32 (rows of at most 64 floats: one element of each row per lane).
//
// Every replay ends at the same step t_target, so two rows owing kA >= kB steps share their LAST kB steps — same step
// numbers, same scalars {A_j, B_j} — and those run as v_pk_fma_f32 / v_pk_mul_f32 on the register pair {row A, row B}: four
// issue slots (+ two v_rcp_f32) for two element-steps instead of eight (+ two).  The kA - kB steps before them run on row A
// alone.  To make that head short, the wave first SORTS its 64 candidates by the number of steps they owe (bitonic over
// the lanes: ~170 instructions against ~6000 of replay) and pairs neighbours: with ~45 rows per wave, 97 % of the
// element-steps are shared ones.  An odd row out is paired with itself (both halves compute and store the same values).
// Same adam1_zero_grad() per element and step as everywhere else, packed or not: bit-identical to the dense kernel.
// ------------------------------------------------------------------------------------------------
template <bool FULL>  // FULL: D == 64, every lane loads and stores its own column
__device__ __forceinline__ void lazy_replay_pairs(bool need, int row, int l0, int D, float *__restrict__ P,
                                                  float *__restrict__ Mo, float *__restrict__ Vo,
                                                  int32_t *__restrict__ last, const float2 *__restrict__ sc, int t_target,
                                                  const LazyCfg &c) {
    const int lane = threadIdx.x & 63;
    const int n_need = __popcll(__ballot(need));
    if (n_need == 0) return;
    int key = need ? t_target - l0 : 0;  // steps owed (> 0 exactly for the candidates that need work)
    int val = row;
    // bitonic sort of the 64 lanes, DESCENDING by key (payload: the row)
#pragma unroll
    for (int size = 2; size <= 64; size <<= 1) {
#pragma unroll
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            const int pk = __shfl_xor(key, stride, 64);
            const int pv = __shfl_xor(val, stride, 64);
            const bool desc = (lane & size) == 0;       // direction of this lane's block (the last merge: all descending)
            const bool lower = (lane & stride) == 0;
            const bool take = (lower == desc) ? (pk > key) : (pk < key);  // the lower lane of a descending pair keeps the max
            key = take ? pk : key;
            val = take ? pv : val;
        }
    }
    const int npairs = (n_need + 1) >> 1;
    const int col = FULL ? lane : (lane < D ? lane : D - 1);  // (a lane beyond the row re-reads its last column, never stores)
    auto pick = [&](int i, int &ra, int &ka, int &rb, int &kb) {
        const int ia = 2 * i, ib = (2 * i + 1 < n_need) ? 2 * i + 1 : 2 * i;  // odd row out: paired with itself
        ra = __builtin_amdgcn_readlane(val, ia);
        ka = __builtin_amdgcn_readlane(key, ia);
        rb = __builtin_amdgcn_readlane(val, ib);
        kb = __builtin_amdgcn_readlane(key, ib);
    };
    auto load = [&](int ra, int rb, f32x2a &p, f32x2a &m, f32x2a &v) {
        const int64_t oa = (int64_t)ra * D + col, ob = (int64_t)rb * D + col;
        p = f32x2a{P[oa], P[ob]};
        m = f32x2a{Mo[oa], Mo[ob]};
        v = f32x2a{Vo[oa], Vo[ob]};
    };
    int ra, ka, rb, kb;
    pick(0, ra, ka, rb, kb);
    f32x2a p, m, v;
    load(ra, rb, p, m, v);
    for (int i = 0; i < npairs; ++i) {
        // the next pair's loads are issued UNCONDITIONALLY (the last pair is simply read once more): see
        // lazy_replay_candidates
        const int inext = (i + 1 < npairs) ? i + 1 : i;
        int rna, kna, rnb, knb;
        pick(inext, rna, kna, rnb, knb);
        f32x2a pn, mn, vn;
        load(rna, rnb, pn, mn, vn);  // in flight during the replay below
        int j = t_target - ka + 1;
        const int jshared = t_target - kb + 1;
#pragma unroll RP_REPLAY_UNROLL
        for (; j < jshared; ++j) {  // row A alone
            const float2 s = sc[j];
            float pa = p.x, ma = m.x, va = v.x;
            adam1_zero_grad<float>(pa, ma, va, c.one_m_b1, c.sqrt_b2, s.x, s.y);
            p.x = pa;
            m.x = ma;
            v.x = va;
        }
#pragma unroll RP_REPLAY_UNROLL
        for (; j <= t_target; ++j) {  // both rows, packed
            const float2 s = sc[j];  // uniform address: scalar load
            adam1_zero_grad<f32x2a>(p, m, v, c.one_m_b1, c.sqrt_b2, s.x, s.y);
        }
        if (FULL || lane < D) {
            const int64_t oa = (int64_t)ra * D + col, ob = (int64_t)rb * D + col;
            P[oa] = p.x;
            Mo[oa] = m.x;
            Vo[oa] = v.x;
            P[ob] = p.y;
            Mo[ob] = m.y;
            Vo[ob] = v.y;
        }
        last[ra] = t_target;  // every lane, same address, same value: one dword write and no branch
        last[rb] = t_target;
        ra = rna;
        ka = kna;
        rb = rnb;
        kb = knb;
        p = pn;
        m = mn;
        v = vn;
    }
}

// candidates = 64 positions of the sorted key list, STRIDED by the number of waves (lane l of wave w looks at position
// l * n_waves + w): the list is field-major, so 64 consecutive positions belong to one table and all carry that
// table's revisit gap — the five 2-10 M-row tables of a Criteo-shaped arena hold 95 % of the replay work in 19 % of
// the positions, i.e. in ~5 waves per SIMD with consecutive chunks, each busy for > 100 us while the rest of the grid
// has long finished.  Strided, every wave gets the same mix of tables (the same expected work) and the launch keeps
// all its resident waves busy to the end.  A position counts when it is a run head (unique rows, no write race) whose
// row has been updated before (last > 0; otherwise m = v = 0 and a zero-gradient step is the identity) and is
// behind t_target.
template <int EPL, bool FULL>
__global__ __launch_bounds__(256) void lazy_replay_wave_kernel(const int32_t *__restrict__ sk, int64_t n, int D,
                                                               float *__restrict__ P, float *__restrict__ Mo,
                                                               float *__restrict__ Vo, int32_t *__restrict__ last,
                                                               const float2 *__restrict__ sc, int t_target, LazyCfg c,
                                                               const int32_t *__restrict__ t_dev) {
    if (t_dev != nullptr) t_target = *t_dev;
    const int64_t n_waves = (int64_t)gridDim.x * 4;
    const int64_t i = (int64_t)(threadIdx.x & 63) * n_waves + ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6));
    int row = 0, l0 = 0;
    bool need = false;
    if (i < n) {
        row = sk[i];
        if (i == 0 || sk[i - 1] != row) {
            l0 = last[row];
            need = l0 > 0 && l0 < t_target;
        }
    }
    if constexpr (EPL == 1) lazy_replay_pairs<FULL>(need, row, l0, D, P, Mo, Vo, last, sc, t_target, c);
    else lazy_replay_candidates<EPL, FULL>(need, row, l0, D, P, Mo, Vo, last, sc, t_target, c);
}

// candidates = 64 consecutive arena rows; the grid strides over the arena
template <int EPL, bool FULL>
__global__ __launch_bounds__(256) void lazy_flush_wave_kernel(int64_t R, int D, float *__restrict__ P,
                                                              float *__restrict__ Mo, float *__restrict__ Vo,
                                                              int32_t *__restrict__ last,
                                                              const float2 *__restrict__ sc, int t_target, LazyCfg c) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t row0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 64; row0 < R; row0 += stride) {
        const int64_t row = row0 + (threadIdx.x & 63);
        int l0 = 0;
        bool need = false;
        if (row < R) {
            l0 = last[row];
            need = l0 > 0 && l0 < t_target;
        }
        if constexpr (EPL == 1) lazy_replay_pairs<FULL>(need, (int)row, l0, D, P, Mo, Vo, last, sc, t_target, c);
        else lazy_replay_candidates<EPL, FULL>(need, (int)row, l0, D, P, Mo, Vo, last, sc, t_target, c);
    }
}

#define RP_LAZY_WAVE_MIN_D 17   // rows this wide or wider replay one row per wave
#define RP_LAZY_WAVE_MAX_D 256  // 4 floats per lane

static int lazy_tpr(int D, int vw) {
    int need = (D + vw - 1) / vw, tpr = 1;
    while (tpr < need && tpr < 64) tpr <<= 1;
    return tpr;
}

#define LAZY_DISPATCH_T(tpr, TY, CALL) \
    switch (tpr) {                      \
        case 1: CALL(1, TY); break;     \
        case 2: CALL(2, TY); break;     \
        case 4: CALL(4, TY); break;     \
        case 8: CALL(8, TY); break;     \
        case 16: CALL(16, TY); break;   \
        case 32: CALL(32, TY); break;   \
        default: CALL(64, TY); break;   \
    }
#define LAZY_DISPATCH(tpr, vw, CALL)                 \
    do {                                             \
        if (vw == 4) {                               \
            LAZY_DISPATCH_T(tpr, f32x4, CALL)        \
        } else {                                     \
            LAZY_DISPATCH_T(tpr, float, CALL)        \
        }                                            \
    } while (0)

// host helper shared with the python side: the two per-step scalars exactly as rp_adam_step derives them
extern "C" int rp_adam_step_scalars(double lr, double beta1, double beta2, double eps, int64_t step, float *sa,
                                    float *sb) {
    RP_REQUIRE(sa && sb && step >= 1, "adam_step_scalars: bad argument");
    adam_scalars(lr, beta1, beta2, eps, step, sa, sb);
    return RP_OK;
}

// n consecutive steps step0 .. step0 + n - 1 in one call: out[2 i] = A, out[2 i + 1] = B of step step0 + i (host array).  The
// python side fills its per-step device table 1024 steps ahead: one C call instead of 1025 (that loop cost the host ~1 ms
// every 1024th training step — the 2 ms step in profiles/r04_trace_periods.txt)
extern "C" int rp_adam_step_scalars_range(double lr, double beta1, double beta2, double eps, int64_t step0, int64_t n,
                                          float *out) {
    RP_REQUIRE(out && step0 >= 1 && n >= 0, "adam_step_scalars_range: bad argument");
    for (int64_t i = 0; i < n; ++i) adam_scalars(lr, beta1, beta2, eps, step0 + i, out + 2 * i, out + 2 * i + 1);
    return RP_OK;
}

// the effective decay factors of one zero-gradient step as the kernels execute it in fp32: m <- m - m * fl(1 - b1),
// s <- s * fl(sqrt(b2)); the closed form uses their exact (double) values so that it tracks the serial replay
static void cf_decay_factors(double beta1, double beta2, double *b1e, double *r) {
    *b1e = 1.0 - (double)(float)(1.0 - beta1);
    *r = (double)(float)std::sqrt(beta2);
}

extern "C" int rp_lazy_adam_cf_terms(double beta1, int *terms) {
    RP_REQUIRE(terms, "lazy_adam_cf_terms: null pointer");
    RP_REQUIRE(beta1 > 0.0 && beta1 < 1.0, "lazy_adam_cf_terms: beta1 must be inside (0, 1)");
    double b1e, r;
    cf_decay_factors(beta1, 0.999, &b1e, &r);
    const double j = std::ceil(std::log(1e-17) / std::log(b1e));
    *terms = j > 1e6 ? 1000000 : (int)j;  // (0.9: 372; the table kernel holds up to 512 terms)
    return RP_OK;
}

extern "C" int rp_lazy_adam_cf_table(const double *ns_d, int64_t t_end, int64_t cf_from, double beta1, double beta2,
                                     float *cf_table, int64_t capacity, int64_t built_to, const int32_t *t_dev,
                                     rp_stream_t stream) {
    RP_REQUIRE(ns_d && cf_table, "lazy_adam_cf_table: null pointer");
    RP_REQUIRE(cf_from >= 1 && t_end < INT32_MAX && capacity >= 1 && capacity < INT32_MAX, "lazy_adam_cf_table: bad step range");
    RP_REQUIRE(t_dev != nullptr || t_end < capacity, "lazy_adam_cf_table: step %lld outside the table (%lld entries)",
               (long long)t_end, (long long)capacity);
    RP_REQUIRE((((uintptr_t)ns_d) & 15u) == 0 && (((uintptr_t)cf_table) & 31u) == 0,
               "lazy_adam_cf_table: tables must be 16 / 32-byte aligned");
    int J;
    if (rp_lazy_adam_cf_terms(beta1, &J) != RP_OK) return RP_ERR_ARG;
    RP_REQUIRE(J <= 512, "lazy_adam_cf_table: beta1 = %g needs %d terms, the closed form holds 512 (use the serial replay)",
               (double)beta1, J);
    double b1e, r;
    cf_decay_factors(beta1, beta2, &b1e, &r);
    hipStream_t s = (hipStream_t)stream;
    CfEntry *out = reinterpret_cast<CfEntry *>(cf_table);
    if (built_to < 0) {  // a fresh buffer: the power columns, then every stamp
        hipLaunchKernelGGL(lazy_cf_pow_kernel, dim3((unsigned)rp_cdiv(capacity, 256)), dim3(256), 0, s, out, (int)capacity, b1e, r);
        RP_LAUNCH_CHECK("lazy_adam_cf_table (powers)");
    }
    int64_t l_lo = built_to < 0 ? cf_from : built_to - J;
    if (l_lo < cf_from) l_lo = cf_from;
    // graph replays (t_dev): consecutive steps, the window of the J + 4 youngest stamps whatever the step number is
    const int64_t span = t_dev != nullptr ? (int64_t)J + 4 : t_end - l_lo;
    if (span <= 0) return RP_OK;
    const unsigned grid = (unsigned)rp_cdiv(span, 4);  // one wave per entry
    hipLaunchKernelGGL(lazy_cf_table_kernel, dim3(grid), dim3(256), 0, s, reinterpret_cast<const double2 *>(ns_d), (int)t_end,
                       (int)cf_from, (int)l_lo, J, b1e, r, out, t_dev);
    RP_LAUNCH_CHECK("lazy_adam_cf_table");
    return RP_OK;
}

extern "C" int rp_lazy_adam_rows(const int32_t *sorted_keys, int64_t n, int D, float *p, float *g, float *m, float *v,
                                 int32_t *last, const float *step_scalars, int64_t t_target, int real_step,
                                 int zero_grad, double beta1, double beta2, double eps, const float *cf_table,
                                 int64_t cf_from, const int32_t *t_dev, rp_stream_t stream) {
    RP_REQUIRE(sorted_keys && p && m && v && last && step_scalars, "lazy_adam_rows: null pointer");
    RP_REQUIRE(!cf_table || (cf_from >= 1 && eps > 0.0 && (((uintptr_t)cf_table) & 31u) == 0),
               "lazy_adam_rows: the closed-form replay needs cf_from >= 1, eps > 0 and a 32-byte aligned table");
    RP_REQUIRE(!real_step || g, "lazy_adam_rows: a real step needs the gradient arena");
    RP_REQUIRE(D >= 1, "lazy_adam_rows: D must be positive");
    const int vw = (D % 4 == 0 && rp_aligned16(p) && rp_aligned16(m) && rp_aligned16(v) && (!g || rp_aligned16(g))) ? 4 : 1;
    RP_REQUIRE(t_dev != nullptr || (t_target >= (real_step ? 1 : 0) && t_target < INT32_MAX), "lazy_adam_rows: bad step");
    if (n == 0) return RP_OK;
    LazyCfg c{(float)(1.0 - beta1), (float)beta2, (float)std::sqrt(beta2), (float)(1.0 - beta2), (float)eps};
    const int tpr = lazy_tpr(D, vw);
    const unsigned grid = (unsigned)rp_cdiv(n, 256 / tpr);
    hipStream_t s = (hipStream_t)stream;
    const float2 *sc = reinterpret_cast<const float2 *>(step_scalars);
    const CfEntry *cf = reinterpret_cast<const CfEntry *>(cf_table);
    // closed form: every row costs the same O(1) work, the launch is an HBM stream -> the vector-lane kernel below
    if (!cf && !real_step && D >= RP_LAZY_WAVE_MIN_D && D <= RP_LAZY_WAVE_MAX_D) {  // serial replay: one row per wave
        const dim3 gw((unsigned)rp_cdiv(n, 256));
#define CALLW(EPL, FULL)                                                                                              \
    hipLaunchKernelGGL((lazy_replay_wave_kernel<EPL, FULL>), gw, dim3(256), 0, s, sorted_keys, n, D, p, m, v, last, sc, \
                       (int)t_target, c, t_dev)
        if (D == 64) CALLW(1, true);
        else if (D < 64) CALLW(1, false);
        else if (D <= 128) CALLW(2, false);
        else CALLW(4, false);
#undef CALLW
        RP_LAUNCH_CHECK("lazy_adam_rows (replay)");
        return RP_OK;
    }
#define CALL(T, TY)                                                                                                       \
    do {                                                                                                                  \
        if (real_step)                                                                                                    \
            hipLaunchKernelGGL((lazy_adam_rows_kernel<T, TY, true>), dim3(grid), dim3(256), 0, s, sorted_keys, n, D, p, g, \
                               m, v, last, sc, (int)t_target, 1, zero_grad, c, cf, (int)cf_from, t_dev);                   \
        else                                                                                                              \
            hipLaunchKernelGGL((lazy_adam_rows_kernel<T, TY, false>), dim3(grid), dim3(256), 0, s, sorted_keys, n, D, p,   \
                               g, m, v, last, sc, (int)t_target, 0, zero_grad, c, cf, (int)cf_from, t_dev);                \
    } while (0)
    LAZY_DISPATCH(tpr, vw, CALL);
#undef CALL
    RP_LAUNCH_CHECK("lazy_adam_rows");
    return RP_OK;
}

extern "C" int rp_lazy_adam_flush(int64_t rows, int D, float *p, float *m, float *v, int32_t *last,
                                  const float *step_scalars, int64_t t_target, double beta1, double beta2, double eps,
                                  const float *cf_table, int64_t cf_from, rp_stream_t stream) {
    RP_REQUIRE(p && m && v && last && step_scalars, "lazy_adam_flush: null pointer");
    RP_REQUIRE(!cf_table || (cf_from >= 1 && eps > 0.0 && (((uintptr_t)cf_table) & 31u) == 0),
               "lazy_adam_flush: the closed-form replay needs cf_from >= 1, eps > 0 and a 32-byte aligned table");
    const CfEntry *cf = reinterpret_cast<const CfEntry *>(cf_table);
    RP_REQUIRE(D >= 1, "lazy_adam_flush: D must be positive");
    const int vw = (D % 4 == 0 && rp_aligned16(p) && rp_aligned16(m) && rp_aligned16(v)) ? 4 : 1;
    if (rows == 0 || t_target <= 0) return RP_OK;
    LazyCfg c{(float)(1.0 - beta1), (float)beta2, (float)std::sqrt(beta2), (float)(1.0 - beta2), (float)eps};
    const int tpr = lazy_tpr(D, vw);
    int64_t nb = rp_cdiv(rows, 256 / tpr);
    if (nb > 65536) nb = 65536;
    hipStream_t s = (hipStream_t)stream;
    const float2 *sc = reinterpret_cast<const float2 *>(step_scalars);
    if (!cf && D >= RP_LAZY_WAVE_MIN_D && D <= RP_LAZY_WAVE_MAX_D) {
        int64_t nw = rp_cdiv(rows, 256);
        if (nw > 65536) nw = 65536;
#define CALLW(EPL, FULL)                                                                                                   \
    hipLaunchKernelGGL((lazy_flush_wave_kernel<EPL, FULL>), dim3((unsigned)nw), dim3(256), 0, s, rows, D, p, m, v, last, sc, \
                       (int)t_target, c)
        if (D == 64) CALLW(1, true);
        else if (D < 64) CALLW(1, false);
        else if (D <= 128) CALLW(2, false);
        else CALLW(4, false);
#undef CALLW
        RP_LAUNCH_CHECK("lazy_adam_flush (wave)");
        return RP_OK;
    }
#define CALL(T, TY)                                                                                              \
    hipLaunchKernelGGL((lazy_adam_flush_kernel<T, TY>), dim3((unsigned)nb), dim3(256), 0, s, rows, D, p, m, v, last, sc, \
                       (int)t_target, c, cf, (int)cf_from)
    LAZY_DISPATCH(tpr, vw, CALL);
#undef CALL
    RP_LAUNCH_CHECK("lazy_adam_flush");
    return RP_OK;
}

// ------------------------------------------------------------------------------------------------
// DEFERRED execution of the real step (opt-in: LazyAdamRows(defer=True), rec_pangu_amd/optim.py).
//
// The lazy execution above touches the rows of a batch twice per training step: the replay before the forward (read
// p, m, s; write p, m, s: 6 rows) and the real step after the backward (read p, m, s, g; write p, m, s, g = 0: 8 rows).
// But a row's real step needs nothing but its own gradient row, which stays where the backward wrote it (the gradient
// arena is dense: one row per table row).  So the step of a row can wait, like its zero-gradient steps do, until the
// row is next needed — and then ONE launch per training step does everything its batch's rows are owed:
//     the pending real step (step number l+1, its own scalars, the stored gradient row; the row is cleared),
//     then the zero-gradient steps l+2 .. t_done (serial up to cf_from, closed form beyond),
// 8 rows of traffic per unique row instead of 14.  Per row the SAME operations run on the SAME values in the SAME order
// as in the immediate execution (and, with the serial replay, as in the dense kernel): bit-identical results after a
// flush; only their time of execution moves.
//   last[row] >= 0 : (p, m, s) are current through step last[row]; no gradient pending (0 = never updated)
//   last[row] <  0 : current through step l = -last[row] - 1, and grad_arena[row] holds the gradient of step l + 1
// A pending gradient is applicable once its step has been taken (l + 1 <= t_done); one whose step is still in progress
// (a second forward before optimizer.step(): gradient accumulation, or an evaluation pass) is left alone.
// mark: this launch precedes a forward whose backward will write the rows' gradients of step t_done + 1: they are
// stamped pending for it (a row that then receives no gradient holds zeros: adam1 with g = 0 IS the zero-gradient step).
// ------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void lazy_owed(T &p, T &m, T &v, bool apply, int l, int t_done, const T g,
                                          const float2 *__restrict__ sc, const LazyCfg &c,
                                          const CfEntry *__restrict__ cf, int cf_from) {
    int lc = l;
    if (apply) {
        const float2 s = sc[l + 1];
        adam1<T>(p, g, m, v, c.one_m_b1, c.b2, c.sqrt_b2, c.one_m_b2, s.x, s.y);
        lc = l + 1;
    }
    if (lc > 0 && lc < t_done) {
        const int t_serial = (cf && cf_from < t_done) ? cf_from : t_done;  // (exact mode: everything)
#pragma unroll 4
        for (int j = lc + 1; j <= t_serial; ++j) {
            const float2 s = sc[j];
            adam1_zero_grad<T>(p, m, v, c.one_m_b1, c.sqrt_b2, s.x, s.y);
        }
        const int l2 = lc > t_serial ? lc : t_serial;
        if (l2 < t_done) adam_zero_grad_closed<T>(p, m, v, cf_lookup(cf, l2, t_done - l2), c.eps);
    }
}

// NCH column chunks of one row per lane (TPR = D / (NCH * VW) lanes per row): every load of the row is issued before
// the first use, and a wave holds 64 / TPR rows.  Measured at D = 64 in the long-run DeepFM step: 16 lanes x 1 chunk 0.239 ms,
// 8 x 2 0.230 ms.
template <int NCH, int TPR, typename T>
__device__ __forceinline__ void catchup_row_chunks(int32_t row, int D, int t, float *__restrict__ P, float *__restrict__ G,
                                                   float *__restrict__ Mo, float *__restrict__ Vo,
                                                   uint16_t *__restrict__ shadow, bool apply, int mark, int l, int t_done,
                                                   const float2 *__restrict__ sc, const LazyCfg &c,
                                                   const CfEntry *__restrict__ cf, int cf_from) {
    constexpr int VW = sizeof(T) / sizeof(float);
    T pp[NCH], mm[NCH], vv[NCH], gg[NCH];
    int64_t off[NCH];
#pragma unroll
    for (int it = 0; it < NCH; ++it) {
        off[it] = (int64_t)row * D + (t + it * TPR) * VW;
        pp[it] = *reinterpret_cast<T *>(P + off[it]);
        mm[it] = *reinterpret_cast<T *>(Mo + off[it]);
        vv[it] = *reinterpret_cast<T *>(Vo + off[it]);
        gg[it] = rp_splat(0.f, pp[it]);
    }
    if (apply) {
#pragma unroll
        for (int it = 0; it < NCH; ++it) gg[it] = *reinterpret_cast<const T *>(G + off[it]);
        if (mark != 2) {  // (mark == 2: the backward this launch precedes overwrites the row, see below)
#pragma unroll
            for (int it = 0; it < NCH; ++it) *reinterpret_cast<T *>(G + off[it]) = rp_splat(0.f, pp[it]);
        }
    }
#pragma unroll
    for (int it = 0; it < NCH; ++it) {
        lazy_owed<T>(pp[it], mm[it], vv[it], apply, l, t_done, gg[it], sc, c, cf, cf_from);
        *reinterpret_cast<T *>(P + off[it]) = pp[it];
        if (shadow != nullptr) rp_store_bf16(shadow + off[it], pp[it]);
        *reinterpret_cast<T *>(Mo + off[it]) = mm[it];
        *reinterpret_cast<T *>(Vo + off[it]) = vv[it];
    }
}

template <int TPR, typename T>
__global__ __launch_bounds__(256) void lazy_adam_catchup_kernel(const int32_t *__restrict__ sk, int64_t n, int D,
                                                                float *__restrict__ P, float *__restrict__ G,
                                                                float *__restrict__ Mo, float *__restrict__ Vo,
                                                                int32_t *__restrict__ last,
                                                                const float2 *__restrict__ sc, int t_done, int mark,
                                                                LazyCfg c, const CfEntry *__restrict__ cf, int cf_from,
                                                                const int32_t *__restrict__ t_dev,
                                                                uint16_t *__restrict__ shadow) {
    if (t_dev != nullptr) t_done = *t_dev;  // completed steps (graph replays)
    constexpr int GPB = 256 / TPR;
    const int t = threadIdx.x % TPR;
    const int64_t i = (int64_t)blockIdx.x * GPB + threadIdx.x / TPR;
    if (i >= n) return;
    const int32_t row = sk[i];
    if (i > 0 && sk[i - 1] == row) return;  // not a run head
    const int raw = last[row];
    const bool pend = raw < 0;
    const int l = pend ? -raw - 1 : raw;
    const bool apply = pend && l + 1 <= t_done && G != nullptr;
    const bool behind = apply || (l > 0 && l < t_done);
    constexpr int VW = sizeof(T) / sizeof(float);
    if (behind && D == 2 * TPR * VW) {
        catchup_row_chunks<2, TPR, T>(row, D, t, P, G, Mo, Vo, shadow, apply, mark, l, t_done, sc, c, cf, cf_from);
    } else if (behind && D == 4 * TPR * VW) {
        catchup_row_chunks<4, TPR, T>(row, D, t, P, G, Mo, Vo, shadow, apply, mark, l, t_done, sc, c, cf, cf_from);
    } else if (behind) {
        for (int cidx = t * VW; cidx < D; cidx += TPR * VW) {
            const int64_t off = (int64_t)row * D + cidx;
            T p = *reinterpret_cast<T *>(P + off);
            T m = *reinterpret_cast<T *>(Mo + off);
            T v = *reinterpret_cast<T *>(Vo + off);
            T g = rp_splat(0.f, p);
            if (apply) {
                g = *reinterpret_cast<const T *>(G + off);
                // mark == 2: the caller promises that the backward this launch precedes OVERWRITES the gradient row of
                // every row it stamps (one non-accumulating gradient launch over the same keys): the clear is dead
                // traffic then, 1 of the 8 row transfers of this launch
                if (mark != 2) *reinterpret_cast<T *>(G + off) = rp_splat(0.f, p);
            }
            lazy_owed<T>(p, m, v, apply, l, t_done, g, sc, c, cf, cf_from);
            *reinterpret_cast<T *>(P + off) = p;
            if (shadow != nullptr) rp_store_bf16(shadow + off, p);
            *reinterpret_cast<T *>(Mo + off) = m;
            *reinterpret_cast<T *>(Vo + off) = v;
        }
    }
    if (t == 0) {
        int nl = raw;
        if (!(pend && !apply)) {  // (a gradient whose step is still in progress stays pending as it is)
            if (mark) nl = -(t_done + 1);
            else if (behind) nl = t_done;
        }
        if (nl != raw) last[row] = nl;
    }
}

// The same launch with the run heads COMPACTED per wave (round 5).  Above, a TPR-lane group is spent on
// every sorted PAIR and only run heads work: 31 % of the groups at Criteo shape (0.53 M unique rows of 1.70 M pairs), so a
// wave carries two or three live rows and as many loads in flight.  Here a wave takes 64 consecutive pairs, finds its run
// heads with one ballot, parks their keys in LDS in rank order and walks them 64 / TPR at a time: every iteration of a wave
// but its last is full.  Same per-row code (catchup_row_chunks), same results bit for bit.
template <int NCH, int TPR, typename T>
__global__ __launch_bounds__(256) void lazy_adam_catchup_wave_kernel(const int32_t *__restrict__ sk, int64_t n, int D,
                                                                     float *__restrict__ P, float *__restrict__ G,
                                                                     float *__restrict__ Mo, float *__restrict__ Vo,
                                                                     int32_t *__restrict__ last,
                                                                     const float2 *__restrict__ sc, int t_done, int mark,
                                                                     LazyCfg c, const CfEntry *__restrict__ cf, int cf_from,
                                                                     const int32_t *__restrict__ t_dev,
                                                                     uint16_t *__restrict__ shadow) {
    if (t_dev != nullptr) t_done = *t_dev;  // completed steps (graph replays)
    __shared__ int32_t heads[4][64];
    constexpr int RPW = 64 / TPR;  // rows a wave works on at once
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool in = i < n;
    const int32_t key = in ? sk[i] : -1;
    const int32_t prev = (in && i > 0) ? sk[i - 1] : -1;
    const bool head = in && (i == 0 || prev != key);
    const unsigned long long mask = __ballot(head);
    const int nh = __popcll(mask);
    if (head) heads[wv][__popcll(mask & ((1ull << lane) - 1ull))] = key;
    __syncthreads();
    const int g = lane / TPR, t = lane % TPR;
    for (int it = 0; it * RPW < nh; ++it) {
        const int q = it * RPW + g;
        if (q >= nh) continue;
        const int32_t row = heads[wv][q];
        const int raw = last[row];
        const bool pend = raw < 0;
        const int l = pend ? -raw - 1 : raw;
        const bool apply = pend && l + 1 <= t_done && G != nullptr;
        const bool behind = apply || (l > 0 && l < t_done);
        if (behind) catchup_row_chunks<NCH, TPR, T>(row, D, t, P, G, Mo, Vo, shadow, apply, mark, l, t_done, sc, c, cf, cf_from);
        if (t == 0) {
            int nl = raw;
            if (!(pend && !apply)) {  // (a gradient whose step is still in progress stays pending as it is)
                if (mark) nl = -(t_done + 1);
                else if (behind) nl = t_done;
            }
            if (nl != raw) last[row] = nl;
        }
    }
}

// every row of the arena: what it is owed through step t_target (checkpoints, state_dict(), evaluation on raw tables)
template <int TPR, typename T>
__global__ __launch_bounds__(256) void lazy_adam_flush_deferred_kernel(int64_t R, int D, float *__restrict__ P,
                                                                       float *__restrict__ G, float *__restrict__ Mo,
                                                                       float *__restrict__ Vo, int32_t *__restrict__ last,
                                                                       const float2 *__restrict__ sc, int t_target,
                                                                       LazyCfg c, const CfEntry *__restrict__ cf,
                                                                       int cf_from, uint16_t *__restrict__ shadow) {
    constexpr int GPB = 256 / TPR;
    constexpr int VW = sizeof(T) / sizeof(float);
    const int t = threadIdx.x % TPR;
    const int64_t stride = (int64_t)gridDim.x * GPB;
    for (int64_t row = (int64_t)blockIdx.x * GPB + threadIdx.x / TPR; row < R; row += stride) {
        const int raw = last[row];
        const bool pend = raw < 0;
        const int l = pend ? -raw - 1 : raw;
        const bool apply = pend && l + 1 <= t_target && G != nullptr;
        const bool behind = apply || (l > 0 && l < t_target);
        if (!behind) continue;
        for (int cidx = t * VW; cidx < D; cidx += TPR * VW) {
            const int64_t off = row * D + cidx;
            T p = *reinterpret_cast<T *>(P + off);
            T m = *reinterpret_cast<T *>(Mo + off);
            T v = *reinterpret_cast<T *>(Vo + off);
            T g = rp_splat(0.f, p);
            if (apply) {
                g = *reinterpret_cast<const T *>(G + off);
                *reinterpret_cast<T *>(G + off) = rp_splat(0.f, p);
            }
            lazy_owed<T>(p, m, v, apply, l, t_target, g, sc, c, cf, cf_from);
            *reinterpret_cast<T *>(P + off) = p;
            if (shadow != nullptr) rp_store_bf16(shadow + off, p);
            *reinterpret_cast<T *>(Mo + off) = m;
            *reinterpret_cast<T *>(Vo + off) = v;
        }
        if (t == 0) last[row] = t_target;
    }
}

extern "C" int rp_lazy_adam_catchup(const int32_t *sorted_keys, int64_t n, int D, float *p, float *g, float *m, float *v,
                                    int32_t *last, const float *step_scalars, int64_t t_done, int mark, double beta1,
                                    double beta2, double eps, const float *cf_table, int64_t cf_from,
                                    const int32_t *t_dev, void *shadow_bf16, rp_stream_t stream) {
    RP_REQUIRE(sorted_keys && p && m && v && last && step_scalars, "lazy_adam_catchup: null pointer");
    RP_REQUIRE(shadow_bf16 == nullptr || (reinterpret_cast<uintptr_t>(shadow_bf16) & 15u) == 0, "lazy_adam_catchup: unaligned bf16 copy");
    RP_REQUIRE(!cf_table || (cf_from >= 1 && eps > 0.0 && (((uintptr_t)cf_table) & 31u) == 0),
               "lazy_adam_catchup: the closed-form replay needs cf_from >= 1, eps > 0 and a 32-byte aligned table");
    RP_REQUIRE(D >= 1, "lazy_adam_catchup: D must be positive");
    RP_REQUIRE(t_dev != nullptr || (t_done >= 0 && t_done < INT32_MAX - 1), "lazy_adam_catchup: bad step");
    if (n == 0) return RP_OK;
    const int vw = (D % 4 == 0 && rp_aligned16(p) && rp_aligned16(m) && rp_aligned16(v) && (!g || rp_aligned16(g))) ? 4 : 1;
    LazyCfg c{(float)(1.0 - beta1), (float)beta2, (float)std::sqrt(beta2), (float)(1.0 - beta2), (float)eps};
    int tpr = lazy_tpr(D, vw);
    // column chunks per lane (catchup_row_chunks): RP_CATCHUP_CHUNKS = 1, 2 (default) or 4
    static const int chunks = getenv("RP_CATCHUP_CHUNKS") ? atoi(getenv("RP_CATCHUP_CHUNKS")) : 2;
    if ((chunks == 2 || chunks == 4) && tpr >= chunks && D == tpr * vw) tpr /= chunks;
    hipStream_t s = (hipStream_t)stream;
    const float2 *sc = reinterpret_cast<const float2 *>(step_scalars);
    const CfEntry *cf = reinterpret_cast<const CfEntry *>(cf_table);
    // D = 64 rows as two float4 chunks per lane (the layers of every BASELINE config): run heads compacted per wave
    // (RP_CATCHUP_WAVE=0: one lane group per sorted pair, as before)
    static const bool wave_form = !(getenv("RP_CATCHUP_WAVE") && getenv("RP_CATCHUP_WAVE")[0] == '0');
    if (wave_form && vw == 4 && D == 64 && tpr == 8) {
        hipLaunchKernelGGL((lazy_adam_catchup_wave_kernel<2, 8, f32x4>), dim3((unsigned)rp_cdiv(n, 256)), dim3(256), 0, s, sorted_keys,
                           n, D, p, g, m, v, last, sc, (int)t_done, mark, c, cf, (int)cf_from, t_dev,
                           reinterpret_cast<uint16_t *>(shadow_bf16));
        RP_LAUNCH_CHECK("lazy_adam_catchup");
        return RP_OK;
    }
    const unsigned grid = (unsigned)rp_cdiv(n, 256 / tpr);
#define CALL(T, TY)                                                                                                      \
    hipLaunchKernelGGL((lazy_adam_catchup_kernel<T, TY>), dim3(grid), dim3(256), 0, s, sorted_keys, n, D, p, g, m, v, last, \
                       sc, (int)t_done, mark, c, cf, (int)cf_from, t_dev, reinterpret_cast<uint16_t *>(shadow_bf16))
    LAZY_DISPATCH(tpr, vw, CALL);
#undef CALL
    RP_LAUNCH_CHECK("lazy_adam_catchup");
    return RP_OK;
}

extern "C" int rp_lazy_adam_flush_deferred(int64_t rows, int D, float *p, float *g, float *m, float *v, int32_t *last,
                                           const float *step_scalars, int64_t t_target, double beta1, double beta2,
                                           double eps, const float *cf_table, int64_t cf_from, void *shadow_bf16,
                                           rp_stream_t stream) {
    RP_REQUIRE(p && m && v && last && step_scalars, "lazy_adam_flush_deferred: null pointer");
    RP_REQUIRE(shadow_bf16 == nullptr || (reinterpret_cast<uintptr_t>(shadow_bf16) & 15u) == 0,
               "lazy_adam_flush_deferred: unaligned bf16 copy");
    RP_REQUIRE(!cf_table || (cf_from >= 1 && eps > 0.0 && (((uintptr_t)cf_table) & 31u) == 0),
               "lazy_adam_flush_deferred: the closed-form replay needs cf_from >= 1, eps > 0 and a 32-byte aligned table");
    RP_REQUIRE(D >= 1 && t_target < INT32_MAX - 1, "lazy_adam_flush_deferred: bad D / step");
    if (rows == 0 || t_target <= 0) return RP_OK;
    const int vw = (D % 4 == 0 && rp_aligned16(p) && rp_aligned16(m) && rp_aligned16(v) && (!g || rp_aligned16(g))) ? 4 : 1;
    LazyCfg c{(float)(1.0 - beta1), (float)beta2, (float)std::sqrt(beta2), (float)(1.0 - beta2), (float)eps};
    const int tpr = lazy_tpr(D, vw);
    int64_t nb = rp_cdiv(rows, 256 / tpr);
    if (nb > 65536) nb = 65536;
    hipStream_t s = (hipStream_t)stream;
    const float2 *sc = reinterpret_cast<const float2 *>(step_scalars);
    const CfEntry *cf = reinterpret_cast<const CfEntry *>(cf_table);
#define CALL(T, TY)                                                                                                       \
    hipLaunchKernelGGL((lazy_adam_flush_deferred_kernel<T, TY>), dim3((unsigned)nb), dim3(256), 0, s, rows, D, p, g, m, v, \
                       last, sc, (int)t_target, c, cf, (int)cf_from, reinterpret_cast<uint16_t *>(shadow_bf16))
    LAZY_DISPATCH(tpr, vw, CALL);
#undef CALL
    RP_LAUNCH_CHECK("lazy_adam_flush_deferred");
    return RP_OK;
}

// shadow[row, :] = bf16(p[row, :]) for the run heads of a sorted key list (the bf16 lookup copy after an update that did
// not go through the deferred kernels: the first step after the optimizer state came into being)
__global__ __launch_bounds__(256) void rows_to_bf16_kernel(const int32_t *__restrict__ sk, int64_t n, int D,
                                                           const float *__restrict__ P, uint16_t *__restrict__ shadow) {
    const int64_t i = (int64_t)blockIdx.x * 16 + threadIdx.x / 16;
    if (i >= n) return;
    const int32_t row = sk[i];
    if (i > 0 && sk[i - 1] == row) return;
    for (int c = threadIdx.x % 16; c < D; c += 16) rp_store_bf16(shadow + (int64_t)row * D + c, P[(int64_t)row * D + c]);
}

extern "C" int rp_rows_to_bf16(const int32_t *sorted_keys, int64_t n, int D, const float *p, void *shadow_bf16,
                               rp_stream_t stream) {
    RP_REQUIRE(sorted_keys && p && shadow_bf16 && D >= 1, "rows_to_bf16: bad argument");
    if (n == 0) return RP_OK;
    hipLaunchKernelGGL(rows_to_bf16_kernel, dim3((unsigned)rp_cdiv(n, 16)), dim3(256), 0, (hipStream_t)stream, sorted_keys, n, D, p,
                       reinterpret_cast<uint16_t *>(shadow_bf16));
    RP_LAUNCH_CHECK("rows_to_bf16");
    return RP_OK;
}

// *counter += delta on the stream (the device-resident step counters of the hipGraph path)
__global__ void counter_add_kernel(int32_t *c, int32_t delta) { *c += delta; }

// the same for up to eight counters in one launch (a captured step advances the table clock and the dense clock together)
struct CounterList {
    int32_t *c[8];
};
__global__ void counters_add_kernel(CounterList l, int n, int32_t delta) {
    if ((int)threadIdx.x < n) *l.c[threadIdx.x] += delta;
}

extern "C" int rp_counters_add(int32_t *const *counters, int n, int32_t delta, rp_stream_t stream) {
    RP_REQUIRE(counters && n >= 1 && n <= 8, "counters_add: 1..8 counters");
    CounterList l;
    for (int i = 0; i < 8; ++i) l.c[i] = i < n ? counters[i] : nullptr;
    for (int i = 0; i < n; ++i) {
        RP_REQUIRE(l.c[i] != nullptr, "counters_add: null pointer");
        for (int j = 0; j < i; ++j) RP_REQUIRE(l.c[i] != l.c[j], "counters_add: the same counter twice");
    }
    hipLaunchKernelGGL(counters_add_kernel, dim3(1), dim3(8), 0, (hipStream_t)stream, l, n, delta);
    RP_LAUNCH_CHECK("counters_add");
    return RP_OK;
}

extern "C" int rp_counter_add(int32_t *counter, int32_t delta, rp_stream_t stream) {
    RP_REQUIRE(counter, "counter_add: null pointer");
    hipLaunchKernelGGL(counter_add_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, counter, delta);
    RP_LAUNCH_CHECK("counter_add");
    return RP_OK;
}
