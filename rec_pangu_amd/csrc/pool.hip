// Multi-id ("bag") embedding lookups with pooling, gfx950 — north_star's "fused CSR/segmented embedding gather +
// sum-pool".  Reference pieces: the `_seq` lookup of EmbeddingLayer.forward (rec_pangu/models/layers/embedding.py:64-71:
// X[name] is [B, L] int64, the result [B, L, D]) followed by MaskedSumPooling (layers/sequence.py:38-59:
// torch.sum(dim=1)) or MaskedAveragePooling (layers/sequence.py:13-36: sum(dim=1) / ((E != 0).sum(dim=1) + 1e-16) —
// the count is PER ELEMENT (b, d), not per row).  Fused here: the [B, L, D] intermediate (L x the output) never exists.
//
// Bags come either dense (ids [B, L], L entries per bag — the reference's input format, padding ids included: the
// reference sums the padding row like any other, "mask by zeros" only holds when that row is zero) or CSR
// (offsets [B+1] into a flat ids [nnz]: ragged bags without padding).
//
// Work decomposition (wave64): TPR lanes x VEC floats own one bag; the bag's rows are summed in id order (fixed order:
// run-to-run identical), four row loads in flight per lane.  HBM-bound: nnz * (D*4 + 8) bytes read, B * D * 4 written.
#include "common.h"

namespace {

template <int VEC>
struct PV;
template <>
struct PV<4> {
    typedef f32x4 T;
    static __device__ __forceinline__ T load(const float *p) { return *reinterpret_cast<const f32x4 *>(p); }
    static __device__ __forceinline__ void store(float *p, T v) { *reinterpret_cast<f32x4 *>(p) = v; }
    static __device__ __forceinline__ T zero() { return f32x4{0.f, 0.f, 0.f, 0.f}; }
    static __device__ __forceinline__ T nz(T v) {
        return f32x4{v.x != 0.f ? 1.f : 0.f, v.y != 0.f ? 1.f : 0.f, v.z != 0.f ? 1.f : 0.f, v.w != 0.f ? 1.f : 0.f};
    }
    static __device__ __forceinline__ T splat(float x) { return f32x4{x, x, x, x}; }
};
template <>
struct PV<1> {
    typedef float T;
    static __device__ __forceinline__ T load(const float *p) { return *p; }
    static __device__ __forceinline__ void store(float *p, T v) { *p = v; }
    static __device__ __forceinline__ T zero() { return 0.f; }
    static __device__ __forceinline__ T nz(T v) { return v != 0.f ? 1.f : 0.f; }
    static __device__ __forceinline__ T splat(float x) { return x; }
};

#define RP_POOL_INFLIGHT 4

template <int TPR, int VEC, bool AVG>
__global__ __launch_bounds__(256) void embed_gather_pool_kernel(
    const float *__restrict__ arena, int64_t row_base, int64_t row_count, const int64_t *__restrict__ ids,
    const int64_t *__restrict__ offsets, int64_t L, int64_t B, int D, float *__restrict__ out, int64_t ldo,
    float *__restrict__ inv_out, int32_t *__restrict__ bag_out, int32_t *__restrict__ err_flag) {
    typedef PV<VEC> V;
    constexpr int BPB = 256 / TPR;
    const int t = threadIdx.x % TPR;
    const int64_t b = (int64_t)blockIdx.x * BPB + threadIdx.x / TPR;
    if (b >= B) return;
    const int64_t beg = offsets ? offsets[b] : b * L;
    const int64_t end = offsets ? offsets[b + 1] : beg + L;
    for (int c = t * VEC; c < D; c += TPR * VEC) {
        typename V::T acc = V::zero(), cnt = V::zero();
        for (int64_t j0 = beg; j0 < end; j0 += RP_POOL_INFLIGHT) {
            typename V::T r[RP_POOL_INFLIGHT];
#pragma unroll
            for (int u = 0; u < RP_POOL_INFLIGHT; ++u) {
                const int64_t j = j0 + u;
                r[u] = V::zero();
                if (j < end) {
                    int64_t id = ids[j];
                    if (id < 0 || id >= row_count) {  // reference: IndexError from nn.Embedding on CPU
                        *err_flag = 1;
                        id = 0;
                    }
                    r[u] = V::load(arena + (row_base + id) * D + c);
                    if (bag_out != nullptr && c == 0) bag_out[j] = (int32_t)b;  // t == 0 only
                }
            }
#pragma unroll
            for (int u = 0; u < RP_POOL_INFLIGHT; ++u) {  // (beyond the bag: zero rows, nothing counted)
                acc += r[u];
                if (AVG) cnt += V::nz(r[u]);
            }
        }
        if (AVG) {
            const typename V::T den = cnt + V::splat(1e-16f);  // sequence.py:34
            acc = acc / den;
            if (inv_out != nullptr) V::store(inv_out + b * D + c, V::splat(1.f) / den);
        }
        V::store(out + b * ldo + c, acc);
    }
}

// the poolings on an EXPLICIT [B, L, D] tensor (the drop-in MaskedSumPooling / MaskedAveragePooling modules)
template <int TPR, int VEC, bool AVG>
__global__ __launch_bounds__(256) void seq_pool_fwd_kernel(const float *__restrict__ e, int64_t B, int64_t L, int D,
                                                           float *__restrict__ out, float *__restrict__ inv_out) {
    typedef PV<VEC> V;
    constexpr int BPB = 256 / TPR;
    const int t = threadIdx.x % TPR;
    const int64_t b = (int64_t)blockIdx.x * BPB + threadIdx.x / TPR;
    if (b >= B) return;
    const float *eb = e + b * L * D;
    for (int c = t * VEC; c < D; c += TPR * VEC) {
        typename V::T acc = V::zero(), cnt = V::zero();
#pragma unroll 4
        for (int64_t l = 0; l < L; ++l) {
            const typename V::T r = V::load(eb + l * D + c);
            acc += r;
            if (AVG) cnt += V::nz(r);
        }
        if (AVG) {
            const typename V::T den = cnt + V::splat(1e-16f);
            acc = acc / den;
            if (inv_out != nullptr) V::store(inv_out + b * D + c, V::splat(1.f) / den);
        }
        V::store(out + b * D + c, acc);
    }
}

// de[b, l, :] = g[b, :] (* inv[b, :])
template <int TPR, int VEC>
__global__ __launch_bounds__(256) void seq_pool_bwd_kernel(const float *__restrict__ g, const float *__restrict__ inv,
                                                           int64_t B, int64_t L, int D, float *__restrict__ de) {
    typedef PV<VEC> V;
    constexpr int BPB = 256 / TPR;
    const int t = threadIdx.x % TPR;
    const int64_t b = (int64_t)blockIdx.x * BPB + threadIdx.x / TPR;
    if (b >= B) return;
    for (int c = t * VEC; c < D; c += TPR * VEC) {
        typename V::T v = V::load(g + b * D + c);
        if (inv != nullptr) v = v * V::load(inv + b * D + c);
        for (int64_t l = 0; l < L; ++l) V::store(de + (b * L + l) * D + c, v);
    }
}

int pool_tpr(int D, int vec) {
    int need = (D + vec - 1) / vec, tpr = 1;
    while (tpr < need && tpr < 64) tpr <<= 1;
    return tpr;
}

}  // namespace

#define RP_POOL_DISPATCH(tpr, vec, CALL)                      \
    do {                                                      \
        if (vec == 4) {                                       \
            switch (tpr) {                                    \
                case 1: { CALL(1, 4); } break;                \
                case 2: { CALL(2, 4); } break;                \
                case 4: { CALL(4, 4); } break;                \
                case 8: { CALL(8, 4); } break;                \
                case 16: { CALL(16, 4); } break;              \
                case 32: { CALL(32, 4); } break;              \
                default: { CALL(64, 4); } break;              \
            }                                                 \
        } else {                                              \
            switch (tpr) {                                    \
                case 1: { CALL(1, 1); } break;                \
                case 2: { CALL(2, 1); } break;                \
                case 4: { CALL(4, 1); } break;                \
                case 8: { CALL(8, 1); } break;                \
                case 16: { CALL(16, 1); } break;              \
                case 32: { CALL(32, 1); } break;              \
                default: { CALL(64, 1); } break;              \
            }                                                 \
        }                                                     \
    } while (0)

extern "C" int rp_embed_gather_pool_fwd(const float *arena, int64_t row_base, int64_t row_count, const int64_t *ids,
                                        const int64_t *offsets, int64_t L, int64_t B, int D, int mode, float *out,
                                        int64_t ldo, float *inv_out, int32_t *bag_out, int32_t *err_flag,
                                        rp_stream_t stream) {
    RP_REQUIRE(arena && ids && out && err_flag, "embed_gather_pool_fwd: null pointer");
    RP_REQUIRE(row_base >= 0 && row_count >= 1 && D >= 1 && B >= 0, "embed_gather_pool_fwd: bad table / D / B");
    RP_REQUIRE(offsets != nullptr || L >= 0, "embed_gather_pool_fwd: dense bags need L >= 0");
    RP_REQUIRE(mode == 0 || mode == 1, "embed_gather_pool_fwd: mode must be 0 (sum) or 1 (masked average)");
    RP_REQUIRE(ldo >= D, "embed_gather_pool_fwd: ldo=%lld < D", (long long)ldo);
    if (B == 0) return RP_OK;
    const bool v4 = (D % 4 == 0) && (ldo % 4 == 0) && rp_aligned16(arena) && rp_aligned16(out) &&
                    (inv_out == nullptr || rp_aligned16(inv_out));
    const int vec = v4 ? 4 : 1;
    const int tpr = pool_tpr(D, vec);
    const unsigned grid = (unsigned)rp_cdiv(B, 256 / tpr);
    hipStream_t s = (hipStream_t)stream;
#define CALL(T, VV)                                                                                                      \
    do {                                                                                                                 \
        if (mode == 1)                                                                                                   \
            hipLaunchKernelGGL((embed_gather_pool_kernel<T, VV, true>), dim3(grid), dim3(256), 0, s, arena, row_base,     \
                               row_count, ids, offsets, L, B, D, out, ldo, inv_out, bag_out, err_flag);                   \
        else                                                                                                             \
            hipLaunchKernelGGL((embed_gather_pool_kernel<T, VV, false>), dim3(grid), dim3(256), 0, s, arena, row_base,    \
                               row_count, ids, offsets, L, B, D, out, ldo, inv_out, bag_out, err_flag);                   \
    } while (0)
    RP_POOL_DISPATCH(tpr, vec, CALL);
#undef CALL
    RP_LAUNCH_CHECK("embed_gather_pool_fwd");
    return RP_OK;
}

extern "C" int rp_seq_pool_fwd(const float *e, int64_t B, int64_t L, int D, int mode, float *out, float *inv_out,
                               rp_stream_t stream) {
    RP_REQUIRE(e && out, "seq_pool_fwd: null pointer");
    RP_REQUIRE(B >= 0 && L >= 0 && D >= 1 && (mode == 0 || mode == 1), "seq_pool_fwd: bad argument");
    if (B == 0) return RP_OK;
    const int vec = (D % 4 == 0 && rp_aligned16(e) && rp_aligned16(out) && (!inv_out || rp_aligned16(inv_out))) ? 4 : 1;
    const int tpr = pool_tpr(D, vec);
    const unsigned grid = (unsigned)rp_cdiv(B, 256 / tpr);
    hipStream_t s = (hipStream_t)stream;
#define CALL(T, VV)                                                                                                        \
    do {                                                                                                                   \
        if (mode == 1)                                                                                                     \
            hipLaunchKernelGGL((seq_pool_fwd_kernel<T, VV, true>), dim3(grid), dim3(256), 0, s, e, B, L, D, out, inv_out);  \
        else                                                                                                               \
            hipLaunchKernelGGL((seq_pool_fwd_kernel<T, VV, false>), dim3(grid), dim3(256), 0, s, e, B, L, D, out, inv_out); \
    } while (0)
    RP_POOL_DISPATCH(tpr, vec, CALL);
#undef CALL
    RP_LAUNCH_CHECK("seq_pool_fwd");
    return RP_OK;
}

extern "C" int rp_seq_pool_bwd(const float *g, const float *inv, int64_t B, int64_t L, int D, float *de,
                               rp_stream_t stream) {
    RP_REQUIRE(g && de, "seq_pool_bwd: null pointer");
    RP_REQUIRE(B >= 0 && L >= 0 && D >= 1, "seq_pool_bwd: bad argument");
    if (B == 0 || L == 0) return RP_OK;
    const int vec = (D % 4 == 0 && rp_aligned16(g) && rp_aligned16(de) && (!inv || rp_aligned16(inv))) ? 4 : 1;
    const int tpr = pool_tpr(D, vec);
    const unsigned grid = (unsigned)rp_cdiv(B, 256 / tpr);
    hipStream_t s = (hipStream_t)stream;
#define CALL(T, VV) hipLaunchKernelGGL((seq_pool_bwd_kernel<T, VV>), dim3(grid), dim3(256), 0, s, g, inv, B, L, D, de)
    RP_POOL_DISPATCH(tpr, vec, CALL);
#undef CALL
    RP_LAUNCH_CHECK("seq_pool_bwd");
    return RP_OK;
}
