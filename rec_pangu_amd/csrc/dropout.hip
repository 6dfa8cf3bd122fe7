// Dropout for the MLP / tower chains (reference: nn.Dropout in layers/deep.py:66-68 and the multi-task towers,
// mmoe.py:55): y = x * keep / (1 - p), keep ~ Bernoulli(1 - p) per element; the keep mask is saved (one byte per
// element) and the backward is dx = dy * keep / (1 - p).
//
// Random bits: Philox4x32-10, counter = (group of 4 consecutive elements, call offset), key = seed — stateless, so the
// mask of a call is a pure function of (seed, offset, element index): the same mask for any grid shape, reproducible
// from torch.manual_seed through the seed / offset the host passes (rec_pangu_amd/hip.py takes them from torch's
// device generator and advances it).  The stream of values is NOT torch's own dropout stream (that one depends on
// torch's launch geometry); train-mode parity with the reference is therefore statistical, as it is for the
// reference itself between CPU and GPU.  HBM-bound: 4 B read + 5 B written per element.
#include "common.h"

__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                              uint32_t (&out)[4]) {
    constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(M0, c0), lo0 = M0 * c0;
        const uint32_t hi1 = __umulhi(M1, c2), lo1 = M1 * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0;
        c1 = n1;
        c2 = n2;
        c3 = n3;
        k0 += W0;
        k1 += W1;
    }
    out[0] = c0;
    out[1] = c1;
    out[2] = c2;
    out[3] = c3;
}

// element e = m * N + n (logical index, independent of the leading dimensions); group = e / 4
template <bool VEC>
__global__ __launch_bounds__(256) void dropout_fwd_kernel(const float *__restrict__ x, int64_t ldx, float *__restrict__ y,
                                                          int64_t ldy, uint8_t *__restrict__ mask, int64_t M, int N,
                                                          uint32_t thresh, float scale, uint64_t seed, uint64_t offset,
                                                          const uint64_t *__restrict__ offset_dev = nullptr) {
    if (offset_dev != nullptr) offset += *offset_dev;  // a captured step: the generator offset at the step's start, on the device
    const int64_t total = M * (int64_t)N;
    const int64_t ngroups = (total + 3) / 4;
    for (int64_t grp = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; grp < ngroups;
         grp += (int64_t)gridDim.x * blockDim.x) {
        uint32_t r[4];
        philox4x32_10((uint32_t)grp, (uint32_t)(grp >> 32), (uint32_t)offset, (uint32_t)(offset >> 32), (uint32_t)seed,
                      (uint32_t)(seed >> 32), r);
        const int64_t e0 = grp * 4;
        if (VEC) {  // N % 4 == 0, 16-byte aligned rows: the four elements sit in one row
            const int64_t m = e0 / N;
            const int n = (int)(e0 - m * N);
            const f32x4 v = *reinterpret_cast<const f32x4 *>(x + m * ldx + n);
            f32x4 o;
            uint32_t mk = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bool keep = r[i] >= thresh;
                o[i] = keep ? v[i] * scale : 0.f;
                mk |= (keep ? 1u : 0u) << (8 * i);
            }
            *reinterpret_cast<f32x4 *>(y + m * ldy + n) = o;
            *reinterpret_cast<uint32_t *>(mask + e0) = mk;
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int64_t e = e0 + i;
                if (e >= total) break;
                const int64_t m = e / N;
                const int n = (int)(e - m * N);
                const bool keep = r[i] >= thresh;
                y[m * ldy + n] = keep ? x[m * ldx + n] * scale : 0.f;
                mask[e] = keep ? 1 : 0;
            }
        }
    }
}

template <bool VEC>
__global__ __launch_bounds__(256) void dropout_bwd_kernel(const float *__restrict__ dy, int64_t lddy,
                                                          const uint8_t *__restrict__ mask, float *__restrict__ dx,
                                                          int64_t lddx, int64_t M, int N, float scale) {
    const int64_t total = M * (int64_t)N;
    const int64_t ngroups = (total + 3) / 4;
    for (int64_t grp = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; grp < ngroups;
         grp += (int64_t)gridDim.x * blockDim.x) {
        const int64_t e0 = grp * 4;
        if (VEC) {
            const int64_t m = e0 / N;
            const int n = (int)(e0 - m * N);
            const f32x4 g = *reinterpret_cast<const f32x4 *>(dy + m * lddy + n);
            const uint32_t mk = *reinterpret_cast<const uint32_t *>(mask + e0);
            f32x4 o;
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = ((mk >> (8 * i)) & 1u) ? g[i] * scale : 0.f;
            *reinterpret_cast<f32x4 *>(dx + m * lddx + n) = o;
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int64_t e = e0 + i;
                if (e >= total) break;
                const int64_t m = e / N;
                const int n = (int)(e - m * N);
                dx[m * lddx + n] = mask[e] ? dy[m * lddy + n] * scale : 0.f;
            }
        }
    }
}

static unsigned drop_grid(int64_t M, int N) {
    int64_t b = rp_cdiv(rp_cdiv(M * (int64_t)N, 4), 256);
    if (b > 8192) b = 8192;
    if (b < 1) b = 1;
    return (unsigned)b;
}

static int dropout_fwd_impl(const float *x, int64_t ldx, float *y, int64_t ldy, uint8_t *mask, int64_t M, int N, float p,
                            uint64_t seed, uint64_t offset, const uint64_t *offset_dev, rp_stream_t stream) {
    RP_REQUIRE(x && y && mask && M >= 0 && N >= 1 && ldx >= N && ldy >= N, "dropout_fwd: bad argument");
    RP_REQUIRE(p >= 0.f && p < 1.f, "dropout_fwd: p = %f outside [0, 1)", (double)p);
    if (M == 0) return RP_OK;
    // keep <=> r >= thresh with r uniform on [0, 2^32): P(drop) = thresh / 2^32 = p
    double t = (double)p * 4294967296.0;
    if (t > 4294967295.0) t = 4294967295.0;
    const uint32_t thresh = (uint32_t)t;
    const float scale = 1.f / (1.f - p);
    const bool vec = (N % 4 == 0) && (ldx % 4 == 0) && (ldy % 4 == 0) && rp_aligned16(x) && rp_aligned16(y) &&
                     ((reinterpret_cast<uintptr_t>(mask) & 3u) == 0);
    hipStream_t s = (hipStream_t)stream;
    if (vec)
        hipLaunchKernelGGL((dropout_fwd_kernel<true>), dim3(drop_grid(M, N)), dim3(256), 0, s, x, ldx, y, ldy, mask, M, N,
                           thresh, scale, seed, offset, offset_dev);
    else
        hipLaunchKernelGGL((dropout_fwd_kernel<false>), dim3(drop_grid(M, N)), dim3(256), 0, s, x, ldx, y, ldy, mask, M, N,
                           thresh, scale, seed, offset, offset_dev);
    RP_LAUNCH_CHECK("dropout_fwd");
    return RP_OK;
}

extern "C" int rp_dropout_fwd(const float *x, int64_t ldx, float *y, int64_t ldy, uint8_t *mask, int64_t M, int N, float p,
                              uint64_t seed, uint64_t offset, rp_stream_t stream) {
    return dropout_fwd_impl(x, ldx, y, ldy, mask, M, N, p, seed, offset, nullptr, stream);
}

// offset = *offset_dev + offset_delta, read on the device: the form a CAPTURED step records (its launch arguments are
// frozen; the generator offset of the step's start lives in device memory and is advanced by rp_counter_add_u64 at the
// step's end, so every replay draws the masks the eager loop would draw at that point of the generator's stream)
extern "C" int rp_dropout_fwd_dev(const float *x, int64_t ldx, float *y, int64_t ldy, uint8_t *mask, int64_t M, int N, float p,
                                  uint64_t seed, uint64_t offset_delta, const uint64_t *offset_dev, rp_stream_t stream) {
    RP_REQUIRE(offset_dev != nullptr, "dropout_fwd_dev: null offset counter");
    return dropout_fwd_impl(x, ldx, y, ldy, mask, M, N, p, seed, offset_delta, offset_dev, stream);
}

__global__ void counter_add_u64_kernel(uint64_t *c, uint64_t delta) { *c += delta; }

extern "C" int rp_counter_add_u64(uint64_t *counter, uint64_t delta, rp_stream_t stream) {
    RP_REQUIRE(counter, "counter_add_u64: null pointer");
    hipLaunchKernelGGL(counter_add_u64_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, counter, delta);
    RP_LAUNCH_CHECK("counter_add_u64");
    return RP_OK;
}

extern "C" int rp_dropout_bwd(const float *dy, int64_t lddy, const uint8_t *mask, float *dx, int64_t lddx, int64_t M, int N,
                              float p, rp_stream_t stream) {
    RP_REQUIRE(dy && mask && dx && M >= 0 && N >= 1 && lddy >= N && lddx >= N, "dropout_bwd: bad argument");
    RP_REQUIRE(p >= 0.f && p < 1.f, "dropout_bwd: p = %f outside [0, 1)", (double)p);
    if (M == 0) return RP_OK;
    const float scale = 1.f / (1.f - p);
    const bool vec = (N % 4 == 0) && (lddy % 4 == 0) && (lddx % 4 == 0) && rp_aligned16(dy) && rp_aligned16(dx) &&
                     ((reinterpret_cast<uintptr_t>(mask) & 3u) == 0);
    hipStream_t s = (hipStream_t)stream;
    if (vec)
        hipLaunchKernelGGL((dropout_bwd_kernel<true>), dim3(drop_grid(M, N)), dim3(256), 0, s, dy, lddy, mask, dx, lddx, M, N,
                           scale);
    else
        hipLaunchKernelGGL((dropout_bwd_kernel<false>), dim3(drop_grid(M, N)), dim3(256), 0, s, dy, lddy, mask, dx, lddx, M,
                           N, scale);
    RP_LAUNCH_CHECK("dropout_bwd");
    return RP_OK;
}
