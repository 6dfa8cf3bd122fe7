// Shared helpers for the gfx950 kernels behind include/rec_pangu_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/rec_pangu_hip.h"

#include <string.h>
#include <tuple>
#include <utility>

// thread-local error text + process-wide launch counter (defined in common.hip)
int rp_fail(int code, const char *fmt, ...);
void rp_count_launch();

// ------------------------------------------------------------------------------------------------------------------
// Every kernel launch of the library goes through rp_launch (the hipLaunchKernelGGL macro is redirected to it below):
// the arguments are packed into one buffer the way the kernel takes them and handed to hipLaunchKernel; while a LAUNCH
// PLAN is being recorded (plan.hip: rp_plan_begin .. rp_plan_end) the packed launch is also appended to the plan, which
// rp_plan_replay later re-issues with the same arguments — the library's own, node-overhead-free form of a captured step
// (rec_pangu_amd/graph_step.py).  gfx950 kernel arguments are at most 4 KB.
// ------------------------------------------------------------------------------------------------------------------
#define RP_MAX_KERNARG 4096
bool rp_plan_recording();
void rp_plan_record(const void *func, dim3 grid, dim3 block, unsigned shmem, hipStream_t stream, const char *blob,
                    size_t blob_bytes, const size_t *offsets, int nargs);

template <typename T>
static inline size_t rp_pack_arg(char *buf, size_t &off, const T &v) {
    off = (off + alignof(T) - 1) & ~(size_t)(alignof(T) - 1);
    const size_t at = off;
    memcpy(buf + at, &v, sizeof(T));
    off += sizeof(T);
    return at;
}

// argument I of the launch converted to the kernel's parameter type K, exactly as a <<< >>> launch would; a kernel
// parameter the launch site leaves out takes K{} — every default argument of this library's kernels is 0 / nullptr
// (embed_grad_reduce_kernel, linear_fwd_bf16_kernel, linear_wgrad_bf16_kernel), and a function pointer carries no defaults
template <size_t I, typename K, typename Tup>
static inline K rp_arg_or_zero(Tup &t) {
    if constexpr (I < std::tuple_size<Tup>::value) return static_cast<K>(std::get<I>(t));
    else return K{};
}

template <typename... KArgs, typename Tup, size_t... I>
static inline void rp_launch_packed(void (*kernel)(KArgs...), dim3 grid, dim3 block, unsigned shmem, hipStream_t stream,
                                    Tup &args, std::index_sequence<I...>) {
    constexpr int N = (int)sizeof...(KArgs);
    static_assert((sizeof(KArgs) + ... + 0) + 16 * sizeof...(KArgs) <= RP_MAX_KERNARG, "kernel arguments exceed the packing buffer");
    alignas(16) char buf[RP_MAX_KERNARG];
    size_t off = 0;
    size_t offs[N ? N : 1] = {rp_pack_arg<KArgs>(buf, off, rp_arg_or_zero<I, KArgs>(args))...};
    void *ptrs[N ? N : 1];
    for (int i = 0; i < N; ++i) ptrs[i] = buf + offs[i];
    if (rp_plan_recording()) rp_plan_record(reinterpret_cast<const void *>(kernel), grid, block, shmem, stream, buf, off, offs, N);
    (void)hipLaunchKernel(reinterpret_cast<const void *>(kernel), grid, block, ptrs, shmem, stream);
}

template <typename... KArgs, typename... Args>
static inline void rp_launch(void (*kernel)(KArgs...), dim3 grid, dim3 block, unsigned shmem, hipStream_t stream,
                             Args &&...args) {
    static_assert(sizeof...(Args) <= sizeof...(KArgs), "kernel launch: too many arguments");
    auto tup = std::forward_as_tuple(std::forward<Args>(args)...);
    rp_launch_packed(kernel, grid, block, shmem, stream, tup, std::make_index_sequence<sizeof...(KArgs)>());
}

#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) rp_launch(kernel, grid, block, shmem, stream, __VA_ARGS__)

#define RP_REQUIRE(cond, ...)                                  \
    do {                                                       \
        if (!(cond)) return rp_fail(RP_ERR_ARG, __VA_ARGS__);  \
    } while (0)

#define RP_LAUNCH_CHECK(name)                                                        \
    do {                                                                             \
        hipError_t e__ = hipGetLastError();                                          \
        if (e__ != hipSuccess)                                                       \
            return rp_fail(RP_ERR_LAUNCH, "%s: %s", name, hipGetErrorString(e__));   \
        rp_count_launch();                                                           \
    } while (0)

static inline bool rp_aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static inline int64_t rp_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// activation of a GEMM epilogue (RP_ACT_MASK is handled at the call site: it needs aux)
__device__ __forceinline__ float rp_act_apply(int act, float v) {
    switch (act) {
        case RP_ACT_RELU: return v > 0.f ? v : 0.f;
        case RP_ACT_TANH: return tanhf(v);
        case RP_ACT_SIGMOID: return 1.f / (1.f + expf(-v));
        case RP_ACT_LEAKY: return v > 0.f ? v : 0.01f * v;
        default: return v;
    }
}
// dy * act'(pre) through the activation's output y
__device__ __forceinline__ float rp_act_grad(int act, float dy, float y) {
    switch (act) {
        case RP_ACT_RELU: return y > 0.f ? dy : 0.f;
        case RP_ACT_TANH: return dy * (1.f - y * y);
        case RP_ACT_SIGMOID: return dy * (y * (1.f - y));
        case RP_ACT_LEAKY: return y > 0.f ? dy : 0.01f * dy;
        default: return dy;
    }
}

// wave64 constants: hard-coded, gfx950 only
#define RP_WAVE 64

// Matrix-core products per flop for a GEMM-shaped launch under the process-wide precision mode (gemm.hip):
// 6 / 3 / 1 bf16 products, or 0 for the f32-input MFMA.  RP_MATMUL_AUTO picks 3 (~2^-16 per product) when the launch is
// matrix-core bound even at 3 products (flops x 3 / algorithmic bytes above the MFMA/HBM ridge) and 6 (fp32-faithful)
// otherwise — HBM-bound launches get the exact products for free.
int rp_matmul_products(double flops, double bytes);

// embed.hip: second pass over a (head, tail) piece list of `nb0` tiles + the sequential walk (rp_embed_grad_reduce's finish)
int rp_int_grad_reduce_finish(int64_t n, int D, int64_t nb0, char *wbase, float *piece0, int32_t *key0, float *grad_arena,
                              int accumulate, hipStream_t s);
