// Shared helpers for the gfx950 kernels behind include/rec_pangu_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/rec_pangu_hip.h"

// thread-local error text + process-wide launch counter (defined in common.hip)
int rp_fail(int code, const char *fmt, ...);
void rp_count_launch();

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

// wave64 constants: hard-coded, gfx950 only
#define RP_WAVE 64

// Matrix-core products per flop for a GEMM-shaped launch under the process-wide precision mode (gemm.hip):
// 6 / 3 / 1 bf16 products, or 0 for the f32-input MFMA.  RP_MATMUL_AUTO picks 3 (~2^-16 per product) when the launch is
// matrix-core bound even at 3 products (flops x 3 / algorithmic bytes above the MFMA/HBM ridge) and 6 (fp32-faithful)
// otherwise — HBM-bound launches get the exact products for free.
int rp_matmul_products(double flops, double bytes);
