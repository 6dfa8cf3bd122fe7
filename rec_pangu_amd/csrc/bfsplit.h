// Split-bf16 operand helpers shared by the matrix-core kernels (gemm.hip, embed.hip): an fp32 value is split into bf16
// pieces x = hi + mid + lo (the subtractions are exact, the conversions are v_cvt_pk_bf16_f32) and a product is a sum of
// bf16 x bf16 MFMA products accumulated in fp32 — see the header comment in gemm.hip.
#pragma once
#include "common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x8 __attribute__((ext_vector_type(8)));

#define BF_LD 40

template <int NPROD>
struct BfProd {
    static constexpr int NP = NPROD == 6 ? 3 : (NPROD == 3 ? 2 : 1);
    // smallest terms first
    __device__ static constexpr int pa(int i) { return NPROD == 6 ? (i == 0 ? 0 : i == 1 ? 2 : i == 2 ? 1 : i == 3 ? 0 : i == 4 ? 1 : 0)
                                                     : NPROD == 3 ? (i == 0 ? 0 : i == 1 ? 1 : 0) : 0; }
    __device__ static constexpr int pb(int i) { return NPROD == 6 ? (i == 0 ? 2 : i == 1 ? 0 : i == 2 ? 1 : i == 3 ? 1 : i == 4 ? 0 : 0)
                                                     : NPROD == 3 ? (i == 0 ? 1 : i == 1 ? 0 : 0) : 0; }
};

template <int NP>
__device__ __forceinline__ void bf_split4(f32x4 v, bf16x4 (&p)[NP]) {
    p[0] = __builtin_convertvector(v, bf16x4);
    if (NP > 1) {
        v -= __builtin_convertvector(p[0], f32x4);
        p[1] = __builtin_convertvector(v, bf16x4);
    }
    if (NP > 2) {
        v -= __builtin_convertvector(p[1], f32x4);
        p[2] = __builtin_convertvector(v, bf16x4);
    }
}

template <int NP>
__device__ __forceinline__ void bf_split8(f32x8 v, bf16x8 (&p)[NP]) {
    p[0] = __builtin_convertvector(v, bf16x8);
    if (NP > 1) {
        v -= __builtin_convertvector(p[0], f32x8);
        p[1] = __builtin_convertvector(v, bf16x8);
    }
    if (NP > 2) {
        v -= __builtin_convertvector(p[1], f32x8);
        p[2] = __builtin_convertvector(v, bf16x8);
    }
}

