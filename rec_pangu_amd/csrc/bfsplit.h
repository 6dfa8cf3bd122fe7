// Split-bf16 operand helpers shared by the matrix-core kernels (gemm.hip, embed.hip, mlp_tail.hip): an fp32 value is split
// into bf16 pieces x = hi + mid + lo (the subtractions are exact) and a product is a sum of bf16 x bf16 MFMA products
// accumulated in fp32 — see the header comment in gemm.hip.
//   one / two pieces (bf16, bf16x3): round-to-nearest conversions (v_cvt_pk_bf16_f32) — the dropped remainder is
//     unbiased;
//   three pieces (bf16x6, the fp32-faithful mode): TRUNCATION — hi = the top 16 bits of x, mid = the top 16 bits of x - hi,
//     lo likewise: three 8-bit windows cover the whole 24-bit significand, so hi + mid + lo == x exactly and nothing is
//     dropped that rounding would have kept, and a piece costs an AND, a subtraction and half a v_perm_b32 instead of two
//     conversions, a shift and a subtraction (the conversion-based split was 41 % of the weight-gradient kernel's
//     issue cycles, more than its 6-product MFMA work: profiles/microbench/wgrad_one.py).
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

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));

// the high halves of two dwords packed into one: (bf16 of `even` by truncation, bf16 of `odd` by truncation)
__device__ __forceinline__ uint32_t bf_pack_hi(uint32_t even, uint32_t odd) { return __builtin_amdgcn_perm(odd, even, 0x07060302u); }

__device__ __forceinline__ void bf_trunc3_4(f32x4 v, bf16x4 (&p)[3]) {
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const u32x4 u = __builtin_bit_cast(u32x4, v);
        u32x2 w;
        w[0] = bf_pack_hi(u[0], u[1]);
        w[1] = bf_pack_hi(u[2], u[3]);
        p[q] = __builtin_bit_cast(bf16x4, w);
        if (q < 2) v -= __builtin_bit_cast(f32x4, u & 0xFFFF0000u);
    }
}

__device__ __forceinline__ void bf_trunc3_8(f32x8 v, bf16x8 (&p)[3]) {
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const u32x8 u = __builtin_bit_cast(u32x8, v);
        u32x4 w;
#pragma unroll
        for (int j = 0; j < 4; ++j) w[j] = bf_pack_hi(u[2 * j], u[2 * j + 1]);
        p[q] = __builtin_bit_cast(bf16x8, w);
        if (q < 2) v -= __builtin_bit_cast(f32x8, u & 0xFFFF0000u);
    }
}

template <int NP>
__device__ __forceinline__ void bf_split4(f32x4 v, bf16x4 (&p)[NP]) {
    if constexpr (NP == 3) {
        bf_trunc3_4(v, p);
        return;
    }
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
    if constexpr (NP == 3) {
        bf_trunc3_8(v, p);
        return;
    }
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

