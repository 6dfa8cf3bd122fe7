// xDeepFM Compressed Interaction Network layer on the exact-fp32 matrix core, gfx950.
//   reference: layers/interaction.py:157-171
//     X_k[b,o,d] = sum_{h<H} sum_{m<M} W[o, h*M+m] * X_0[b,h,d] * X_{k-1}[b,m,d] + bias[o]      (no activation)
//   The reference materialises the outer product [B, H*M, D] (11 GB / 56 GB fp32 at B=65536) and runs a
//   Conv1d(k=1) over it.  Here nothing is materialised: for a fixed (sample, output channel o) the double sum
//   factors as   X_k[o,d] = sum_h X_0[h,d] * T_o[h,d],   T_o[h,d] = sum_m W[o,h,m] * X_{k-1}[m,d],
//   and T_o is ONE 32x32 MFMA tile chain: rows = field h (H <= 32), columns = 32 embedding lanes d,
//   contraction over m (v_mfma_f32_32x32x2_f32, M/2 instructions).  The epilogue multiplies the accumulator by
//   X_0 in the C layout (row = (r&3)+8(r>>2)+4(lane>>5), col = lane&31), sums the 16 rows a lane holds and
//   finishes the row reduction with one cross-half shuffle.  X_0 and X_{k-1} of S samples are staged in LDS
//   (row stride D+1: the forward reads rows of consecutive d, the backward reads columns — both conflict-free);
//   the A fragments (W[o], H*M floats) are loaded once per output channel into registers and reused for all
//   S samples x D/32 column tiles.  Sum pooling over d rides along.  MFMA-bound: 2*H*M*D flop per (b,o) on a
//   32-row tile = H/32 utilisation (81 % at H=26).
// The last CIN layer never runs at its full width: with no activation and linear sum-pooling + fc after it,
//   sum_o c[o] * sum_d X_L[b,o,d] = sum_d sum_{h,m} V[h,m] X_0[b,h,d] X_{L-1}[b,m,d] + const,  V = sum_o c[o] W_L[o],
// so the host collapses it to a single output channel (O = 1) — same kernel, 1/O_L of the work, exact algebra.
#include "common.h"

#define CIN_MAXH 32
#define CIN_LDS_BUDGET (156 * 1024)

// One MFMA accumulation chain of n2 steps: acc += sum_k A[.,k] B[k,.] with both operands read from LDS at
// pa + k*sa / pb + k*sb (float strides).  Operands are fetched four steps ahead of the MFMAs that consume them and
// the addresses are walked, not recomputed: the first version spent ~15 VALU instructions per MFMA on index
// arithmetic (rocprofv3 SQ_INSTS_VALU / SQ_INSTS_MFMA), which capped the matrix pipe at ~34 %.
__device__ __forceinline__ void cin_chain(f32x16 &acc, const float *pa, int sa, const float *pb, int sb, int n2,
                                          float bscale) {
    int k = 0;
    for (; k + 4 <= n2; k += 4) {
        const float a0 = pa[0], a1 = pa[sa], a2 = pa[2 * sa], a3 = pa[3 * sa];
        const float b0 = pb[0], b1 = pb[sb], b2 = pb[2 * sb], b3 = pb[3 * sb];
        pa += 4 * sa;
        pb += 4 * sb;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bscale * b0, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bscale * b1, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, bscale * b2, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a3, bscale * b3, acc, 0, 0, 0);
    }
    for (; k < n2; ++k) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[0], bscale * pb[0], acc, 0, 0, 0);
        pa += sa;
        pb += sb;
    }
}

// Two column tiles at once: they share the A operand (one LDS read feeds two MFMAs) and give the matrix pipe two
// independent accumulators to alternate between.
__device__ __forceinline__ void cin_chain2(f32x16 &acc0, f32x16 &acc1, const float *pa, int sa, const float *pb,
                                           int sb, int n2) {
    int k = 0;
    for (; k + 4 <= n2; k += 4) {
        const float a0 = pa[0], a1 = pa[sa], a2 = pa[2 * sa], a3 = pa[3 * sa];
        const float b0 = pb[0], b1 = pb[sb], b2 = pb[2 * sb], b3 = pb[3 * sb];
        const float c0 = pb[32], c1 = pb[sb + 32], c2 = pb[2 * sb + 32], c3 = pb[3 * sb + 32];
        pa += 4 * sa;
        pb += 4 * sb;
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, c0, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, c1, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, b2, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, c2, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a3, b3, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a3, c3, acc1, 0, 0, 0);
    }
    for (; k < n2; ++k) {
        const float a = pa[0];
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, pb[0], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, pb[32], acc1, 0, 0, 0);
        pa += sa;
        pb += sb;
    }
}

// W[o] ([H][M] contiguous in global memory) -> registers -> this wave's padded LDS tile [32][Ms].  The global
// loads for channel o+4 are issued before the MFMA work of channel o and only consumed (stored to LDS) after it,
// so their ~1 us L2 latency is hidden; without this the staging was ~2/3 of the kernel time.
#define CIN_WREGS 16  // H*M <= 1024 floats per channel (M <= 32 layers); larger tiles stage synchronously
struct CinWRegs {
    float r[CIN_WREGS];
};
__device__ __forceinline__ void cin_w_load(CinWRegs &w, const float *__restrict__ Wo, int HM, int lane) {
#pragma unroll
    for (int u = 0; u < CIN_WREGS; ++u) {
        const int i = lane + 64 * u;
        w.r[u] = (i < HM) ? Wo[i] : 0.f;
    }
}
struct CinWOff {
    int off[CIN_WREGS];  // LDS offset of element lane + 64u of a [H][M] tile inside the padded [32][Ms] tile, or -1
};
__device__ __forceinline__ void cin_w_offsets(CinWOff &o, int HM, int M, int Ms, int lane) {
#pragma unroll
    for (int u = 0; u < CIN_WREGS; ++u) {
        const int i = lane + 64 * u;
        const int h = i / M;
        o.off[u] = (i < HM) ? h * Ms + (i - h * M) : -1;
    }
}
__device__ __forceinline__ void cin_w_store(const CinWRegs &w, const CinWOff &o, float *Wl) {
#pragma unroll
    for (int u = 0; u < CIN_WREGS; ++u)
        if (o.off[u] >= 0) Wl[o.off[u]] = w.r[u];
}

// LDS plan shared by host and device.  All tiles are zero-padded so the MFMA loop carries no predicates:
//   X0s [S][32][Dpp]       rows >= H and columns >= D are zero            (Dpp = 32*ceil(D/32) + 1, odd)
//   Xps [S][Mr][Dpp]       Mr = M rounded up to even (pad row zero); aliases X0s for the first layer (X_{k-1} = X_0)
//   Wl  [4 waves][32][Ms]  the current output channel's W[o] as [h][m], Ms odd >= M+1 (pad column / rows zero)
//   poolL [S][O]
struct CinPlan {
    int S, Dpp, Mr, Ms, same;
    int offXp, offW, offPool, total;  // float offsets
};

__host__ __device__ static inline CinPlan cin_plan(int H, int M, int O, int D, int S, int same) {
    CinPlan p;
    p.S = S;
    p.same = same;
    p.Dpp = ((D + 31) / 32) * 32 + 1;
    p.Mr = (M + 1) & ~1;
    p.Ms = (M & 1) ? M + 2 : M + 1;
    int o = S * 32 * p.Dpp;
    p.offXp = same ? 0 : o;
    if (!same) o += S * p.Mr * p.Dpp;
    p.offW = o;
    o += 4 * 32 * p.Ms;
    p.offPool = o;
    o += S * O;
    p.total = o;
    return p;
}

__global__ __launch_bounds__(256) void cin_layer_fwd_kernel(const float *__restrict__ x0, int64_t ld0,
                                                            const float *__restrict__ xp, int64_t ldp,
                                                            const float *__restrict__ W, const float *__restrict__ bias,
                                                            float *__restrict__ out, float *__restrict__ pooled,
                                                            int64_t ldpool, int H, int M, int O, int D, int64_t B,
                                                            CinPlan P) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int S = P.S, Dpp = P.Dpp, Mr = P.Mr, Ms = P.Ms;
    float *X0s = smem;
    float *Xps = smem + P.offXp;
    float *poolL = smem + P.offPool;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, half = lane >> 5;
    float *Wl = smem + P.offW + wave * 32 * Ms;
    const int ND = (D + 31) / 32, M2 = Mr / 2;
    const int64_t b0 = (int64_t)blockIdx.x * S;
    for (int i = tid; i < P.offW; i += 256) smem[i] = 0.f;  // zero padding of X0s / Xps
    for (int i = tid; i < 4 * 32 * Ms; i += 256) smem[P.offW + i] = 0.f;
    for (int i = tid; i < S * O; i += 256) poolL[i] = 0.f;
    __syncthreads();
    for (int i = tid; i < S * H * D; i += 256) {
        const int s = i / (H * D), r = i - s * H * D, h = r / D, d = r - h * D;
        if (b0 + s < B) X0s[(s * 32 + h) * Dpp + d] = x0[(b0 + s) * ld0 + r];
    }
    if (!P.same)
        for (int i = tid; i < S * M * D; i += 256) {
            const int s = i / (M * D), r = i - s * M * D, m = r / D, d = r - m * D;
            if (b0 + s < B) Xps[(s * Mr + m) * Dpp + d] = xp[(b0 + s) * ldp + r];
        }
    __syncthreads();
    const int xrows = P.same ? 32 : Mr;  // row pitch (in rows) of one sample inside Xps
    // work split: with >= 4 output channels a wave owns channels o = wave, wave+4, ... (W[o] staged once, reused
    // over all S*ND tiles); with fewer (the collapsed last layer) the (sample, column tile) items are dealt out
    const bool by_channel = (O >= 4);
    const int nitems = S * ND;
    const int HM = H * M, ostep = by_channel ? 4 : 1;
    const bool pre = HM <= 64 * CIN_WREGS;
    CinWRegs wr;
    CinWOff wo;
    if (pre) cin_w_offsets(wo, HM, M, Ms, lane);
    if (pre && (by_channel ? wave : 0) < O) cin_w_load(wr, W + (int64_t)(by_channel ? wave : 0) * HM, HM, lane);
    for (int o = by_channel ? wave : 0; o < O; o += ostep) {
        if (pre) {
            cin_w_store(wr, wo, Wl);
            if (o + ostep < O) cin_w_load(wr, W + (int64_t)(o + ostep) * HM, HM, lane);  // in flight during the MFMAs
        } else {
            for (int i = lane; i < HM; i += 64) {
                const int h = i / M, m = i - h * M;
                Wl[h * Ms + m] = W[(int64_t)o * HM + i];
            }
        }
        __builtin_amdgcn_wave_barrier();
        const float bo = (bias != nullptr) ? bias[o] : 0.f;
        if (by_channel && (ND % 2) == 0) {
            // two 32-column tiles of a sample per chain
            for (int s = 0; s < S; ++s)
                for (int dp = 0; dp < ND; dp += 2) {
                    const int d = dp * 32 + li;
                    f32x16 acc0, acc1;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        acc0[r] = 0.f;
                        acc1[r] = 0.f;
                    }
                    cin_chain2(acc0, acc1, Wl + li * Ms + half, 2, Xps + (s * xrows + half) * Dpp + d, 2 * Dpp, M2);
                    const float *x0c = X0s + (s * 32 + 4 * half) * Dpp + d;
                    float v0 = 0.f, v1 = 0.f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int ro = ((r & 3) + 8 * (r >> 2)) * Dpp;
                        v0 += acc0[r] * x0c[ro];
                        v1 += acc1[r] * x0c[ro + 32];
                    }
                    v0 += __shfl_xor(v0, 32, 64);
                    v1 += __shfl_xor(v1, 32, 64);
                    v0 += bo;
                    v1 += bo;
                    const bool bok = (b0 + s) < B;
                    const bool k0 = (half == 0) && (d < D), k1 = (half == 0) && (d + 32 < D);
                    if (out != nullptr && bok) {
                        if (k0) out[((b0 + s) * O + o) * D + d] = v0;
                        if (k1) out[((b0 + s) * O + o) * D + d + 32] = v1;
                    }
                    float pv = (k0 ? v0 : 0.f) + (k1 ? v1 : 0.f);
#pragma unroll
                    for (int off = 16; off > 0; off >>= 1) pv += __shfl_xor(pv, off, 64);
                    if (lane == 0) atomicAdd(&poolL[s * O + o], pv);
                }
            __builtin_amdgcn_wave_barrier();
            continue;
        }
        for (int it = by_channel ? 0 : wave; it < nitems; it += by_channel ? 1 : 4) {
            const int s = it / ND, dh = it - s * ND;
            const int d = dh * 32 + li;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            cin_chain(acc, Wl + li * Ms + half, 2, Xps + (s * xrows + half) * Dpp + d, 2 * Dpp, M2, 1.f);
            const float *x0c = X0s + (s * 32 + 4 * half) * Dpp + d;
            float v = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) v += acc[r] * x0c[((r & 3) + 8 * (r >> 2)) * Dpp];
            v += __shfl_xor(v, 32, 64);
            v += bo;
            const bool keep = (half == 0) && (d < D);
            if (out != nullptr && keep && (b0 + s) < B) out[((b0 + s) * O + o) * D + d] = v;
            float pv = keep ? v : 0.f;
#pragma unroll
            for (int off = 16; off > 0; off >>= 1) pv += __shfl_xor(pv, off, 64);
            if (lane == 0) atomicAdd(&poolL[s * O + o], pv);
        }
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    if (pooled != nullptr)
        for (int i = tid; i < S * O; i += 256) {
            const int s = i / O, o = i - s * O;
            if (b0 + s < B) pooled[(b0 + s) * ldpool + o] = poolL[i];
        }
}

static int cin_pick_plan(int H, int M, int O, int D, int same, CinPlan *out) {
    for (int S = 4; S >= 1; S >>= 1) {
        CinPlan p = cin_plan(H, M, O, D, S, same);
        const size_t bytes = (size_t)p.total * sizeof(float);
        if (bytes <= (S > 1 ? CIN_LDS_BUDGET / 2 : CIN_LDS_BUDGET) || (S == 1 && bytes <= CIN_LDS_BUDGET)) {
            *out = p;
            return RP_OK;
        }
    }
    // nothing fits twice per CU: take the largest S that fits once
    for (int S = 4; S >= 1; S >>= 1) {
        CinPlan p = cin_plan(H, M, O, D, S, same);
        if ((size_t)p.total * sizeof(float) <= CIN_LDS_BUDGET) {
            *out = p;
            return RP_OK;
        }
    }
    return rp_fail(RP_ERR_UNSUPPORTED, "cin: H=%d M=%d O=%d D=%d does not fit in LDS", H, M, O, D);
}

extern "C" int rp_cin_layer_fwd(const float *x0, int64_t ld0, const float *xp, int64_t ldp, const float *W,
                                const float *bias, float *out, float *pooled, int64_t ldpool, int H, int M, int O,
                                int D, int64_t B, rp_stream_t stream) {
    RP_REQUIRE(x0 && xp && W && B >= 0, "cin_layer_fwd: null pointer");
    RP_REQUIRE(out != nullptr || pooled != nullptr, "cin_layer_fwd: nothing to produce");
    RP_REQUIRE(H >= 1 && M >= 1 && O >= 1 && D >= 1 && ld0 >= (int64_t)H * D && ldp >= (int64_t)M * D,
               "cin_layer_fwd: bad dims");
    RP_REQUIRE(pooled == nullptr || ldpool >= O, "cin_layer_fwd: ldpool < O");
    if (H > CIN_MAXH) return rp_fail(RP_ERR_UNSUPPORTED, "cin: H=%d fields (max %d) unsupported", H, CIN_MAXH);
    if (B == 0) return RP_OK;
    const int same = (xp == x0) && (ldp == ld0) && (M == H);
    CinPlan P;
    int rc = cin_pick_plan(H, M, O, D, same, &P);
    if (rc != RP_OK) return rc;
    const size_t lds = (size_t)P.total * sizeof(float);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(cin_layer_fwd_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(cin_layer_fwd_kernel, dim3((unsigned)rp_cdiv(B, P.S)), dim3(256), lds, (hipStream_t)stream, x0,
                       ld0, xp, ldp, W, bias, out, pooled, ldpool, H, M, O, D, B, P);
    RP_LAUNCH_CHECK("cin_layer_fwd");
    return RP_OK;
}

// ------------------------------------------------------------------------------------------------
// Backward, part 1: gradients of the two factors.
//   G[b,o,d] = g_out[b,o,d] + g_pool[b,o]
//   dX_0[h,d]     = sum_o G[o,d] * T_o[h,d],          T_o[h,d] = sum_m W[o,h,m] X_{k-1}[m,d]   (the forward chain)
//   dX_{k-1}[m,d] = sum_o sum_h W[o,h,m] * (G[o,d] X_0[h,d])           (A = W[o]^T, B = G-scaled X_0: all o and h
//                                                                       accumulate into the same 32x32 tile)
// Same staging as the forward.  With O >= 4 a wave owns channels o = wave, wave+4, ... and keeps the S*ND
// accumulator tiles of both gradients in registers across its channels (needs M <= 32); the four waves' partial
// tiles are then summed through LDS.  With O < 4 (the collapsed last layer, any M) the (sample, column tile)
// items are dealt to the waves, which finish their tiles alone.
// ------------------------------------------------------------------------------------------------
#define CIN_BX_MAXT 4  // S * ND accumulator tile pairs per wave in by-channel mode

__device__ __forceinline__ void cin_store_tile(const f32x16 &acc, float *__restrict__ dst, int rows_valid, int D,
                                               int d, int half, bool add) {
    // C layout -> dst[row][d], row = (r&3)+8(r>>2)+4*half, rows >= rows_valid / columns >= D skipped
    if (d >= D) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
        if (row < rows_valid) {
            float *p = dst + (int64_t)row * D + d;
            *p = add ? (*p + acc[r]) : acc[r];
        }
    }
}

__global__ __launch_bounds__(256) void cin_layer_bwd_x_kernel(
    const float *__restrict__ x0, int64_t ld0, const float *__restrict__ xp, int64_t ldp, const float *__restrict__ W,
    const float *__restrict__ g_out, const float *__restrict__ g_pool, int64_t ldgp, float *__restrict__ dx0,
    int64_t lddx0, int add_dx0, float *__restrict__ dxp, int64_t lddxp, int H, int M, int O, int D, int64_t B,
    CinPlan P, int offRed) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int S = P.S, Dpp = P.Dpp, Mr = P.Mr, Ms = P.Ms;
    float *X0s = smem;
    float *Xps = smem + P.offXp;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, half = lane >> 5;
    float *Wl = smem + P.offW + wave * 32 * Ms;
    float *Red = smem + offRed;  // [4 waves][S*ND][32*32] (by-channel mode only)
    const int ND = (D + 31) / 32, M2 = Mr / 2, H2 = (H + 1) / 2, MT = (M + 31) / 32;
    const int64_t b0 = (int64_t)blockIdx.x * S;
    for (int i = tid; i < P.offW + 4 * 32 * Ms; i += 256) smem[i] = 0.f;
    __syncthreads();
    for (int i = tid; i < S * H * D; i += 256) {
        const int s = i / (H * D), r = i - s * H * D, h = r / D, d = r - h * D;
        if (b0 + s < B) X0s[(s * 32 + h) * Dpp + d] = x0[(b0 + s) * ld0 + r];
    }
    if (!P.same)
        for (int i = tid; i < S * M * D; i += 256) {
            const int s = i / (M * D), r = i - s * M * D, m = r / D, d = r - m * D;
            if (b0 + s < B) Xps[(s * Mr + m) * Dpp + d] = xp[(b0 + s) * ldp + r];
        }
    __syncthreads();
    const int xrows = P.same ? 32 : Mr;
    const bool by_channel = (O >= 4);
    const int nitems = S * ND;

    if (by_channel) {
        f32x16 a0[CIN_BX_MAXT], ap[CIN_BX_MAXT];
#pragma unroll
        for (int it = 0; it < CIN_BX_MAXT; ++it)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                a0[it][r] = 0.f;
                ap[it][r] = 0.f;
            }
        const int HM = H * M;  // M <= 32 here, so H*M <= 1024 always fits the prefetch registers
        CinWRegs wr;
        CinWOff wo;
        cin_w_offsets(wo, HM, M, Ms, lane);
        if (wave < O) cin_w_load(wr, W + (int64_t)wave * HM, HM, lane);
        for (int o = wave; o < O; o += 4) {
            cin_w_store(wr, wo, Wl);
            if (o + 4 < O) cin_w_load(wr, W + (int64_t)(o + 4) * HM, HM, lane);
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int it = 0; it < CIN_BX_MAXT; ++it) {
                if (it < nitems) {
                    const int s = it / ND, dh = it - s * ND;
                    const int d = dh * 32 + li;
                    const bool ok = (d < D) && (b0 + s < B);
                    float g = 0.f;
                    if (ok) {
                        if (g_out != nullptr) g += g_out[((b0 + s) * O + o) * D + d];
                        if (g_pool != nullptr) g += g_pool[(b0 + s) * ldgp + o];
                    }
                    // (i) forward chain T_o, then dX0 += G * T_o
                    f32x16 t;
#pragma unroll
                    for (int r = 0; r < 16; ++r) t[r] = 0.f;
                    cin_chain(t, Wl + li * Ms + half, 2, Xps + (s * xrows + half) * Dpp + d, 2 * Dpp, M2, 1.f);
#pragma unroll
                    for (int r = 0; r < 16; ++r) a0[it][r] += g * t[r];
                    // (ii) dXp tile (m = li, single m tile): A[m][h] = W[o][h][m], B[h][d] = G * X0[h][d]
                    cin_chain(ap[it], Wl + half * Ms + li, 2 * Ms, X0s + (s * 32 + half) * Dpp + d, 2 * Dpp, H2, g);
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        // cross-wave sums through LDS: first dX0 tiles, then dXp tiles
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            __syncthreads();
#pragma unroll
            for (int it = 0; it < CIN_BX_MAXT; ++it)
                if (it < nitems) {
                    float *dst = Red + ((wave * nitems + it) * 1024);
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        dst[((r & 3) + 8 * (r >> 2) + 4 * half) * 32 + li] = (pass == 0 ? a0[it][r] : ap[it][r]);
                }
            __syncthreads();
            const int rows = (pass == 0) ? H : M;
            for (int i = tid; i < nitems * rows * 32; i += 256) {
                const int it = i / (rows * 32), rem = i - it * rows * 32, row = rem / 32, col = rem - row * 32;
                const int s = it / ND, dh = it - s * ND, d = dh * 32 + col;
                if (d >= D || b0 + s >= B) continue;
                float v = 0.f;
                for (int w = 0; w < 4; ++w) v += Red[(w * nitems + it) * 1024 + row * 32 + col];
                if (pass == 0 || P.same) {
                    float *p = dx0 + (b0 + s) * lddx0 + (int64_t)row * D + d;
                    *p = (add_dx0 || pass == 1) ? (*p + v) : v;
                } else {
                    dxp[(b0 + s) * lddxp + (int64_t)row * D + d] = v;
                }
            }
        }
    } else {
        for (int it = wave; it < nitems; it += 4) {
            const int s = it / ND, dh = it - s * ND;
            const int d = dh * 32 + li;
            const bool bok = b0 + s < B;
            f32x16 a0;
#pragma unroll
            for (int r = 0; r < 16; ++r) a0[r] = 0.f;
            for (int mt = -1; mt < MT; ++mt) {
                // mt = -1: the dX0 tile; mt >= 0: the dXp tile of rows m in [32 mt, 32 mt + 32)
                f32x16 ap;
#pragma unroll
                for (int r = 0; r < 16; ++r) ap[r] = 0.f;
                for (int o = 0; o < O; ++o) {
                    __builtin_amdgcn_wave_barrier();
                    for (int i = lane; i < H * M; i += 64) {
                        const int h = i / M, m = i - h * M;
                        Wl[h * Ms + m] = W[(int64_t)o * H * M + i];
                    }
                    __builtin_amdgcn_wave_barrier();
                    float g = 0.f;
                    if (d < D && bok) {
                        if (g_out != nullptr) g += g_out[((b0 + s) * O + o) * D + d];
                        if (g_pool != nullptr) g += g_pool[(b0 + s) * ldgp + o];
                    }
                    if (mt < 0) {
                        f32x16 t;
#pragma unroll
                        for (int r = 0; r < 16; ++r) t[r] = 0.f;
                        cin_chain(t, Wl + li * Ms + half, 2, Xps + (s * xrows + half) * Dpp + d, 2 * Dpp, M2, 1.f);
#pragma unroll
                        for (int r = 0; r < 16; ++r) a0[r] += g * t[r];
                    } else {
                        const int mcol = mt * 32 + li;  // tile rows beyond M read a clamped column and are never stored
                        const float *wt = Wl + half * Ms + (mcol < Ms ? mcol : Ms - 1);
                        cin_chain(ap, wt, 2 * Ms, X0s + (s * 32 + half) * Dpp + d, 2 * Dpp, H2, g);
                    }
                }
                if (!bok) continue;
                if (mt < 0) {
                    cin_store_tile(a0, dx0 + (b0 + s) * lddx0, H, D, d, half, add_dx0 != 0);
                } else {
                    const int rows_valid = (M - mt * 32) < 32 ? (M - mt * 32) : 32;
                    if (P.same)
                        cin_store_tile(ap, dx0 + (b0 + s) * lddx0 + (int64_t)mt * 32 * D, rows_valid, D, d, half, true);
                    else
                        cin_store_tile(ap, dxp + (b0 + s) * lddxp + (int64_t)mt * 32 * D, rows_valid, D, d, half, false);
                }
            }
        }
    }
}

extern "C" int rp_cin_layer_bwd_x(const float *x0, int64_t ld0, const float *xp, int64_t ldp, const float *W,
                                  const float *g_out, const float *g_pool, int64_t ldgp, float *dx0, int64_t lddx0,
                                  int add_dx0, float *dxp, int64_t lddxp, int H, int M, int O, int D, int64_t B,
                                  rp_stream_t stream) {
    RP_REQUIRE(x0 && xp && W && dx0 && B >= 0, "cin_layer_bwd_x: null pointer");
    RP_REQUIRE(g_out != nullptr || g_pool != nullptr, "cin_layer_bwd_x: no incoming gradient");
    RP_REQUIRE(H >= 1 && M >= 1 && O >= 1 && D >= 1 && ld0 >= (int64_t)H * D && ldp >= (int64_t)M * D &&
                   lddx0 >= (int64_t)H * D, "cin_layer_bwd_x: bad dims");
    const int same = (xp == x0) && (ldp == ld0) && (M == H);
    RP_REQUIRE(same || (dxp && lddxp >= (int64_t)M * D), "cin_layer_bwd_x: dxp missing");
    if (H > CIN_MAXH) return rp_fail(RP_ERR_UNSUPPORTED, "cin: H=%d fields (max %d) unsupported", H, CIN_MAXH);
    if (O >= 4 && M > 32)
        return rp_fail(RP_ERR_UNSUPPORTED, "cin backward: a middle layer with M=%d > 32 input maps and O=%d is not "
                                           "supported by the register-tiled kernel", M, O);
    if (B == 0) return RP_OK;
    const int ND = (D + 31) / 32;
    CinPlan P;
    int found = 0;
    int offRed = 0;
    size_t lds = 0;
    for (int S = 4; S >= 1; S >>= 1) {
        if (O >= 4 && S * ND > CIN_BX_MAXT) continue;
        P = cin_plan(H, M, O, D, S, same);
        offRed = P.offW + 4 * 32 * P.Ms;
        lds = ((size_t)offRed + (O >= 4 ? (size_t)4 * S * ND * 1024 : 0)) * sizeof(float);
        if (lds <= CIN_LDS_BUDGET) {
            found = 1;
            break;
        }
    }
    if (!found) return rp_fail(RP_ERR_UNSUPPORTED, "cin backward: H=%d M=%d O=%d D=%d does not fit in LDS", H, M, O, D);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(cin_layer_bwd_x_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(cin_layer_bwd_x_kernel, dim3((unsigned)rp_cdiv(B, P.S)), dim3(256), lds, (hipStream_t)stream, x0,
                       ld0, xp, ldp, W, g_out, g_pool, ldgp, dx0, lddx0, add_dx0, dxp, lddxp, H, M, O, D, B, P, offRed);
    RP_LAUNCH_CHECK("cin_layer_bwd_x");
    return RP_OK;
}

// ------------------------------------------------------------------------------------------------
// Backward, part 2: weight and bias gradients.
//   dW[o,h,m] = sum_b sum_d (G[b,o,d] X_0[b,h,d]) * X_{k-1}[b,m,d]        dbias[o] = sum_b sum_d G[b,o,d]
// Contraction over (sample, d): per (sample, o) one MFMA chain of D/2 steps per 32-column tile of m, A[h][d] =
// G X_0 read column-wise from the (odd-stride) LDS tile, B[d][m] = X_{k-1} likewise.  A workgroup owns 4 output
// channels (one per wave) for a chunk of samples and keeps the [32 x 32*MT] accumulator in registers over the whole
// chunk; chunk partials go to the workspace and are summed in fixed order (deterministic, no atomics).
// ------------------------------------------------------------------------------------------------
template <int MT>
__global__ __launch_bounds__(256) void cin_layer_bwd_w_kernel(const float *__restrict__ x0, int64_t ld0,
                                                              const float *__restrict__ xp, int64_t ldp,
                                                              const float *__restrict__ g_out,
                                                              const float *__restrict__ g_pool, int64_t ldgp,
                                                              float *__restrict__ Pw, float *__restrict__ Pb, int H,
                                                              int M, int O, int D, int64_t B, int64_t chunk, int S,
                                                              int same) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int ND = (D + 31) / 32, Dpp = ND * 32 + 1, MR = MT * 32;
    float *X0s = smem;                                   // [S][32][Dpp]
    float *Xps = same ? X0s : smem + S * 32 * Dpp;       // [S][MR][Dpp]
    float *Gs = smem + S * 32 * Dpp + (same ? 0 : S * MR * Dpp);  // [4][S][Dpp]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, half = lane >> 5;
    const int o = blockIdx.x * 4 + wave;
    const bool o_ok = o < O;
    const int D2 = (D + 1) / 2;
    const int total = S * 32 * Dpp + (same ? 0 : S * MR * Dpp) + 4 * S * Dpp;
    f32x16 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;
    float bsum = 0.f;
    const int64_t bbeg = (int64_t)blockIdx.y * chunk;
    int64_t bend = bbeg + chunk;
    if (bend > B) bend = B;
    for (int i = tid; i < total; i += 256) smem[i] = 0.f;  // padding rows/columns stay zero for the whole chunk
    for (int64_t b0 = bbeg; b0 < bend; b0 += S) {
        __syncthreads();
        for (int i = tid; i < S * H * D; i += 256) {
            const int s = i / (H * D), r = i - s * H * D, h = r / D, d = r - h * D;
            X0s[(s * 32 + h) * Dpp + d] = (b0 + s < bend) ? x0[(b0 + s) * ld0 + r] : 0.f;
        }
        if (!same)
            for (int i = tid; i < S * M * D; i += 256) {
                const int s = i / (M * D), r = i - s * M * D, m = r / D, d = r - m * D;
                Xps[(s * MR + m) * Dpp + d] = (b0 + s < bend) ? xp[(b0 + s) * ldp + r] : 0.f;
            }
        if (o_ok)
            for (int i = lane; i < S * D; i += 64) {
                const int s = i / D, d = i - s * D;
                float g = 0.f;
                if (b0 + s < bend) {
                    if (g_out != nullptr) g += g_out[((b0 + s) * O + o) * D + d];
                    if (g_pool != nullptr) g += g_pool[(b0 + s) * ldgp + o];
                }
                Gs[(wave * S + s) * Dpp + d] = g;
                bsum += g;
            }
        __syncthreads();
        if (o_ok) {
            for (int s = 0; s < S; ++s) {
                const float *gs = Gs + (wave * S + s) * Dpp;
                const float *xa = X0s + (s * 32 + li) * Dpp;
                const float *xb = Xps + (s * (same ? 32 : MR) + li) * Dpp;
                int s2 = 0;
                for (; s2 + 4 <= D2; s2 += 4) {  // operands four steps ahead of their MFMAs (see cin_chain)
                    float a4[4], b4[4][MT];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int d = 2 * (s2 + u) + half;
                        a4[u] = gs[d] * xa[d];
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) b4[u][mt] = (same && mt > 0) ? 0.f : xb[(mt * 32) * Dpp + d];
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u)
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
                            acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[u], b4[u][mt], acc[mt], 0, 0, 0);
                }
                for (; s2 < D2; ++s2) {
                    const int d = 2 * s2 + half;
                    const float a = gs[d] * xa[d];
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        const float bv = (same && mt > 0) ? 0.f : xb[(mt * 32) * Dpp + d];
                        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc[mt], 0, 0, 0);
                    }
                }
            }
        }
    }
    if (!o_ok) return;
    float *dst = Pw + ((int64_t)blockIdx.y * O + o) * H * M;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int m = mt * 32 + li;
        if (m < M) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int h = (r & 3) + 8 * (r >> 2) + 4 * half;
                if (h < H) dst[h * M + m] = acc[mt][r];
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) bsum += __shfl_xor(bsum, off, 64);
    if (lane == 0) Pb[(int64_t)blockIdx.y * O + o] = bsum;
}

__global__ __launch_bounds__(256) void cin_partial_sum_kernel(const float *__restrict__ partial, int nchunk,
                                                              int64_t n, float *__restrict__ out) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    float s = 0.f;
    for (int k = 0; k < nchunk; ++k) s += partial[(int64_t)k * n + e];
    out[e] = s;
}

static void cin_bw_plan(int64_t B, int O, int64_t *chunk, int *nchunk, int S) {
    const int64_t og = (O + 3) / 4;
    int64_t nc = 1024 / og;
    if (nc < 1) nc = 1;
    int64_t ch = (B + nc - 1) / nc;
    ch = ((ch + S - 1) / S) * S;
    if (ch < 64) ch = 64;
    *chunk = ch;
    *nchunk = (int)((B + ch - 1) / ch);
}

extern "C" int rp_cin_layer_bwd_w_workspace_bytes(int64_t B, int H, int M, int O, size_t *bytes) {
    RP_REQUIRE(bytes && B >= 1 && H >= 1 && M >= 1 && O >= 1, "cin_layer_bwd_w_workspace_bytes: bad argument");
    int64_t chunk;
    int nchunk;
    cin_bw_plan(B, O, &chunk, &nchunk, 8);
    *bytes = ((size_t)nchunk * O * H * M + (size_t)nchunk * O) * sizeof(float) + 256;
    return RP_OK;
}

extern "C" int rp_cin_layer_bwd_w(const float *x0, int64_t ld0, const float *xp, int64_t ldp, const float *g_out,
                                  const float *g_pool, int64_t ldgp, float *dW, float *dbias, int H, int M, int O,
                                  int D, int64_t B, void *workspace, size_t workspace_bytes, rp_stream_t stream) {
    RP_REQUIRE(x0 && xp && dW && workspace && B >= 1, "cin_layer_bwd_w: null pointer");
    RP_REQUIRE(g_out != nullptr || g_pool != nullptr, "cin_layer_bwd_w: no incoming gradient");
    RP_REQUIRE(H >= 1 && M >= 1 && O >= 1 && D >= 1 && ld0 >= (int64_t)H * D && ldp >= (int64_t)M * D,
               "cin_layer_bwd_w: bad dims");
    if (H > CIN_MAXH || M > 256)
        return rp_fail(RP_ERR_UNSUPPORTED, "cin: H=%d (max %d) / M=%d (max 256) unsupported", H, CIN_MAXH, M);
    size_t need = 0;
    rp_cin_layer_bwd_w_workspace_bytes(B, H, M, O, &need);
    RP_REQUIRE(workspace_bytes >= need, "cin_layer_bwd_w: workspace %zu < %zu", workspace_bytes, need);
    const int same = (xp == x0) && (ldp == ld0) && (M == H);
    const int MT = (M + 31) / 32, MTt = MT <= 1 ? 1 : (MT <= 4 ? 4 : 8);
    const int ND = (D + 31) / 32, Dpp = ND * 32 + 1;
    int S = 8;  // samples staged per barrier pair: more of them amortise the staging latency
    size_t lds = 0;
    for (; S >= 1; S >>= 1) {
        lds = ((size_t)S * 32 * Dpp + (same ? 0 : (size_t)S * MTt * 32 * Dpp) + (size_t)4 * S * Dpp) * sizeof(float);
        if (lds <= CIN_LDS_BUDGET / 2 || S == 1) break;
    }
    if (lds > CIN_LDS_BUDGET) return rp_fail(RP_ERR_UNSUPPORTED, "cin backward-w: M=%d D=%d does not fit in LDS", M, D);
    int64_t chunk;
    int nchunk;
    cin_bw_plan(B, O, &chunk, &nchunk, 8);
    float *Pw = reinterpret_cast<float *>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
    float *Pb = Pw + (size_t)nchunk * O * H * M;
    dim3 grid((unsigned)((O + 3) / 4), (unsigned)nchunk);
    hipStream_t st = (hipStream_t)stream;
#define CIN_BW(MTT)                                                                                                   \
    do {                                                                                                              \
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(cin_layer_bwd_w_kernel<MTT>),                        \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                              \
        hipLaunchKernelGGL((cin_layer_bwd_w_kernel<MTT>), grid, dim3(256), lds, st, x0, ld0, xp, ldp, g_out, g_pool,  \
                           ldgp, Pw, Pb, H, M, O, D, B, chunk, S, same);                                              \
    } while (0)
    if (MTt == 1) CIN_BW(1);
    else if (MTt == 4) CIN_BW(4);
    else CIN_BW(8);
#undef CIN_BW
    RP_LAUNCH_CHECK("cin_layer_bwd_w");
    const int64_t n = (int64_t)O * H * M;
    hipLaunchKernelGGL(cin_partial_sum_kernel, dim3((unsigned)rp_cdiv(n, 256)), dim3(256), 0, st, Pw, nchunk, n, dW);
    RP_LAUNCH_CHECK("cin_layer_bwd_w reduce");
    if (dbias != nullptr) {
        hipLaunchKernelGGL(cin_partial_sum_kernel, dim3((unsigned)rp_cdiv((int64_t)O, 256)), dim3(256), 0, st, Pb,
                           nchunk, (int64_t)O, dbias);
        RP_LAUNCH_CHECK("cin_layer_bwd_w reduce bias");
    }
    return RP_OK;
}
