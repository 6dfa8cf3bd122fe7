// xDeepFM CIN layer on the bf16 matrix core (split-bf16 operands), "B-stationary" form, gfx950.
//   reference: layers/interaction.py:157-171
//     X_k[b,o,d] = sum_{h<H} X_0[b,h,d] * T_o[h,(b,d)],    T_o[h,(b,d)] = sum_{m<M} W[o,h,m] X_{k-1}[b,m,d]    (+ bias[o])
// T_o is a GEMM tile: rows h (<= 32), columns r = (b,d), contraction over m (<= 32 here: a first layer, where
// X_{k-1} = X_0).  The f32-MFMA kernels of cin.hip stage the samples in LDS and re-read them for every channel; here
// the roles are turned around: a wave keeps the X_{k-1} fragments of its column tiles IN REGISTERS for the whole
// kernel (B operand: lane = column r, 8 consecutive m per k-octet; 24 VGPRs per 32-column tile) and the 128 channels'
// weights stream past — pre-split into bf16 pieces on the host side ([O][3][32][32] bf16, 6 KB per channel, L2-resident),
// four channels per LDS buffer, double buffered, one barrier per four channels.  Per (channel, column tile):
// 2 k-steps x 6 products = 12 v_mfma_f32_32x32x16_bf16, then the epilogue in the C layout
// (col = lane & 31 = column r, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) = h):
//   MODE 0 (forward)    y[o,r] = sum_h X_0[h,r] * acc[h,r]   (16 fma + one cross-half shuffle), out[b,o,d], pooled[b,o]
//   MODE 1 (backward-x) dX[h,r] += G[o,r] * acc[h,r]          (G = dL/dX_k[b,o,d] + dL/dpooled[b,o])
// Backward-x with weights W[o,h,m] gives the gradient of the X_0 role, with W[o,m,h] that of the X_{k-1} role; for a
// first layer both are X_0 and ONE pass with W[o,h,m] + W[o,m,h] gives the whole gradient.
// A wave owns the two 32-column tiles of one sample (D = 64) or of two samples (D = 32): 4 waves = 4 (8) samples
// per workgroup.  FULL = all of them exist: straight-line channel loop (counted vmcnt, see gemm.hip); the tail of the
// batch runs the guarded instantiation.
#include "common.h"

typedef __bf16 cbbf8 __attribute__((ext_vector_type(8)));
typedef float cbf8 __attribute__((ext_vector_type(8)));

#define CB_OG 4      // channels per LDS buffer
#define CB_LD 40     // bf16 per LDS row (32 data + 8 pad -> 80 B, conflict-free ds_read_b128)

__device__ __forceinline__ void cb_split(cbf8 v, cbbf8 (&p)[3]) {
    p[0] = __builtin_convertvector(v, cbbf8);
    v -= __builtin_convertvector(p[0], cbf8);
    p[1] = __builtin_convertvector(v, cbbf8);
    v -= __builtin_convertvector(p[1], cbf8);
    p[2] = __builtin_convertvector(v, cbbf8);
}

// xk   : contraction operand X_{k-1} rows, [B, ldk] with Mc rows of D floats per sample (Mc <= 32)
// xe   : MODE 0: X_0 ([B, lde], Hr rows of D floats) for the epilogue;  MODE 1: unused
// wp   : weights as bf16 pieces [O][3][32][32] (row = output row of the tile, col = contraction index), zero padded
// MODE 0: bias [O] or null, out [B, O*D] (ldo = O*D) or null, pooled [B, O] or null
// MODE 1: gout [B, O*D] or null, gpool [B, O] or null, dx [B, lddx] (Hr rows of D floats written)
template <int MODE, bool FULL>
__global__ __launch_bounds__(256, 2) void cin_bs_kernel(const float *__restrict__ xk, int64_t ldk, int Mc,
                                                     const float *__restrict__ xe, int64_t lde, int Hr,
                                                     const __bf16 *__restrict__ wp, int O, int D,
                                                     const float *__restrict__ bias, float *__restrict__ out,
                                                     float *__restrict__ pooled, const float *__restrict__ gout,
                                                     const float *__restrict__ gpool, float *__restrict__ dx,
                                                     int64_t lddx, int64_t B) {
    __shared__ __attribute__((aligned(16))) __bf16 Wl[2][CB_OG][3][32][CB_LD];
    const int t = threadIdx.x;
    const int w = t >> 6, l = t & 63, i = l & 31, hh = l >> 5;
    const int tps = D / 32;                    // 32-column tiles per sample (1 or 2)
    const int spw = 2 / tps;                   // samples per wave (2 tiles per wave)
    const int64_t bw = ((int64_t)blockIdx.x * 4 + w) * spw;  // first sample of this wave
    int64_t bt[2];
    int dt[2];
    bool ok[2];
#pragma unroll
    for (int ti = 0; ti < 2; ++ti) {
        bt[ti] = bw + (tps == 2 ? 0 : ti);
        dt[ti] = (tps == 2 ? ti * 32 : 0) + i;
        ok[ti] = FULL || bt[ti] < B;
        if (!ok[ti]) bt[ti] = B - 1;  // clamp: loads stay in range, stores are suppressed
    }
    // B fragments: lane = column (b, d), k = 16*ks + 8*hh + e
    cbbf8 bf[2][2][3];
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            cbf8 v;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int m = ks * 16 + 8 * hh + e;
                v[e] = (m < Mc) ? xk[bt[ti] * ldk + (int64_t)m * D + dt[ti]] : 0.f;
            }
            cb_split(v, bf[ti][ks]);
        }
    // epilogue operands in the C layout: row h = (r&3) + 8*(r>>2) + 4*hh
    float ep[2][16];
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int h = (r & 3) + 8 * (r >> 2) + 4 * hh;
            if (MODE == 0) ep[ti][r] = (h < Hr) ? xe[bt[ti] * lde + (int64_t)h * D + dt[ti]] : 0.f;
            else ep[ti][r] = 0.f;  // dX accumulator
        }

    // weight stream: 4 channels = 4*3*32 rows of 64 B = 1536 16-byte chunks per buffer -> 6 per thread
    const int ngrp = (O + CB_OG - 1) / CB_OG;
    f32x4 wr[6];
    auto w_load = [&](int grp) {
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            const int chunk = t + 256 * u;           // [oo][q][row][c16]: row-major 16-byte chunks
            const int oo = chunk / 384;              // 3*32*4 chunks per channel
            int o = grp * CB_OG + oo;
            if (o >= O) o = O - 1;                   // (harmless re-read; such channels are never consumed)
            wr[u] = *reinterpret_cast<const f32x4 *>(reinterpret_cast<const char *>(wp) +
                                                     ((int64_t)o * 384 + (chunk - oo * 384)) * 16);
        }
    };
    auto w_store = [&](int buf) {
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            const int chunk = t + 256 * u;
            const int oo = chunk / 384, rem = chunk - oo * 384;
            const int q = rem / 128, rr = (rem - q * 128) >> 2, c16 = rem & 3;
            *reinterpret_cast<f32x4 *>(&Wl[buf][oo][q][rr][c16 * 8]) = wr[u];
        }
    };
    // MODE 1: the upstream gradients of a channel group are fetched together at the top of the group
    float gc[CB_OG][2];
    auto g_load = [&](int grp) {
#pragma unroll
        for (int oo = 0; oo < CB_OG; ++oo) {
            int o = grp * CB_OG + oo;
            if (o >= O) o = O - 1;
#pragma unroll
            for (int ti = 0; ti < 2; ++ti) {
                float gv = 0.f;
                if (gout != nullptr) gv = gout[(bt[ti] * O + o) * D + dt[ti]];
                if (gpool != nullptr) gv += gpool[bt[ti] * O + o];
                gc[oo][ti] = gv;
            }
        }
    };
    w_load(0);
    for (int grp = 0; grp < ngrp; ++grp) {
        const int buf = grp & 1;
        w_store(buf);
        // LDS-only barrier (no release fence: the output stores of the previous group must not be drained here)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        w_load(grp + 1 < ngrp ? grp + 1 : grp);
        if (MODE == 1) g_load(grp);  // consumed one to four channels of MFMA work later
#pragma unroll
        for (int oo = 0; oo < CB_OG; ++oo) {
            const int o = grp * CB_OG + oo;
            if (!FULL && o >= O) break;
            cbbf8 a[2][3];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int q = 0; q < 3; ++q)
                    a[ks][q] = *reinterpret_cast<const cbbf8 *>(&Wl[buf][oo][q][i][ks * 16 + 8 * hh]);
            float ysum = 0.f;
#pragma unroll
            for (int ti = 0; ti < 2; ++ti) {
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    // hi*lo + lo*hi + mid*mid + hi*mid + mid*hi + hi*hi  (smallest terms first)
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks][0], bf[ti][ks][2], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks][2], bf[ti][ks][0], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks][1], bf[ti][ks][1], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks][0], bf[ti][ks][1], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks][1], bf[ti][ks][0], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks][0], bf[ti][ks][0], acc, 0, 0, 0);
                }
                if (MODE == 0) {
                    float y = 0.f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) y += ep[ti][r] * acc[r];
                    y += __shfl_xor(y, 32, 64);  // the other 16 rows live in the other lane half
                    const float yb = y + (bias != nullptr ? bias[o] : 0.f);
                    // FULL: both lane halves hold the same value and store it to the same address (a lane-dependent
                    // guard would put the store behind a branch and the weight prefetch behind a vmcnt(0))
                    if (out != nullptr && (FULL || (hh == 0 && ok[ti]))) out[(bt[ti] * O + o) * D + dt[ti]] = yb;
                    if (pooled != nullptr) {
                        if (tps == 2) {
                            ysum += yb;  // both tiles belong to this wave's sample
                        } else {
                            float s = (hh == 0) ? yb : 0.f;
#pragma unroll
                            for (int sh = 32; sh > 0; sh >>= 1) s += __shfl_xor(s, sh, 64);  // every lane: the total
                            if (FULL || (l == 0 && ok[ti])) pooled[bt[ti] * O + o] = s;
                        }
                    }
                } else {
                    const float gv = gc[oo][ti];
#pragma unroll
                    for (int r = 0; r < 16; ++r) ep[ti][r] += gv * acc[r];
                }
            }
            if (MODE == 0 && pooled != nullptr && tps == 2) {
                float s = (hh == 0) ? ysum : 0.f;
#pragma unroll
                for (int sh = 32; sh > 0; sh >>= 1) s += __shfl_xor(s, sh, 64);  // every lane: the total
                if (FULL || (l == 0 && ok[0])) pooled[bt[0] * O + o] = s;
            }
            // keep the scheduler from interleaving the four channels of a group: it buys nothing (the matrix core is
            // busy either way) and the extra live accumulators cost the second wave per SIMD
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (MODE == 1) {
#pragma unroll
        for (int ti = 0; ti < 2; ++ti)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int h = (r & 3) + 8 * (r >> 2) + 4 * hh;
                if (h < Hr && ok[ti]) dx[bt[ti] * lddx + (int64_t)h * D + dt[ti]] = ep[ti][r];
            }
    }
}

extern "C" int rp_cin_bs_fits(int H, int M, int D) { return (H >= 1 && H <= 32 && M >= 1 && M <= 32 && (D == 32 || D == 64)) ? 1 : 0; }

static int cb_check(int Hr, int Mc, int O, int D) {
    if (!rp_cin_bs_fits(Hr, Mc, D) || O < 1)
        return rp_fail(RP_ERR_UNSUPPORTED, "cin_bs: rows=%d contraction=%d (<=32) D=%d (32|64) O=%d unsupported", Hr, Mc, D, O);
    return RP_OK;
}

template <int MODE>
static void cb_launch(const float *xk, int64_t ldk, int Mc, const float *xe, int64_t lde, int Hr, const void *wp, int O,
                      int D, const float *bias, float *out, float *pooled, const float *gout, const float *gpool,
                      float *dx, int64_t lddx, int64_t B, hipStream_t s) {
    const int spb = 4 * (D == 64 ? 1 : 2);  // samples per workgroup
    // the straight-line kernel also wants whole channel groups; MODE 1 has no stores in its channel loop (nothing for a
    // conservative vmcnt to drain) and its fully unrolled FULL form spills, so it always runs the guarded form
    const int64_t Bf = (MODE == 0 && O % CB_OG == 0) ? (B / spb) * spb : 0;
    const __bf16 *w = reinterpret_cast<const __bf16 *>(wp);
    if (Bf > 0)
        hipLaunchKernelGGL((cin_bs_kernel<MODE, true>), dim3((unsigned)(Bf / spb)), dim3(256), 0, s, xk, ldk, Mc, xe, lde, Hr,
                           w, O, D, bias, out, pooled, gout, gpool, dx, lddx, Bf);
    if (B > Bf) {  // tail samples (or every sample when O is not a multiple of 4): guarded instantiation
        const int64_t o1 = Bf;
        hipLaunchKernelGGL((cin_bs_kernel<MODE, false>), dim3((unsigned)rp_cdiv(B - Bf, spb)), dim3(256), 0, s, xk + o1 * ldk, ldk, Mc,
                           xe ? xe + o1 * lde : nullptr, lde, Hr, w, O, D, bias, out ? out + o1 * (int64_t)O * D : nullptr,
                           pooled ? pooled + o1 * O : nullptr, gout ? gout + o1 * (int64_t)O * D : nullptr,
                           gpool ? gpool + o1 * O : nullptr, dx ? dx + o1 * lddx : nullptr, lddx, B - Bf);
    }
}

// forward of one CIN layer with at most 32 x 32 (field, map) pairs per channel.
//   x0 [B, ld0]: H rows of D floats; xp [B, ldp]: M rows of D floats (may alias x0); wp: [O][3][32][32] bf16 pieces of
//   W[o, h, m] (row h, column m, zero padded); out [B, O, D] and/or pooled [B, O].
extern "C" int rp_cin_bs_fwd(const float *x0, int64_t ld0, const float *xp, int64_t ldp, const void *wp,
                             const float *bias, int H, int M, int O, int D, float *out, float *pooled, int64_t B,
                             rp_stream_t stream) {
    RP_REQUIRE(x0 && xp && wp && (out || pooled) && B >= 0, "cin_bs_fwd: null pointer");
    int rc = cb_check(H, M, O, D);
    if (rc != RP_OK) return rc;
    RP_REQUIRE(ld0 >= (int64_t)H * D && ldp >= (int64_t)M * D, "cin_bs_fwd: ld too small");
    if (B == 0) return RP_OK;
    cb_launch<0>(xp, ldp, M, x0, ld0, H, wp, O, D, bias, out, pooled, nullptr, nullptr, nullptr, 0, B, (hipStream_t)stream);
    RP_LAUNCH_CHECK("cin_bs_fwd");
    return RP_OK;
}

// dx[b, r, d] = sum_o (gout[b,o,d] + gpool[b,o]) * sum_c Wp[o][r][c] * xk[b, c, d]     (r < R rows, c < C contraction)
//   X_0-role gradient : xk = X_{k-1}, (R, C) = (H, M), wp from W[o,h,m]
//   X_{k-1}-role grad : xk = X_0,     (R, C) = (M, H), wp from W[o,m,h] (transposed)
//   first layer       : xk = X_0,     (R, C) = (H, H), wp from W[o,h,m] + W[o,m,h]: the whole gradient in one pass
extern "C" int rp_cin_bs_bwd_x(const float *xk, int64_t ldk, const void *wp, const float *gout, const float *gpool, int R,
                               int Cn, int O, int D, float *dx, int64_t lddx, int64_t B, rp_stream_t stream) {
    RP_REQUIRE(xk && wp && dx && (gout || gpool) && B >= 0, "cin_bs_bwd_x: null pointer");
    int rc = cb_check(R, Cn, O, D);
    if (rc != RP_OK) return rc;
    RP_REQUIRE(ldk >= (int64_t)Cn * D && lddx >= (int64_t)R * D, "cin_bs_bwd_x: ld too small");
    if (B == 0) return RP_OK;
    cb_launch<1>(xk, ldk, Cn, nullptr, 0, R, wp, O, D, nullptr, nullptr, nullptr, gout, gpool, dx, lddx, B,
                 (hipStream_t)stream);
    RP_LAUNCH_CHECK("cin_bs_bwd_x");
    return RP_OK;
}

// ------------------------------------------------------------------------------------------------ weight gradient
// dW[o,h,m] = sum_{b,d} G[b,o,d] X_0[b,h,d] X_{k-1}[b,m,d],  db[o] = sum_{b,d} G[b,o,d]        (G = gout + gpool)
// A TN GEMM per channel (32 x 32 output tile, contraction index r = (b,d)) on the bf16 matrix core — with NO LDS and
// NO barriers: every operand is "contraction-contiguous" in memory, so a lane can load its MFMA fragment directly:
//   A fragment, lane (row h, k-octet)  = G[b,o,d0..d0+7] * X_0[b,h,d0..d0+7]   (two 32-byte reads, the G one broadcast)
//   B fragment, lane (col m, k-octet)  = X_{k-1}[b,m,d0..d0+7]                 (the X_0 octet itself in a first layer)
// Each wave owns CW_CH = 4 channels (4 accumulator tiles) and walks over its workgroup's chunk of samples; the four
// waves of a workgroup take four channel groups of the SAME samples, so their X_0 reads share the L1.  Partials per
// sample chunk go to a workspace; a second kernel sums them in a fixed order (deterministic) and strips the padding.
#define CW_CH 4
__global__ __launch_bounds__(256) void cin_bs_bwd_w_kernel(const float *__restrict__ x0, int64_t ld0,
                                                           const float *__restrict__ xp, int64_t ldp, int H, int M,
                                                           int O, int D, const float *__restrict__ gout,
                                                           const float *__restrict__ gpool, float *__restrict__ P,
                                                           float *__restrict__ Pb, int64_t B, int64_t b_per_blk) {
    const int t = threadIdx.x;
    const int w = t >> 6, l = t & 63, i = l & 31, hh = l >> 5;
    const int o0 = (blockIdx.y * 4 + w) * CW_CH;  // this wave's channels
    const int64_t bbeg = (int64_t)blockIdx.x * b_per_blk;
    int64_t bend = bbeg + b_per_blk;
    if (bend > B) bend = B;
    const bool same = (xp == x0) && (ldp == ld0);
    f32x16 acc[CW_CH];
#pragma unroll
    for (int u = 0; u < CW_CH; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[u][r] = 0.f;
    float bsum[CW_CH];
#pragma unroll
    for (int u = 0; u < CW_CH; ++u) bsum[u] = 0.f;
    const int nks = D / 16;  // k-steps per sample
    if (o0 < O) {
        for (int64_t b = bbeg; b < bend; ++b) {
            for (int ks = 0; ks < nks; ++ks) {
                const int d0 = ks * 16 + 8 * hh;
                cbf8 vx, vp;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    vx[e] = 0.f;
                    vp[e] = 0.f;
                }
                if (i < H) {
                    const f32x4 q0 = *reinterpret_cast<const f32x4 *>(x0 + b * ld0 + (int64_t)i * D + d0);
                    const f32x4 q1 = *reinterpret_cast<const f32x4 *>(x0 + b * ld0 + (int64_t)i * D + d0 + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        vx[e] = q0[e];
                        vx[4 + e] = q1[e];
                    }
                }
                if (same) {
                    vp = vx;
                } else if (i < M) {
                    const f32x4 q0 = *reinterpret_cast<const f32x4 *>(xp + b * ldp + (int64_t)i * D + d0);
                    const f32x4 q1 = *reinterpret_cast<const f32x4 *>(xp + b * ldp + (int64_t)i * D + d0 + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        vp[e] = q0[e];
                        vp[4 + e] = q1[e];
                    }
                }
                cbbf8 bq[3];
                cb_split(vp, bq);
#pragma unroll
                for (int u = 0; u < CW_CH; ++u) {
                    int o = o0 + u;
                    if (o >= O) o = O - 1;  // a channel past the end recomputes the last one; its tile is never stored
                    cbf8 vg;
#pragma unroll
                    for (int e = 0; e < 8; ++e) vg[e] = 0.f;
                    if (gout != nullptr) {
                        const f32x4 q0 = *reinterpret_cast<const f32x4 *>(gout + (b * O + o) * D + d0);
                        const f32x4 q1 = *reinterpret_cast<const f32x4 *>(gout + (b * O + o) * D + d0 + 4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            vg[e] = q0[e];
                            vg[4 + e] = q1[e];
                        }
                    }
                    if (gpool != nullptr) {
                        const float gp = gpool[b * O + o];
#pragma unroll
                        for (int e = 0; e < 8; ++e) vg[e] += gp;
                    }
                    bsum[u] += ((vg[0] + vg[1]) + (vg[2] + vg[3])) + ((vg[4] + vg[5]) + (vg[6] + vg[7]));
                    cbbf8 a[3];
                    cb_split(vg * vx, a);
                    acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], bq[2], acc[u], 0, 0, 0);
                    acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], bq[0], acc[u], 0, 0, 0);
                    acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], bq[1], acc[u], 0, 0, 0);
                    acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], bq[1], acc[u], 0, 0, 0);
                    acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], bq[0], acc[u], 0, 0, 0);
                    acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], bq[0], acc[u], 0, 0, 0);
                }
            }
        }
    }
    // partials: P[chunk][o][h][m] (32 x 32 padded); C layout: col (m) = lane & 31, row (h) = (r&3) + 8*(r>>2) + 4*hh
#pragma unroll
    for (int u = 0; u < CW_CH; ++u) {
        const int o = o0 + u;
        if (o >= O) continue;
        float *Pz = P + ((int64_t)blockIdx.x * O + o) * 1024;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int h = (r & 3) + 8 * (r >> 2) + 4 * hh;
            Pz[h * 32 + i] = acc[u][r];
        }
        if (Pb != nullptr) {
            // every lane of a half summed the same G octets: lane 0 holds the octets with 8*0, lane 32 those with 8*1
            const float other = __shfl_xor(bsum[u], 32, 64);
            if (l == 0) Pb[(int64_t)blockIdx.x * O + o] = bsum[u] + other;
        }
    }
}

// dW[o, h*M + m] = sum_chunk P[chunk][o][h][m];  db[o] = sum_chunk Pb[chunk][o]   (16 x 16 threads, fixed order)
__global__ __launch_bounds__(256) void cin_bs_wsum_kernel(const float *__restrict__ P, const float *__restrict__ Pb, int S,
                                                          int O, int H, int M, float *__restrict__ dW,
                                                          float *__restrict__ db) {
    __shared__ float red[16][17];
    const int c = threadIdx.x & 15, q = threadIdx.x >> 4;
    const int64_t nw = (int64_t)O * H * M;
    const int64_t e = (int64_t)blockIdx.x * 16 + c;
    const bool is_w = e < nw, is_b = !is_w && e < nw + O && db != nullptr;
    float s = 0.f;
    if (is_w) {
        const int o = (int)(e / (H * M)), rem = (int)(e - (int64_t)o * H * M), h = rem / M, m = rem - h * M;
        const int64_t src = (int64_t)o * 1024 + h * 32 + m;
        for (int z = q; z < S; z += 16) s += P[(int64_t)z * O * 1024 + src];
    } else if (is_b) {
        const int o = (int)(e - nw);
        for (int z = q; z < S; z += 16) s += Pb[(int64_t)z * O + o];
    }
    red[q][c] = s;
    __syncthreads();
    if (q != 0) return;
#pragma unroll
    for (int j = 1; j < 16; ++j) s += red[j][c];
    if (is_w) dW[e] = s;
    else if (is_b) db[e - nw] = s;
}

static int64_t cw_chunks(int64_t B) {
    int64_t n = B < 256 ? B : 256;
    return n < 1 ? 1 : n;
}

extern "C" int rp_cin_bs_bwd_w_workspace_bytes(int64_t B, int O, size_t *bytes) {
    RP_REQUIRE(bytes && B >= 0 && O >= 1, "cin_bs_bwd_w_workspace_bytes: bad argument");
    *bytes = (size_t)cw_chunks(B) * O * (1024 + 1) * sizeof(float) + 256;
    return RP_OK;
}

extern "C" int rp_cin_bs_bwd_w(const float *x0, int64_t ld0, const float *xp, int64_t ldp, const float *gout,
                               const float *gpool, int H, int M, int O, int D, float *dW, float *db, int64_t B,
                               void *workspace, size_t workspace_bytes, rp_stream_t stream) {
    RP_REQUIRE(x0 && xp && dW && workspace && (gout || gpool) && B >= 1, "cin_bs_bwd_w: bad argument");
    int rc = cb_check(H, M, O, D);
    if (rc != RP_OK) return rc;
    RP_REQUIRE(ld0 >= (int64_t)H * D && ldp >= (int64_t)M * D && ld0 % 4 == 0 && ldp % 4 == 0 && rp_aligned16(x0) &&
                   rp_aligned16(xp) && (gout == nullptr || rp_aligned16(gout)),
               "cin_bs_bwd_w: rows must be 16-byte aligned (ld multiple of 4 floats)");
    size_t need = 0;
    rp_cin_bs_bwd_w_workspace_bytes(B, O, &need);
    RP_REQUIRE(workspace_bytes >= need, "cin_bs_bwd_w: workspace %zu < %zu", workspace_bytes, need);
    float *P = reinterpret_cast<float *>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
    const int64_t nc = cw_chunks(B);
    const int64_t per = rp_cdiv(B, nc);
    const int64_t ncx = rp_cdiv(B, per);
    float *Pb = P + (size_t)ncx * O * 1024;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(cin_bs_bwd_w_kernel, dim3((unsigned)ncx, (unsigned)rp_cdiv(O, 4 * CW_CH)), dim3(256), 0, s, x0, ld0, xp, ldp,
                       H, M, O, D, gout, gpool, P, db ? Pb : nullptr, B, per);
    RP_LAUNCH_CHECK("cin_bs_bwd_w");
    const int64_t total = (int64_t)O * H * M + O;
    hipLaunchKernelGGL(cin_bs_wsum_kernel, dim3((unsigned)rp_cdiv(total, 16)), dim3(256), 0, s, P, Pb, (int)ncx, O, H, M, dW, db);
    RP_LAUNCH_CHECK("cin_bs_bwd_w reduce");
    return RP_OK;
}

// ------------------------------------------------------------------------------------------------ weight gradient, pair form
// First layer only (X_{k-1} = X_0):  dW[o,h,m] = sum_r G[o,r] X_0[h,r] X_0[m,r],  r = (b,d), is symmetric in (h,m).
// Form the products once, not once per channel:
//     Pr[p, r] = X_0[h_p, r] * X_0[m_p, r]      for the NPAIR = H(H+1)/2 pairs p = (h_p <= m_p)
//     dWs[o, p] = sum_r G[o, r] Pr[p, r]         one TN GEMM, M = O rows, N = NPAIR columns, contraction r
// The A operand (G) needs no scaling at all (split only, shared by every pair tile), the B operand is formed once per
// workgroup stage and shared by all O rows, and the matrix core does 2.9x fewer passes than the per-channel 32 x 32
// form (no padding of 26 -> 32 in either dimension, half the pairs).  dW[o,h,m] = dW[o,m,h] = dWs[o, p(h,m)].
// Workgroup tile 128 (o) x 128 (pairs), EIGHT waves 4 x 2, each 32 x 64 (2 MFMA tiles; see cin_pair_fwd_kernel); stage = 32 contraction rows = a
// sample's d-half: X_0's half-sample goes through LDS once (fp32), the pair products are formed from it.
// The next stage's loads are issued before this stage's MFMA block.  (Loading G 8 threads per 128-byte row, whole lines
// per wave instruction, measured SLOWER than one 32-byte octet per thread: twice the LDS store instructions.)
#define CP_LD 40
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// products of the split-bf16 evaluation, smallest terms first (as BfProd in gemm.hip): piece 0 = hi, 1 = mid, 2 = lo
template <int NPROD>
struct CpProd {
    __device__ static constexpr int pa(int i) { return NPROD == 6 ? (i == 1 ? 2 : ((i == 2 || i == 4) ? 1 : 0))
                                                     : NPROD == 3 ? (i == 1 ? 1 : 0) : 0; }
    __device__ static constexpr int pb(int i) { return NPROD == 6 ? (i == 0 ? 2 : ((i == 2 || i == 3) ? 1 : 0))
                                                     : NPROD == 3 ? (i == 0 ? 1 : 0) : 0; }
};
template <int NPROD>
__global__ __launch_bounds__(512, 4) void cin_pair_bwd_w_kernel(const float *__restrict__ x0, int64_t ld0, int H, int O,
                                                                int D, int npair, const float *__restrict__ gout,
                                                                const float *__restrict__ gpool,
                                                                float *__restrict__ P, float *__restrict__ Pb, int64_t B,
                                                                int64_t b_per_blk) {
    constexpr int NPC = NPROD == 6 ? 3 : (NPROD == 3 ? 2 : 1);  // bf16 pieces per operand that are used
    __shared__ __attribute__((aligned(16))) __bf16 At[3][128][CP_LD];  // G[o][r]
    __shared__ __attribute__((aligned(16))) __bf16 Bt[3][128][CP_LD];  // Pr[p][r]
    __shared__ __attribute__((aligned(16))) float Xs[32][36];           // X_0[h][r] of the stage (fp32)
    __shared__ unsigned char ph[128], pm[128];                          // this tile's pairs
    const int t = threadIdx.x;
    const int w = t >> 6, l = t & 63, i = l & 31, hh = l >> 5;
    const int wa = (w & 3) * 32, wb = (w >> 2) * 64;  // 8 waves, 4 (rows o) x 2 (pair columns), each 32 x 64
    const int c = t & 127, oct = t >> 7;               // A image: row c; B image: pair c; contraction octet oct
    const int gr = (t >> 3) & 31, gs = t & 7;          // X_0 stage (threads 0..255): row gr, floats 4 gs .. 4 gs + 3
    const int p0 = blockIdx.y * 128;
    const int64_t bbeg = (int64_t)blockIdx.x * b_per_blk;
    int64_t bend = bbeg + b_per_blk;
    if (bend > B) bend = B;
    if (t < 128) {  // pair p -> (h, m), h <= m, row-major over the upper triangle
        int p = p0 + t, h = 0;
        if (p >= npair) p = npair - 1;  // padding pairs repeat the last one; their columns are never stored
        int rem = p;
        while (rem >= H - h) {
            rem -= H - h;
            ++h;
        }
        ph[t] = (unsigned char)h;
        pm[t] = (unsigned char)(h + rem);
    }
    f32x16 acc[2];
#pragma unroll
    for (int v = 0; v < 2; ++v)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[v][r] = 0.f;
    float bsum = 0.f;
    const int nhalf = D / 32;
    const bool do_bias = (Pb != nullptr) && (blockIdx.y == 0);
    __syncthreads();
    const int h0 = ph[c], m0_ = pm[c];
    const int64_t nstage = (bend - bbeg) * nhalf;
    f32x4 gq[2], xq;
    float gpq;
    int64_t lb = bbeg;  // (sample, d-half) of the next stage to load
    int lh = 0;
    auto load_stage = [&]() {
        const int64_t b = lb;
        const int d0 = lh * 32;
        if (++lh == nhalf) {
            lh = 0;
            ++lb;
        }
        gq[0] = gq[1] = f32x4{0.f, 0.f, 0.f, 0.f};
        gpq = 0.f;
        if (c < O) {  // row c, contraction octet oct
            if (gout != nullptr) {
                gq[0] = *reinterpret_cast<const f32x4 *>(gout + (b * O + c) * D + d0 + 8 * oct);
                gq[1] = *reinterpret_cast<const f32x4 *>(gout + (b * O + c) * D + d0 + 8 * oct + 4);
            }
            if (gpool != nullptr) gpq = gpool[b * O + c];
        }
        xq = f32x4{0.f, 0.f, 0.f, 0.f};
        if (t < 256 && gr < H) xq = *reinterpret_cast<const f32x4 *>(x0 + b * ld0 + (int64_t)gr * D + d0 + 4 * gs);
    };
    if (nstage > 0) load_stage();
    for (int64_t st = 0; st < nstage; ++st) {
        cbf8 vg;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            vg[e] = gq[0][e] + gpq;
            vg[4 + e] = gq[1][e] + gpq;
        }
        if (do_bias) bsum += ((vg[0] + vg[1]) + (vg[2] + vg[3])) + ((vg[4] + vg[5]) + (vg[6] + vg[7]));
        __syncthreads();  // previous stage's fragment reads (and Xs reads) are done
        if (t < 256) *reinterpret_cast<f32x4 *>(&Xs[gr][4 * gs]) = xq;
        cbbf8 pc[3];
        cb_split(vg, pc);
#pragma unroll
        for (int q = 0; q < NPC; ++q) *reinterpret_cast<cbbf8 *>(&At[q][c][8 * oct]) = pc[q];
        __syncthreads();  // Xs complete
        {
            cbf8 a0, b0;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                a0[e] = Xs[h0][8 * oct + e];
                b0[e] = Xs[m0_][8 * oct + e];
            }
            cb_split(a0 * b0, pc);
#pragma unroll
            for (int q = 0; q < NPC; ++q) *reinterpret_cast<cbbf8 *>(&Bt[q][c][8 * oct]) = pc[q];
        }
        if (st + 1 < nstage) load_stage();
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            cbbf8 a[3], bq[2][3];
#pragma unroll
            for (int q = 0; q < NPC; ++q) {
                a[q] = *reinterpret_cast<const cbbf8 *>(&At[q][wa + i][ks * 16 + 8 * hh]);
#pragma unroll
                for (int u = 0; u < 2; ++u)
                    bq[u][q] = *reinterpret_cast<const cbbf8 *>(&Bt[q][wb + 32 * u + i][ks * 16 + 8 * hh]);
            }
            // product-major: consecutive MFMAs alternate between the two accumulators (smallest terms first)
#pragma unroll
            for (int pr = 0; pr < NPROD; ++pr) {
                const int qa = CpProd<NPROD>::pa(pr);
                const int qb = CpProd<NPROD>::pb(pr);
#pragma unroll
                for (int v = 0; v < 2; ++v)
                    acc[v] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[qa], bq[v][qb], acc[v], 0, 0, 0);
            }
        }
    }
    // partials P[chunk][o][pair]; C layout: col (pair) = lane & 31, row (o) = (r&3) + 8*(r>>2) + 4*hh
    float *Pz = P + (int64_t)blockIdx.x * O * npair;
#pragma unroll
    for (int v = 0; v < 2; ++v) {
        const int p = p0 + wb + 32 * v + i;
        if (p >= npair) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int o = wa + (r & 3) + 8 * (r >> 2) + 4 * hh;
            if (o < O) Pz[(int64_t)o * npair + p] = acc[v][r];
        }
    }
    if (do_bias) {  // row c: its four octet owners hold the partial sums
        __syncthreads();
        float *bred = reinterpret_cast<float *>(&At[0][0][0]);  // [4][128]
        bred[oct * 128 + c] = bsum;
        __syncthreads();
        if (t < 128 && t < O) Pb[(int64_t)blockIdx.x * O + t] = (bred[t] + bred[128 + t]) + (bred[256 + t] + bred[384 + t]);
    }
}

// dW[o, h*H + m] = dW[o, m*H + h] = sum_chunk P[chunk][o][pair(h,m)];  db[o] = sum_chunk Pb[chunk][o]
__global__ __launch_bounds__(256) void cin_pair_wsum_kernel(const float *__restrict__ P, const float *__restrict__ Pb,
                                                            int S, int O, int H, int npair, float *__restrict__ dW,
                                                            float *__restrict__ db) {
    __shared__ float red[16][17];
    const int c = threadIdx.x & 15, q = threadIdx.x >> 4;
    const int64_t nw = (int64_t)O * H * H;
    const int64_t e = (int64_t)blockIdx.x * 16 + c;
    const bool is_w = e < nw, is_b = !is_w && e < nw + O && db != nullptr;
    float s = 0.f;
    if (is_w) {
        const int o = (int)(e / (H * H)), rem = (int)(e - (int64_t)o * H * H);
        int h = rem / H, m = rem - h * H;
        if (h > m) {
            const int tmp = h;
            h = m;
            m = tmp;
        }
        const int p = h * H - h * (h - 1) / 2 + (m - h);  // row-major upper triangle
        const int64_t src = (int64_t)o * npair + p;
        for (int z = q; z < S; z += 16) s += P[(int64_t)z * O * npair + src];
    } else if (is_b) {
        const int o = (int)(e - nw);
        for (int z = q; z < S; z += 16) s += Pb[(int64_t)z * O + o];
    }
    red[q][c] = s;
    __syncthreads();
    if (q != 0) return;
#pragma unroll
    for (int j = 1; j < 16; ++j) s += red[j][c];
    if (is_w) dW[e] = s;
    else if (is_b) db[e - nw] = s;
}

static int64_t cp_chunks(int64_t B, int ntile) {
    int64_t n = 512 / ntile;  // <= 512 workgroups = one resident round (2 per CU); one more would double the time
    if (n > B) n = B;
    return n < 1 ? 1 : n;
}

extern "C" int rp_cin_pair_bwd_w_workspace_bytes(int64_t B, int H, int O, size_t *bytes) {
    RP_REQUIRE(bytes && B >= 0 && H >= 1 && O >= 1, "cin_pair_bwd_w_workspace_bytes: bad argument");
    const int npair = H * (H + 1) / 2;
    *bytes = (size_t)cp_chunks(B, (int)rp_cdiv(npair, 128)) * O * (npair + 1) * sizeof(float) + 256;
    return RP_OK;
}

// first-layer weight gradient in the pair form; O <= 128
extern "C" int rp_cin_pair_bwd_w(const float *x0, int64_t ld0, const float *gout, const float *gpool, int H, int O, int D,
                                 float *dW, float *db, int64_t B, void *workspace, size_t workspace_bytes,
                                 rp_stream_t stream) {
    RP_REQUIRE(x0 && dW && workspace && (gout || gpool) && B >= 1, "cin_pair_bwd_w: bad argument");
    if (H < 1 || H > 32 || O < 1 || O > 128 || (D != 32 && D != 64))
        return rp_fail(RP_ERR_UNSUPPORTED, "cin_pair_bwd_w: H=%d (<=32) O=%d (<=128) D=%d (32|64) unsupported", H, O, D);
    RP_REQUIRE(ld0 >= (int64_t)H * D && ld0 % 4 == 0 && rp_aligned16(x0) && (gout == nullptr || rp_aligned16(gout)),
               "cin_pair_bwd_w: rows must be 16-byte aligned (ld multiple of 4 floats)");
    size_t need = 0;
    rp_cin_pair_bwd_w_workspace_bytes(B, H, O, &need);
    RP_REQUIRE(workspace_bytes >= need, "cin_pair_bwd_w: workspace %zu < %zu", workspace_bytes, need);
    const int npair = H * (H + 1) / 2;
    const int ntile = (int)rp_cdiv(npair, 128);
    float *P = reinterpret_cast<float *>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
    const int64_t nc = cp_chunks(B, ntile);
    const int64_t per = rp_cdiv(B, nc);
    const int64_t ncx = rp_cdiv(B, per);
    float *Pb = P + (size_t)ncx * O * npair;
    hipStream_t s = (hipStream_t)stream;
    const int nprod = rp_matmul_products(2.0 * O * npair * D * (double)B, 4.0 * (double)B * D * (O + H));
#define CPW(NP_)                                                                                                         \
    hipLaunchKernelGGL((cin_pair_bwd_w_kernel<NP_>), dim3((unsigned)ncx, (unsigned)ntile), dim3(512), 0, s, x0, ld0, H, O, D, \
                       npair, gout, gpool, P, db ? Pb : nullptr, B, per)
    if (nprod == 6) CPW(6);
    else if (nprod == 3) CPW(3);
    else CPW(1);
#undef CPW
    RP_LAUNCH_CHECK("cin_pair_bwd_w");
    const int64_t total = (int64_t)O * H * H + O;
    hipLaunchKernelGGL(cin_pair_wsum_kernel, dim3((unsigned)rp_cdiv(total, 16)), dim3(256), 0, s, P, Pb, (int)ncx, O, H, npair, dW,
                       db);
    RP_LAUNCH_CHECK("cin_pair_bwd_w reduce");
    return RP_OK;
}

// ------------------------------------------------------------------------------------------------ forward, pair form
// First layer (X_{k-1} = X_0):  X_1[b,o,d] = bias[o] + sum_p Ws[o,p] Pr[p,(b,d)],   Pr[p,(b,d)] = X_0[b,h_p,d] X_0[b,m_p,d],
// Ws[o,(h,m)] = W[o,h,m] + W[o,m,h] (h < m), W[o,h,h] — one NN GEMM with M = O <= 128 rows, K = NPAIR, N = B*D columns.
// wsp: Ws as bf16 pieces [3][128][KP] (KP = NPAIR rounded up to 32, zero padded rows/columns), contraction-contiguous.
// Workgroup: all 128 rows x 128 columns (128/D samples); X_0 of those samples sits in LDS (fp32) for the whole kernel,
// each stage (32 pairs) copies the A tile L2 -> LDS and forms the B tile from LDS X_0.  EIGHT waves, 4 (rows) x 2
// (columns), each 32 x 64: hipcc schedules a stage as "all VALU / LDS, then the MFMAs back to back", so the overlap has
// to come from other waves — four half-size waves per SIMD (120 VGPRs) beat two full-size ones (3.9 -> 3.35 ms).
template <int NPROD>
__global__ __launch_bounds__(512, 4) void cin_pair_fwd_kernel(const float *__restrict__ x0, int64_t ld0, int H, int O, int D,
                                                              int npair, int KP, const __bf16 *__restrict__ wsp,
                                                              const float *__restrict__ bias, float *__restrict__ out,
                                                              float *__restrict__ pooled, int64_t B) {
    constexpr int NPC = NPROD == 6 ? 3 : (NPROD == 3 ? 2 : 1);  // bf16 pieces per operand that are used
    __shared__ __attribute__((aligned(16))) __bf16 At[3][128][CP_LD];  // Ws[o][p]
    __shared__ __attribute__((aligned(16))) __bf16 Bt[3][128][CP_LD];  // Pr[col][p]
    __shared__ __attribute__((aligned(16))) float Xs[32][132];          // X_0[h][col]
    __shared__ __attribute__((aligned(16))) unsigned tab[544];          // pair p -> (h*132) | (m*132) << 16
    const int t = threadIdx.x;
    const int w = t >> 6, l = t & 63, i = l & 31, hh = l >> 5;
    const int wa = (w & 3) * 32, wb = (w >> 2) * 64;  // 8 waves: 4 (rows) x 2 (columns), each 32 x 64
    const int col = t & 127, oq = t >> 7;             // B image: column, pair octet oq (0..3)
    const int spb = 128 / D;  // samples per workgroup
    const int64_t bs0 = (int64_t)blockIdx.x * spb;
    for (int p = t; p < KP; p += 512) {
        int pp = p < npair ? p : 0, h = 0;  // padding pairs: any valid rows (their weights are zero)
        while (pp >= H - h) {
            pp -= H - h;
            ++h;
        }
        tab[p] = (unsigned)(h * 132) | ((unsigned)((h + pp) * 132) << 16);
    }
    {
        const int64_t b = bs0 + col / D;
        const int d = col % D;
        for (int h = oq; h < H; h += 4) Xs[h][col] = (b < B) ? x0[b * ld0 + (int64_t)h * D + d] : 0.f;
    }
    f32x16 acc[1][2];
#pragma unroll
    for (int u = 0; u < 1; ++u)
#pragma unroll
        for (int v = 0; v < 2; ++v)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[u][v][r] = 0.f;
    // A tile copy: 3 pieces x 128 rows x 4 sixteen-byte chunks = 1536 chunks, 3 per thread
    f32x4 aq[3];
    auto load_a = [&](int st) {
#pragma unroll
        for (int u = 0; u < NPC; ++u) {
            const int id = t + 512 * u;
            const int q = id >> 9, row = (id & 511) >> 2, ch = id & 3;
            aq[u] = *reinterpret_cast<const f32x4 *>(wsp + ((int64_t)(q * 128 + row) * KP + st * 32 + ch * 8));
        }
    };
    const int nst = KP / 32;
    load_a(0);
    __syncthreads();  // Xs, tab
    const float *xcol = &Xs[0][col];
    for (int st = 0; st < nst; ++st) {
        // B tile: this thread's column, pair octets 2 oq and 2 oq + 1
        cbbf8 pb[1][3];
#pragma unroll
        for (int j = 0; j < 1; ++j) {
            const int k0 = st * 32 + 8 * oq;
            const u32x4 t0 = *reinterpret_cast<const u32x4 *>(&tab[k0]);
            const u32x4 t1 = *reinterpret_cast<const u32x4 *>(&tab[k0 + 4]);
            cbf8 pr;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                pr[e] = xcol[t0[e] & 0xffffu] * xcol[t0[e] >> 16];
                pr[4 + e] = xcol[t1[e] & 0xffffu] * xcol[t1[e] >> 16];
            }
            cb_split(pr, pb[j]);
        }
        __syncthreads();  // previous stage's fragment reads are done
#pragma unroll
        for (int u = 0; u < NPC; ++u) {
            const int id = t + 512 * u;
            const int q = id >> 9, row = (id & 511) >> 2, ch = id & 3;
            *reinterpret_cast<f32x4 *>(&At[q][row][ch * 8]) = aq[u];
        }
#pragma unroll
        for (int q = 0; q < NPC; ++q) *reinterpret_cast<cbbf8 *>(&Bt[q][col][8 * oq]) = pb[0][q];
        if (st + 1 < nst) load_a(st + 1);
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            cbbf8 a[1][3], bq[2][3];
#pragma unroll
            for (int q = 0; q < NPC; ++q) {
                a[0][q] = *reinterpret_cast<const cbbf8 *>(&At[q][wa + i][ks * 16 + 8 * hh]);
#pragma unroll
                for (int u = 0; u < 2; ++u)
                    bq[u][q] = *reinterpret_cast<const cbbf8 *>(&Bt[q][wb + 32 * u + i][ks * 16 + 8 * hh]);
            }
            // product-major: consecutive MFMAs go to four different accumulators (smallest terms first per accumulator)
#pragma unroll
            for (int pr = 0; pr < NPROD; ++pr) {
                const int qa = CpProd<NPROD>::pa(pr);
                const int qb = CpProd<NPROD>::pb(pr);
#pragma unroll
                for (int u = 0; u < 1; ++u)
#pragma unroll
                    for (int v = 0; v < 2; ++v)
                        acc[u][v] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[u][qa], bq[v][qb], acc[u][v], 0, 0, 0);
            }
        }
    }
    // epilogue: C layout col = wb + 32 v + i, row o = wa + 32 u + (r&3) + 8*(r>>2) + 4*hh
#pragma unroll
    for (int u = 0; u < 1; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int o = wa + 32 * u + (r & 3) + 8 * (r >> 2) + 4 * hh;
            const float bo = (bias != nullptr && o < O) ? bias[o] : 0.f;
            float val[2], ps[2];
#pragma unroll
            for (int v = 0; v < 2; ++v) {
                val[v] = acc[u][v][r] + bo;
                const int cc = wb + 32 * v + i;
                const int64_t b = bs0 + cc / D;
                if (out != nullptr && o < O && b < B) out[(b * O + o) * D + (cc % D)] = val[v];
            }
            if (pooled == nullptr) continue;
            if (D == 64) {
                ps[0] = val[0] + val[1];
                ps[1] = 0.f;
            } else {
                ps[0] = val[0];
                ps[1] = val[1];
            }
#pragma unroll
            for (int v = 0; v < 2; ++v) {
                if (v == 1 && D == 64) break;
                float s = ps[v];
                s += __shfl_xor(s, 1);
                s += __shfl_xor(s, 2);
                s += __shfl_xor(s, 4);
                s += __shfl_xor(s, 8);
                s += __shfl_xor(s, 16);
                const int64_t b = bs0 + (wb + 32 * v) / D;
                if (i == 0 && o < O && b < B) pooled[b * O + o] = s;
            }
        }
}

extern "C" int rp_cin_pair_fits(int H, int O, int D) { return H >= 1 && H <= 32 && O >= 1 && O <= 128 && (D == 32 || D == 64); }

// wsp: bf16 [3][128][KP], KP = 32 * ceil(H(H+1)/2 / 32)
extern "C" int rp_cin_pair_fwd(const float *x0, int64_t ld0, const void *wsp, const float *bias, int H, int O, int D, float *out,
                               float *pooled, int64_t B, rp_stream_t stream) {
    RP_REQUIRE(x0 && wsp && (out || pooled) && B >= 0, "cin_pair_fwd: bad argument");
    if (!rp_cin_pair_fits(H, O, D))
        return rp_fail(RP_ERR_UNSUPPORTED, "cin_pair_fwd: H=%d (<=32) O=%d (<=128) D=%d (32|64) unsupported", H, O, D);
    RP_REQUIRE(ld0 >= (int64_t)H * D && rp_aligned16(wsp), "cin_pair_fwd: bad leading dimension / alignment");
    if (B == 0) return RP_OK;
    const int npair = H * (H + 1) / 2;
    const int KP = (int)rp_cdiv(npair, 32) * 32;
    const int64_t nblk = rp_cdiv(B * D, 128);
    const int nprod = rp_matmul_products(2.0 * O * npair * D * (double)B, 4.0 * (double)B * D * (O + H));
#define CPF(NP_)                                                                                                          \
    hipLaunchKernelGGL((cin_pair_fwd_kernel<NP_>), dim3((unsigned)nblk), dim3(512), 0, (hipStream_t)stream, x0, ld0, H, O, D, \
                       npair, KP, reinterpret_cast<const __bf16 *>(wsp), bias, out, pooled, B)
    if (nprod == 6) CPF(6);
    else if (nprod == 3) CPF(3);
    else CPF(1);
#undef CPF
    RP_LAUNCH_CHECK("cin_pair_fwd");
    return RP_OK;
}

// ------------------------------------------------------------------------------------------------ input gradient, pair form
// First layer (both roles are X_0):
//     T[p,(b,d)]   = sum_o Ws[o,p] G[o,(b,d)]                        one GEMM, M = NPAIR rows, K = O, N = B*D columns
//     dX_0[b,h,d]  = sum_{p = (h,m) or (m,h)} T[p,(b,d)] X_0[b,m,d]   (the diagonal pair counts twice: 2 W[h,h] x_h)
// wst: Ws^T as bf16 pieces [3][KPT][128] (KPT = NPAIR rounded up to 128, K = o contiguous, zero padded).
// Workgroup: 128 columns (128/D samples), EIGHT waves 4 (pair rows) x 2 (columns) of 32 x 64 (see cin_pair_fwd_kernel), all
// pair tiles one after the other, K = O in 32-row stages.
//   * G is contraction-STRIDED for this GEMM (its rows are o): each stage's [32 o][128 col] slab is loaded row-wise
//     (16-byte loads), parked in LDS as fp32 and read back column-wise (conflict-free) to form the B operand.
//   * after a pair tile, T (in the accumulators) goes to LDS 64 pairs at a time and every thread GATHERS the terms of
//     its own (h, column) outputs from a host-built list  (tile, half, h) -> [(local pair row, m)]  — wave-uniform
//     scalar loads, no atomics, fixed summation order; dX_0 stays in registers until the end.
// (A first version scattered T with LDS float atomics and loaded G with strided dword loads: 21 ms.)
template <int NPROD>
__global__ __launch_bounds__(512, 4) void cin_pair_bwd_x_kernel(const float *__restrict__ x0, int64_t ld0, int H, int O,
                                                                int D, int KPT, const __bf16 *__restrict__ wst,
                                                                const float *__restrict__ gout,
                                                                const float *__restrict__ gpool, float *__restrict__ dx,
                                                                int64_t lddx, int64_t B, const int *__restrict__ lstart,
                                                                const int *__restrict__ lent, int accumulate) {
    constexpr int NPC = NPROD == 6 ? 3 : (NPROD == 3 ? 2 : 1);  // bf16 pieces per operand that are used
    __shared__ __attribute__((aligned(16))) char smem[2 * 3 * 128 * CP_LD * 2 + 32 * 132 * 4];
    typedef __bf16(*Tile)[128][CP_LD];
    Tile At = reinterpret_cast<Tile>(smem);                                   // Ws^T[p][o]
    Tile Bt = reinterpret_cast<Tile>(smem + 3 * 128 * CP_LD * 2);            // G[col][o]
    float *Gs = reinterpret_cast<float *>(smem + 2 * 3 * 128 * CP_LD * 2);   // G[o][col] fp32, [32][132]
    float *Ts = reinterpret_cast<float *>(smem);                              // epilogue: T[64 pairs][132]
    float *Xe = reinterpret_cast<float *>(smem + 64 * 132 * 4);               // epilogue: X_0[32][132]
    const int t = threadIdx.x;
    const int w = t >> 6, l = t & 63, i = l & 31, hh = l >> 5;
    const int wr = w & 3;                             // 8 waves, 4 (pair rows) x 2 (columns), each 32 x 64
    const int wa = wr * 32, wb = (w >> 2) * 64;
    const int col = t & 127, oq = __builtin_amdgcn_readfirstlane(t >> 7);  // oq 0..3
    const int go = (t >> 3) & 31, c16 = t & 7, gj = (t >> 8) * 2;  // G slab: row go, columns 16 c16 + 4 (gj + {0,1})
    const int spb = 128 / D;
    const int64_t bs0 = (int64_t)blockIdx.x * spb;
    const int64_t bcol = bs0 + col / D;  // this thread's sample in the column role
    const int dcol = col % D;
    const bool colok = bcol < B;
    const int64_t bg = bs0 + (16 * c16) / D;  // ... and in the slab-loading role
    const int dg = (16 * c16) % D;
    const bool gok = bg < B;
    const int nks = (O + 31) / 32, ntile = KPT / 128, nst = nks * ntile;
    f32x4 aq[3], gq[2];
    float gp;
    auto load_stage = [&](int st) {
        const int tile = st / nks, ks = st - tile * nks;
#pragma unroll
        for (int u = 0; u < NPC; ++u) {
            const int id = t + 512 * u;
            const int q = id >> 9, row = (id & 511) >> 2, ch = id & 3;
            aq[u] = *reinterpret_cast<const f32x4 *>(wst + ((int64_t)(q * KPT + tile * 128 + row) * 128 + ks * 32 + ch * 8));
        }
        const int o = ks * 32 + go;
        const bool ok = gok && o < O;
        gp = (ok && gpool != nullptr) ? gpool[bg * O + o] : 0.f;
#pragma unroll
        for (int j = 0; j < 2; ++j)
            gq[j] = (ok && gout != nullptr) ? *reinterpret_cast<const f32x4 *>(gout + (bg * O + o) * D + dg + 4 * (gj + j))
                                            : f32x4{0.f, 0.f, 0.f, 0.f};
    };
    float dacc[8];
#pragma unroll
    for (int jh = 0; jh < 8; ++jh) dacc[jh] = 0.f;
    f32x16 acc[2];
    load_stage(0);
    for (int st = 0; st < nst; ++st) {
        const int tile = st / nks, ks_ = st - tile * nks;
        if (ks_ == 0) {
#pragma unroll
            for (int v = 0; v < 2; ++v)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[v][r] = 0.f;
        }
        __syncthreads();  // previous stage's fragment reads / previous tile's epilogue are done
#pragma unroll
        for (int j = 0; j < 2; ++j) *reinterpret_cast<f32x4 *>(&Gs[go * 132 + 16 * c16 + 4 * (gj + j)]) = gq[j] + gp;
#pragma unroll
        for (int u = 0; u < NPC; ++u) {
            const int id = t + 512 * u;
            const int q = id >> 9, row = (id & 511) >> 2, ch = id & 3;
            *reinterpret_cast<f32x4 *>(&At[q][row][ch * 8]) = aq[u];
        }
        __syncthreads();  // the slab is complete
        {
            cbf8 gv;
#pragma unroll
            for (int e = 0; e < 8; ++e) gv[e] = Gs[(8 * oq + e) * 132 + col];
            cbbf8 pc[3];
            cb_split(gv, pc);
#pragma unroll
            for (int q = 0; q < NPC; ++q) *reinterpret_cast<cbbf8 *>(&Bt[q][col][8 * oq]) = pc[q];
        }
        if (st + 1 < nst) load_stage(st + 1);
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            cbbf8 a[3], bq[2][3];
#pragma unroll
            for (int q = 0; q < NPC; ++q) {
                a[q] = *reinterpret_cast<const cbbf8 *>(&At[q][wa + i][ks * 16 + 8 * hh]);
#pragma unroll
                for (int u = 0; u < 2; ++u)
                    bq[u][q] = *reinterpret_cast<const cbbf8 *>(&Bt[q][wb + 32 * u + i][ks * 16 + 8 * hh]);
            }
#pragma unroll
            for (int pr = 0; pr < NPROD; ++pr) {
                const int qa = CpProd<NPROD>::pa(pr);
                const int qb = CpProd<NPROD>::pb(pr);
#pragma unroll
                for (int v = 0; v < 2; ++v)
                    acc[v] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[qa], bq[v][qb], acc[v], 0, 0, 0);
            }
        }
        if (ks_ != nks - 1) continue;
        // ---- this pair tile's T is complete: fold it into the dX_0 registers, 64 pairs at a time
        __syncthreads();  // At / Bt are free
        for (int h = oq; h < H; h += 4) Xe[h * 132 + col] = colok ? x0[bcol * ld0 + (int64_t)h * D + dcol] : 0.f;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            if ((wr >> 1) == half) {
#pragma unroll
                for (int v = 0; v < 2; ++v)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        Ts[(32 * (wr & 1) + (r & 3) + 8 * (r >> 2) + 4 * hh) * 132 + wb + 32 * v + i] = acc[v][r];
            }
            __syncthreads();
            const int lbase = (tile * 2 + half) * H;
#pragma unroll
            for (int jh = 0; jh < 8; ++jh) {
                const int h = oq + 4 * jh;
                if (h < H) {
                    const int e0 = lstart[lbase + h], e1 = lstart[lbase + h + 1];
                    float a_ = dacc[jh];
                    int e = e0;
                    for (; e + 4 <= e1; e += 4) {  // four entries' scalar loads and LDS reads in flight together
                        int en[4];
                        float tv[4], xv[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) en[u] = lent[e + u];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            tv[u] = Ts[(en[u] & 255) * 132 + col];
                            xv[u] = Xe[(en[u] >> 8) * 132 + col];
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u) a_ = __builtin_fmaf(tv[u], xv[u], a_);
                    }
                    for (; e < e1; ++e) {
                        const int en = lent[e];
                        a_ = __builtin_fmaf(Ts[(en & 255) * 132 + col], Xe[(en >> 8) * 132 + col], a_);
                    }
                    dacc[jh] = a_;
                }
            }
            __syncthreads();
        }
    }
    if (colok) {
#pragma unroll
        for (int jh = 0; jh < 8; ++jh) {
            const int h = oq + 4 * jh;
            if (h < H) {
                float *dst = dx + bcol * lddx + (int64_t)h * D + dcol;
                *dst = accumulate ? *dst + dacc[jh] : dacc[jh];
            }
        }
    }
}

// wst: bf16 [3][KPT][128], KPT = 128 * ceil(H(H+1)/2 / 128).  lstart [2*(KPT/128)*H + 1] / lent: for pair tile t, half
// c (64 pairs) and field h, entries lstart[(2t+c)*H + h] .. lstart[(2t+c)*H + h + 1] of lent list the pairs of that
// half that contain h as  (local pair row 0..63) | (other field m) << 8  (the diagonal pair (h,h) is listed twice).
// dx rows [B, lddx]: the first H*D floats of each row are written.
extern "C" int rp_cin_pair_bwd_x(const float *x0, int64_t ld0, const void *wst, const float *gout, const float *gpool,
                                 const int32_t *lstart, const int32_t *lent, int H, int O, int D, float *dx, int64_t lddx,
                                 int64_t B, int accumulate, rp_stream_t stream) {
    RP_REQUIRE(x0 && wst && dx && lstart && lent && (gout || gpool) && B >= 0, "cin_pair_bwd_x: bad argument");
    if (!rp_cin_pair_fits(H, O, D))
        return rp_fail(RP_ERR_UNSUPPORTED, "cin_pair_bwd_x: H=%d (<=32) O=%d (<=128) D=%d (32|64) unsupported", H, O, D);
    RP_REQUIRE(ld0 >= (int64_t)H * D && lddx >= (int64_t)H * D && rp_aligned16(wst) && (gout == nullptr || rp_aligned16(gout)),
               "cin_pair_bwd_x: bad leading dimension / alignment");
    if (B == 0) return RP_OK;
    const int npair = H * (H + 1) / 2;
    const int KPT = (int)rp_cdiv(npair, 128) * 128;
    const int64_t nblk = rp_cdiv(B * D, 128);
    const int nprod = rp_matmul_products(2.0 * O * npair * D * (double)B, 4.0 * (double)B * D * (O + 2 * H));
#define CPX(NP_)                                                                                                            \
    hipLaunchKernelGGL((cin_pair_bwd_x_kernel<NP_>), dim3((unsigned)nblk), dim3(512), 0, (hipStream_t)stream, x0, ld0, H, O, D, \
                       KPT, reinterpret_cast<const __bf16 *>(wst), gout, gpool, dx, lddx, B, lstart, lent, accumulate)
    if (nprod == 6) CPX(6);
    else if (nprod == 3) CPX(3);
    else CPX(1);
#undef CPX
    RP_LAUNCH_CHECK("cin_pair_bwd_x");
    return RP_OK;
}
