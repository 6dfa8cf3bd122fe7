// DCN-v1 CrossNet, all layers in one pass per direction, gfx950.
//   X_{l+1} = X_l + (X_l . w_l) * X_0 + b_l          (reference: layers/interaction.py:119-141)
// HBM-bound: the forward reads each [d]-row of X_0 once and keeps it in registers through all L layers
// (the reference re-reads and re-writes [B,d] about 5 times per layer); it writes X_L only if asked, and
// can finish with the fc dot product (DCN's `fc(cross_out)`, dcn.py:64) so only a logit leaves the chip.
// The backward recomputes X_l from X_0 and the saved per-layer scalars s_l = X_l . w_l (no [B,d]
// activations are stored), writes dX_0 once and accumulates the parameter gradients per block in
// registers; a second kernel sums the per-block partials in fixed order (deterministic).
//
// One 256-thread workgroup per row, element e of the row on thread e % 256 (coalesced 1-KiB segments);
// a row of up to 256*CROSS_J floats lives in CROSS_J registers per thread.  Dot products are a wave
// shuffle reduction + one LDS exchange between the 4 waves.
#include "common.h"
#include <cstdlib>

#define CROSS_J 8  // d <= 2048
#define CROSS_MAXL 6

__device__ __forceinline__ float block_allsum_256(float v, float *sh) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();  // sh may still be read from the previous reduction
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// Forward: one 64-lane wave per row, persistent workgroups.  W (L rows), wfc and — when they fit in 64 KB of LDS
// next to them — the biases are staged once per workgroup, zero-padded to 256-column chunks; lane l owns columns
// 256j + 4l .. +3 of chunk j (dwordx4 global accesses, conflict-free ds_read_b128).  The row stays in registers
// through all L layers; each layer is one dot product (wave shuffles, no barrier) and one axpy.
static int64_t cross_nb_max(int64_t dflt) {  // (RP_CROSS_NB: experiment knob for the persistent grid size)
    static const int64_t v = getenv("RP_CROSS_NB") ? atoll(getenv("RP_CROSS_NB")) : 0;
    return v > 0 ? v : dflt;
}

// Stage `nrows` weight rows of d floats into LDS, zero padded to dp (a multiple of 256) columns per row.  A 256-column
// chunk lies inside one row, so the row (and its source pointer) is wave-uniform; four chunks are requested per iteration
// with UNCONDITIONAL, clamped loads (the element-by-element loop this replaces issued one guarded load at a time; measured:
// no difference at Criteo shape, 0.224 / 0.265 ms forward / backward either way — the staging is not what bounds these
// kernels).  row_ptr(l) returns the source of row l (NULL = a row of zeros).
template <typename RowPtr>
__device__ __forceinline__ void cross_stage_rows(float *__restrict__ wl, int nrows, int d, int dp, RowPtr row_ptr) {
    const int cpr = dp / 256, nchunk = nrows * cpr;
    for (int c0 = 0; c0 < nchunk; c0 += 4) {
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = (c0 + u < nchunk) ? c0 + u : nchunk - 1;  // (uniform; the surplus chunks re-read the last one)
            const int l = c / cpr;
            const int e = (c - l * cpr) * 256 + (int)threadIdx.x;
            const float *src = row_ptr(l);
            const float t = (src != nullptr ? src : row_ptr(0))[e < d ? e : d - 1];
            v[u] = (src != nullptr && e < d) ? t : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (c0 + u < nchunk) wl[(c0 + u) * 256 + threadIdx.x] = v[u];
    }
}

#define CROSS_NJ 8  // 256-column chunks per row: d <= 2048
template <bool VEC, bool B_LDS>
__global__ __launch_bounds__(256) void crossnet_fwd_kernel(const float *__restrict__ x0, int64_t ldx, int d, int L,
                                                           const float *__restrict__ W, const float *__restrict__ Bv,
                                                           const float *__restrict__ wfc, const float *__restrict__ bfc,
                                                           float *__restrict__ xout, int64_t ldo,
                                                           float *__restrict__ logit, float *__restrict__ s_out,
                                                           int64_t B) {
    extern __shared__ __attribute__((aligned(16))) float wl[];  // [L][dp] W | [dp] wfc | (B_LDS: [L][dp] biases)
    const int dp = ((d + 255) / 256) * 256;
    const int nrows = B_LDS ? 2 * L + 1 : L + 1;
    cross_stage_rows(wl, nrows, d, dp, [&](int l) -> const float * {
        return l < L ? W + (int64_t)l * d : (l == L ? wfc : Bv + (int64_t)(l - L - 1) * d);
    });
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int nj = dp / 256;
    const float *bl = wl + (L + 1) * dp;
    // the next row's loads are issued before this row's arithmetic (register double buffer): a wave otherwise
    // spends most of its time waiting for its own single row
    // VEC: every chunk is ONE unconditional dwordx4 load from a clamped address (the row is ldx >= d floats of valid memory),
    // the lanes beyond d are zeroed with selects — a guarded load (the chunk that straddles d = 1677 took four scalar loads
    // behind per-lane branches) turns every wait of the row loop into vmcnt(0): measured 0.278 ms for a 440 MB read
    auto load_row = [&](int64_t b, f32x4 (&dst)[CROSS_NJ]) {
#pragma unroll
        for (int j = 0; j < CROSS_NJ; ++j) {
            const int e = 256 * j + 4 * lane;
            f32x4 xv = {0.f, 0.f, 0.f, 0.f};
            if (VEC) {
                const int ec = e + 4 <= (int)ldx ? e : (int)ldx - 4;
                const f32x4 raw = *reinterpret_cast<const f32x4 *>(x0 + b * ldx + ec);
#pragma unroll
                for (int k = 0; k < 4; ++k) xv[k] = (e + k < d) ? raw[k] : 0.f;
            } else if (j < nj) {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (e + k < d) xv[k] = x0[b * ldx + e + k];
            }
            dst[j] = xv;
        }
    };
    const int64_t bstep = (int64_t)gridDim.x * 4;
    int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    f32x4 nx[CROSS_NJ];
    if (b < B) load_row(b, nx);
    for (; b < B; b += bstep) {
        f32x4 r0[CROSS_NJ], r[CROSS_NJ];
#pragma unroll
        for (int j = 0; j < CROSS_NJ; ++j) {
            r0[j] = nx[j];
            r[j] = nx[j];
        }
        if (b + bstep < B) load_row(b + bstep, nx);
        for (int l = 0; l < L; ++l) {
            float part = 0.f;
#pragma unroll
            for (int j = 0; j < CROSS_NJ; ++j) {
                if (j < nj) {
                    const f32x4 pr = r[j] * *reinterpret_cast<const f32x4 *>(&wl[l * dp + 256 * j + 4 * lane]);
                    part += (pr.x + pr.y) + (pr.z + pr.w);
                }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
            const float sv = part;
            if (lane == 0 && s_out != nullptr) s_out[b * L + l] = sv;
#pragma unroll
            for (int j = 0; j < CROSS_NJ; ++j) {
                if (j < nj) {
                    const int e = 256 * j + 4 * lane;
                    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
                    if (B_LDS) {
                        bv = *reinterpret_cast<const f32x4 *>(&bl[l * dp + e]);
                    } else {
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            if (e + k < d) bv[k] = Bv[(int64_t)l * d + e + k];
                    }
                    r[j] = r[j] + (sv * r0[j] + bv);
                }
            }
        }
        if (xout != nullptr) {
#pragma unroll
            for (int j = 0; j < CROSS_NJ; ++j) {
                const int e = 256 * j + 4 * lane;
                if (j < nj) {
                    if (VEC && e + 4 <= d) {
                        *reinterpret_cast<f32x4 *>(xout + b * ldo + e) = r[j];
                    } else {
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            if (e + k < d) xout[b * ldo + e + k] = r[j][k];
                    }
                }
            }
        }
        if (logit != nullptr) {
            float part = 0.f;
#pragma unroll
            for (int j = 0; j < CROSS_NJ; ++j) {
                if (j < nj) {
                    const f32x4 pr = r[j] * *reinterpret_cast<const f32x4 *>(&wl[L * dp + 256 * j + 4 * lane]);
                    part += (pr.x + pr.y) + (pr.z + pr.w);
                }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
            if (lane == 0) logit[b] = part + (bfc != nullptr ? bfc[0] : 0.f);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Backward, streaming form.  CrossNet is low-rank in disguise: by induction X_l = A_l X_0 + C_l with the per-sample
// scalar A_l = 1 + sum_{k<l} s_k and the per-feature vector C_l = sum_{k<l} b_k.  Hence
//   t_l := dL/ds_l = g_{l+1} . X_0            g_l = g_{l+1} + t_l w_l            dX_0 = g_0 + sum_l s_l g_{l+1}
//   dW_l = sum_b t_l X_l = X_0^T (t_l A_l) + C_l sum_b t_l        (a [B,L]^T x [B,d] product: rp_linear_wgrad)
// so the per-row kernel needs NO parameter-gradient accumulators: one wave per row, the row (d <= 2048 -> 32
// floats per lane) in registers, dots by wave shuffles, no barriers; it writes dX_0 and the 2L+2 scalars
// V[b] = [t_l A_l (l<L) | gl A_L | t_l (l<L) | gl] whose product with X_0 (and column sums) give every parameter
// gradient.  HBM traffic: X_0 read once, dX_0 written once (+ the incoming gradient row when it is not the fc).
// ------------------------------------------------------------------------------------------------
// Execution: persistent workgroups; W (L rows) and wfc are staged ONCE per workgroup into LDS, zero-padded to a
// multiple of 256 columns (every row used to re-fetch (L+1)*d floats from L2 through the 64 B/clk L1 — twice the
// bytes of the HBM stream itself); lane l owns the 4 consecutive columns 256j + 4l .. +3 of chunk j, so X_0 / g / dX_0
// move as dwordx4 and the W reads are conflict-free ds_read_b128.
template <bool VEC>
__global__ __launch_bounds__(256) void crossnet_bwd_rows_kernel(const float *__restrict__ x0, int64_t ldx, int d, int L,
                                                                const float *__restrict__ W,
                                                                const float *__restrict__ wfc,
                                                                const float *__restrict__ s_in,
                                                                const float *__restrict__ g_x, int64_t ldg,
                                                                const float *__restrict__ g_logit,
                                                                float *__restrict__ dx0, int64_t lddx,
                                                                float *__restrict__ V, int64_t B) {
    extern __shared__ __attribute__((aligned(16))) float wl[];  // [(L+1)][dp]
    const int dp = ((d + 255) / 256) * 256;
    cross_stage_rows(wl, L + 1, d, dp, [&](int l) -> const float * { return l < L ? W + (int64_t)l * d : wfc; });
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int nj = dp / 256;
    // (VEC: unconditional clamped dwordx4 loads + selects, as in the forward)
    auto load_row = [&](int64_t b, f32x4 (&dx_)[CROSS_NJ], f32x4 (&dg_)[CROSS_NJ]) {
#pragma unroll
        for (int j = 0; j < CROSS_NJ; ++j) {
            const int e = 256 * j + 4 * lane;
            f32x4 xv = {0.f, 0.f, 0.f, 0.f}, gv = {0.f, 0.f, 0.f, 0.f};
            if (VEC) {
                const int ec = e + 4 <= (int)ldx ? e : (int)ldx - 4;
                const f32x4 raw = *reinterpret_cast<const f32x4 *>(x0 + b * ldx + ec);
#pragma unroll
                for (int k = 0; k < 4; ++k) xv[k] = (e + k < d) ? raw[k] : 0.f;
                if (g_x != nullptr) {  // (uniform)
                    const int gc = e + 4 <= (int)ldg ? e : (int)ldg - 4;
                    const f32x4 rg = *reinterpret_cast<const f32x4 *>(g_x + b * ldg + gc);
#pragma unroll
                    for (int k = 0; k < 4; ++k) gv[k] = (e + k < d) ? rg[k] : 0.f;
                }
            } else if (j < nj) {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (e + k < d) {
                        xv[k] = x0[b * ldx + e + k];
                        if (g_x != nullptr) gv[k] = g_x[b * ldg + e + k];
                    }
            }
            dx_[j] = xv;
            dg_[j] = gv;
        }
    };
    const int64_t bstep = (int64_t)gridDim.x * 4;
    int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    f32x4 nx[CROSS_NJ], ng[CROSS_NJ];
    if (b < B) load_row(b, nx, ng);
    for (; b < B; b += bstep) {
        f32x4 r0[CROSS_NJ], g[CROSS_NJ], gx0[CROSS_NJ];
        const float gl = (g_logit != nullptr) ? g_logit[b] : 0.f;
#pragma unroll
        for (int j = 0; j < CROSS_NJ; ++j) {
            r0[j] = nx[j];
            g[j] = ng[j];
            if (j < nj) g[j] += gl * *reinterpret_cast<const f32x4 *>(&wl[L * dp + 256 * j + 4 * lane]);  // wfc
            gx0[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if (b + bstep < B) load_row(b + bstep, nx, ng);  // in flight during this row's arithmetic
        float a_run = 1.f;  // A_L = 1 + sum_k s_k, then peeled back layer by layer
        for (int l = 0; l < L; ++l) a_run += s_in[b * L + l];
        float *Vb = V + b * (2 * L + 2);
        if (lane == 0) {
            Vb[L] = gl * a_run;
            Vb[2 * L + 1] = gl;
        }
        for (int l = L - 1; l >= 0; --l) {
            const float sl = s_in[b * L + l];
            a_run -= sl;  // A_l
            float part = 0.f;
#pragma unroll
            for (int j = 0; j < CROSS_NJ; ++j) {
                const f32x4 pr = g[j] * r0[j];
                part += (pr.x + pr.y) + (pr.z + pr.w);
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
            const float tl = part;
            if (lane == 0) {
                Vb[l] = tl * a_run;
                Vb[L + 1 + l] = tl;
            }
#pragma unroll
            for (int j = 0; j < CROSS_NJ; ++j) {
                if (j < nj) {
                    gx0[j] += sl * g[j];
                    g[j] += tl * *reinterpret_cast<const f32x4 *>(&wl[l * dp + 256 * j + 4 * lane]);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < CROSS_NJ; ++j) {
            const int e = 256 * j + 4 * lane;
            if (j < nj) {
                const f32x4 o4 = gx0[j] + g[j];
                if (VEC) {
                    // whole dwordx4 stores up to the row's leading dimension: the columns beyond d receive exact zeros (the
                    // gradient of x's padding), no per-element branches around the stores
                    f32x4 z4;
#pragma unroll
                    for (int k = 0; k < 4; ++k) z4[k] = (e + k < d) ? o4[k] : 0.f;
                    if (e + 4 <= (int)lddx) *reinterpret_cast<f32x4 *>(dx0 + b * lddx + e) = z4;
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (e + k < d) dx0[b * lddx + e + k] = o4[k];
                }
            }
        }
    }
}

extern "C" int rp_crossnet_bwd_rows(const float *x0, int64_t ldx, int d, int L, const float *W, const float *wfc,
                                    const float *s_in, const float *g_x, int64_t ldg, const float *g_logit,
                                    float *dx0, int64_t lddx, float *V, int64_t B, rp_stream_t stream) {
    RP_REQUIRE(x0 && W && s_in && dx0 && V && B >= 0, "crossnet_bwd_rows: bad argument");
    RP_REQUIRE(g_x != nullptr || g_logit != nullptr, "crossnet_bwd_rows: no incoming gradient");
    RP_REQUIRE(g_logit == nullptr || wfc != nullptr, "crossnet_bwd_rows: logit gradient needs wfc");
    if (d < 1 || d > 256 * CROSS_NJ || L < 1 || L > CROSS_MAXL)
        return rp_fail(RP_ERR_UNSUPPORTED, "crossnet: d=%d (max %d) / L=%d (max %d) unsupported", d, 256 * CROSS_NJ, L,
                       CROSS_MAXL);
    if (B == 0) return RP_OK;
    const int dp = ((d + 255) / 256) * 256;
    const size_t lds = (size_t)(L + 1) * dp * sizeof(float);  // <= 7 * 2048 * 4 = 56 KB
    const bool vec = (ldx % 4 == 0) && (lddx % 4 == 0) && rp_aligned16(x0) && rp_aligned16(dx0) &&
                     (g_x == nullptr || ((ldg % 4 == 0) && rp_aligned16(g_x)));
    int64_t nb = rp_cdiv(B, 4);
    const int64_t nb_max = cross_nb_max(2048);
    if (nb > nb_max) nb = nb_max;  // persistent: each wave walks over B / (4 * nb) rows, W staged once per workgroup
    if (vec)
        hipLaunchKernelGGL((crossnet_bwd_rows_kernel<true>), dim3((unsigned)nb), dim3(256), lds, (hipStream_t)stream, x0,
                           ldx, d, L, W, wfc, s_in, g_x, ldg, g_logit, dx0, lddx, V, B);
    else
        hipLaunchKernelGGL((crossnet_bwd_rows_kernel<false>), dim3((unsigned)nb), dim3(256), lds, (hipStream_t)stream,
                           x0, ldx, d, L, W, wfc, s_in, g_x, ldg, g_logit, dx0, lddx, V, B);
    RP_LAUNCH_CHECK("crossnet_bwd_rows");
    return RP_OK;
}

// The parameter gradients of the CrossNet from what the streaming backward left (round 5: this [L, d] arithmetic was a
// dozen ATen launches — cat, cumsum, flip, mul, add — the only foreign launches of a DCN step, which kept it from being
// replayed as a launch plan).  With P [2L+2, d] = V^T X_0 (rp_linear_wgrad over the rows kernel's V), cs [2L+2] = the column
// sums of V (st_l = cs[L+1+l] = sum_b t_l, sgl = cs[2L+1] = sum_b g_logit), C_l = sum_{k<l} b_k:
//     dW[l] = P[l] + C_l st_l          dB[l] = sum_{k>l} w_k st_k + extra          extra = wfc sgl (fused fc) | colg (else)
//     dwfc  = P[L] + C_L sgl   (fused fc)
__global__ __launch_bounds__(256) void crossnet_param_grads_kernel(const float *__restrict__ P, int64_t ldp,
                                                                   const float *__restrict__ cs, const float *__restrict__ W,
                                                                   const float *__restrict__ Bv, const float *__restrict__ wfc,
                                                                   const float *__restrict__ colg, int L, int d,
                                                                   float *__restrict__ dW, float *__restrict__ dB,
                                                                   float *__restrict__ dwfc) {
    const int j = (int)blockIdx.x * 256 + (int)threadIdx.x;
    if (j >= d) return;
    const float sgl = cs[2 * L + 1];
    float c = 0.f;
    for (int l = 0; l < L; ++l) {
        dW[(int64_t)l * d + j] = P[(int64_t)l * ldp + j] + c * cs[L + 1 + l];
        c += Bv[(int64_t)l * d + j];
    }
    const float extra = wfc != nullptr ? wfc[j] * sgl : (colg != nullptr ? colg[j] : 0.f);
    float suffix = 0.f;
    for (int l = L - 1; l >= 0; --l) {
        dB[(int64_t)l * d + j] = suffix + extra;
        suffix += W[(int64_t)l * d + j] * cs[L + 1 + l];
    }
    if (dwfc != nullptr) dwfc[j] = P[(int64_t)L * ldp + j] + c * sgl;
}

extern "C" int rp_crossnet_param_grads(const float *P, int64_t ldp, const float *cs, const float *W, const float *Bv,
                                       const float *wfc, const float *colg, int L, int d, float *dW, float *dB, float *dwfc,
                                       rp_stream_t stream) {
    RP_REQUIRE(P && cs && W && Bv && dW && dB && L >= 1 && d >= 1 && ldp >= d, "crossnet_param_grads: bad argument");
    RP_REQUIRE((wfc == nullptr) == (dwfc == nullptr), "crossnet_param_grads: wfc and dwfc go together");
    hipLaunchKernelGGL(crossnet_param_grads_kernel, dim3((unsigned)rp_cdiv(d, 256)), dim3(256), 0, (hipStream_t)stream, P, ldp, cs, W,
                       Bv, wfc, colg, L, d, dW, dB, dwfc);
    RP_LAUNCH_CHECK("crossnet_param_grads");
    return RP_OK;
}

extern "C" int rp_crossnet_fwd(const float *x0, int64_t ldx, int d, int L, const float *W, const float *Bv,
                               const float *wfc, const float *bfc, float *xout, int64_t ldo, float *logit,
                               float *s_out, int64_t B, rp_stream_t stream) {
    RP_REQUIRE(x0 && W && Bv && B >= 0 && ldx >= d, "crossnet_fwd: bad argument");
    RP_REQUIRE(xout != nullptr || logit != nullptr, "crossnet_fwd: nothing to produce");
    RP_REQUIRE(logit == nullptr || wfc != nullptr, "crossnet_fwd: logit needs wfc");
    if (d < 1 || d > 256 * CROSS_NJ || L < 1 || L > CROSS_MAXL)
        return rp_fail(RP_ERR_UNSUPPORTED, "crossnet: d=%d (max %d) / L=%d (max %d) unsupported", d, 256 * CROSS_NJ, L,
                       CROSS_MAXL);
    if (B == 0) return RP_OK;
    const int dp = ((d + 255) / 256) * 256;
    const bool b_lds = (size_t)(2 * L + 1) * dp * sizeof(float) <= 64 * 1024;
    const size_t lds = (size_t)(b_lds ? 2 * L + 1 : L + 1) * dp * sizeof(float);
    const bool vec = (ldx % 4 == 0) && rp_aligned16(x0) && (xout == nullptr || ((ldo % 4 == 0) && rp_aligned16(xout)));
    int64_t nb = rp_cdiv(B, 4);
    const int64_t nb_max = cross_nb_max(2048);
    if (nb > nb_max) nb = nb_max;
    hipStream_t st = (hipStream_t)stream;
#define CF(VEC, BL)                                                                                                   \
    hipLaunchKernelGGL((crossnet_fwd_kernel<VEC, BL>), dim3((unsigned)nb), dim3(256), lds, st, x0, ldx, d, L, W, Bv, wfc, \
                       bfc, xout, ldo, logit, s_out, B)
    if (vec && b_lds) CF(true, true);
    else if (vec) CF(true, false);
    else if (b_lds) CF(false, true);
    else CF(false, false);
#undef CF
    RP_LAUNCH_CHECK("crossnet_fwd");
    return RP_OK;
}
