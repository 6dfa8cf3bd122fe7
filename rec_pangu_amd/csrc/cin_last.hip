// The collapsed LAST layer of the xDeepFM CIN, forward and backward, gfx950.
//   reference: layers/interaction.py:157-171.  With no activation and linear sum-pooling + fc after it, the last CIN
//   layer enters the logit only through
//       p[b] = sum_d sum_{h<H} sum_{m<M} V[h,m] X_0[b,h,d] X_{L-1}[b,m,d],        V = sum_o c[o] W_L[o]   (host, tiny)
//   i.e. ONE output channel.  The generic layer kernels (cin.hip) are built around many channels sharing staged
//   operands; run with O = 1 they spent 67 ms per step on 28 GFLOP.  The real cost of this layer is streaming
//   X_{L-1} ([B, M, D]: 2.1 GB at M = 128), so it gets its own HBM-bound kernels:
//     forward / bwd_x : one wave per sample, LANE = embedding column d (D <= 64).  X_0[b,:,d] lives in 32 registers,
//                       X_{L-1}[b,m,d] streams through (one coalesced 4*D-byte row per m), V^T[m,:] is wave-uniform
//                       and comes through the scalar cache:
//                           T[h]   += V[h,m] * xp           (dX_0[b,h,d] = g[b] T[h],  p[b] = sum_d sum_h X_0 T)
//                           dXp[m]  = g[b] * sum_h V[h,m] X_0[b,h,d]
//                       fp32 VALU, 2*H*M*D MAC per sample: ~0.4 ms (fwd) / ~0.8 ms (bwd) of VALU time at B = 65536.
//     bwd_v           : dV[h,m] = sum_{b,d} g[b] X_0[b,h,d] X_{L-1}[b,m,d] — a TN GEMM whose contraction index is
//                       (b,d), on the bf16 matrix core with split-bf16 operands (the rp_linear_wgrad machinery; the
//                       operands are already "contraction-contiguous": 8 consecutive d of one (b,h) are one 32-byte
//                       read).  Batch split over workgroups, deterministic second-stage sum.
#include "common.h"

#define CL_MAXH 32

// vt: V transposed and zero-padded to [M][32]
template <bool BWD>
__global__ __launch_bounds__(256) void cin_last_kernel(const float *__restrict__ x0, int64_t ld0,
                                                       const float *__restrict__ xp, int64_t ldp,
                                                       const float *__restrict__ vt, const float *__restrict__ g,
                                                       int H, int M, int D, float *__restrict__ pooled,
                                                       float *__restrict__ dx0, int64_t ldd0, float *__restrict__ dxp,
                                                       int64_t lddp, int64_t B) {
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool dok = lane < D;
    for (int64_t b = (int64_t)blockIdx.x * 4 + wid; b < B; b += (int64_t)gridDim.x * 4) {
        const float *x0b = x0 + b * ld0, *xpb = xp + b * ldp;
        float x[CL_MAXH], T[CL_MAXH];
#pragma unroll
        for (int h = 0; h < CL_MAXH; ++h) {
            x[h] = (h < H && dok) ? x0b[(int64_t)h * D + lane] : 0.f;
            T[h] = 0.f;
        }
        const float gb = BWD ? g[b] : 0.f;
        // X_{L-1} rows four at a time, the next four in flight while these are consumed (one 256-byte row per wave
        // load: without this the loop is latency-bound at ~2.4 TB/s)
        float xn[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) xn[j] = (j < M && dok) ? xpb[(int64_t)j * D + lane] : 0.f;
        for (int m0 = 0; m0 < M; m0 += 4) {
            float xc[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                xc[j] = xn[j];
                xn[j] = (m0 + 4 + j < M && dok) ? xpb[(int64_t)(m0 + 4 + j) * D + lane] : 0.f;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int m = m0 + j;
                if (m >= M) break;
                const float *vm = vt + (int64_t)m * CL_MAXH;  // wave-uniform -> scalar loads
                const float xv = xc[j];
                float dsum = 0.f;
#pragma unroll
                for (int h = 0; h < CL_MAXH; ++h) {
                    const float v = vm[h];
                    T[h] += v * xv;
                    if (BWD) dsum += v * x[h];
                }
                if (BWD && dok) dxp[b * lddp + (int64_t)m * D + lane] = gb * dsum;
            }
        }
        if (BWD) {
#pragma unroll
            for (int h = 0; h < CL_MAXH; ++h)
                if (h < H && dok) dx0[b * ldd0 + (int64_t)h * D + lane] = gb * T[h];
        } else {
            float s = 0.f;
#pragma unroll
            for (int h = 0; h < CL_MAXH; ++h) s += x[h] * T[h];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
            if (lane == 0) pooled[b] = s;
        }
    }
}

static int cl_check(int H, int M, int D) {
    if (H < 1 || H > CL_MAXH || M < 1 || D < 1 || D > 64)
        return rp_fail(RP_ERR_UNSUPPORTED, "cin_last: H=%d (<=%d) M=%d D=%d (<=64) unsupported", H, CL_MAXH, M, D);
    return RP_OK;
}

extern "C" int rp_cin_last_fits(int H, int M, int D) { return (H >= 1 && H <= CL_MAXH && M >= 1 && D >= 1 && D <= 64) ? 1 : 0; }

static unsigned cl_grid(int64_t B) {
    const int64_t nb = rp_cdiv(B, 4);
    return (unsigned)(nb < 8192 ? nb : 8192);
}

extern "C" int rp_cin_last_fwd(const float *x0, int64_t ld0, const float *xp, int64_t ldp, const float *vt, int H, int M,
                               int D, float *pooled, int64_t B, rp_stream_t stream) {
    RP_REQUIRE(x0 && xp && vt && pooled && B >= 0, "cin_last_fwd: null pointer");
    int rc = cl_check(H, M, D);
    if (rc != RP_OK) return rc;
    RP_REQUIRE(ld0 >= (int64_t)H * D && ldp >= (int64_t)M * D, "cin_last_fwd: ld too small");
    if (B == 0) return RP_OK;
    hipLaunchKernelGGL((cin_last_kernel<false>), dim3(cl_grid(B)), dim3(256), 0, (hipStream_t)stream, x0, ld0, xp, ldp, vt,
                       nullptr, H, M, D, pooled, nullptr, 0, nullptr, 0, B);
    RP_LAUNCH_CHECK("cin_last_fwd");
    return RP_OK;
}

extern "C" int rp_cin_last_bwd_x(const float *x0, int64_t ld0, const float *xp, int64_t ldp, const float *vt,
                                 const float *g, int H, int M, int D, float *dx0, int64_t ldd0, float *dxp,
                                 int64_t lddp, int64_t B, rp_stream_t stream) {
    RP_REQUIRE(x0 && xp && vt && g && dx0 && dxp && B >= 0, "cin_last_bwd_x: null pointer");
    int rc = cl_check(H, M, D);
    if (rc != RP_OK) return rc;
    RP_REQUIRE(ld0 >= (int64_t)H * D && ldp >= (int64_t)M * D && ldd0 >= (int64_t)H * D && lddp >= (int64_t)M * D,
               "cin_last_bwd_x: ld too small");
    if (B == 0) return RP_OK;
    hipLaunchKernelGGL((cin_last_kernel<true>), dim3(cl_grid(B)), dim3(256), 0, (hipStream_t)stream, x0, ld0, xp, ldp, vt,
                       g, H, M, D, nullptr, dx0, ldd0, dxp, lddp, B);
    RP_LAUNCH_CHECK("cin_last_bwd_x");
    return RP_OK;
}

// ------------------------------------------------------------------------------------------------ dV
// TN GEMM over the contraction index r = (b, d):  dV[h, m] = sum_r (g[b] X0[b,h,d]) * Xp[b,m,d].
// Output tile 32 (h) x 128 (m) per workgroup pass; stage = 32 contraction rows = half a sample's d range
// (D <= 32: the whole sample, zero padded).  Thread (column c = t & 63, octet o = t >> 6) reads the 8 consecutive d
// 8o..8o+7 of X0 row h = c (c < 32) and of the two Xp rows m = 2c, 2c+1 (c < 64) as 2 x dwordx4 each, splits them
// into bf16 pieces and writes ONE 16-byte LDS segment per piece — the LDS images are [h][r] / [m][r] and the MFMA
// fragments (8 consecutive r per lane) are ds_read_b128.  Waves: 4 = the four 32-column m tiles.
typedef __bf16 clbf8 __attribute__((ext_vector_type(8)));
typedef float clf8 __attribute__((ext_vector_type(8)));
#define CLV_LD 40

__device__ __forceinline__ void cl_split(clf8 v, clbf8 (&p)[3]) {
    p[0] = __builtin_convertvector(v, clbf8);
    v -= __builtin_convertvector(p[0], clf8);
    p[1] = __builtin_convertvector(v, clbf8);
    v -= __builtin_convertvector(p[1], clf8);
    p[2] = __builtin_convertvector(v, clbf8);
}

__global__ __launch_bounds__(256) void cin_last_bwd_v_kernel(const float *__restrict__ x0, int64_t ld0,
                                                             const float *__restrict__ xp, int64_t ldp,
                                                             const float *__restrict__ g, int H, int M, int D,
                                                             float *__restrict__ P, int64_t B, int64_t b_per_blk) {
    __shared__ __attribute__((aligned(16))) __bf16 At[3][32][CLV_LD];    // (g X0)[h][r]
    __shared__ __attribute__((aligned(16))) __bf16 Bt[3][128][CLV_LD];   // Xp[m][r]
    const int t = threadIdx.x;
    const int w = t >> 6, l = t & 63, i = l & 31, hh = l >> 5;
    const int c = t & 63, o = t >> 6;
    const int m0 = blockIdx.y * 128;
    const int64_t bbeg = (int64_t)blockIdx.x * b_per_blk;
    int64_t bend = bbeg + b_per_blk;
    if (bend > B) bend = B;
    const int nhalf = (D + 31) / 32;  // stages per sample
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const bool vec = (D % 4 == 0) && (ld0 % 4 == 0) && (ldp % 4 == 0) && ((reinterpret_cast<uintptr_t>(x0) & 15u) == 0) &&
                     ((reinterpret_cast<uintptr_t>(xp) & 15u) == 0);
    for (int64_t b = bbeg; b < bend; ++b) {
        const float gb = g[b];
        for (int half = 0; half < nhalf; ++half) {
            const int d0 = half * 32 + 8 * o;
            clf8 va, vb0, vb1;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                va[e] = 0.f;
                vb0[e] = 0.f;
                vb1[e] = 0.f;
            }
            auto load8 = [&](const float *row, clf8 &dst) {
                if (vec && d0 + 8 <= D) {
                    const f32x4 q0 = *reinterpret_cast<const f32x4 *>(row + d0);
                    const f32x4 q1 = *reinterpret_cast<const f32x4 *>(row + d0 + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        dst[e] = q0[e];
                        dst[4 + e] = q1[e];
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (d0 + e < D) dst[e] = row[d0 + e];
                }
            };
            if (c < H) load8(x0 + b * ld0 + (int64_t)c * D, va);
            const int ma = m0 + 2 * c, mb = ma + 1;
            if (ma < M) load8(xp + b * ldp + (int64_t)ma * D, vb0);
            if (mb < M) load8(xp + b * ldp + (int64_t)mb * D, vb1);
            va *= gb;
            __syncthreads();  // the previous stage's fragment reads are done
            clbf8 pc[3];
            if (c < 32) {
                cl_split(va, pc);
#pragma unroll
                for (int q = 0; q < 3; ++q) *reinterpret_cast<clbf8 *>(&At[q][c][8 * o]) = pc[q];
            }
            cl_split(vb0, pc);
#pragma unroll
            for (int q = 0; q < 3; ++q) *reinterpret_cast<clbf8 *>(&Bt[q][2 * c][8 * o]) = pc[q];
            cl_split(vb1, pc);
#pragma unroll
            for (int q = 0; q < 3; ++q) *reinterpret_cast<clbf8 *>(&Bt[q][2 * c + 1][8 * o]) = pc[q];
            __syncthreads();
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                clbf8 a[3], bq[3];
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    a[q] = *reinterpret_cast<const clbf8 *>(&At[q][i][ks * 16 + 8 * hh]);
                    bq[q] = *reinterpret_cast<const clbf8 *>(&Bt[q][32 * w + i][ks * 16 + 8 * hh]);
                }
                // hi*lo + lo*hi + mid*mid + hi*mid + mid*hi + hi*hi  (smallest terms first)
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], bq[2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], bq[0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], bq[1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], bq[1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], bq[0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], bq[0], acc, 0, 0, 0);
            }
        }
    }
    // partial[blockIdx.x][h][m]; C layout: col (m) = lane & 31, row (h) = (r&3) + 8*(r>>2) + 4*(lane>>5)
    float *Pz = P + (int64_t)blockIdx.x * H * M;
    const int m = m0 + 32 * w + i;
    if (m < M) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int h = (r & 3) + 8 * (r >> 2) + 4 * hh;
            if (h < H) Pz[(int64_t)h * M + m] = acc[r];
        }
    }
}

// out[e] = sum_z P[z][e]: 16 elements x 16 slices per workgroup, fixed order
__global__ __launch_bounds__(256) void cin_last_sum_kernel(const float *__restrict__ P, int S, int64_t n,
                                                           float *__restrict__ out) {
    __shared__ float red[16][17];
    const int c = threadIdx.x & 15, q = threadIdx.x >> 4;
    const int64_t e = (int64_t)blockIdx.x * 16 + c;
    float s = 0.f;
    if (e < n)
        for (int z = q; z < S; z += 16) s += P[(int64_t)z * n + e];
    red[q][c] = s;
    __syncthreads();
    if (q != 0 || e >= n) return;
#pragma unroll
    for (int j = 1; j < 16; ++j) s += red[j][c];
    out[e] = s;
}

static int64_t clv_blocks(int64_t B) {
    int64_t nb = B < 1024 ? B : 1024;
    return nb < 1 ? 1 : nb;
}

extern "C" int rp_cin_last_bwd_v_workspace_bytes(int64_t B, int H, int M, size_t *bytes) {
    RP_REQUIRE(bytes && B >= 0 && H >= 1 && M >= 1, "cin_last_bwd_v_workspace_bytes: bad argument");
    *bytes = (size_t)clv_blocks(B) * H * M * sizeof(float) + 256;
    return RP_OK;
}

extern "C" int rp_cin_last_bwd_v(const float *x0, int64_t ld0, const float *xp, int64_t ldp, const float *g, int H, int M,
                                 int D, float *dV, int64_t B, void *workspace, size_t workspace_bytes,
                                 rp_stream_t stream) {
    RP_REQUIRE(x0 && xp && g && dV && workspace && B >= 1, "cin_last_bwd_v: bad argument");
    int rc = cl_check(H, M, D);
    if (rc != RP_OK) return rc;
    size_t need = 0;
    rp_cin_last_bwd_v_workspace_bytes(B, H, M, &need);
    RP_REQUIRE(workspace_bytes >= need, "cin_last_bwd_v: workspace %zu < %zu", workspace_bytes, need);
    float *P = reinterpret_cast<float *>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
    const int64_t nb = clv_blocks(B);
    const int64_t per = rp_cdiv(B, nb);
    const int64_t nbx = rp_cdiv(B, per);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(cin_last_bwd_v_kernel, dim3((unsigned)nbx, (unsigned)rp_cdiv(M, 128)), dim3(256), 0, s, x0, ld0, xp, ldp,
                       g, H, M, D, P, B, per);
    RP_LAUNCH_CHECK("cin_last_bwd_v");
    const int64_t n = (int64_t)H * M;
    hipLaunchKernelGGL(cin_last_sum_kernel, dim3((unsigned)rp_cdiv(n, 16)), dim3(256), 0, s, P, (int)nbx, n, dV);
    RP_LAUNCH_CHECK("cin_last_bwd_v reduce");
    return RP_OK;
}
