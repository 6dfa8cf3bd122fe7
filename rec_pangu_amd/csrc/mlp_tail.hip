// The narrow tail of the reference's default MLP (layers/deep.py:62-72 with hidden_units [64, 64, 64], output_dim 1:
// DeepFM's dnn.net.{2,4,6}) as ONE forward and ONE backward launch, gfx950.
//
// After the first layer (K = F*D + ND = 1677 -> 64, its own GEMM) the chain is  h1 -> [Linear 64x64 + ReLU] x L ->
// Linear 64 -> 1 : per sample 16 K MACs over 256-byte activations.  As separate launches (forward: L + 1 GEMMs;
// backward: L + 1 dgrads, L + 1 wgrads with their second stages, L + 1 transposes) each moves 16.7 MB at batch 65536 and
// costs 10-20 us of ramp-up and drain: ~0.18 ms of a 1.8 ms DeepFM step for ~3 % of its bytes.  Here a workgroup takes
// 128 rows through the whole chain:
//   * weights of one layer at a time in LDS as split-bf16 pieces (six products: fp32-faithful, as every HBM-bound GEMM);
//   * a wave owns 32 rows; the activation tile never leaves the CU between layers: the MFMA C layout (lane = column) is
//     turned into the A layout of the next layer (lane = row) through a 32 x 64 fp32 LDS tile;
//   * backward: dgrads in the same A-stationary form on W^T (transposed while staged: no host-side transposes), weight
//     gradients as TN products over the workgroup's 128 rows with the 64 x 64 output split over the four waves
//     (32 x 32 each, 16 accumulator registers per layer), per-workgroup partials summed in a fixed order by a second
//     small launch (deterministic); bias / head gradients as column sums of the same tiles.
#include "common.h"
#include "bfsplit.h"

#define MT_LD 72   // bf16 row of 64 + 8 pad (144 B, odd multiple of 16 B): conflict-free ds_read_b128 fragment reads
#define MT_CT 68   // fp32 row of an activation tile
#define MT_MAXL 3  // hidden 64 x 64 layers in the tail

struct TailFwdArgs {
    const float *W[MT_MAXL];  // [64, 64] row-major (out, in), row stride ldw
    const float *b[MT_MAXL];  // [64] or null
    float *h[MT_MAXL];        // outputs of the hidden layers [M, 64] (saved for the backward), row stride 64
    int64_t ldw[MT_MAXL];
};

// W [64 out][64 in] fp32 -> LDS pieces Wt[q][n = out][k = in]  (the NT "W" operand: contraction over the input index)
__device__ __forceinline__ void mt_stage_w(const float *__restrict__ W, int64_t ldw, __bf16 (*Wt)[64][MT_LD]) {
    const int d = threadIdx.x >> 2, c = (threadIdx.x & 3) * 16;
    const float *src = W + (int64_t)d * ldw + c;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const f32x4 v0 = *reinterpret_cast<const f32x4 *>(src + 8 * u), v1 = *reinterpret_cast<const f32x4 *>(src + 8 * u + 4);
        f32x8 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[e] = v0[e];
            v[4 + e] = v1[e];
        }
        bf16x8 pc[3];
        bf_split8<3>(v, pc);
#pragma unroll
        for (int q = 0; q < 3; ++q) *reinterpret_cast<bf16x8 *>(&Wt[q][d][c + 8 * u]) = pc[q];
    }
}

// W [64 out][64 in] fp32 -> LDS pieces Wt[q][n = in][k = out]  (dgrad: contraction over the OUTPUT index), transposed
// while staged (2-byte scattered LDS writes, once per workgroup and layer)
__device__ __forceinline__ void mt_stage_wt(const float *__restrict__ W, int64_t ldw, __bf16 (*Wt)[64][MT_LD]) {
    const int k = threadIdx.x >> 2, c = (threadIdx.x & 3) * 16;  // row k = output index, 16 input columns from c
    const float *src = W + (int64_t)k * ldw + c;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(src + 4 * u);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float x = v[e];
            const __bf16 hi = (__bf16)x;
            const float r1 = x - (float)hi;
            const __bf16 mid = (__bf16)r1;
            const __bf16 lo = (__bf16)(r1 - (float)mid);
            const int n = c + 4 * u + e;
            Wt[0][n][k] = hi;
            Wt[1][n][k] = mid;
            Wt[2][n][k] = lo;
        }
    }
}

// The loss head folded into the forward (rp_mlp_tail_fwd_bce): z = sum of the other logit addends (DeepFM: the FM term) + this
// tail's logit, pred = sigmoid(z), and the workgroup's partial sum of the BCE terms — the arithmetic of rp_sigmoid_bce_fwd
// (csrc/loss.hip) term by term, so pred is bit-identical to the two-launch form; only the ORDER of the loss sum differs (one
// partial per 128 rows here).  rp_loss_finish turns the partials into the scalar.
struct TailBce {
    const float *add[3];
    int n_add;
    const float *label;
    float p_eps;
    float *pred;
    float *partial;  // [workgroups]
};

template <int L, bool BCE>
__global__ __launch_bounds__(256, 2) void mlp_tail_fwd_kernel(const float *__restrict__ hin, int64_t ldin, TailFwdArgs a,
                                                              const float *__restrict__ wout, const float *__restrict__ bout,
                                                              float *__restrict__ logit, int64_t M, TailBce bc) {
    __shared__ __attribute__((aligned(16))) __bf16 Wt[3][64][MT_LD];
    __shared__ __attribute__((aligned(16))) float Ct[4][32][MT_CT];  // per wave: the activation tile between layers
    const int wv = threadIdx.x >> 6, l = threadIdx.x & 63, i = l & 31, h = l >> 5;
    const int64_t row = (int64_t)blockIdx.x * 128 + 32 * wv + i;  // A layout: lane = row, 8 k per lane and k-step
    const int64_t rowc = row < M ? row : M - 1;                    // clamped for loads (never stored)
    f32x8 av[4];  // this lane's row of the current activation, fp32, k = 16 ks + 8 h .. + 7
    {
        const float *src = hin + rowc * ldin + 8 * h;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const f32x4 v0 = *reinterpret_cast<const f32x4 *>(src + ks * 16), v1 = *reinterpret_cast<const f32x4 *>(src + ks * 16 + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                av[ks][e] = v0[e];
                av[ks][4 + e] = v1[e];
            }
        }
    }
#pragma unroll
    for (int ly = 0; ly < L; ++ly) {
        if (ly > 0) __syncthreads();  // the previous layer's fragment reads of Wt are done
        mt_stage_w(a.W[ly], a.ldw[ly], Wt);
        __syncthreads();
        f32x16 acc[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8 af[3];
            bf_split8<3>(av[ks], af);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                bf16x8 b[3];
#pragma unroll
                for (int q = 0; q < 3; ++q) b[q] = *reinterpret_cast<const bf16x8 *>(&Wt[q][nt * 32 + i][ks * 16 + 8 * h]);
#pragma unroll
                for (int pr = 0; pr < 6; ++pr)
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[BfProd<6>::pa(pr)], b[BfProd<6>::pb(pr)], acc[nt], 0, 0, 0);
            }
        }
        // epilogue in the C layout (col = nt*32 + i, row = (r&3) + 8*(r>>2) + 4*h of the wave's 32): bias + ReLU, store the
        // layer's output (the backward needs it) and park it in the wave's LDS tile for the A layout of the next layer
        float *hout = a.h[ly];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int n = nt * 32 + i;
            const float bv = a.b[ly] != nullptr ? a.b[ly][n] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rl = (r & 3) + 8 * (r >> 2) + 4 * h;
                const float v = fmaxf(acc[nt][r] + bv, 0.f);
                Ct[wv][rl][n] = v;
                const int64_t m = (int64_t)blockIdx.x * 128 + 32 * wv + rl;
                if (m < M) hout[m * 64 + n] = v;
            }
        }
        __syncthreads();  // (a wave-level barrier would do: the tile is private to the wave)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const f32x4 v0 = *reinterpret_cast<const f32x4 *>(&Ct[wv][i][ks * 16 + 8 * h]);
            const f32x4 v1 = *reinterpret_cast<const f32x4 *>(&Ct[wv][i][ks * 16 + 8 * h + 4]);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                av[ks][e] = v0[e];
                av[ks][4 + e] = v1[e];
            }
        }
    }
    // head: logit[row] = a_L[row, :] . wout + bout   (plain fp32 fma chain over the lane's 32 columns, then the two halves)
    float part = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int e = 0; e < 8; ++e) part = __builtin_fmaf(av[ks][e], wout[ks * 16 + 8 * h + e], part);
    part += __shfl_xor(part, 32, 64);
    const float lg = part + (bout != nullptr ? bout[0] : 0.f);
    if constexpr (!BCE) {
        if (h == 0 && row < M) logit[row] = lg;
    } else {
        __shared__ float lred[128];
        if (h == 0) {
            float term = 0.f;
            if (row < M) {
                float z = lg;
                if (bc.n_add > 0) {  // (the addends first, this tail's logit last: the order of the model's logit list)
                    z = bc.add[0][row];
                    for (int q = 1; q < bc.n_add; ++q) z += bc.add[q][row];
                    z += lg;
                }
                const float p = 1.f / (1.f + expf(-z));
                bc.pred[row] = p;
                const float pe = p + bc.p_eps, y = bc.label[row];
                const float lp = fmaxf(logf(pe), -100.f), l1p = fmaxf(log1pf(-pe), -100.f);
                term = -(y * lp + (1.f - y) * l1p);
            }
            lred[32 * wv + i] = term;
        }
        __syncthreads();
        if (threadIdx.x < 64) {  // fixed-order tree over the workgroup's 128 rows
            float v = lred[threadIdx.x] + lred[threadIdx.x + 64];
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
            if (threadIdx.x == 0) bc.partial[blockIdx.x] = v;
        }
    }
}

extern "C" int rp_mlp_tail_fits(int n_hidden, int width, int64_t ldin) {
    return (n_hidden >= 1 && n_hidden <= MT_MAXL && width == 64 && ldin % 4 == 0) ? 1 : 0;
}

// logit[M,1] = head(relu-chain(hin)), hidden outputs h_out[l] [M,64] saved.  W_hidden[l]: [64,64] (out,in), ldw[l] floats per row.
static int tail_fwd_launch(const float *hin, int64_t ldin, int n_hidden, const float *const *W_hidden, const int64_t *ldw,
                           const float *const *b_hidden, float *const *h_out, const float *w_out, const float *b_out,
                           float *logit, int64_t M, const TailBce *bce, rp_stream_t stream) {
    RP_REQUIRE(hin && W_hidden && ldw && b_hidden && h_out && w_out && (logit || bce), "mlp_tail_fwd: null pointer");
    if (!rp_mlp_tail_fits(n_hidden, 64, ldin) || !rp_aligned16(hin))
        return rp_fail(RP_ERR_UNSUPPORTED, "mlp_tail_fwd: 1..3 hidden layers of width 64, 16-byte aligned rows");
    if (M == 0) return RP_OK;
    TailFwdArgs a;
    for (int l = 0; l < MT_MAXL; ++l) {
        a.W[l] = nullptr;
        a.b[l] = nullptr;
        a.h[l] = nullptr;
        a.ldw[l] = 0;
    }
    for (int l = 0; l < n_hidden; ++l) {
        RP_REQUIRE(W_hidden[l] && h_out[l] && ldw[l] >= 64 && ldw[l] % 4 == 0 && rp_aligned16(W_hidden[l]),
                   "mlp_tail_fwd: layer %d invalid", l);
        a.W[l] = W_hidden[l];
        a.b[l] = b_hidden[l];
        a.h[l] = h_out[l];
        a.ldw[l] = ldw[l];
    }
    const dim3 grid((unsigned)rp_cdiv(M, 128));
    hipStream_t s = (hipStream_t)stream;
    TailBce bc{};
    if (bce != nullptr) bc = *bce;
#define TAIL_FWD(LL, BB) hipLaunchKernelGGL((mlp_tail_fwd_kernel<LL, BB>), grid, dim3(256), 0, s, hin, ldin, a, w_out, b_out, logit, M, bc)
    if (bce == nullptr) {
        if (n_hidden == 1) TAIL_FWD(1, false);
        else if (n_hidden == 2) TAIL_FWD(2, false);
        else TAIL_FWD(3, false);
    } else {
        if (n_hidden == 1) TAIL_FWD(1, true);
        else if (n_hidden == 2) TAIL_FWD(2, true);
        else TAIL_FWD(3, true);
    }
#undef TAIL_FWD
    RP_LAUNCH_CHECK("mlp_tail_fwd");
    return RP_OK;
}

extern "C" int rp_mlp_tail_fwd(const float *hin, int64_t ldin, int n_hidden, const float *const *W_hidden, const int64_t *ldw,
                               const float *const *b_hidden, float *const *h_out, const float *w_out, const float *b_out,
                               float *logit, int64_t M, rp_stream_t stream) {
    RP_REQUIRE(logit != nullptr, "mlp_tail_fwd: null pointer");
    return tail_fwd_launch(hin, ldin, n_hidden, W_hidden, ldw, b_hidden, h_out, w_out, b_out, logit, M, nullptr, stream);
}

// partial sums the loss head writes: one per 128 rows
extern "C" int rp_mlp_tail_loss_partials(int64_t M) { return (int)rp_cdiv(M > 0 ? M : 1, 128); }

// The tail's forward with the model's loss head inside: pred [M] = sigmoid(addends[0] + .. + tail logit), partial
// [rp_mlp_tail_loss_partials(M)] = the workgroups' sums of the BCE terms of (pred + p_eps, label) — rp_loss_finish(partial, n,
// weight / M) gives the mean loss.  (deepfm.py:61-66 on top of layers/deep.py:61-84: three launches of the step in one.)
extern "C" int rp_mlp_tail_fwd_bce(const float *hin, int64_t ldin, int n_hidden, const float *const *W_hidden, const int64_t *ldw,
                                   const float *const *b_hidden, float *const *h_out, const float *w_out, const float *b_out,
                                   const float *const *addends, int n_addends, const float *label, float p_eps, float *pred,
                                   float *partial, int64_t M, rp_stream_t stream) {
    RP_REQUIRE(label && pred && partial, "mlp_tail_fwd_bce: null pointer");
    RP_REQUIRE(n_addends >= 0 && n_addends <= 3 && (n_addends == 0 || addends != nullptr), "mlp_tail_fwd_bce: 0..3 other logit addends");
    TailBce bc{};
    for (int q = 0; q < n_addends; ++q) {
        RP_REQUIRE(addends[q] != nullptr, "mlp_tail_fwd_bce: null addend %d", q);
        bc.add[q] = addends[q];
    }
    bc.n_add = n_addends;
    bc.label = label;
    bc.p_eps = p_eps;
    bc.pred = pred;
    bc.partial = partial;
    return tail_fwd_launch(hin, ldin, n_hidden, W_hidden, ldw, b_hidden, h_out, w_out, b_out, nullptr, M, &bc, stream);
}

// ------------------------------------------------------------------------------------------------ backward
struct TailBwdArgs {
    const float *W[MT_MAXL];   // hidden weights [64 out, 64 in]
    const float *act[MT_MAXL + 1];  // act[0] = hin (input of the tail), act[l] = output of hidden layer l (l = 1..L), [M,64]
    int64_t ldw[MT_MAXL];
    int64_t ldact0;
};

// partial workspace layout per workgroup g:  P[g][L][64*64] (dW), then Pb[g][L][64] (db), then Pw[g][64] (dw_out), Pz[g] (db_out)
// (round 4: the W_l^T operand of the dgrad lives in REGISTERS — 24 fragments per lane, read from global memory with the
//  transposition in the addressing (8 coalesced 128-byte row segments per fragment), split in place — instead of in 27 KB
//  of LDS written with 48 two-byte stores per thread and layer: the two activation tiles are then all the LDS a workgroup
//  holds (70 KB), TWO workgroups fit a CU and all 512 of a 65536-row launch are resident at once.  With one workgroup per
//  CU — one wave per SIMD — nothing hid the kernel's chains of LDS round trips: 67 us for ~6 us of matrix work.)
// the loss head's backward folded into the tail's (rp_mlp_tail_bwd_bce): dz[row] is formed from (pred, label, the loss
// gradient) with rp_sigmoid_bce_bwd's arithmetic instead of being read, and written out for the other logit addends
struct TailDz {
    const float *pred, *label, *gloss;  // pred == nullptr: read dz
    float p_eps, scale;
    float *dz_out;
};

template <int L>
__global__ __launch_bounds__(256, 2) void mlp_tail_bwd_kernel(const float *__restrict__ dz, TailBwdArgs a,
                                                              const float *__restrict__ wout, float *__restrict__ dhin,
                                                              int64_t lddh, float *__restrict__ P, int64_t M, int nwg, TailDz tz) {
    __shared__ __attribute__((aligned(16))) float T0[4][32][MT_CT];      // dpre_l  (rows of the four waves back to back)
    __shared__ __attribute__((aligned(16))) float T1[4][32][MT_CT];      // a_{l-1}
    __shared__ float dzs[128];                                           // dz of the workgroup's rows (0 beyond M)
    const int wv = threadIdx.x >> 6, l = threadIdx.x & 63, i = l & 31, h = l >> 5;
    const int64_t row0 = (int64_t)blockIdx.x * 128;
    const int64_t row = row0 + 32 * wv + i;
    const bool rok = row < M;
    const int64_t rowc = rok ? row : M - 1;
    const int mt = wv & 1, ntw = wv >> 1;  // this wave's 32 x 32 tile of every dW_l: out rows mt*32.., in columns ntw*32..
    float (*T0f)[MT_CT] = reinterpret_cast<float (*)[MT_CT]>(&T0[0][0][0]);  // [128][MT_CT]
    float (*T1f)[MT_CT] = reinterpret_cast<float (*)[MT_CT]>(&T1[0][0][0]);
    float *Pg = P + (int64_t)blockIdx.x * ((int64_t)L * 64 * 64 + (int64_t)L * 64 + 64 + 1);
    // ---- head: dpre_L = dz * wout, masked by a_L > 0; dw_out, db_out partials
    float dzr = 0.f;
    if (tz.pred == nullptr) {
        dzr = rok ? dz[row] : 0.f;
    } else if (rok) {
        const float g = tz.gloss[0] * tz.scale;
        const float p = tz.pred[row], y = tz.label[row], pe = p + tz.p_eps;
        dzr = (pe - y) / fmaxf((1.f - pe) * pe, 1e-12f) * g;
        dzr *= p * (1.f - p);
        if (h == 0 && tz.dz_out != nullptr) tz.dz_out[row] = dzr;
    }
    if (h == 0) dzs[32 * wv + i] = dzr;
    f32x8 dp[4];  // dpre of the current layer, A layout (lane = row, k = 16 ks + 8 h ..)
    {
        const float *src = a.act[L] + rowc * 64 + 8 * h;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const f32x4 v0 = *reinterpret_cast<const f32x4 *>(src + ks * 16), v1 = *reinterpret_cast<const f32x4 *>(src + ks * 16 + 4);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float av = e < 4 ? v0[e] : v1[e - 4];
                dp[ks][e] = av > 0.f ? dzr * wout[ks * 16 + 8 * h + e] : 0.f;
                T1[wv][i][ks * 16 + 8 * h + e] = rok ? dzr * av : 0.f;  // dz * a_L: its column sums are dw_out
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < 64) {  // dw_out[k] partial = sum over the workgroup's 128 rows, fixed order
        float s = 0.f;
        for (int r = 0; r < 128; ++r) s += T1f[r][threadIdx.x];
        Pg[(int64_t)L * 64 * 64 + (int64_t)L * 64 + threadIdx.x] = s;
    } else if (threadIdx.x == 64) {
        float s = 0.f;
        for (int r = 0; r < 128; ++r) s += dzs[r];  // (the same order as a serial walk over dz)
        Pg[(int64_t)L * 64 * 64 + (int64_t)L * 64 + 64] = s;
    }
#pragma unroll
    for (int ly = L; ly >= 1; --ly) {
        __syncthreads();  // T0 / T1 / Wt of the previous layer are free
        // dpre_ly -> T0 (row-major = the A layout), a_{ly-1} -> T1, W_ly^T -> Wt
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int e = 0; e < 8; ++e) T0[wv][i][ks * 16 + 8 * h + e] = dp[ks][e];
        {
            const float *src = (ly - 1 == 0 ? a.act[0] + rowc * a.ldact0 : a.act[ly - 1] + rowc * 64) + 8 * h;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const f32x4 v0 = *reinterpret_cast<const f32x4 *>(src + ks * 16), v1 = *reinterpret_cast<const f32x4 *>(src + ks * 16 + 4);
                *reinterpret_cast<f32x4 *>(&T1[wv][i][ks * 16 + 8 * h]) = rok ? v0 : f32x4{0.f, 0.f, 0.f, 0.f};
                *reinterpret_cast<f32x4 *>(&T1[wv][i][ks * 16 + 8 * h + 4]) = rok ? v1 : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        // W_ly^T fragments: lane (i, h) of fragment (nt, ks) holds W[out = 16 ks + 8 h + e][in = 32 nt + i], e = 0..7
        bf16x8 wf[2][4][3];
        {
            const float *Wl = a.W[ly - 1];
            const int64_t ldw = a.ldw[ly - 1];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    f32x8 v;
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = Wl[(int64_t)(ks * 16 + 8 * h + e) * ldw + nt * 32 + i];
                    bf_split8<3>(v, wf[nt][ks]);
                }
        }
        __syncthreads();
        // ---- weight gradient tile of this wave: dW[out = mt*32 + .., in = ntw*32 + ..] = sum over 128 rows
        //      A = dpre^T (lane = out column, 8 consecutive rows per k-step), B = a^T (lane = in column)
        f32x16 wacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) wacc[r] = 0.f;
        float bsum = 0.f;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            f32x8 va, vb;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                va[e] = T0f[ks * 16 + 8 * h + e][mt * 32 + i];
                vb[e] = T1f[ks * 16 + 8 * h + e][ntw * 32 + i];
            }
            bsum += ((va[0] + va[1]) + (va[2] + va[3])) + ((va[4] + va[5]) + (va[6] + va[7]));
            bf16x8 pa_[3], pb_[3];
            bf_split8<3>(va, pa_);
            bf_split8<3>(vb, pb_);
#pragma unroll
            for (int pr = 0; pr < 6; ++pr)
                wacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa_[BfProd<6>::pa(pr)], pb_[BfProd<6>::pb(pr)], wacc, 0, 0, 0);
        }
        {   // partials: C layout col (in) = ntw*32 + i, row (out) = mt*32 + (r&3) + 8*(r>>2) + 4*h
            float *Pw = Pg + (int64_t)(ly - 1) * 64 * 64;
#pragma unroll
            for (int r = 0; r < 16; ++r) Pw[(mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * 64 + ntw * 32 + i] = wacc[r];
            bsum += __shfl_xor(bsum, 32, 64);  // the two row halves
            if (ntw == 0 && h == 0) Pg[(int64_t)L * 64 * 64 + (int64_t)(ly - 1) * 64 + mt * 32 + i] = bsum;
        }
        // ---- dgrad: da_{ly-1} = dpre_ly . W_ly  (contraction over the output index), masked by a_{ly-1} > 0
        f32x16 acc[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8 af[3];
            bf_split8<3>(dp[ks], af);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
                for (int pr = 0; pr < 6; ++pr)
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[BfProd<6>::pa(pr)], wf[nt][ks][BfProd<6>::pb(pr)], acc[nt], 0, 0, 0);
            }
        }
        __syncthreads();  // every wave has finished reading T0 (wgrad): it now takes the masked gradient
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int n = nt * 32 + i;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rl = (r & 3) + 8 * (r >> 2) + 4 * h;
                const float v = T1[wv][rl][n] > 0.f ? acc[nt][r] : 0.f;
                if (ly - 1 == 0) {
                    const int64_t m = row0 + 32 * wv + rl;
                    if (m < M) dhin[m * lddh + n] = v;
                } else {
                    T0[wv][rl][n] = v;
                }
            }
        }
        if (ly - 1 > 0) {
            __syncthreads();
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const f32x4 v0 = *reinterpret_cast<const f32x4 *>(&T0[wv][i][ks * 16 + 8 * h]);
                const f32x4 v1 = *reinterpret_cast<const f32x4 *>(&T0[wv][i][ks * 16 + 8 * h + 4]);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    dp[ks][e] = v0[e];
                    dp[ks][4 + e] = v1[e];
                }
            }
        }
    }
}

// out[e] = sum over workgroups of P[g][e], fixed order (16 slices x 16 elements per block, as the wgrad second stage)
__global__ __launch_bounds__(256) void mlp_tail_reduce_kernel(const float *__restrict__ P, int nwg, int64_t per, float *__restrict__ out) {
    __shared__ float red[16][17];
    const int c = threadIdx.x & 15, q = threadIdx.x >> 4;
    const int64_t e = (int64_t)blockIdx.x * 16 + c;
    float s = 0.f;
    if (e < per)
        for (int g = q; g < nwg; g += 16) s += P[(int64_t)g * per + e];
    red[q][c] = s;
    __syncthreads();
    if (q != 0 || e >= per) return;
#pragma unroll
    for (int j = 1; j < 16; ++j) s += red[j][c];
    out[e] = s;
}

extern "C" int rp_mlp_tail_bwd_workspace_bytes(int64_t M, int n_hidden, size_t *bytes) {
    RP_REQUIRE(bytes && M >= 0 && n_hidden >= 1 && n_hidden <= MT_MAXL, "mlp_tail_bwd_workspace_bytes: bad argument");
    const int64_t per = (int64_t)n_hidden * 64 * 64 + (int64_t)n_hidden * 64 + 64 + 1;
    *bytes = (size_t)(rp_cdiv(M > 0 ? M : 1, 128) * per) * sizeof(float) + 256;
    return RP_OK;
}

// dz [M] (gradient of the logit) -> dhin [M, 64] (gradient w.r.t. the tail's input, already masked by hin > 0: hin is a
// ReLU output), and `grads` = [dW_0 (64*64) | .. | dW_{L-1} | db_0 (64) | .. | db_{L-1} | dw_out (64) | db_out (1)] packed.
// acts[0] = hin (row stride ldact0), acts[l] = output of hidden layer l (row stride 64).
// parts: 1 = the per-workgroup launch (dhin + partial sums into the workspace), 2 = the second stage (partials -> grads),
// 3 = both.  The second stage depends on nothing but the workspace: a captured step issues it beside the kernels that
// follow the first one (rec_pangu_amd/hip.py: mlp_tail_bwd inside a launch plan).
static int tail_bwd_launch(const float *dz, const TailDz &tz, int n_hidden, const float *const *W_hidden, const int64_t *ldw,
                           const float *const *acts, int64_t ldact0, const float *w_out, float *dhin, int64_t lddh,
                           float *grads, int64_t M, void *workspace, size_t workspace_bytes, int parts, rp_stream_t stream) {
    RP_REQUIRE(parts >= 1 && parts <= 3, "mlp_tail_bwd_parts: parts must be 1, 2 or 3");
    RP_REQUIRE((dz || tz.pred) && W_hidden && ldw && acts && w_out && dhin && grads && workspace, "mlp_tail_bwd: null pointer");
    if (!rp_mlp_tail_fits(n_hidden, 64, ldact0) || lddh < 64)
        return rp_fail(RP_ERR_UNSUPPORTED, "mlp_tail_bwd: 1..3 hidden layers of width 64");
    RP_REQUIRE(M >= 1, "mlp_tail_bwd: M must be positive");
    size_t need = 0;
    rp_mlp_tail_bwd_workspace_bytes(M, n_hidden, &need);
    RP_REQUIRE(workspace_bytes >= need, "mlp_tail_bwd: workspace %zu < %zu bytes", workspace_bytes, need);
    TailBwdArgs a;
    for (int l = 0; l < MT_MAXL; ++l) {
        a.W[l] = nullptr;
        a.ldw[l] = 0;
    }
    for (int l = 0; l <= MT_MAXL; ++l) a.act[l] = nullptr;
    for (int l = 0; l < n_hidden; ++l) {
        RP_REQUIRE(W_hidden[l] && ldw[l] >= 64 && ldw[l] % 4 == 0 && rp_aligned16(W_hidden[l]), "mlp_tail_bwd: layer %d invalid", l);
        a.W[l] = W_hidden[l];
        a.ldw[l] = ldw[l];
    }
    for (int l = 0; l <= n_hidden; ++l) {
        RP_REQUIRE(acts[l] && rp_aligned16(acts[l]), "mlp_tail_bwd: activation %d invalid", l);
        a.act[l] = acts[l];
    }
    a.ldact0 = ldact0;
    float *P = reinterpret_cast<float *>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
    const int nwg = (int)rp_cdiv(M, 128);
    const int64_t per = (int64_t)n_hidden * 64 * 64 + (int64_t)n_hidden * 64 + 64 + 1;
    hipStream_t s = (hipStream_t)stream;
    if (parts & 1) {
        if (n_hidden == 1) hipLaunchKernelGGL((mlp_tail_bwd_kernel<1>), dim3(nwg), dim3(256), 0, s, dz, a, w_out, dhin, lddh, P, M, nwg, tz);
        else if (n_hidden == 2) hipLaunchKernelGGL((mlp_tail_bwd_kernel<2>), dim3(nwg), dim3(256), 0, s, dz, a, w_out, dhin, lddh, P, M, nwg, tz);
        else hipLaunchKernelGGL((mlp_tail_bwd_kernel<3>), dim3(nwg), dim3(256), 0, s, dz, a, w_out, dhin, lddh, P, M, nwg, tz);
        RP_LAUNCH_CHECK("mlp_tail_bwd");
    }
    if (parts & 2) {
        hipLaunchKernelGGL(mlp_tail_reduce_kernel, dim3((unsigned)rp_cdiv(per, 16)), dim3(256), 0, s, P, nwg, per, grads);
        RP_LAUNCH_CHECK("mlp_tail_bwd (partials)");
    }
    return RP_OK;
}

extern "C" int rp_mlp_tail_bwd_parts(const float *dz, int n_hidden, const float *const *W_hidden, const int64_t *ldw,
                                     const float *const *acts, int64_t ldact0, const float *w_out, float *dhin, int64_t lddh,
                                     float *grads, int64_t M, void *workspace, size_t workspace_bytes, int parts,
                                     rp_stream_t stream) {
    RP_REQUIRE(dz != nullptr, "mlp_tail_bwd: null pointer");
    TailDz tz{};
    return tail_bwd_launch(dz, tz, n_hidden, W_hidden, ldw, acts, ldact0, w_out, dhin, lddh, grads, M, workspace, workspace_bytes,
                           parts, stream);
}

// The backward of rp_mlp_tail_fwd_bce: the gradient of the logit is formed per row from (pred, label, gloss[0] * weight / M) —
// rp_sigmoid_bce_bwd's arithmetic, bit for bit — and also written to dz_out [M] (may be NULL) for the other logit addends.
// Everything else as rp_mlp_tail_bwd_parts.
extern "C" int rp_mlp_tail_bwd_bce(const float *pred, const float *label, const float *gloss, float p_eps, float weight,
                                   float *dz_out, int n_hidden, const float *const *W_hidden, const int64_t *ldw,
                                   const float *const *acts, int64_t ldact0, const float *w_out, float *dhin, int64_t lddh,
                                   float *grads, int64_t M, void *workspace, size_t workspace_bytes, int parts,
                                   rp_stream_t stream) {
    RP_REQUIRE(pred && label && gloss && M >= 1, "mlp_tail_bwd_bce: null pointer");
    TailDz tz{};
    tz.pred = pred;
    tz.label = label;
    tz.gloss = gloss;
    tz.p_eps = p_eps;
    tz.scale = weight / (float)M;
    tz.dz_out = dz_out;
    return tail_bwd_launch(nullptr, tz, n_hidden, W_hidden, ldw, acts, ldact0, w_out, dhin, lddh, grads, M, workspace,
                           workspace_bytes, parts, stream);
}

extern "C" int rp_mlp_tail_bwd(const float *dz, int n_hidden, const float *const *W_hidden, const int64_t *ldw,
                               const float *const *acts, int64_t ldact0, const float *w_out, float *dhin, int64_t lddh,
                               float *grads, int64_t M, void *workspace, size_t workspace_bytes, rp_stream_t stream) {
    return rp_mlp_tail_bwd_parts(dz, n_hidden, W_hidden, ldw, acts, ldact0, w_out, dhin, lddh, grads, M, workspace,
                                 workspace_bytes, 3, stream);
}
