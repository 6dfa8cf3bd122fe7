// Gather backward of the TINY tables, sample-major (round 4; D = 64, fused with the first Linear's dgrad like
// rp_embed_grad_gemm — reference: aten::embedding_dense_backward of layers/embedding.py:61-63 + the autograd of
// interaction.py:38-44 and deep.py:62).
//
// The gradient row of table row r of field f is linear in per-sample quantities:
//     G[r] = (sum_b dH[b]) . W1_f^T  +  sum_b g_fm[b] S[b]  -  (sum_b g_fm[b]) v_r        (b over the samples with id_f[b] = r)
// i.e. G[r] = Sh[r] . W1_f^T + Su[r] - Sg[r] v_r with the 129-wide row sums [Sh | Su | Sg] of z[b] = [dH[b] | g_fm[b] S[b] |
// g_fm[b]].  For a table of a few rows (Criteo: 8 of the 26 fields have 3 .. 105 rows, 31 % of all (sample, field) pairs)
// those sums are ONE-HOT GEMMs,  [Sh | Su | Sg] = onehot_f^T [rows x B] . Z [B x 129]:  the one-hot operand is built in
// registers from the ids (exact in bf16), Z is split into three bf16 pieces (exact), the matrix core accumulates in fp32 in
// a fixed order — deterministic, no atomics — and Z is read ONCE for all tiny tables together (33.5 MB at B = 65536)
// instead of once per pair (the row-sorted kernel spends 128 x 64 x 64 x 6 MFMA products and two gathers per 128 pairs).
//   rp_embed_grad_tiny      partial sums per block of samples ([blocks, 224, 160] fp32 workspace), then one small launch:
//                           fixed-order sum over the blocks, the [129] -> [64] map with W1_f^T in fp32 FMAs, the FM term,
//                           the store into the gradient arena (rows somebody looked up in this batch: column 129 counts).
// rp_embed_grad_gemm then leaves those fields out (skip_mask).  Up to 16 tables of <= 254 rows, 224 rows together.
#include "common.h"
#include "bfsplit.h"

#define ET_MAXF 16
#define ET_ROWS 224   // accumulator rows: 7 m-tiles of 32
#define ET_COLS 160   // 64 (dH sums) + 64 (g S sums) + 1 (g sums) + 1 (lookups of the row) + zero padding: 5 n-tiles of 32
#define ET_LD 72      // bf16 per LDS row: 64 samples + 8 pad (144 B = 9 x 16 B: conflict-free ds_read_b128)
#define ET_CHUNK 64
#define ET_NOROW 254  // rloc of an accumulator row no table owns
#define ET_NOSMP 255  // rid of a sample slot beyond the block's range

struct TinyTables {
    int n, total;
    int field[ET_MAXF];  // field index: the pair of (field, sample b) is keys[field * B + b]
    int base[ET_MAXF];   // first arena row of the table
    int rows[ET_MAXF];   // rows of the table
    int acc0[ET_MAXF];   // its first accumulator row
};

__device__ __forceinline__ void tiny_split3(float v, __bf16 *p0, __bf16 *p1, __bf16 *p2) {
    // truncation split: hi + mid + lo == v exactly (bfsplit.h)
    uint32_t u = __float_as_uint(v);
    *reinterpret_cast<uint16_t *>(p0) = (uint16_t)(u >> 16);
    v -= __uint_as_float(u & 0xFFFF0000u);
    u = __float_as_uint(v);
    *reinterpret_cast<uint16_t *>(p1) = (uint16_t)(u >> 16);
    v -= __uint_as_float(u & 0xFFFF0000u);
    u = __float_as_uint(v);
    *reinterpret_cast<uint16_t *>(p2) = (uint16_t)(u >> 16);
}

__global__ __launch_bounds__(256, 2) void embed_grad_tiny_partial_kernel(
    const int32_t *__restrict__ keys, int64_t B, TinyTables tt, const float *__restrict__ dh, int64_t lddh,
    const float *__restrict__ sum_in, const float *__restrict__ gfm, int64_t per_blk, float *__restrict__ partial) {
    __shared__ __attribute__((aligned(16))) __bf16 Zt[3][ET_COLS][ET_LD];      // [piece][column of z][sample of the chunk]
    __shared__ __attribute__((aligned(8))) uint8_t rid[ET_MAXF][ET_CHUNK];      // local row of every sample, per tiny table
    __shared__ uint8_t rfld[ET_ROWS], rloc[ET_ROWS];                            // accumulator row -> (table slot, local row)
    const int t = threadIdx.x, wv = t >> 6, l = t & 63, i = l & 31, h = l >> 5;
    for (int m = t; m < ET_ROWS; m += 256) {
        int slot = 0, loc = ET_NOROW;
        for (int j = 0; j < tt.n; ++j)
            if (m >= tt.acc0[j] && m < tt.acc0[j] + tt.rows[j]) {
                slot = j;
                loc = m - tt.acc0[j];
            }
        rfld[m] = (uint8_t)slot;
        rloc[m] = (uint8_t)loc;
    }
    for (int e = t; e < 3 * (ET_COLS - 130) * ET_LD; e += 256) {  // the padding columns stay zero for the whole launch
        const int p = e / ((ET_COLS - 130) * ET_LD), rest = e - p * ((ET_COLS - 130) * ET_LD);
        Zt[p][130 + rest / ET_LD][rest % ET_LD] = (__bf16)0.f;
    }
    f32x16 acc[5];
#pragma unroll
    for (int nt = 0; nt < 5; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
    const int64_t s_begin = (int64_t)blockIdx.x * per_blk;
    const int64_t s_end = s_begin + per_blk < B ? s_begin + per_blk : B;
    __syncthreads();
    // the 7 accumulator row tiles are split over TWO workgroups per block of samples (blockIdx.y: tiles 0..3 / 4..6, one per
    // wave): both stage the same rows of z (the second read is an L2 hit), each does half the matrix work
    const int mt = 4 * (int)blockIdx.y + wv;
    const bool mok = mt < ET_ROWS / 32;
    const int fl0 = mok ? rfld[32 * mt + i] : 0, lc0 = mok ? rloc[32 * mt + i] : ET_NOROW;
    const int s = t >> 2, q = t & 3;  // staging: sample s of the chunk, columns 16 q .. 16 q + 15 of dH and of S
    // the chunk's rows travel global -> registers one chunk AHEAD: the loads of chunk c + 1 are issued before the matrix
    // phase of chunk c (one workgroup per CU: nothing else would hide their latency)
    f32x4 vd[4], vs[4];
    float g = 0.f, okf = 0.f;
    int rloc0 = ET_NOSMP, rloc1 = ET_NOSMP;  // this thread's (up to two) entries of the chunk's local-row table
    auto fetch = [&](int64_t c0) {
        const int64_t b = c0 + s;
        const bool ok = b < s_end;
        const int64_t bc = ok ? b : s_end - 1;
        okf = ok ? 1.f : 0.f;
        g = (gfm != nullptr) ? gfm[bc] * okf : 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) vd[e] = *reinterpret_cast<const f32x4 *>(dh + bc * lddh + 16 * q + 4 * e);
        if (sum_in != nullptr) {
#pragma unroll
            for (int e = 0; e < 4; ++e) vs[e] = *reinterpret_cast<const f32x4 *>(sum_in + bc * 64 + 16 * q + 4 * e);
        }
        rloc0 = rloc1 = ET_NOSMP;
        {
            const int idx = t, j = idx >> 6, ss = idx & 63;
            if (idx < tt.n * ET_CHUNK && c0 + ss < s_end) rloc0 = keys[(int64_t)tt.field[j] * B + c0 + ss] - tt.base[j];
        }
        {
            const int idx = t + 256, j = idx >> 6, ss = idx & 63;
            if (idx < tt.n * ET_CHUNK && c0 + ss < s_end) rloc1 = keys[(int64_t)tt.field[j] * B + c0 + ss] - tt.base[j];
        }
        // (tables 9 .. 16: two more entries per thread)
    };
    fetch(s_begin);
    for (int64_t c0 = s_begin; c0 < s_end; c0 += ET_CHUNK) {
        {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int col = 16 * q + e;
                tiny_split3(vd[e >> 2][e & 3] * okf, &Zt[0][col][s], &Zt[1][col][s], &Zt[2][col][s]);
                const float uval = (sum_in != nullptr) ? g * vs[e >> 2][e & 3] : 0.f;
                tiny_split3(uval, &Zt[0][64 + col][s], &Zt[1][64 + col][s], &Zt[2][64 + col][s]);
            }
            if (q == 0) tiny_split3(g, &Zt[0][128][s], &Zt[1][128][s], &Zt[2][128][s]);
            if (q == 1) tiny_split3(okf, &Zt[0][129][s], &Zt[1][129][s], &Zt[2][129][s]);  // column 129 counts the lookups
            if (t < tt.n * ET_CHUNK) rid[t >> 6][t & 63] = (uint8_t)rloc0;
            if (t + 256 < tt.n * ET_CHUNK) rid[(t + 256) >> 6][t & 63] = (uint8_t)rloc1;
            for (int idx = t + 512; idx < tt.n * ET_CHUNK; idx += 256) {  // more than 8 tables: straight from memory
                const int j = idx >> 6, ss = idx & 63;
                int loc = ET_NOSMP;
                if (c0 + ss < s_end) loc = keys[(int64_t)tt.field[j] * B + c0 + ss] - tt.base[j];
                rid[j][ss] = (uint8_t)loc;
            }
        }
        if (c0 + ET_CHUNK < s_end) fetch(c0 + ET_CHUNK);  // in flight across the matrix phase below
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            // A operand: lane (i, h) holds onehot[accumulator row 32 mt + i][samples 16 ks + 8 h .. + 7] (bf16 1.0 = 0x3F80)
            const uint64_t r8 = *reinterpret_cast<const uint64_t *>(&rid[fl0][16 * ks + 8 * h]);
            u32x4 aw;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t e0 = (uint32_t)((r8 >> (16 * j)) & 0xFF), e1 = (uint32_t)((r8 >> (16 * j + 8)) & 0xFF);
                aw[j] = (e0 == (uint32_t)lc0 ? 0x00003F80u : 0u) | (e1 == (uint32_t)lc0 ? 0x3F800000u : 0u);
            }
            const bf16x8 a = __builtin_bit_cast(bf16x8, aw);
            if (mok) {  // wave-uniform
#pragma unroll
                for (int nt = 0; nt < 5; ++nt)
#pragma unroll
                    for (int p = 2; p >= 0; --p) {  // smallest pieces first
                        const bf16x8 bq = *reinterpret_cast<const bf16x8 *>(&Zt[p][32 * nt + i][16 * ks + 8 * h]);
                        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bq, acc[nt], 0, 0, 0);
                    }
            }
        }
        __syncthreads();
    }
    // acc[nt][r] = element (row 32 mt + (r & 3) + 8 (r >> 2) + 4 h, column 32 nt + i)
    float *P = partial + (int64_t)blockIdx.x * ET_ROWS * ET_COLS;
    if (mok) {
#pragma unroll
        for (int nt = 0; nt < 5; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                P[(int64_t)(32 * mt + (r & 3) + 8 * (r >> 2) + 4 * h) * ET_COLS + 32 * nt + i] = acc[nt][r];
    }
}

// one workgroup per accumulator row: S = the fixed-order sum of the blocks' partial rows (four independent chains per
// column, combined in a fixed order), then  G[row, d] (+)= sum_h S[h] W1[h, f 64 + d] + S[64 + d] - S[128] v[row, d]
#define ET_FIN_PARTS 4
__global__ __launch_bounds__(ET_COLS * ET_FIN_PARTS) void embed_grad_tiny_finish_kernel(
    const float *__restrict__ partial, int nblk, TinyTables tt, const float *__restrict__ wt, int64_t ldwt,
    const float *__restrict__ arena, float *__restrict__ G, int accumulate, int has_fm, float *__restrict__ hs_out) {
    __shared__ float Sp[ET_FIN_PARTS][ET_COLS];
    __shared__ float S[ET_COLS];
    const int m = blockIdx.x, c = threadIdx.x % ET_COLS, part = threadIdx.x / ET_COLS;
    int slot = 0;
    for (int j = 0; j < tt.n; ++j)
        if (m >= tt.acc0[j] && m < tt.acc0[j] + tt.rows[j]) slot = j;
    const int64_t arow = (int64_t)tt.base[slot] + (m - tt.acc0[slot]);
    {
        const int per = (nblk + ET_FIN_PARTS - 1) / ET_FIN_PARTS;
        const int b0 = part * per, b1 = (b0 + per < nblk) ? b0 + per : nblk;
        float s = 0.f;
#pragma unroll 8
        for (int b = b0; b < b1; ++b) s += partial[((int64_t)b * ET_ROWS + m) * ET_COLS + c];
        Sp[part][c] = s;
    }
    __syncthreads();
    if (threadIdx.x < ET_COLS) S[c] = (Sp[0][c] + Sp[1][c]) + (Sp[2][c] + Sp[3][c]);
    __syncthreads();
    if (hs_out != nullptr && threadIdx.x >= 64 && threadIdx.x < 128) hs_out[m * 64 + (threadIdx.x - 64)] = S[threadIdx.x - 64];
    if (threadIdx.x < 64) {
        const int d = threadIdx.x;
        const float *w = wt + ((int64_t)tt.field[slot] * 64 + d) * ldwt;  // row f 64 + d of W1^T = column of W1
        float a = 0.f;
#pragma unroll 8
        for (int hh = 0; hh < 64; ++hh) a = __builtin_fmaf(S[hh], w[hh], a);
        a += S[64 + d];
        if (has_fm) a -= S[128] * arena[arow * 64 + d];
        float *dst = G + arow * 64 + d;
        // a row nobody looked up in THIS batch is left alone, like the row-sorted path leaves it: under the deferred
        // optimizer it may still hold the waiting gradient of an earlier step
        if (S[129] > 0.f) *dst = accumulate ? *dst + a : a;
    }
}

// The tiny tables' share of the first layer's WEIGHT gradient (round 5, the companion of rp_embed_grad_seg: the forward
// stores no activation):  dw[hidden, f*64 + d] = sum over the table's rows r of  Sh[r, hidden] v_r[d]  with the dH row sums
// Sh the finish launch left in the workspace — a few hundred rows in all: fp32 FMAs in row order.
__global__ __launch_bounds__(256) void embed_grad_tiny_dw_kernel(const float *__restrict__ hs, TinyTables tt,
                                                                 const float *__restrict__ arena, float *__restrict__ dw,
                                                                 int64_t lddw) {
    const int slot = (int)blockIdx.x;
    const int e = (int)blockIdx.y * 256 + (int)threadIdx.x, d = e & 63, hh = e >> 6;  // (d fastest: coalesced row reads)
    const float *hrow = hs + (int64_t)tt.acc0[slot] * 64 + hh;
    const float *vrow = arena + (int64_t)tt.base[slot] * 64 + d;
    float a = 0.f;
    const int nr = tt.rows[slot];
    int r = 0;
    for (; r + 8 <= nr; r += 8) {  // (eight rows' loads in flight: the plain loop waited for every pair of loads in turn)
        float hv[8], vv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            hv[u] = hrow[(int64_t)(r + u) * 64];
            vv[u] = vrow[(int64_t)(r + u) * 64];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) a = __builtin_fmaf(hv[u], vv[u], a);
    }
    for (; r < nr; ++r) a = __builtin_fmaf(hrow[(int64_t)r * 64], vrow[(int64_t)r * 64], a);
    dw[(int64_t)hh * lddw + (int64_t)tt.field[slot] * 64 + d] = a;
}

static int64_t tiny_per_block(int64_t B) {
    int64_t per = rp_cdiv(rp_cdiv(B, 128), ET_CHUNK) * ET_CHUNK;  // ~128 blocks, whole chunks
    return per < ET_CHUNK ? ET_CHUNK : per;
}

extern "C" int rp_embed_grad_tiny_workspace_bytes(int64_t B, size_t *bytes) {
    RP_REQUIRE(bytes && B >= 1, "embed_grad_tiny_workspace_bytes: bad argument");
    *bytes = (size_t)rp_cdiv(B, tiny_per_block(B)) * ET_ROWS * ET_COLS * sizeof(float) + (size_t)ET_ROWS * 64 * sizeof(float) + 256;
    return RP_OK;
}

extern "C" int rp_embed_grad_tiny(const int32_t *keys, int64_t B, const int32_t *tiny_field, const int64_t *tiny_base,
                                  const int32_t *tiny_rows, int n_tiny, const float *dh, int64_t lddh, const float *wt,
                                  int64_t ldwt, const float *gfm, const float *sum_in, const float *arena,
                                  float *grad_arena, int accumulate, float *dw, int64_t lddw, void *workspace,
                                  size_t workspace_bytes, rp_stream_t stream) {
    RP_REQUIRE(keys && tiny_field && tiny_base && tiny_rows && dh && wt && grad_arena && workspace, "embed_grad_tiny: null pointer");
    RP_REQUIRE(n_tiny >= 1 && n_tiny <= ET_MAXF, "embed_grad_tiny: %d tables (1..%d)", n_tiny, ET_MAXF);
    RP_REQUIRE(B >= 1 && B < INT32_MAX, "embed_grad_tiny: bad B");
    RP_REQUIRE(gfm == nullptr || arena != nullptr, "embed_grad_tiny: the FM term needs the arena");
    RP_REQUIRE(lddh % 4 == 0 && ldwt >= 64 && rp_aligned16(dh) && (sum_in == nullptr || rp_aligned16(sum_in)),
               "embed_grad_tiny: dh / sum_in rows must be 16-byte aligned");
    TinyTables tt;
    tt.n = n_tiny;
    int total = 0;
    for (int j = 0; j < ET_MAXF; ++j) {
        tt.field[j] = tt.base[j] = tt.rows[j] = tt.acc0[j] = 0;
        if (j >= n_tiny) continue;
        RP_REQUIRE(tiny_rows[j] >= 1 && tiny_rows[j] <= ET_NOROW, "embed_grad_tiny: table %d has %d rows (1..%d)", j, tiny_rows[j],
                   ET_NOROW);
        RP_REQUIRE(tiny_base[j] >= 0 && tiny_base[j] < INT32_MAX && tiny_field[j] >= 0, "embed_grad_tiny: bad table %d", j);
        tt.field[j] = tiny_field[j];
        tt.base[j] = (int)tiny_base[j];
        tt.rows[j] = tiny_rows[j];
        tt.acc0[j] = total;
        total += tiny_rows[j];
    }
    RP_REQUIRE(total <= ET_ROWS, "embed_grad_tiny: %d rows together (<= %d)", total, ET_ROWS);
    tt.total = total;
    const int64_t per = tiny_per_block(B);
    const int64_t nblk = rp_cdiv(B, per);
    size_t need = 0;
    rp_embed_grad_tiny_workspace_bytes(B, &need);
    RP_REQUIRE(workspace_bytes >= need, "embed_grad_tiny: workspace %zu < %zu bytes", workspace_bytes, need);
    float *partial = reinterpret_cast<float *>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(embed_grad_tiny_partial_kernel, dim3((unsigned)nblk, 2), dim3(256), 0, s, keys, B, tt, dh, lddh, sum_in, gfm, per,
                       partial);
    RP_LAUNCH_CHECK("embed_grad_tiny (partial sums)");
    RP_REQUIRE(dw == nullptr || arena != nullptr, "embed_grad_tiny: the weight gradient needs the arena");
    float *hs = dw ? partial + (size_t)nblk * ET_ROWS * ET_COLS : nullptr;  // [total][64]: the rows' dH sums
    hipLaunchKernelGGL(embed_grad_tiny_finish_kernel, dim3((unsigned)total), dim3(ET_COLS * ET_FIN_PARTS), 0, s, partial, (int)nblk, tt, wt, ldwt, arena,
                       grad_arena, accumulate, gfm != nullptr ? 1 : 0, hs);
    RP_LAUNCH_CHECK("embed_grad_tiny (finish)");
    if (dw != nullptr) {
        for (int j = 0; j < n_tiny; ++j)
            RP_REQUIRE(lddw >= ((int64_t)tt.field[j] + 1) * 64, "embed_grad_tiny: dw rows shorter than field %d's columns", tt.field[j]);
        hipLaunchKernelGGL(embed_grad_tiny_dw_kernel, dim3((unsigned)n_tiny, 16), dim3(256), 0, s, hs, tt, arena, dw, lddw);
        RP_LAUNCH_CHECK("embed_grad_tiny (weight gradient)");
    }
    return RP_OK;
}
