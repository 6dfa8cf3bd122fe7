// rp_embed_grad_smp (round 6): the first layer's backward on the embedding columns of the BIG tables, SAMPLE-major.
// (reference: aten::embedding_dense_backward of layers/embedding.py:61-63, the `** 2` backward of interaction.py:38-44, the
//  first Linear's dgrad and the embedding columns of its weight gradient, deep.py:62-72 on the input of deepfm.py:57-59)
//
// rp_embed_grad_seg walks the (row-sorted) pair list field by field: every sample's dH and S row is gathered once per FIELD
// (18 non-tiny Criteo fields x 65536 x 512 B = 604 MB of re-reads for 33.5 MB of distinct data: 2/3 of that launch's
// traffic), and the sort buys nothing for a table whose runs are singletons — at batch 65536 the five 2-10 M-row Criteo
// tables hold ~200-1000 duplicate pairs each, the 93 k / 143 k / 286 k-row ones 20-50 %.  For those tables this launch goes
// through the batch in SAMPLE order instead: a unit = 128 consecutive samples x one field,
//     C[b, :]  = dH[b, :] . W1[:, f*64:(f+1)*64]  +  g_fm[b] (S[b, :] - v[b, :])          (one matrix pass: the dgrad)
//     dW1[:, f*64:(f+1)*64] += dH^T . V                                                    (a second one, K = the 128 samples)
// with dH staged ONCE per unit in LDS (coalesced), v gathered straight into the MATRIX-CORE ACCUMULATOR LAYOUT (lane = column
// d, registers = samples: 128 contiguous bytes per half-wave and row) where the same registers are the FM term's operand, the
// weight gradient's B fragments (the k-order of that product is permuted to match) and the layout C leaves the matrix core
// in — so v, S and the gradient rows never pass through LDS, no wave waits for another one inside a unit (one LDS-only
// barrier per unit for the double-buffered dH stage), and the row of a pair goes
//     * straight to grad_arena[key] when the pair is the only one of its table row in the batch (one writer per row), or
//     * to row c of a side buffer when its run in the sorted pair list has two or more pairs (c = its number among those
//       pairs in sorted order; rp_embed_grad_smp_mark, from the sorted list), which rp_embed_grad_reduce_rows then sums per
//       run in list order.
// Deterministic: fixed summation orders, no atomics.  Split-bf16 x6 products (fp32-faithful), as rp_embed_grad_seg.
#include "common.h"
#include "bfsplit.h"

#define SM_MAXF 16
#define SM_LD 68  // floats per staged dH row (272 B: conflict-free ds_read_b128 A fragments)
#define SM_WLD 68  // bf16 per row of the staged W1 pieces (136 B)
struct SmpFields {
    int n;
    int field[SM_MAXF];
    int base[SM_MAXF];  // first arena row of the field's table
};
#define SM_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

// The duplicate marks, COMPACT (three short launches; they depend on the batch's ids only and run with the sort, a step ahead):
//   entry j of field fi's sorted range is a DUPLICATE when its run has two or more pairs; the duplicates are numbered
//   c = 0, 1, ... in (fi, j) order — sorted order, so the pairs of a run are consecutive —
//   dupq[fi * B + b] = c of pair (field[fi], sample b), or -1 when the pair is alone in its run
//   dupkeys[c]       = the key of duplicate c; -1 from the number of duplicates on (the reduce behind the main launch walks
//                      this list: with all n_fields * B entries in it — 14 % duplicates at Criteo shape — that reduce was a
//                      30 us launch over 4096 workgroups of "no entry" keys, 100 us beside the step's other launches)
__device__ __forceinline__ bool smp_is_dup(const int32_t *__restrict__ sk, int64_t n, int64_t q, int32_t k) {
    return (q > 0 && sk[q - 1] == k) || (q + 1 < n && sk[q + 1] == k);
}

// counts[fi * nbk + blk] = duplicates among the 256 entries of block blk of field fi
__global__ __launch_bounds__(256) void embed_grad_smp_count_kernel(const int32_t *__restrict__ sk, int64_t n, int Bi,
                                                                   SmpFields sf, int nbk, int32_t *__restrict__ counts) {
    __shared__ int32_t wc[4];
    const int fi = (int)blockIdx.y, j = (int)blockIdx.x * 256 + (int)threadIdx.x;
    bool dup = false;
    if (j < Bi) {
        const int64_t q = (int64_t)sf.field[fi] * Bi + j;
        dup = smp_is_dup(sk, n, q, sk[q]);
    }
    const uint64_t bal = __ballot(dup);
    if ((threadIdx.x & 63) == 0) wc[threadIdx.x >> 6] = __builtin_popcountll(bal);
    __syncthreads();
    if (threadIdx.x == 0) counts[fi * nbk + (int)blockIdx.x] = (wc[0] + wc[1]) + (wc[2] + wc[3]);
}

// counts -> exclusive offsets in place; counts[nblocks] = the total (one workgroup)
__global__ __launch_bounds__(256) void embed_grad_smp_scan_kernel(int32_t *__restrict__ counts, int nblocks) {
    __shared__ int32_t part[256];
    const int t = (int)threadIdx.x;
    const int per = (nblocks + 255) / 256, a = t * per, b = (a + per < nblocks) ? a + per : nblocks;
    int s = 0;
    for (int e = a; e < b; ++e) s += counts[e];
    part[t] = s;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {  // (Hillis-Steele over 256 partial sums)
        const int v = (t >= d) ? part[t - d] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int run = (t > 0) ? part[t - 1] : 0;
    for (int e = a; e < b; ++e) {
        const int c = counts[e];
        counts[e] = run;
        run += c;
    }
    if (t == 255) counts[nblocks] = part[255];
}

__global__ __launch_bounds__(256) void embed_grad_smp_mark_kernel(const int32_t *__restrict__ sk, const int32_t *__restrict__ sp,
                                                                  int64_t n, int Bi, SmpFields sf, int nbk,
                                                                  const int32_t *__restrict__ offs, int32_t *__restrict__ dupq,
                                                                  int32_t *__restrict__ dupkeys) {
    __shared__ int32_t wc[4];
    const int fi = (int)blockIdx.y, j = (int)blockIdx.x * 256 + (int)threadIdx.x;
    const int wv = (int)threadIdx.x >> 6, l = (int)threadIdx.x & 63;
    const int f = sf.field[fi];
    bool dup = false;
    int32_t k = -1, p = 0;
    if (j < Bi) {
        const int64_t q = (int64_t)f * Bi + j;
        k = sk[q];
        p = sp[q];
        dup = smp_is_dup(sk, n, q, k);
    }
    const uint64_t bal = __ballot(dup);
    if (l == 0) wc[wv] = __builtin_popcountll(bal);
    __syncthreads();
    const int nblocks = sf.n * nbk;
    const int total = offs[nblocks];
    if (j < Bi) {
        int c = offs[fi * nbk + (int)blockIdx.x] + __builtin_popcountll(bal & ((l == 0) ? 0ull : (~0ull >> (64 - l))));
        for (int w2 = 0; w2 < wv; ++w2) c += wc[w2];
        int b = p - f * Bi;
        b = b < 0 ? 0 : (b >= Bi ? Bi - 1 : b);  // (a position outside the field's range would be a caller error)
        dupq[(int64_t)fi * Bi + b] = dup ? c : -1;
        if (dup) dupkeys[c] = k;
        const int g = fi * Bi + j;  // the list's tail: "no entry"
        if (g >= total) dupkeys[g] = -1;
    }
}

#define SM_ROWS 64  // samples per unit
// Addressing: a unit's table rows and gradient rows are addressed as (the field's base: scalar) + (row inside the table x 256 B
// + the lane's column: ONE 32-bit vector operation per row) — the 64-bit form cost eight vector instructions per gathered row
// and the launch was bound by vector-instruction issue (860 per unit and wave, the matrix pipe a third busy).  Hence tables
// of < 2^24 rows (4 GB) only.  LDS per unit: the row INSIDE the table (0 beyond the batch) and the slot: >= 0 the side
// buffer's row, -1 a pair that is alone in its run, -2 a row beyond the batch.
template <bool HAS_FM, bool ACC>
__global__ __launch_bounds__(256, 2) void embed_grad_smp_kernel(
    const int32_t *__restrict__ keys, const int32_t *__restrict__ dupq, int Bi, const float *__restrict__ dh, int64_t lddh,
    const float *__restrict__ w, int64_t ldw, const float *__restrict__ gfm, const float *__restrict__ sum_in,
    const float *__restrict__ arena, float *__restrict__ G, float *__restrict__ dupbuf, float *__restrict__ dwpart,
    SmpFields sf, int fpw, int bpw, int nranges) {
    constexpr int D = 64, R = SM_ROWS;
    __shared__ __attribute__((aligned(16))) float dHs[3][R][SM_LD];
    __shared__ __attribute__((aligned(16))) int32_t kS[3][R];
    __shared__ __attribute__((aligned(16))) int32_t qS[3][R];
    __shared__ __attribute__((aligned(16))) float gS[3][R];
    // the field's W1 slice as three bf16 pieces, [piece][column d][hidden]: the dgrad's B fragments are 16-byte reads (it lived
    // in 48 registers per lane at first: 56 of them spilled beside the prefetched rows)
    __shared__ __attribute__((aligned(16))) __bf16 Wp[3][D][SM_WLD];
    const int tid = threadIdx.x, wv = tid >> 6, l = tid & 63, i = l & 31, h = l >> 5;
    const int mh = wv >> 1, nb = wv & 1;  // this wave: samples 32 mh .. + 31 of the unit x columns 32 nb .. + 31
    const int range = (int)blockIdx.x % nranges, fgrp = (int)blockIdx.x / nranges;
    const int nblk = (Bi + R - 1) / R;
    const int blk0 = range * bpw, blk1 = (blk0 + bpw < nblk) ? blk0 + bpw : nblk;
    const int f_lo = fgrp * fpw, f_hi = (f_lo + fpw < sf.n) ? f_lo + fpw : sf.n;
    const int nbl = blk1 - blk0, nu = (f_hi - f_lo) * nbl;
    if (nu <= 0) return;
    const bool want_dw = dwpart != nullptr;
    // ---- staging of a unit (all 256 threads): dH rows (tid >> 4) + 16 j, float4 column 4 (tid & 15); thread x < 64: the row
    //      and the slot of sample x, thread 64 + x: g_fm of sample x.  A unit's stage travels global -> registers two units
    //      ahead and registers -> LDS one unit ahead (three LDS buffers: one barrier per unit), so that the table rows of
    //      unit u + 1 — whose keys come out of that stage — are in flight across the whole of unit u --------------------------
    const int sr = tid >> 4, sc = 4 * (tid & 15);
    f32x4 st[R / 16];
    int32_t stx = 0;  // wave 0: the key of sample l, wave 1: its slot, wave 2: its g_fm (bits)
    // RAW loads only, from clamped (always valid) addresses, no branch around them and no arithmetic on their results: the
    // masks are applied a unit later, in stash().  (The first version multiplied by the in-range mask right here, inside an
    // `if (u + 2 < nu)`: the compiler then has to wait for the loads at the end of that block — s_waitcnt vmcnt(0), which also
    // drains the gathered rows and the stores issued before: one exposed trip to memory per unit, 3 us of the unit's 7.)
    auto fetch = [&](int u) {
        const int fi = f_lo + u / nbl, blk = blk0 + u % nbl;
        const int b0 = blk * R;
#pragma unroll
        for (int j = 0; j < R / 16; ++j) {
            int b = b0 + sr + 16 * j;
            b = b < Bi ? b : Bi - 1;
            st[j] = *reinterpret_cast<const f32x4 *>(dh + (int64_t)b * lddh + sc);
        }
        int b = b0 + l;
        b = b < Bi ? b : Bi - 1;
        const int32_t *src = (wv == 0) ? keys + (int64_t)sf.field[fi] * Bi
                                       : ((wv == 1) ? dupq + (int64_t)fi * Bi : reinterpret_cast<const int32_t *>(gfm));
        if (wv < (HAS_FM ? 3 : 2)) stx = src[b];  // (wave-uniform)
    };
    auto stash = [&](int buf, int u) {
        const int fi = f_lo + u / nbl, blk = blk0 + u % nbl;
        const int b0 = blk * R;
#pragma unroll
        for (int j = 0; j < R / 16; ++j) {
            const float m = (b0 + sr + 16 * j < Bi) ? 1.f : 0.f;
            *reinterpret_cast<f32x4 *>(&dHs[buf][sr + 16 * j][sc]) = m * st[j];
        }
        const bool ok = b0 + l < Bi;
        if (wv == 0) kS[buf][l] = ok ? stx - sf.base[fi] : 0;
        else if (wv == 1) qS[buf][l] = ok ? stx : -2;
        else if (wv == 2) gS[buf][l] = ok ? __int_as_float(stx) : 0.f;
    };
    const int m0 = 32 * mh;
    const uint32_t col = (uint32_t)(32 * nb + i);
    // the table rows of this wave's 32 samples, in the accumulator layout: register r = sample m0 + (r & 3) + 8 (r >> 2) + 4 h,
    // this lane's column (rows beyond the batch: the table's row 0, finite, never used)
    auto rows_of = [&](int buf, const float *__restrict__ tab, float (&v)[16]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int4 k4 = *reinterpret_cast<const int4 *>(&kS[buf][m0 + 4 * h + 8 * j]);
            const int32_t kk[4] = {k4.x, k4.y, k4.z, k4.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) v[4 * j + e] = tab[((uint32_t)kk[e] << 6) + col];
        }
    };
    // the FM sum rows of a unit in the same layout, loaded one unit ahead as well (in front of the previous unit's weight-
    // gradient pass: issued inside their own unit they were an exposed trip to the Infinity Cache per unit)
    float sv[16];
    auto sums_of = [&](int bb0) {
        const uint32_t r0 = (uint32_t)(bb0 + m0 + 4 * h);
        if (bb0 + R <= Bi) {  // (workgroup-uniform: a full unit needs no clamp, the row offsets are immediates)
            const float *src = sum_in + (r0 * (uint32_t)D + col);
#pragma unroll
            for (int r = 0; r < 16; ++r) sv[r] = src[((r & 3) + 8 * (r >> 2)) * D];
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                uint32_t b = r0 + (uint32_t)((r & 3) + 8 * (r >> 2));
                b = b < (uint32_t)Bi ? b : (uint32_t)Bi - 1u;
                sv[r] = sum_in[b * (uint32_t)D + col];
            }
        }
    };
    f32x16 dwacc[2];  // the weight gradient: dW1[hidden 32 hb + .., f*64 + 32 nb + i] over this wave's samples
    float v[16];
    fetch(0);
    stash(0, 0);
    fetch(nu > 1 ? 1 : 0);
    SM_BARRIER();
    rows_of(0, arena + (int64_t)sf.base[f_lo] * D, v);
    if (HAS_FM) sums_of(blk0 * R);
    for (int u = 0; u < nu; ++u) {
        const int buf = u % 3;
        const int fi = f_lo + u / nbl, ub = u % nbl;
        const int f = sf.field[fi];
        const bool full = (blk0 + ub) * R + R <= Bi;
        const int u1 = (u + 1 < nu) ? u + 1 : nu - 1, u2 = (u + 2 < nu) ? u + 2 : nu - 1;  // (the last units fetch themselves again)
        stash((u + 1) % 3, u1);  // (behind the last unit: into a buffer nobody reads)
        fetch(u2);               // in flight across this unit's work
        if (ub == 0) {
            if (u > 0) SM_BARRIER();  // (a new field: every wave has finished the last unit's reads of the old slice)
            const int wn = tid & 63, wk = 16 * (tid >> 6);
            const float *wsrc = w + (int64_t)f * D + wn;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                f32x8 wv8;
#pragma unroll
                for (int e = 0; e < 8; ++e) wv8[e] = wsrc[(uint32_t)(wk + 8 * half + e) * (uint32_t)ldw];  // (64 x ldw floats < 2^32)
                bf16x8 pc[3];
                bf_split8<3>(wv8, pc);
#pragma unroll
                for (int q = 0; q < 3; ++q) *reinterpret_cast<bf16x8 *>(&Wp[q][wn][wk + 8 * half]) = pc[q];
            }
#pragma unroll
            for (int hb = 0; hb < 2; ++hb)
#pragma unroll
                for (int r = 0; r < 16; ++r) dwacc[hb][r] = 0.f;
        }
        SM_BARRIER();  // the next unit's stage is complete (and every wave has left the unit before this one: its buffer is free)
        float vn[16];
        rows_of((u + 1) % 3, arena + (int64_t)sf.base[f_lo + u1 / nbl] * D, vn);  // (no branch around the loads)
        // ---- dgrad: C[sample, d] = sum_hidden dH[sample, hidden] W1[hidden, f*64 + d] -------------------------------------
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        {
            const float *arow = &dHs[buf][m0 + i][8 * h];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const f32x4 v0 = *reinterpret_cast<const f32x4 *>(arow + 16 * ks);
                const f32x4 v1 = *reinterpret_cast<const f32x4 *>(arow + 16 * ks + 4);
                f32x8 av;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    av[e] = v0[e];
                    av[4 + e] = v1[e];
                }
                bf16x8 a[3], wq[3];
                bf_split8<3>(av, a);
#pragma unroll
                for (int q = 0; q < 3; ++q) wq[q] = *reinterpret_cast<const bf16x8 *>(&Wp[q][32 * nb + i][16 * ks + 8 * h]);
#pragma unroll
                for (int pr = 0; pr < 6; ++pr)
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[BfProd<6>::pa(pr)], wq[BfProd<6>::pb(pr)], acc, 0, 0, 0);
            }
        }
        // ---- the pairs' gradient rows: C + g (S - v) -> the table row's gradient row, or its slot of the side buffer.  One
        //      exec-masked store per destination (short blocks: no skip branches) ----------------------------------------------
        {
            float *__restrict__ Gf = G + (int64_t)sf.base[fi] * D;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int4 q4 = *reinterpret_cast<const int4 *>(&qS[buf][m0 + 4 * h + 8 * j]);
                const int4 k4 = *reinterpret_cast<const int4 *>(&kS[buf][m0 + 4 * h + 8 * j]);
                const int32_t qq[4] = {q4.x, q4.y, q4.z, q4.w}, kk[4] = {k4.x, k4.y, k4.z, k4.w};
                f32x4 g4 = {0.f, 0.f, 0.f, 0.f};
                if (HAS_FM) g4 = *reinterpret_cast<const f32x4 *>(&gS[buf][m0 + 4 * h + 8 * j]);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * j + e;
                    float val = acc[r];
                    if (HAS_FM) val += g4[e] * (sv[r] - v[r]);
                    if (qq[e] == -1) {  // alone in its run: the only writer of this gradient row
                        float *dst = Gf + (((uint32_t)kk[e] << 6) + col);
                        *dst = ACC ? *dst + val : val;
                    }
                    if (qq[e] >= 0) dupbuf[((uint32_t)qq[e] << 6) + col] = val;  // (added per run by rp_embed_grad_reduce_rows)
                }
            }
        }
        if (HAS_FM) sums_of((blk0 + u1 % nbl) * R);
        // ---- weight gradient: dW1[hidden, f*64 + d] += sum_sample dH[sample, hidden] v[sample, d] on the f32-INPUT matrix
        //      instruction (32x32x2: k-step r pairs the two samples register r stands for in the two lane halves, so the B
        //      operand IS register r of the gathered rows and the A operand one LDS word — no bf16 split at all: with the
        //      split-bf16 form this launch was bound by VALU issue, 970 instructions per unit and wave of which 440 were
        //      this product's splits; the f32 instruction runs at 1/16 of the bf16 rate, which the matrix pipe has to spare).
        //      Rows beyond the batch: their staged dH rows are zero, their v is masked (the table's row 0 may hold anything) ----
        if (want_dw) {
            if (!full) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int4 q4 = *reinterpret_cast<const int4 *>(&qS[buf][m0 + 4 * h + 8 * j]);
                    const int32_t qq[4] = {q4.x, q4.y, q4.z, q4.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[4 * j + e] = (qq[e] != -2) ? v[4 * j + e] : 0.f;
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float *arow = &dHs[buf][m0 + (r & 3) + 8 * (r >> 2) + 4 * h][i];
                dwacc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(arow[0], v[r], dwacc[0], 0, 0, 0);
                dwacc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(arow[32], v[r], dwacc[1], 0, 0, 0);
            }
        }
        if (want_dw && ub == nbl - 1) {
            // partial [field][range][mh][hidden 64][d 64]; summed in a fixed order by embed_grad_smp_dw_kernel
            float *P = dwpart + (((int64_t)fi * nranges + range) * 2 + mh) * (D * D);
#pragma unroll
            for (int hb = 0; hb < 2; ++hb)
#pragma unroll
                for (int r = 0; r < 16; ++r) P[(32 * hb + (r & 3) + 8 * (r >> 2) + 4 * h) * D + 32 * nb + i] = dwacc[hb][r];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = vn[r];
    }
}

// dw[hidden, field*64 + d] = the fixed-order sum of the field's partials [np][hidden][d].  One workgroup = 64 consecutive
// elements x four quarters of the partial list (a wave each, eight loads in flight per lane), added in order through LDS.
__global__ __launch_bounds__(256) void embed_grad_smp_dw_kernel(const float *__restrict__ dwpart, SmpFields sf, int np,
                                                                float *__restrict__ dw, int64_t lddw) {
    __shared__ float part[4][64];
    const int fi = (int)blockIdx.x, f = sf.field[fi];
    const int q = (int)threadIdx.x >> 6, l = (int)threadIdx.x & 63;
    const int e = (int)blockIdx.y * 64 + l;  // element (hidden = e >> 6, d = e & 63)
    const int per = (np + 3) / 4, c0 = q * per, c1 = (c0 + per < np) ? c0 + per : np;
    const float *p = dwpart + ((int64_t)fi * np) * 4096 + e;
    float s0 = 0.f, s1 = 0.f;
    int ci = c0;
    for (; ci + 8 <= c1; ci += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = p[(int64_t)(ci + u) * 4096];
#pragma unroll
        for (int u = 0; u < 8; u += 2) {
            s0 += v[u];
            s1 += v[u + 1];
        }
    }
    for (; ci < c1; ++ci) s0 += p[(int64_t)ci * 4096];
    part[q][l] = s0 + s1;
    __syncthreads();
    if (q == 0) dw[(int64_t)(e >> 6) * lddw + (int64_t)f * 64 + (e & 63)] = (part[0][l] + part[1][l]) + (part[2][l] + part[3][l]);
}

// blocks of SM_ROWS samples per workgroup / fields per workgroup: ~512 workgroups (two per CU), a field's weight-gradient
// partial covers `bpw` blocks (RP_SMP_BPW / RP_SMP_FPW override)
static void smp_shape(int64_t B, int nf, int *fpw, int *bpw, int *nranges) {
    static const int e_bpw = []() {
        const char *e = getenv("RP_SMP_BPW");
        return e ? atoi(e) : 0;
    }();
    static const int e_fpw = []() {
        const char *e = getenv("RP_SMP_FPW");
        return e ? atoi(e) : 0;
    }();
    const int nblk = (int)rp_cdiv(B, SM_ROWS);
    int f = e_fpw > 0 ? e_fpw : 2;
    f = f > nf ? nf : f;
    const int fgroups = (int)rp_cdiv(nf, f);
    int b = e_bpw > 0 ? e_bpw : (int)rp_cdiv((int64_t)nblk * fgroups, 512);
    b = b < 1 ? 1 : (b > nblk ? nblk : b);
    *fpw = f;
    *bpw = b;
    *nranges = (int)rp_cdiv(nblk, b);
}

static int smp_fields(const int32_t *fields, int n_fields, int F, const int64_t *field_base, const int64_t *field_rows,
                      SmpFields *sf) {
    RP_REQUIRE(fields && n_fields >= 1 && n_fields <= SM_MAXF, "embed_grad_smp: %d fields (1..%d)", n_fields, SM_MAXF);
    sf->n = n_fields;
    for (int j = 0; j < SM_MAXF; ++j) sf->field[j] = sf->base[j] = 0;
    for (int j = 0; j < n_fields; ++j) {
        RP_REQUIRE(fields[j] >= 0 && fields[j] < F && (j == 0 || fields[j] > fields[j - 1]),
                   "embed_grad_smp: fields must be ascending and < F");
        sf->field[j] = fields[j];
        if (field_base != nullptr) {
            RP_REQUIRE(field_base[j] >= 0 && field_base[j] < INT32_MAX && field_rows && field_rows[j] >= 1 &&
                           field_rows[j] < ((int64_t)1 << 24),
                       "embed_grad_smp: table %d needs 1 .. 2^24 - 1 rows inside an int32 arena (32-bit row offsets)", j);
            sf->base[j] = (int)field_base[j];
        }
    }
    return RP_OK;
}

extern "C" int rp_embed_grad_smp_fits(int D, int hidden, int64_t lddh) {
    return (D == 64 && hidden == 64 && lddh % 4 == 0) ? 1 : 0;
}

extern "C" int rp_embed_grad_smp_mark_scratch(int64_t B, int n_fields, size_t *n_int32) {
    RP_REQUIRE(n_int32 && B >= 1 && n_fields >= 1 && n_fields <= SM_MAXF, "embed_grad_smp_mark_scratch: bad argument");
    *n_int32 = (size_t)n_fields * (size_t)rp_cdiv(B, 256) + 8;
    return RP_OK;
}

extern "C" int rp_embed_grad_smp_mark(const int32_t *sorted_keys, const int32_t *sorted_pos, int64_t n, int64_t B,
                                      const int32_t *fields, int n_fields, int32_t *dupq, int32_t *dupkeys, int32_t *scratch,
                                      rp_stream_t stream) {
    RP_REQUIRE(sorted_keys && sorted_pos && dupq && dupkeys && scratch, "embed_grad_smp_mark: null pointer");
    RP_REQUIRE(B >= 1 && B < INT32_MAX && n >= 0 && n < INT32_MAX && n % B == 0 && n / B <= 64,
               "embed_grad_smp_mark: needs field-major positions (n = F * B, F <= 64)");
    RP_REQUIRE((int64_t)n_fields * B < INT32_MAX, "embed_grad_smp_mark: n_fields * B overflows int32");
    SmpFields sf;
    if (int rc = smp_fields(fields, n_fields, (int)(n / B), nullptr, nullptr, &sf)) return rc;
    const int nbk = (int)rp_cdiv(B, 256);
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)nbk, (unsigned)n_fields);
    hipLaunchKernelGGL(embed_grad_smp_count_kernel, grid, dim3(256), 0, s, sorted_keys, n, (int)B, sf, nbk, scratch);
    RP_LAUNCH_CHECK("embed_grad_smp_mark (counts)");
    hipLaunchKernelGGL(embed_grad_smp_scan_kernel, dim3(1), dim3(256), 0, s, scratch, n_fields * nbk);
    RP_LAUNCH_CHECK("embed_grad_smp_mark (scan)");
    hipLaunchKernelGGL(embed_grad_smp_mark_kernel, grid, dim3(256), 0, s, sorted_keys, sorted_pos, n, (int)B, sf, nbk, scratch, dupq,
                       dupkeys);
    RP_LAUNCH_CHECK("embed_grad_smp_mark");
    return RP_OK;
}

extern "C" int rp_embed_grad_reduce_workspace_bytes(int64_t n, int D, size_t *bytes);
extern "C" int rp_embed_grad_reduce_rows(const int32_t *keys, const float *rows, int64_t n, int D, float *grad_arena,
                                         int accumulate, void *workspace, size_t workspace_bytes, rp_stream_t stream);

// workspace: [side buffer: n_fields * B rows of 64 floats][weight-gradient partials][the reduce's piece lists]
static void smp_layout(int64_t B, int nf, size_t *o_dw, size_t *o_red, size_t *total) {
    int fpw, bpw, nranges;
    smp_shape(B, nf, &fpw, &bpw, &nranges);
    size_t off = ((size_t)nf * B * 64 * sizeof(float) + 255) & ~(size_t)255;
    *o_dw = off;
    off += (((size_t)nf * nranges * 2 * 4096 * sizeof(float) + 255) & ~(size_t)255);
    *o_red = off;
    size_t red = 0;
    rp_embed_grad_reduce_workspace_bytes((int64_t)nf * B, 64, &red);
    *total = off + red + 512;
}

extern "C" int rp_embed_grad_smp_workspace_bytes(int64_t B, int n_fields, size_t *bytes) {
    RP_REQUIRE(bytes && B >= 1 && n_fields >= 1 && n_fields <= SM_MAXF, "embed_grad_smp_workspace_bytes: bad argument");
    size_t a, b;
    smp_layout(B, n_fields, &a, &b, bytes);
    return RP_OK;
}

extern "C" int rp_embed_grad_smp(const int32_t *keys, const int32_t *dupq, const int32_t *dupkeys, int64_t B, int F,
                                 const int32_t *fields, const int64_t *field_base, const int64_t *field_rows, int n_fields,
                                 const float *dh, int64_t lddh, const float *w, int64_t ldw, const float *gfm, const float *sum_in, const float *arena, float *grad_arena,
                                 int accumulate, float *dw, int64_t lddw, int phases, void *workspace, size_t workspace_bytes,
                                 rp_stream_t stream) {
    RP_REQUIRE(keys && dupq && dupkeys && dh && w && arena && grad_arena && workspace && field_base && field_rows,
               "embed_grad_smp: null pointer");
    RP_REQUIRE((int64_t)n_fields * B < ((int64_t)1 << 24), "embed_grad_smp: n_fields * B must stay below 2^24 (32-bit side-buffer offsets)");
    RP_REQUIRE(B >= 1 && B < INT32_MAX && F >= 1 && F <= 64 && (int64_t)F * B < INT32_MAX, "embed_grad_smp: bad B / F");
    RP_REQUIRE((gfm == nullptr) == (sum_in == nullptr), "embed_grad_smp: the FM term needs both gfm and sum_in");
    RP_REQUIRE(ldw >= (int64_t)F * 64 && (dw == nullptr || lddw >= (int64_t)F * 64), "embed_grad_smp: weight rows shorter than F * 64");
    if (!rp_embed_grad_smp_fits(64, 64, lddh) || !rp_aligned16(dh) || !rp_aligned16(grad_arena) || !rp_aligned16(arena))
        return rp_fail(RP_ERR_UNSUPPORTED, "embed_grad_smp: needs D = 64, a 64-wide layer and 16-byte aligned operands");
    SmpFields sf;
    if (int rc = smp_fields(fields, n_fields, F, field_base, field_rows, &sf)) return rc;
    size_t o_dw, o_red, need;
    smp_layout(B, n_fields, &o_dw, &o_red, &need);
    RP_REQUIRE(workspace_bytes >= need, "embed_grad_smp: workspace %zu < %zu bytes", workspace_bytes, need);
    char *wbase = reinterpret_cast<char *>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
    float *dupbuf = reinterpret_cast<float *>(wbase);
    float *dwpart = dw ? reinterpret_cast<float *>(wbase + o_dw) : nullptr;
    int fpw, bpw, nranges;
    smp_shape(B, n_fields, &fpw, &bpw, &nranges);
    const unsigned grid = (unsigned)(rp_cdiv(n_fields, fpw) * nranges);
    hipStream_t s = (hipStream_t)stream;
    RP_REQUIRE(phases >= 1 && phases <= 3, "embed_grad_smp: phases = 1 (the main launch), 2 (the launches behind it) or 3 (both)");
    if (phases & 1) {
#define SMP_LAUNCH(FM, AC)                                                                                                  \
    hipLaunchKernelGGL((embed_grad_smp_kernel<FM, AC>), dim3(grid), dim3(256), 0, s, keys, dupq, (int)B, dh, lddh, w, ldw, gfm, \
                       sum_in, arena, grad_arena, dupbuf, dwpart, sf, fpw, bpw, nranges)
        if (gfm != nullptr) {
            if (accumulate) SMP_LAUNCH(true, true);
            else SMP_LAUNCH(true, false);
        } else {
            if (accumulate) SMP_LAUNCH(false, true);
            else SMP_LAUNCH(false, false);
        }
#undef SMP_LAUNCH
        RP_LAUNCH_CHECK("embed_grad_smp");
    }
    // the launches behind it (they touch other rows than rp_embed_grad_seg / _tiny: the caller may run them beside those)
    if (!(phases & 2)) return RP_OK;
    if (dw != nullptr) {
        hipLaunchKernelGGL(embed_grad_smp_dw_kernel, dim3((unsigned)n_fields, 64), dim3(256), 0, s, dwpart, sf, nranges * 2, dw, lddw);
        RP_LAUNCH_CHECK("embed_grad_smp (weight-gradient partials)");
    }
    return rp_embed_grad_reduce_rows(dupkeys, dupbuf, (int64_t)n_fields * B, 64, grad_arena, accumulate, wbase + o_red,
                                     workspace_bytes - (size_t)((wbase + o_red) - reinterpret_cast<char *>(workspace)), stream);
}
