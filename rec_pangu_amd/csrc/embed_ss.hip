// rp_embed_grad_ss (round 6): the first layer's backward on the embedding columns of the MID-SIZE tables in two launches —
// segment sums first as a STREAMING pass, the matrix passes second, over the run pieces only.
// (reference: aten::embedding_dense_backward of layers/embedding.py:61-63, the `** 2` backward of interaction.py:38-44, the
//  first Linear's dgrad and the embedding columns of its weight gradient, deep.py:62-72 on the input of deepfm.py:57-59)
//
// rp_embed_grad_seg does both inside one workgroup-synchronous tile loop: gather 128 x (dH, S, v) rows -> segment sums ->
// two matrix passes -> C tile -> stores, five barriers per tile, two workgroups per CU (77 KB of LDS, 256 registers): the
// gathers of a tile are in flight only during its first phase, and the launch moves its 2 x 256 B per pair out of the
// Infinity Cache at 2.3 TB/s where profiles/microbench/rowgather.hip measures 8 (profiles/r06 notes).  Split:
//   1. embed_segsum_kernel: NO LDS, no barrier, ~100 registers — a 16-lane group walks a chunk of 32 sorted positions, eight
//      at a time: dH row, S row and g_fm of eight pairs in flight per lane, segment sums in registers, and one RECORD
//      [sum dH | sum g S | sum g | key] per run PIECE (a run cut at chunk borders) stored at the sorted index of the piece's
//      last position; every other position gets key -1.  Sixteen waves per CU keep the gathers in flight.
//   2. embed_ss_urows_kernel: one workgroup per tile of 128 UNIQUE rows of a field, taken from the field's unique-row list
//      (first sorted position + key of every run: rp_embed_grad_ss_mark — three short launches that depend on the batch's ids
//      only and run with the sort, a step ahead; made inside this call when the caller has none).  The pieces of a row (the
//      records are linear: a run of 215 positions of a 305-row table is ~7 chunk pieces) are summed in position order — every
//      row's e-th piece in flight together; a run of more than SS_LONG pieces by the whole workgroup —, then
//      rp_embed_grad_seg's second half over the 128 rows: dgrad  Hs . W1_f  and the weight gradient  V^T . Hs  on the matrix
//      core (split-bf16 x6), C + u - s v -> the row's ONE writer.  No run is cut at a tile border: no piece list, no finish
//      launches.  (The first form of this pass walked tiles of 256 sorted POSITIONS: a 305-row table's tile held one or two
//      rows — 2560 mostly empty matrix tiles, eight barriers and three dependent trips each, head / tail pieces for the runs
//      cut at tile borders and two more launches to sum those: 88 + 22 + 5 us inside the step for 47 k unique rows.)
// Same contract and determinism as rp_embed_grad_seg (field-major positions, skip_fields).
#include "common.h"
#include "bfsplit.h"

#define SS_CH 32   // sorted positions per 16-lane group of the segment-sum pass
#define SS_HT 68   // floats per row of the Hs / C tile (272 B: conflict-free ds_read_b128 A fragments)
struct SsFields {
    int n;                    // kept fields
    unsigned char sched[64];  // chunk block -> field: the kept fields, largest table first
    unsigned char rank[64];   // field -> its ordinal among the kept fields in ascending order (record / mark index base)
    unsigned char kept[64];   // ordinal -> field (ascending)
};
#ifndef SS_RB
#define SS_RB 128    // unique rows per workgroup of the second pass (64: -DSS_RB=64, an A/B build)
#endif
#define SS_LONG 16   // a row of more pieces than this is summed by the whole workgroup
struct SsTiles {
    int base[65];  // first tile of sched[o]; base[n] = the tile count (static bound: cdiv(min(B, table rows), SS_RB) per field)
};
#define SS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

// ---------------------------------------------------------------------------------------------------------------------
// pass 1: records.  rh / ru [n_kept * B][64], rs [n_kept * B], at the sorted position (kept fields' compact index space)
// where the piece ENDS; the other positions' slots are not written (pass 2 derives the piece ends from the sorted keys)
// ---------------------------------------------------------------------------------------------------------------------
template <bool HAS_FM>
__global__ __launch_bounds__(256, 4) void embed_segsum_kernel(const int32_t *__restrict__ sk, const int32_t *__restrict__ sp,
                                                              int Bi, const float *__restrict__ dh, int64_t lddh,
                                                              const float *__restrict__ gfm, const float *__restrict__ sum_in,
                                                              SsFields sf, int wpf, float *__restrict__ rh,
                                                              float *__restrict__ ru, float *__restrict__ rs) {
    constexpr int D = 64;
    const int tid = threadIdx.x, t = tid & 15, grp = tid >> 4, c = 4 * t;
    const int o = (int)blockIdx.x / wpf, wt = (int)blockIdx.x - o * wpf;
    const int f = sf.sched[o];
    const int x0 = (wt * 16 + grp) * SS_CH;  // first position of this group's chunk inside the field
    if (x0 >= Bi) return;
    const int cnt = (Bi - x0 < SS_CH) ? Bi - x0 : SS_CH;
    const int64_t q0 = (int64_t)f * Bi + x0;             // sorted index
    const int64_t ob = (int64_t)sf.rank[f] * Bi + x0;    // record index
    f32x4 accH = {0.f, 0.f, 0.f, 0.f}, accU = {0.f, 0.f, 0.f, 0.f};
    float gs = 0.f;
#pragma unroll 1
    for (int it = 0; it < SS_CH / 8; ++it) {
        const int n_it = cnt - 8 * it;  // positions left (<= 0: nothing)
        if (n_it <= 0) break;
        int32_t k[9];
        int b[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const bool ok = j < n_it;
            const int64_t q = q0 + 8 * it + (ok ? j : 0);
            k[j] = ok ? sk[q] : -1;
            int bb = sp[q] - f * Bi;
            bb = bb < 0 ? 0 : (bb >= Bi ? Bi - 1 : bb);  // (a position outside the field's range would be a caller error)
            b[j] = bb;
        }
        // the key behind these eight: the chunk's next position, or "none" at the chunk's end (a piece ends there anyway)
        k[8] = (8 * it + 8 < cnt) ? sk[q0 + 8 * it + 8] : -1;
        f32x4 vh[8], vs[8];
        float g[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            vh[j] = *reinterpret_cast<const f32x4 *>(dh + ((uint32_t)b[j] * (uint32_t)lddh + (uint32_t)c));
            if (HAS_FM) {
                vs[j] = *reinterpret_cast<const f32x4 *>(sum_in + ((uint32_t)b[j] * (uint32_t)D + (uint32_t)c));
                g[j] = gfm[(uint32_t)b[j]];
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const bool ok = j < n_it;
            const float okf = ok ? 1.f : 0.f;
            accH += okf * vh[j];
            if (HAS_FM) {
                const float gj = okf * g[j];
                accU += gj * vs[j];
                gs += gj;
            }
            // a piece ends where the key changes or the chunk ends (k[8] = -1 there; positions beyond n_it hold -1 too)
            const bool ends = ok && (k[j + 1] != k[j]);
            if (ends) {
                const int64_t at = ob + 8 * it + j;
                *reinterpret_cast<f32x4 *>(rh + at * D + c) = accH;
                if (HAS_FM) {
                    *reinterpret_cast<f32x4 *>(ru + at * D + c) = accU;
                    if (t == 0) rs[at] = gs;
                }
            }
            const float keepf = ends ? 0.f : 1.f;
            accH = keepf * accH;
            if (HAS_FM) {
                accU = keepf * accU;
                gs = keepf * gs;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// the unique-row lists of the kept fields, from the sorted keys (they depend on the batch's ids only).  Kept field r (ascending
// ordinal): a run STARTS at position x of the field's sorted range where the key differs from the one in front of it (or
// x = 0); the runs are numbered j = 0, 1, ... in position order:
//     ustart[r * B + j] = x,  ukey[r * B + j] = the run's key,  offs[r * nbk + blk] = runs in front of block blk of 256
//     positions (over ALL kept fields in ordinal order): the field's run count = offs[(r + 1) * nbk] - offs[r * nbk]
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool ss_is_start(const int32_t *__restrict__ sk, int64_t q, int j) { return j == 0 || sk[q] != sk[q - 1]; }

__global__ __launch_bounds__(256) void embed_ss_count_kernel(const int32_t *__restrict__ sk, int Bi, SsFields sf, int nbk,
                                                             int32_t *__restrict__ counts) {
    __shared__ int32_t wc[4];
    const int r = (int)blockIdx.y, j = (int)blockIdx.x * 256 + (int)threadIdx.x;
    bool st = false;
    if (j < Bi) st = ss_is_start(sk, (int64_t)sf.kept[r] * Bi + j, j);
    const uint64_t bal = __ballot(st);
    if ((threadIdx.x & 63) == 0) wc[threadIdx.x >> 6] = __builtin_popcountll(bal);
    __syncthreads();
    if (threadIdx.x == 0) counts[r * nbk + (int)blockIdx.x] = (wc[0] + wc[1]) + (wc[2] + wc[3]);
}

// counts -> exclusive offsets in place; counts[nblocks] = the total (one workgroup)
__global__ __launch_bounds__(256) void embed_ss_scan_kernel(int32_t *__restrict__ counts, int nblocks) {
    __shared__ int32_t part[256];
    const int t = (int)threadIdx.x;
    const int per = (nblocks + 255) / 256, a = t * per, b = (a + per < nblocks) ? a + per : nblocks;
    int s = 0;
    for (int e = a; e < b; ++e) s += counts[e];
    part[t] = s;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
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

__global__ __launch_bounds__(256) void embed_ss_mark_kernel(const int32_t *__restrict__ sk, int Bi, SsFields sf, int nbk,
                                                            const int32_t *__restrict__ offs, int32_t *__restrict__ ustart,
                                                            int32_t *__restrict__ ukey) {
    __shared__ int32_t wc[4];
    const int r = (int)blockIdx.y, j = (int)blockIdx.x * 256 + (int)threadIdx.x;
    const int wv = (int)threadIdx.x >> 6, l = (int)threadIdx.x & 63;
    bool st = false;
    int32_t k = -1;
    if (j < Bi) {
        const int64_t q = (int64_t)sf.kept[r] * Bi + j;
        k = sk[q];
        st = ss_is_start(sk, q, j);
    }
    const uint64_t bal = __ballot(st);
    if (l == 0) wc[wv] = __builtin_popcountll(bal);
    __syncthreads();
    if (st) {
        int c = offs[r * nbk + (int)blockIdx.x] - offs[r * nbk] + __builtin_popcountll(bal & ((l == 0) ? 0ull : (~0ull >> (64 - l))));
        for (int w2 = 0; w2 < wv; ++w2) c += wc[w2];
        ustart[(int64_t)r * Bi + c] = j;
        ukey[(int64_t)r * Bi + c] = k;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// pass 2: 128 unique rows of one field per workgroup, on the matrix core
// ---------------------------------------------------------------------------------------------------------------------
template <bool HAS_FM>
__global__ __launch_bounds__(256, 2) void embed_ss_urows_kernel(
    const float *__restrict__ rh, const float *__restrict__ ru, const float *__restrict__ rs, const int32_t *__restrict__ ustart,
    const int32_t *__restrict__ ukey, const int32_t *__restrict__ offs, int nbk, int Bi, const float *__restrict__ w, int64_t ldw,
    const float *__restrict__ arena, float *__restrict__ G, int accumulate, float *__restrict__ dwpart, SsFields sf, SsTiles st) {
    constexpr int D = 64, RB = SS_RB, MB = RB / 64, NU = RB / 16;  // (RB = 128 or 64)
    __shared__ __attribute__((aligned(16))) float HsT[RB][SS_HT];  // the rows' dH sums; after the matrix passes: the C tile
    __shared__ __attribute__((aligned(16))) float VT[RB][D];       // the rows' table rows (zero rows behind them)
    __shared__ __attribute__((aligned(16))) float red[16][2 * D + 4];  // a long row's partial sums per group
    __shared__ int32_t longrow[RB];
    __shared__ int32_t nlong;
    const int tid = threadIdx.x, t = tid & 15, grp = tid >> 4, c = 4 * t;
    const int wv = tid >> 6, l = tid & 63, i = l & 31, h = l >> 5;
    const int tile = (int)blockIdx.x;
    int o = 0;
    while (o + 1 < sf.n && tile >= st.base[o + 1]) ++o;  // (scalar: <= 63 steps)
    const int f = sf.sched[o], r = sf.rank[f];
    const int ucount = offs[(r + 1) * nbk] - offs[r * nbk];
    const int j0 = (tile - st.base[o]) * RB;
    if (j0 >= ucount) return;  // (workgroup-uniform: the tile count is a static bound)
    const int Mb = (ucount - j0 < RB) ? ucount - j0 : RB;
    const int64_t ob = (int64_t)r * Bi;  // record / mark index base of the field
    const bool want_dw = dwpart != nullptr;
    const int nb = wv & 1, mh = wv >> 1;
    // the rows of this group: m = grp + 16 u — their position ranges and keys first (everything else depends on them)
    int xs[NU], xe[NU], npc[NU];
    int32_t rk[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int m = grp + 16 * u;
        const bool ok = m < Mb;
        const int64_t at = ob + j0 + (ok ? m : 0);
        xs[u] = ustart[at];
        rk[u] = ukey[at];
        const int nx = (ok && j0 + m + 1 < ucount) ? ustart[at + 1] : Bi;
        xe[u] = nx - 1;
        npc[u] = ok ? (xe[u] >> 5) - (xs[u] >> 5) + 1 : 0;  // pieces: one per chunk of SS_CH = 32 positions the run touches
    }
    for (int e = tid; e < RB * SS_HT / 4; e += 256) reinterpret_cast<f32x4 *>(&HsT[0][0])[e] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int e = tid; e < RB * D / 4; e += 256) reinterpret_cast<f32x4 *>(&VT[0][0])[e] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (tid == 0) nlong = 0;
    __syncthreads();
    // ---- (a) the rows' sums: piece e of each of this group's (up to 8) rows in flight together, e = 0, 1, ... in position
    //      order; a row of more than SS_LONG pieces is left to the whole workgroup (b) --------------------------------------
    f32x4 aH[NU], aU[NU], vv[NU];
    float aS[NU];
    int maxnp = 0;
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        aH[u] = aU[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        aS[u] = 0.f;
        vv[u] = *reinterpret_cast<const f32x4 *>(arena + (int64_t)((npc[u] > 0) ? rk[u] : 0) * D + c);
        const bool lg = npc[u] > SS_LONG;
        if (lg && t == 0) longrow[atomicAdd(&nlong, 1)] = grp + 16 * u;
        if (lg) npc[u] = -npc[u];  // (marked: skipped below)
        maxnp = npc[u] > maxnp ? npc[u] : maxnp;
    }
    for (int e = 0; e < maxnp; ++e) {
        f32x4 vh[NU], vu[NU];
        float s1[NU];
#pragma unroll
        for (int u = 0; u < NU; ++u)
            if (e < npc[u]) {
                const int pe = (xs[u] | (SS_CH - 1)) + SS_CH * e;
                const int64_t at = ob + (pe < xe[u] ? pe : xe[u]);
                vh[u] = *reinterpret_cast<const f32x4 *>(rh + at * D + c);
                if (HAS_FM) {
                    vu[u] = *reinterpret_cast<const f32x4 *>(ru + at * D + c);
                    s1[u] = rs[at];
                }
            }
#pragma unroll
        for (int u = 0; u < NU; ++u)
            if (e < npc[u]) {
                aH[u] += vh[u];
                if (HAS_FM) {
                    aU[u] += vu[u];
                    aS[u] += s1[u];
                }
            }
    }
    __syncthreads();
    // (the field's W1 slice as bf16 pieces — loaded here, behind the piece loop: in front of it the 48 registers spilled)
    bf16x8 wp[4][3];
    {
        const float *wsrc = w + (int64_t)f * D + 32 * nb + i;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            f32x8 v;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = wsrc[(uint32_t)(16 * ks + 8 * h + e) * (uint32_t)ldw];
            bf_split8<3>(v, wp[ks]);
        }
    }
    // ---- (b) long rows: group g sums pieces g, g + 16, ... (four in flight), the owner adds the 16 partial sums in order ----
    const int nl = nlong;
    for (int q = 0; q < nl; ++q) {
        // (the list's order depends on which group's atomic came first; the sums do not: every row is summed the same way)
        const int m = longrow[q], og = m & 15, ou = m >> 4;
        int lxs = 0, lxe = 0;
        {
            const int64_t at = ob + j0 + m;
            lxs = ustart[at];
            lxe = ((j0 + m + 1 < ucount) ? ustart[at + 1] : Bi) - 1;
        }
        const int np = (lxe >> 5) - (lxs >> 5) + 1;
        f32x4 pH = {0.f, 0.f, 0.f, 0.f}, pU = {0.f, 0.f, 0.f, 0.f};
        float pS = 0.f;
        for (int e0 = grp; e0 < np; e0 += 64) {
            f32x4 vh[4], vu[4];
            float s1[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int e = e0 + 16 * k;
                const int pe = (lxs | (SS_CH - 1)) + SS_CH * (e < np ? e : 0);
                const int64_t at = ob + (pe < lxe ? pe : lxe);
                vh[k] = *reinterpret_cast<const f32x4 *>(rh + at * D + c);
                if (HAS_FM) {
                    vu[k] = *reinterpret_cast<const f32x4 *>(ru + at * D + c);
                    s1[k] = rs[at];
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float okf = (e0 + 16 * k < np) ? 1.f : 0.f;
                pH += okf * vh[k];
                if (HAS_FM) {
                    pU += okf * vu[k];
                    pS += okf * s1[k];
                }
            }
        }
        *reinterpret_cast<f32x4 *>(&red[grp][c]) = pH;
        *reinterpret_cast<f32x4 *>(&red[grp][D + c]) = pU;
        if (t == 0) red[grp][2 * D] = pS;
        __syncthreads();
        if (grp == og) {
            f32x4 sH = {0.f, 0.f, 0.f, 0.f}, sU = {0.f, 0.f, 0.f, 0.f};
            float sS = 0.f;
            for (int g2 = 0; g2 < 16; ++g2) {
                sH += *reinterpret_cast<const f32x4 *>(&red[g2][c]);
                sU += *reinterpret_cast<const f32x4 *>(&red[g2][D + c]);
                sS += red[g2][2 * D];
            }
#pragma unroll
            for (int u = 0; u < NU; ++u)
                if (u == ou) {
                    aH[u] = sH;
                    aU[u] = sU;
                    aS[u] = sS;
                }
        }
        __syncthreads();
    }
    f32x4 E[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int m = grp + 16 * u;
        if (m < Mb) {  // (LDS stores only)
            *reinterpret_cast<f32x4 *>(&HsT[m][c]) = aH[u];
            *reinterpret_cast<f32x4 *>(&VT[m][c]) = vv[u];
        }
        E[u] = HAS_FM ? aU[u] - aS[u] * vv[u] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    SS_BARRIER();  // (A) the tiles are complete
    // ---- dgrad on the matrix core: C[row, d] = sum_hidden Hs[row, hidden] W1[hidden, f*64 + d] ------------------------
    f32x16 ag[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
#pragma unroll
        for (int r2 = 0; r2 < 16; ++r2) ag[mb][r2] = 0.f;
        if (32 * (mh + 2 * mb) >= Mb) continue;  // (workgroup-uniform)
        const float *arow = &HsT[32 * (mh + 2 * mb) + i][8 * h];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const f32x4 v0 = *reinterpret_cast<const f32x4 *>(arow + 16 * ks);
            const f32x4 v1 = *reinterpret_cast<const f32x4 *>(arow + 16 * ks + 4);
            f32x8 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] = v0[e];
                v[4 + e] = v1[e];
            }
            bf16x8 a[3];
            bf_split8<3>(v, a);
#pragma unroll
            for (int pr = 0; pr < 6; ++pr)
                ag[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[BfProd<6>::pa(pr)], wp[ks][BfProd<6>::pb(pr)], ag[mb], 0, 0, 0);
        }
    }
    // ---- weight gradient: dW1^T[d, hidden] = sum_row V[row, d] Hs[row, hidden]  (K = the tile's rows) ------------------
    if (want_dw) {
        f32x16 dwacc;
#pragma unroll
        for (int r2 = 0; r2 < 16; ++r2) dwacc[r2] = 0.f;
        const int dblk = wv >> 1, hblk = wv & 1;
        const int nks = (Mb + 15) >> 4;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
        for (int ks = 0; ks < nks; ++ks) {
            f32x8 va, vb;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                va[e] = VT[16 * ks + 8 * h + e][32 * dblk + i];
                vb[e] = HsT[16 * ks + 8 * h + e][32 * hblk + i];
            }
            bf16x8 a[3], bq[3];
            bf_split8<3>(va, a);
            bf_split8<3>(vb, bq);
#pragma unroll
            for (int pr = 0; pr < 6; ++pr)
                dwacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[BfProd<6>::pa(pr)], bq[BfProd<6>::pb(pr)], dwacc, 0, 0, 0);
        }
        float *P = dwpart + (int64_t)tile * (D * D);  // [d][hidden]
#pragma unroll
        for (int r2 = 0; r2 < 16; ++r2) P[(32 * dblk + (r2 & 3) + 8 * (r2 >> 2) + 4 * h) * D + 32 * hblk + i] = dwacc[r2];
    }
    SS_BARRIER();  // (B) every wave is done reading the Hs tile: it becomes the C tile
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
        if (32 * (mh + 2 * mb) < Mb) {
#pragma unroll
            for (int r2 = 0; r2 < 16; ++r2) HsT[32 * (mh + 2 * mb) + (r2 & 3) + 8 * (r2 >> 2) + 4 * h][32 * nb + i] = ag[mb][r2];
        }
    SS_BARRIER();  // (C)
    // ---- the rows' gradient: C + u - s v, one writer per row ---------------------------------------------------------------
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int m = grp + 16 * u;
        if (m < Mb) {
            const f32x4 val = *reinterpret_cast<const f32x4 *>(&HsT[m][c]) + E[u];
            float *dst = G + (int64_t)rk[u] * D + c;
            const f32x4 out = accumulate ? *reinterpret_cast<const f32x4 *>(dst) + val : val;
            *reinterpret_cast<f32x4 *>(dst) = out;
        }
    }
}

// dw[hidden, f*64 + d] = the fixed-order sum of the field's tile partials [d][hidden] (as many as the field has unique rows for)
__global__ __launch_bounds__(256) void embed_ss_dw_kernel(const float *__restrict__ dwpart, SsFields sf, SsTiles st,
                                                          const int32_t *__restrict__ offs, int nbk, float *__restrict__ dw,
                                                          int64_t lddw) {
    __shared__ float part[4][64];
    const int o = (int)blockIdx.x, f = sf.sched[o], r = sf.rank[f];
    const int q = (int)threadIdx.x >> 6, l = (int)threadIdx.x & 63;
    const int e = (int)blockIdx.y * 64 + l;  // element (d = e >> 6, hidden = e & 63) of the field's block
    const int ucount = offs[(r + 1) * nbk] - offs[r * nbk];
    const int cpf = (ucount + SS_RB - 1) / SS_RB;
    const int per = (cpf + 3) / 4, c0 = q * per, c1 = (c0 + per < cpf) ? c0 + per : cpf;
    const float *p = dwpart + (int64_t)st.base[o] * 4096 + e;
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
    if (q == 0) dw[(int64_t)(e & 63) * lddw + (int64_t)f * 64 + (e >> 6)] = (part[0][l] + part[1][l]) + (part[2][l] + part[3][l]);
}

static int ss_fields(int F, uint64_t skip_fields, const int64_t *field_rows, int64_t B, SsFields *sf, SsTiles *st) {
    memset(sf, 0, sizeof(*sf));
    memset(st, 0, sizeof(*st));
    for (int f = 0; f < F; ++f)
        if (!((skip_fields >> f) & 1u)) {
            sf->rank[f] = (unsigned char)sf->n;
            sf->kept[sf->n++] = (unsigned char)f;
        }
    for (int a = 0; a < sf->n; ++a) {  // largest table first (stable)
        const int f = sf->kept[a];
        int b = a;
        while (b > 0 && field_rows != nullptr && field_rows[sf->sched[b - 1]] < field_rows[f]) {
            sf->sched[b] = sf->sched[b - 1];
            --b;
        }
        sf->sched[b] = (unsigned char)f;
    }
    for (int o = 0; o < sf->n; ++o) {
        int64_t rows = B;  // unique rows of a field <= min(B, table rows)
        if (field_rows != nullptr && field_rows[sf->sched[o]] >= 1 && field_rows[sf->sched[o]] < rows) rows = field_rows[sf->sched[o]];
        st->base[o + 1] = st->base[o] + (int)rp_cdiv(rows, SS_RB);
    }
    return sf->n;
}

// the marks: ustart / ukey [n_kept * B] int32, offs [n_kept * cdiv(B, 256) + 8] int32
extern "C" int rp_embed_grad_ss_mark_sizes(int64_t B, int n_kept, size_t *n_rows, size_t *n_offs) {
    RP_REQUIRE(n_rows && n_offs && B >= 1 && n_kept >= 0 && n_kept <= 64, "embed_grad_ss_mark_sizes: bad argument");
    *n_rows = (size_t)(n_kept > 0 ? n_kept : 1) * (size_t)B;
    *n_offs = (size_t)n_kept * (size_t)rp_cdiv(B, 256) + 8;
    return RP_OK;
}

static int ss_mark_launch(const int32_t *sorted_keys, int64_t B, const SsFields &sf, int32_t *ustart, int32_t *ukey, int32_t *offs,
                          hipStream_t s) {
    const int nbk = (int)rp_cdiv(B, 256);
    const dim3 grid((unsigned)nbk, (unsigned)sf.n);
    hipLaunchKernelGGL(embed_ss_count_kernel, grid, dim3(256), 0, s, sorted_keys, (int)B, sf, nbk, offs);
    RP_LAUNCH_CHECK("embed_grad_ss_mark (counts)");
    hipLaunchKernelGGL(embed_ss_scan_kernel, dim3(1), dim3(256), 0, s, offs, sf.n * nbk);
    RP_LAUNCH_CHECK("embed_grad_ss_mark (scan)");
    hipLaunchKernelGGL(embed_ss_mark_kernel, grid, dim3(256), 0, s, sorted_keys, (int)B, sf, nbk, (const int32_t *)offs, ustart, ukey);
    RP_LAUNCH_CHECK("embed_grad_ss_mark");
    return RP_OK;
}

extern "C" int rp_embed_grad_ss_mark(const int32_t *sorted_keys, int64_t n, int64_t B, uint64_t skip_fields, int32_t *ustart,
                                     int32_t *ukey, int32_t *offs, rp_stream_t stream) {
    RP_REQUIRE(sorted_keys && ustart && ukey && offs, "embed_grad_ss_mark: null pointer");
    RP_REQUIRE(B >= 1 && B < INT32_MAX && n >= 0 && n < INT32_MAX && n % B == 0 && n / B <= 64,
               "embed_grad_ss_mark: needs field-major positions (n = F * B, F <= 64)");
    SsFields sf;
    SsTiles st;
    if (ss_fields((int)(n / B), skip_fields, nullptr, B, &sf, &st) == 0) return RP_OK;
    return ss_mark_launch(sorted_keys, B, sf, ustart, ukey, offs, (hipStream_t)stream);
}

// workspace: [weight-gradient partials: one per tile][records: rh, ru, rs over n_kept * B positions][marks made here]
static void ss_layout(int64_t B, const SsFields &sf, const SsTiles &st, size_t *o_rec, size_t *o_mark, size_t *total) {
    size_t off = ((size_t)st.base[sf.n] * 4096 * sizeof(float) + 255) & ~(size_t)255;
    *o_rec = off;
    const size_t np = (size_t)sf.n * (size_t)B;
    off += 2 * ((np * 64 * sizeof(float) + 255) & ~(size_t)255) + ((np * 4 + 255) & ~(size_t)255);
    *o_mark = off;
    size_t nr = 0, no = 0;
    rp_embed_grad_ss_mark_sizes(B, sf.n, &nr, &no);
    off += 2 * ((nr * 4 + 255) & ~(size_t)255) + ((no * 4 + 255) & ~(size_t)255);
    *total = off + 256;
}

extern "C" int rp_embed_grad_ss_workspace_bytes(int64_t n, int64_t B, int D, uint64_t skip_fields, size_t *bytes) {
    RP_REQUIRE(bytes && n >= 0 && B >= 1 && D == 64 && n % B == 0 && n / B <= 64,
               "embed_grad_ss_workspace_bytes: needs D = 64 and field-major positions (n = F * B, F <= 64)");
    SsFields sf;
    SsTiles st;
    ss_fields((int)(n / B), skip_fields, nullptr, B, &sf, &st);  // (no table sizes: the bound of B rows per field)
    size_t a, b;
    ss_layout(B, sf, st, &a, &b, bytes);
    return RP_OK;
}

extern "C" int rp_embed_grad_ss(const int32_t *sorted_keys, const int32_t *sorted_pos, int64_t n, int64_t B, int D,
                                const float *dh, int64_t lddh, const float *w, int64_t ldw, const float *gfm,
                                const float *sum_in, const float *arena, float *grad_arena, int accumulate,
                                uint64_t skip_fields, const int64_t *field_rows, float *dw, int64_t lddw,
                                const int32_t *ustart, const int32_t *ukey, const int32_t *offs, int phases,
                                void *workspace, size_t workspace_bytes, rp_stream_t stream) {
    RP_REQUIRE(sorted_keys && sorted_pos && dh && w && arena && grad_arena && workspace, "embed_grad_ss: null pointer");
    RP_REQUIRE(phases >= 1 && phases <= 3, "embed_grad_ss: phases = 1 (the segment-sum launch), 2 (the launches behind it) or 3 (both)");
    RP_REQUIRE(B >= 1 && B < INT32_MAX && n >= 0 && n < INT32_MAX, "embed_grad_ss: bad B / n");
    RP_REQUIRE(n % B == 0 && n / B <= 64, "embed_grad_ss: needs field-major positions (n = F * B, F <= 64)");
    RP_REQUIRE((gfm == nullptr) == (sum_in == nullptr), "embed_grad_ss: the FM term needs both gfm and sum_in");
    RP_REQUIRE((ustart == nullptr) == (ukey == nullptr) && (ustart == nullptr) == (offs == nullptr),
               "embed_grad_ss: the marks are (ustart, ukey, offs) of rp_embed_grad_ss_mark, all three or none");
    const int F = (int)(n / B);
    RP_REQUIRE(ldw >= (int64_t)F * 64 && (dw == nullptr || lddw >= (int64_t)F * 64), "embed_grad_ss: weight rows shorter than F * 64");
    if (D != 64 || lddh % 4 != 0 || !rp_aligned16(dh) || !rp_aligned16(grad_arena) || !rp_aligned16(arena) ||
        (sum_in && !rp_aligned16(sum_in)))
        return rp_fail(RP_ERR_UNSUPPORTED, "embed_grad_ss: needs D = 64, a 64-wide layer and 16-byte aligned operands");
    if (n == 0) return RP_OK;
    SsFields sf, sf0;
    SsTiles st, st0;
    if (ss_fields(F, skip_fields, field_rows, B, &sf, &st) == 0) return RP_OK;  // every field is handled elsewhere
    ss_fields(F, skip_fields, nullptr, B, &sf0, &st0);  // (the workspace is sized without the table sizes)
    size_t o_rec, o_mark, need;
    ss_layout(B, sf0, st0, &o_rec, &o_mark, &need);
    RP_REQUIRE(workspace_bytes >= need, "embed_grad_ss: workspace %zu < %zu bytes", workspace_bytes, need);
    char *wbase = reinterpret_cast<char *>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
    float *dwpart = dw ? reinterpret_cast<float *>(wbase) : nullptr;  // [tiles][64 d][64 hidden]
    const size_t np = (size_t)sf.n * (size_t)B;
    const size_t rowb = (np * 64 * sizeof(float) + 255) & ~(size_t)255;
    float *rh = reinterpret_cast<float *>(wbase + o_rec);
    float *ru = reinterpret_cast<float *>(wbase + o_rec + rowb);
    float *rs = reinterpret_cast<float *>(wbase + o_rec + 2 * rowb);
    hipStream_t s = (hipStream_t)stream;
    const int nbk = (int)rp_cdiv(B, 256);
    const int wpf = (int)rp_cdiv(B, 16 * SS_CH);  // workgroups of the first pass per field
    const unsigned grid1 = (unsigned)(sf.n * wpf), grid2 = (unsigned)st.base[sf.n];
    if (phases & 1) {
        if (ustart == nullptr) {  // nobody marked this sort: here, in front of the pass that needs the marks least
            size_t nr = 0, no = 0;
            rp_embed_grad_ss_mark_sizes(B, sf.n, &nr, &no);
            const size_t mb = (nr * 4 + 255) & ~(size_t)255;
            if (int rc = ss_mark_launch(sorted_keys, B, sf, reinterpret_cast<int32_t *>(wbase + o_mark),
                                        reinterpret_cast<int32_t *>(wbase + o_mark + mb),
                                        reinterpret_cast<int32_t *>(wbase + o_mark + 2 * mb), s))
                return rc;
        }
        if (gfm != nullptr)
            hipLaunchKernelGGL((embed_segsum_kernel<true>), dim3(grid1), dim3(256), 0, s, sorted_keys, sorted_pos, (int)B, dh, lddh,
                               gfm, sum_in, sf, wpf, rh, ru, rs);
        else
            hipLaunchKernelGGL((embed_segsum_kernel<false>), dim3(grid1), dim3(256), 0, s, sorted_keys, sorted_pos, (int)B, dh, lddh,
                               gfm, sum_in, sf, wpf, rh, ru, rs);
        RP_LAUNCH_CHECK("embed_grad_ss (segment sums)");
    }
    if (!(phases & 2)) return RP_OK;
    if (ustart == nullptr) {
        size_t nr = 0, no = 0;
        rp_embed_grad_ss_mark_sizes(B, sf.n, &nr, &no);
        const size_t mb = (nr * 4 + 255) & ~(size_t)255;
        ustart = reinterpret_cast<const int32_t *>(wbase + o_mark);
        ukey = reinterpret_cast<const int32_t *>(wbase + o_mark + mb);
        offs = reinterpret_cast<const int32_t *>(wbase + o_mark + 2 * mb);
    }
    if (gfm != nullptr)
        hipLaunchKernelGGL((embed_ss_urows_kernel<true>), dim3(grid2), dim3(256), 0, s, rh, ru, rs, ustart, ukey, offs, nbk, (int)B, w,
                           ldw, arena, grad_arena, accumulate, dwpart, sf, st);
    else
        hipLaunchKernelGGL((embed_ss_urows_kernel<false>), dim3(grid2), dim3(256), 0, s, rh, ru, rs, ustart, ukey, offs, nbk, (int)B, w,
                           ldw, arena, grad_arena, accumulate, dwpart, sf, st);
    RP_LAUNCH_CHECK("embed_grad_ss (rows)");
    if (dw != nullptr) {
        hipLaunchKernelGGL(embed_ss_dw_kernel, dim3((unsigned)sf.n, 64), dim3(256), 0, s, dwpart, sf, st, offs, nbk, dw, lddw);
        RP_LAUNCH_CHECK("embed_grad_ss (weight-gradient partials)");
    }
    return RP_OK;
}
