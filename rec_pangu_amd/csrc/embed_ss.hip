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
//   2. embed_ss_rows_kernel: per tile of 128 sorted positions — compact the tile's pieces (two ballots), merge the pieces
//      of equal keys (the records are linear: a run of 215 positions of a 305-row table is ~7 chunk pieces), then exactly
//      rp_embed_grad_seg's second half over the tile's UNIQUE rows: dgrad  Hs . W1_f  and the weight gradient  V^T . Hs  on
//      the matrix core (split-bf16 x6), C + u - s v -> one writer per row; the tile's first / last row goes to the
//      (head, tail) piece list when its run continues in the neighbour tile (rp_embed_grad_reduce's finish, shared).
// Same contract, determinism and workspace protocol as rp_embed_grad_seg (field-major positions, skip_fields).
#include "common.h"
#include "bfsplit.h"

#define SS_CH 32   // sorted positions per 16-lane group of the segment-sum pass
#define SS_HT 68   // floats per row of the Hs / C tile (272 B: conflict-free ds_read_b128 A fragments)
struct SsFields {
    int n;                    // kept fields
    unsigned char sched[64];  // chunk block -> field: the kept fields, largest table first
    unsigned char rank[64];   // field -> its ordinal among the kept fields in ascending order (record / tile index base)
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
// pass 2: the unique rows of a tile of SS_TILE sorted positions on the matrix core, 128 rows at a time.  The rows and their
// pieces come straight from the sorted keys: a row = a maximal run inside the tile (one ballot + prefix count over the
// positions); its pieces end at the chunk borders inside it and at its last position — no piece list is read.
// ---------------------------------------------------------------------------------------------------------------------
#define SS_TILE 256
template <bool HAS_FM>
__global__ __launch_bounds__(256, 2) void embed_ss_rows_kernel(
    const float *__restrict__ rh, const float *__restrict__ ru, const float *__restrict__ rs, const int32_t *__restrict__ sk,
    int Bi, const float *__restrict__ w, int64_t ldw, const float *__restrict__ arena, float *__restrict__ G, int accumulate,
    float *__restrict__ gpiece, int32_t *__restrict__ gkey, float *__restrict__ dwpart, SsFields sf, int tpf, int T, int cpf) {
    constexpr int D = 64, TILE = SS_TILE, RB = 128, MB = 2;
    __shared__ __attribute__((aligned(16))) float HsT[RB][SS_HT];  // the rows' dH sums; after the matrix passes: the C tile
    __shared__ __attribute__((aligned(16))) float VT[RB][D];       // the rows' table rows (zero rows behind them)
    __shared__ int32_t rxs[TILE + 1];  // row -> its first position inside the tile ([M]: the tile's end)
    __shared__ int32_t rowkey[TILE];
    __shared__ int32_t wcnt[4];
    const int tid = threadIdx.x, t = tid & 15, grp = tid >> 4, c = 4 * t;
    const int wv = tid >> 6, l = tid & 63, i = l & 31, h = l >> 5;
    const int chunk = (int)blockIdx.x;
    const int o = chunk / cpf;
    const int f = sf.sched[o];
    const int j0 = (chunk - o * cpf) * T;
    const int jn = (j0 + T < tpf) ? j0 + T : tpf;
    const int64_t fbase = (int64_t)f * Bi;
    const int64_t ob = (int64_t)sf.rank[f] * Bi;
    const int64_t tile_id0 = (int64_t)sf.rank[f] * tpf;
    const bool want_dw = dwpart != nullptr;
    const int nb = wv & 1, mh = wv >> 1;
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
    f32x16 dwacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) dwacc[r] = 0.f;
    for (int e = tid; e < RB * SS_HT / 4; e += 256) reinterpret_cast<f32x4 *>(&HsT[0][0])[e] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int e = tid; e < RB * D / 4; e += 256) reinterpret_cast<f32x4 *>(&VT[0][0])[e] = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    const uint64_t lt_mask = (l == 0) ? 0ull : (~0ull >> (64 - l));
    for (int j = j0; j < jn; ++j) {
        const int x0 = TILE * j;
        const int npos = (Bi - x0 < TILE) ? Bi - x0 : TILE;
        // ---- (a) the tile's rows: position tid starts one where its key differs from the key in front of it ---------------
        const bool in = tid < npos;
        const int32_t mykey = in ? sk[fbase + x0 + tid] : -1;
        const int32_t prevkey = (in && x0 + tid > 0) ? sk[fbase + x0 + tid - 1] : -1;
        const bool newrow = in && (tid == 0 || mykey != prevkey);
        const uint64_t bal = __ballot(newrow);
        if (l == 0) wcnt[wv] = __builtin_popcountll(bal);
        // the keys in front of and behind the tile (the runs that cross its borders)
        const int32_t kfront = (x0 > 0) ? sk[fbase + x0 - 1] : -1;
        const int32_t kback = (x0 + TILE < Bi) ? sk[fbase + x0 + TILE] : -1;
        SS_BARRIER();
        const int M = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
        if (newrow) {
            int m = __builtin_popcountll(bal & lt_mask);
#pragma unroll
            for (int q = 0; q < 3; ++q) m += (q < wv) ? wcnt[q] : 0;
            rxs[m] = tid;
            rowkey[m] = mykey;
        }
        if (tid == 0) rxs[M] = npos;
        SS_BARRIER();
        const int32_t key0r = rowkey[0], keyLr = rowkey[M - 1];
        const bool head_open = kfront >= 0 && kfront == key0r;
        const bool tail_open = kback >= 0 && kback == keyLr;
        const bool both = M == 1 && head_open && tail_open;  // one run over the whole tile: the head entry carries it
        const int64_t tile_id = tile_id0 + j;
        float *hp = gpiece + tile_id * 2 * D, *tp = hp + D;
        for (int m0 = 0; m0 < M; m0 += RB) {
            const int Mb = (M - m0 < RB) ? M - m0 : RB;  // rows of this batch
            // ---- (b) the rows' sums: the first piece of each of this group's (up to 8) rows in flight together, the
            //      further pieces of a row (a run across chunk borders) behind them ------------------------------------------
            f32x4 E[8];
            {
                f32x4 vh[8], vu[8], vv[8];
                float s1[8];
                int xs[8], xe[8];
                int32_t rk[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int m = grp + 16 * u;
                    const bool ok = m < Mb;
                    xs[u] = ok ? rxs[m0 + m] : 0;
                    xe[u] = ok ? rxs[m0 + m + 1] - 1 : 0;
                    rk[u] = ok ? rowkey[m0 + m] : 0;
                    const int p0 = (xs[u] | (SS_CH - 1)) < xe[u] ? (xs[u] | (SS_CH - 1)) : xe[u];  // the first piece's end
                    const int64_t at = ob + x0 + p0;
                    vh[u] = *reinterpret_cast<const f32x4 *>(rh + at * D + c);
                    if (HAS_FM) {
                        vu[u] = *reinterpret_cast<const f32x4 *>(ru + at * D + c);
                        s1[u] = rs[at];
                    }
                    vv[u] = *reinterpret_cast<const f32x4 *>(arena + (int64_t)rk[u] * D + c);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int m = grp + 16 * u;
                    const bool ok = m < Mb;  // (a row beyond Mb read record 0 of the tile: possibly never written — selects, not factors)
                    f32x4 aH = vh[u], aU = HAS_FM ? vu[u] : f32x4{0.f, 0.f, 0.f, 0.f};
                    float aS = HAS_FM ? s1[u] : 0.f;
                    for (int p = (xs[u] | (SS_CH - 1)) + SS_CH; p - SS_CH < xe[u]; p += SS_CH) {  // (rare: the run's other chunks)
                        const int64_t at = ob + x0 + (p < xe[u] ? p : xe[u]);
                        aH += *reinterpret_cast<const f32x4 *>(rh + at * D + c);
                        if (HAS_FM) {
                            aU += *reinterpret_cast<const f32x4 *>(ru + at * D + c);
                            aS += rs[at];
                        }
                    }
                    if (ok) {  // (LDS stores only)
                        *reinterpret_cast<f32x4 *>(&HsT[m][c]) = aH;
                        *reinterpret_cast<f32x4 *>(&VT[m][c]) = vv[u];
                    }
                    E[u] = HAS_FM ? aU - aS * vv[u] : f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
            // the weight gradient's last k-step reads rows Mb .. (its multiple of 16): zero table rows there
            if (grp < ((Mb + 15) & ~15) - Mb) *reinterpret_cast<f32x4 *>(&VT[Mb + grp][c]) = f32x4{0.f, 0.f, 0.f, 0.f};
            SS_BARRIER();  // (A) the tiles are complete
            // ---- dgrad on the matrix core: C[row, d] = sum_hidden Hs[row, hidden] W1[hidden, f*64 + d] ------------------------
            f32x16 ag[MB];
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
#pragma unroll
                for (int r = 0; r < 16; ++r) ag[mb][r] = 0.f;
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
            // ---- weight gradient: dW1^T[d, hidden] += sum_row V[row, d] Hs[row, hidden]  (K = the batch's rows) ------------
            if (want_dw) {
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
            }
            SS_BARRIER();  // (B) every wave is done reading the Hs tile: it becomes the C tile
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
                if (32 * (mh + 2 * mb) < Mb) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) HsT[32 * (mh + 2 * mb) + (r & 3) + 8 * (r >> 2) + 4 * h][32 * nb + i] = ag[mb][r];
                }
            SS_BARRIER();  // (C)
            // ---- the rows' gradient: C + u - s v.  A run that continues in the neighbour tile goes to the tile's (head, tail)
            //      entry of the piece list instead (the finish launches sum the chains); everything else has one writer ----------
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int m = grp + 16 * u;
                if (m < Mb) {
                    const f32x4 val = *reinterpret_cast<const f32x4 *>(&HsT[m][c]) + E[u];
                    const bool is_head = m0 + m == 0 && head_open;
                    const bool is_tail = m0 + m == M - 1 && tail_open && !both;
                    if (is_head) {
                        *reinterpret_cast<f32x4 *>(hp + c) = val;
                    } else if (is_tail) {
                        *reinterpret_cast<f32x4 *>(tp + c) = val;
                    } else {
                        float *dst = G + (int64_t)rowkey[m0 + m] * D + c;
                        const f32x4 out = accumulate ? *reinterpret_cast<const f32x4 *>(dst) + val : val;
                        *reinterpret_cast<f32x4 *>(dst) = out;
                    }
                }
            }
            SS_BARRIER();  // (D) the C tile is free (the index arrays stay until the tile's last batch is through)
        }
        // entries nobody filled: zero rows (the finish adds them to nothing: key -1)
        if (grp == 1 && !head_open) *reinterpret_cast<f32x4 *>(hp + c) = f32x4{0.f, 0.f, 0.f, 0.f};
        if (grp == 2 && (!tail_open || both)) *reinterpret_cast<f32x4 *>(tp + c) = f32x4{0.f, 0.f, 0.f, 0.f};
        if (tid == 0) {
            gkey[2 * tile_id] = head_open ? key0r : -1;
            gkey[2 * tile_id + 1] = tail_open ? keyLr : -1;
        }
    }
    if (want_dw) {
        float *P = dwpart + (int64_t)chunk * (D * D);
        const int dblk = wv >> 1, hblk = wv & 1;
#pragma unroll
        for (int r = 0; r < 16; ++r) P[(32 * dblk + (r & 3) + 8 * (r >> 2) + 4 * h) * D + 32 * hblk + i] = dwacc[r];
    }
}

// dw[hidden, f*64 + d] = the fixed-order sum of the chunk partials [d][hidden] of field f (as embed_grad_seg_dw_kernel)
__global__ __launch_bounds__(256) void embed_ss_dw_kernel(const float *__restrict__ dwpart, SsFields sf, int cpf,
                                                          float *__restrict__ dw, int64_t lddw) {
    __shared__ float part[4][64];
    const int o = (int)blockIdx.x, f = sf.sched[o];
    const int q = (int)threadIdx.x >> 6, l = (int)threadIdx.x & 63;
    const int e = (int)blockIdx.y * 64 + l;  // element (d = e >> 6, hidden = e & 63) of the field's block
    const int per = (cpf + 3) / 4, c0 = q * per, c1 = (c0 + per < cpf) ? c0 + per : cpf;
    const float *p = dwpart + ((int64_t)o * cpf) * 4096 + e;
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

extern "C" int rp_embed_grad_reduce_workspace_bytes(int64_t n, int D, size_t *bytes);

static int ss_tiles_per_chunk(int tpf) {
    static const int forced = []() {
        const char *e = getenv("RP_SS_TILES");
        return e ? atoi(e) : 0;
    }();
    int T = forced > 0 ? forced : (tpf + 32) / 64;  // ~64 chunks per field
    T = T < 1 ? 1 : (T > 64 ? 64 : T);
    return T > tpf ? tpf : T;
}

static int64_t ss_n_eff(int64_t n, int64_t B) {  // (the piece region is sized through rp_embed_grad_reduce's formula)
    const int64_t F = B > 0 ? n / B : 0;
    const int64_t tiles = F * rp_cdiv(B, SS_TILE);
    return n > tiles * 32 ? n : tiles * 32;
}

// workspace: [the reduce's piece lists][weight-gradient partials][records: rh, ru, rs, pk over n_kept * B positions]
static void ss_layout(int64_t n, int64_t B, int n_kept, size_t *o_dw, size_t *o_rec, size_t *total) {
    size_t red = 0;
    rp_embed_grad_reduce_workspace_bytes(ss_n_eff(n, B), 64, &red);
    const int tpf = (int)rp_cdiv(B, SS_TILE);
    const int64_t cpf = rp_cdiv(tpf, ss_tiles_per_chunk(tpf));
    size_t off = (red + 255) & ~(size_t)255;
    *o_dw = off;
    off += ((size_t)(n / B) * cpf * 4096 * sizeof(float) + 255) & ~(size_t)255;
    *o_rec = off;
    const size_t np = (size_t)n_kept * (size_t)B;
    off += 2 * ((np * 64 * sizeof(float) + 255) & ~(size_t)255) + 2 * ((np * 4 + 255) & ~(size_t)255);
    *total = off + 256;
}

extern "C" int rp_embed_grad_ss_workspace_bytes(int64_t n, int64_t B, int D, uint64_t skip_fields, size_t *bytes) {
    RP_REQUIRE(bytes && n >= 0 && B >= 1 && D == 64 && n % B == 0 && n / B <= 64,
               "embed_grad_ss_workspace_bytes: needs D = 64 and field-major positions (n = F * B, F <= 64)");
    int kept = 0;
    for (int f = 0; f < (int)(n / B); ++f) kept += ((skip_fields >> f) & 1u) ? 0 : 1;
    size_t a, b;
    ss_layout(n, B, kept, &a, &b, bytes);
    return RP_OK;
}

extern "C" int rp_embed_grad_ss(const int32_t *sorted_keys, const int32_t *sorted_pos, int64_t n, int64_t B, int D,
                                const float *dh, int64_t lddh, const float *w, int64_t ldw, const float *gfm,
                                const float *sum_in, const float *arena, float *grad_arena, int accumulate,
                                uint64_t skip_fields, const int64_t *field_rows, float *dw, int64_t lddw, int phases,
                                void *workspace, size_t workspace_bytes, rp_stream_t stream) {
    RP_REQUIRE(sorted_keys && sorted_pos && dh && w && arena && grad_arena && workspace, "embed_grad_ss: null pointer");
    RP_REQUIRE(phases >= 1 && phases <= 3, "embed_grad_ss: phases = 1 (the segment-sum launch), 2 (the launches behind it) or 3 (both)");
    RP_REQUIRE(B >= 1 && B < INT32_MAX && n >= 0 && n < INT32_MAX, "embed_grad_ss: bad B / n");
    RP_REQUIRE(n % B == 0 && n / B <= 64, "embed_grad_ss: needs field-major positions (n = F * B, F <= 64)");
    RP_REQUIRE((gfm == nullptr) == (sum_in == nullptr), "embed_grad_ss: the FM term needs both gfm and sum_in");
    const int F = (int)(n / B);
    RP_REQUIRE(ldw >= (int64_t)F * 64 && (dw == nullptr || lddw >= (int64_t)F * 64), "embed_grad_ss: weight rows shorter than F * 64");
    if (D != 64 || lddh % 4 != 0 || !rp_aligned16(dh) || !rp_aligned16(grad_arena) || !rp_aligned16(arena) ||
        (sum_in && !rp_aligned16(sum_in)))
        return rp_fail(RP_ERR_UNSUPPORTED, "embed_grad_ss: needs D = 64, a 64-wide layer and 16-byte aligned operands");
    if (n == 0) return RP_OK;
    SsFields sf;
    memset(&sf, 0, sizeof(sf));
    int kept[64];
    for (int f = 0; f < F; ++f)
        if (!((skip_fields >> f) & 1u)) {
            sf.rank[f] = (unsigned char)sf.n;
            kept[sf.n++] = f;
        }
    if (sf.n == 0) return RP_OK;  // every field is handled elsewhere
    for (int a = 0; a < sf.n; ++a) {  // largest table first (stable)
        const int f = kept[a];
        int b = a;
        while (b > 0 && field_rows != nullptr && field_rows[sf.sched[b - 1]] < field_rows[f]) {
            sf.sched[b] = sf.sched[b - 1];
            --b;
        }
        sf.sched[b] = (unsigned char)f;
    }
    size_t o_dw, o_rec, need;
    ss_layout(n, B, sf.n, &o_dw, &o_rec, &need);
    RP_REQUIRE(workspace_bytes >= need, "embed_grad_ss: workspace %zu < %zu bytes", workspace_bytes, need);
    char *wbase = reinterpret_cast<char *>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
    const int tpf = (int)rp_cdiv(B, SS_TILE);
    const int T = ss_tiles_per_chunk(tpf);
    const int cpf = (int)rp_cdiv(tpf, T);
    const int64_t n_eff = ss_n_eff(n, B);
    const int64_t nb0 = (int64_t)sf.n * tpf;
    float *piece0 = reinterpret_cast<float *>(wbase);
    int32_t *key0 = reinterpret_cast<int32_t *>(piece0 + nb0 * 2 * D);
    float *dwpart = dw ? reinterpret_cast<float *>(wbase + o_dw) : nullptr;  // [chunks][64 d][64 hidden]
    const size_t np = (size_t)sf.n * (size_t)B;
    const size_t rowb = (np * 64 * sizeof(float) + 255) & ~(size_t)255;
    float *rh = reinterpret_cast<float *>(wbase + o_rec);
    float *ru = reinterpret_cast<float *>(wbase + o_rec + rowb);
    float *rs = reinterpret_cast<float *>(wbase + o_rec + 2 * rowb);
    hipStream_t s = (hipStream_t)stream;
    const int wpf = (int)rp_cdiv(B, 16 * SS_CH);  // workgroups of the first pass per field
    const unsigned grid1 = (unsigned)(sf.n * wpf), grid2 = (unsigned)(sf.n * cpf);
    if (phases & 1) {
        if (gfm != nullptr)
            hipLaunchKernelGGL((embed_segsum_kernel<true>), dim3(grid1), dim3(256), 0, s, sorted_keys, sorted_pos, (int)B, dh, lddh,
                               gfm, sum_in, sf, wpf, rh, ru, rs);
        else
            hipLaunchKernelGGL((embed_segsum_kernel<false>), dim3(grid1), dim3(256), 0, s, sorted_keys, sorted_pos, (int)B, dh, lddh,
                               gfm, sum_in, sf, wpf, rh, ru, rs);
        RP_LAUNCH_CHECK("embed_grad_ss (segment sums)");
    }
    if (!(phases & 2)) return RP_OK;
    if (gfm != nullptr)
        hipLaunchKernelGGL((embed_ss_rows_kernel<true>), dim3(grid2), dim3(256), 0, s, rh, ru, rs, sorted_keys, (int)B, w, ldw,
                           arena, grad_arena, accumulate, piece0, key0, dwpart, sf, tpf, T, cpf);
    else
        hipLaunchKernelGGL((embed_ss_rows_kernel<false>), dim3(grid2), dim3(256), 0, s, rh, ru, rs, sorted_keys, (int)B, w, ldw,
                           arena, grad_arena, accumulate, piece0, key0, dwpart, sf, tpf, T, cpf);
    RP_LAUNCH_CHECK("embed_grad_ss (rows)");
    if (dw != nullptr) {
        hipLaunchKernelGGL(embed_ss_dw_kernel, dim3((unsigned)sf.n, 64), dim3(256), 0, s, dwpart, sf, cpf, dw, lddw);
        RP_LAUNCH_CHECK("embed_grad_ss (weight-gradient partials)");
    }
    return rp_int_grad_reduce_finish(n_eff, D, nb0, wbase, piece0, key0, grad_arena, accumulate, s);
}
