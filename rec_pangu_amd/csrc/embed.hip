// Multi-table embedding gather (+ dense concat + FM second order) and its backward, gfx950.
//
// HBM layout: every table of a model lives back to back in ONE arena [R_total, D] (fp32,
// row = D*4 bytes = 256 B at D=64 -> two 128-B lines, 16-B aligned), table f starting at row
// row_base[f].  A lookup is therefore arena[row_base[f] + id].  The forward writes straight into
// the MLP input buffer x[B, ldx] (embeddings first, then the dense columns, then zero padding up
// to ldx, which the host keeps a multiple of 32 floats so every row starts on a 128-B line), so
// the reference's stack / flatten / cat copies (embedding.py:63, deepfm.py:57-58) never exist.
//
// Work decomposition (wave64): TPR lanes own one sample; each lane moves VEC=4 floats (16 B) of
// every row, so a 64-float row is one 256-B coalesced segment per 16 lanes and a wave issues
// 4 rows per load instruction.  The F rows of a sample are F independent loads per lane (memory
// level parallelism comes from the unrolled field loop); sum_f v and sum_f v^2 stay in registers,
// so FM costs no extra HBM traffic.  HBM-bound: algorithmic bytes/sample =
// F*(D*4 + 8) read + (F*D+ND)*4 written.
#include "common.h"
#include <cstdlib>
#include "bfsplit.h"

struct IdxPtrs {
    const int64_t *p[RP_MAX_FIELDS];
};
struct DensePtrs {
    const float *p[RP_MAX_FIELDS];
};

template <int VEC>
struct Vec;
template <>
struct Vec<4> {
    typedef f32x4 T;
    static __device__ __forceinline__ T load(const float *p) { return *reinterpret_cast<const f32x4 *>(p); }
    static __device__ __forceinline__ void store(float *p, T v) { *reinterpret_cast<f32x4 *>(p) = v; }
    static __device__ __forceinline__ T zero() { return f32x4{0.f, 0.f, 0.f, 0.f}; }
    static __device__ __forceinline__ float hsum(T v) { return (v.x + v.y) + (v.z + v.w); }
};
template <>
struct Vec<1> {
    typedef float T;
    static __device__ __forceinline__ T load(const float *p) { return *p; }
    static __device__ __forceinline__ void store(float *p, T v) { *p = v; }
    static __device__ __forceinline__ T zero() { return 0.f; }
    static __device__ __forceinline__ float hsum(T v) { return v; }
};

template <int TPR, int VEC>
__global__ __launch_bounds__(256) void embed_gather_fwd_kernel(
    const float *__restrict__ arena, const int64_t *__restrict__ row_base, const int64_t *__restrict__ row_count,
    IdxPtrs idx, int F, DensePtrs dense, int ND, int64_t B, int D, float *__restrict__ x, int64_t ldx,
    float *__restrict__ fm_out, float *__restrict__ sum_out, int32_t *__restrict__ keys_out,
    int32_t *__restrict__ err_flag) {
    typedef Vec<VEC> V;
    constexpr int SPB = 256 / TPR;
    const int t = threadIdx.x % TPR;
    const int64_t b = (int64_t)blockIdx.x * SPB + threadIdx.x / TPR;
    if (b >= B) return;  // groups are TPR-aligned inside the wave: whole groups leave together
    float *xrow = x + b * ldx;
    float fm_acc = 0.f;
    for (int c = t * VEC; c < D; c += TPR * VEC) {
        typename V::T s = V::zero(), q = V::zero();
#pragma unroll 4
        for (int f = 0; f < F; ++f) {
            int64_t id = idx.p[f][b];
            if (id < 0 || id >= row_count[f]) {  // reference: IndexError from nn.Embedding on CPU
                *err_flag = 1;
                id = 0;
            }
            const int64_t key = row_base[f] + id;
            const typename V::T v = V::load(arena + key * D + c);
            V::store(xrow + (int64_t)f * D + c, v);
            s += v;
            q += v * v;
            if (keys_out != nullptr && c == 0) keys_out[(int64_t)f * B + b] = (int32_t)key;  // t == 0 only
        }
        if (sum_out != nullptr) V::store(sum_out + b * D + c, s);
        fm_acc += V::hsum(s * s - q);
    }
    const int64_t dense0 = (int64_t)F * D;
    for (int j = t; j < ND; j += TPR) xrow[dense0 + j] = dense.p[j][b];
    for (int64_t j = dense0 + ND + t; j < ldx; j += TPR) xrow[j] = 0.f;
    if (fm_out != nullptr) {
#pragma unroll
        for (int o = TPR / 2; o > 0; o >>= 1) fm_acc += __shfl_xor(fm_acc, o, TPR);
        if (t == 0) fm_out[b] = 0.5f * fm_acc;
    }
}

// Segmented reduction of the per-pair gradient rows into the dense gradient arena — DETERMINISTIC: every gradient row
// has exactly one writer and every sum a fixed order (ascending sorted position), no floating-point atomics.
// One TPR-lane group owns RP_SEG consecutive SORTED positions: it loads all their keys/positions, issues all
// their row loads at once (RP_SEG independent 256-B loads per group keep the memory pipe full; a per-run serial
// walk was latency-bound at 0.8 TB/s) and then folds equal keys together in registers.
//   * a run that starts and ends inside the segment is stored by its group;
//   * the piece of a run that crosses a segment border goes to LDS; after a barrier the first group of the
//     workgroup merges neighbouring pieces with equal keys in segment order (a hot row of a tiny table fills whole
//     workgroups with one key: 16 segments x 8 positions collapse into ONE piece) and stores the runs that lie
//     inside the workgroup;
//   * a merged run that reaches the first / last position of the workgroup may continue in the neighbour: it is
//     written to the workgroup's (head, tail) slot of a small global piece list instead, and embed_grad_fix_kernel
//     sums the chains of equal keys across workgroups in workgroup order — one writer per row again.
//   * those chains can be long (a 3-row table at batch 65536: one run over 170 workgroups), so the piece list — itself a
//     sorted (key, row) list — first goes through THIS kernel once more (LEVEL1: positions are the list indices, key -1
//     = "no piece" is skipped), which leaves chains of a handful of entries for the sequential walk.
#define RP_SEG 8
// POOL (rp_embed_pool_bwd, the backward of the pooled multi-id lookup of pool.hip): position p is the p-th id of a flat
// id list, its gradient row is the pooled output's gradient row of the BAG it belongs to — dx[bag, :] (times
// scale[bag, :] for the masked average) with bag = bag_of[p] (CSR) or p / Bi (dense bags of Bi ids).
template <int TPR, int VEC, bool LEVEL1 = false, bool POOL = false>
__global__ __launch_bounds__(256) void embed_grad_reduce_kernel(
    const int32_t *__restrict__ sk, const int32_t *__restrict__ sp, int64_t n, int Bi, int D,
    const float *__restrict__ dx, int64_t ldx, const float *__restrict__ gfm, const float *__restrict__ sum_in,
    const float *__restrict__ arena, float *__restrict__ G, int accumulate, float *__restrict__ gpiece,
    int32_t *__restrict__ gkey, const int32_t *__restrict__ bag_of = nullptr, const float *__restrict__ scale = nullptr) {
    typedef Vec<VEC> V;
    constexpr int GPB = 256 / TPR;
    constexpr int W = TPR * VEC;  // columns handled per pass
    __shared__ __attribute__((aligned(16))) float piece[GPB][2][W];
    __shared__ int32_t pkey[GPB][2];   // -1: no piece
    __shared__ int32_t pcont[GPB];     // the group's last run continues beyond its segment
    const int t = threadIdx.x % TPR, grp = threadIdx.x / TPR;
    const int64_t start = ((int64_t)blockIdx.x * GPB + grp) * RP_SEG;
    const bool active = start < n;
    const int cnt = active ? (int)((n - start) < RP_SEG ? (n - start) : RP_SEG) : 0;
    int32_t k[RP_SEG];
    int bb[RP_SEG], ff[RP_SEG];
#pragma unroll
    for (int j = 0; j < RP_SEG; ++j) {
        const bool ok = j < cnt;
        k[j] = ok ? sk[start + j] : -1;
        if (LEVEL1) {  // the piece list of the first pass: row start + j of a [n, D] buffer
            ff[j] = 0;
            bb[j] = (int)(start + j);
        } else if (POOL) {
            const int32_t p = ok ? sp[start + j] : 0;
            ff[j] = 0;
            bb[j] = bag_of ? (ok ? bag_of[p] : 0) : p / Bi;
        } else {
            const int32_t p = ok ? sp[start + j] : 0;
            ff[j] = p / Bi;
            bb[j] = p - ff[j] * Bi;
        }
    }
    const int32_t kprev = (active && start > 0) ? sk[start - 1] : -1;
    const int32_t knext = (active && start + cnt < n) ? sk[start + cnt] : -1;
    if (LEVEL1) {
        // a workgroup of "no entry" keys (round 6: the tail of rp_embed_grad_smp's duplicate list, most of it): nothing to sum,
        // nothing to hand on — its two piece entries say so and it is done
        bool any = false;
#pragma unroll
        for (int j = 0; j < RP_SEG; ++j) any |= k[j] >= 0;
        if (!__syncthreads_or((int)any)) {
            if (threadIdx.x == 0) {
                gkey[2 * (int64_t)blockIdx.x] = -1;
                gkey[2 * (int64_t)blockIdx.x + 1] = -1;
            }
            return;
        }
    }
    // the global piece list is written once per workgroup (all column passes fill the same two rows)
    float *hp = gpiece + (int64_t)blockIdx.x * 2 * D, *tp = hp + D;
    for (int c0 = 0; c0 < D; c0 += W) {
        const int c = c0 + t * VEC;
        const bool col_ok = c < D;
        typename V::T r[RP_SEG], w[RP_SEG];
        float gf[RP_SEG];
#pragma unroll
        for (int j = 0; j < RP_SEG; ++j) {
            r[j] = V::zero();
            w[j] = V::zero();
            gf[j] = 0.f;
            if (j < cnt && col_ok && !(LEVEL1 && k[j] < 0)) {
                if (dx != nullptr) r[j] = V::load(dx + (int64_t)bb[j] * ldx + (int64_t)ff[j] * D + c);
                if (POOL && scale != nullptr) r[j] = r[j] * V::load(scale + (int64_t)bb[j] * D + c);
                if (gfm != nullptr) {
                    gf[j] = gfm[bb[j]];
                    if (sum_in != nullptr) r[j] += gf[j] * V::load(sum_in + (int64_t)bb[j] * D + c);  // else: folded upstream
                    // the looked-up row itself (FM: -g * v) — used once per run, at its last position in the segment
                    if (j + 1 >= cnt || k[j + 1] != k[j]) w[j] = V::load(arena + (int64_t)k[j] * D + c);
                }
            }
        }
        if (t == 0) {
            pkey[grp][0] = -1;
            pkey[grp][1] = -1;
            pcont[grp] = 0;
        }
        typename V::T acc = V::zero();
        float gs = 0.f;
        bool run_head = (k[0] != kprev);
#pragma unroll
        for (int j = 0; j < RP_SEG; ++j) {
            if (j < cnt) {
                acc += r[j];
                gs += gf[j];
                const bool last_of_run = (j + 1 < cnt) ? (k[j + 1] != k[j]) : true;
                if (last_of_run) {
                    const bool run_ends_here = (j + 1 < cnt) ? true : (k[j] != knext);
                    if (gfm != nullptr) acc -= gs * w[j];
                    if (LEVEL1 && k[j] < 0) {
                        // "no piece" entries of the first pass: nothing to add anywhere
                    } else if (run_head && run_ends_here) {
                        if (col_ok) {
                            float *dst = G + (int64_t)k[j] * D + c;
                            V::store(dst, accumulate ? V::load(dst) + acc : acc);  // the only writer of this row
                        }
                    } else {
                        // border piece: slot 0 if the run came in from the previous segment, else slot 1
                        const int slot = run_head ? 1 : 0;
                        V::store(&piece[grp][slot][t * VEC], acc);
                        if (t == 0) {
                            pkey[grp][slot] = k[j];
                            if (!run_ends_here) pcont[grp] = 1;
                        }
                    }
                    acc = V::zero();
                    gs = 0.f;
                    run_head = true;
                }
            }
        }
        __syncthreads();
        if (grp == 0) {
            // does the first / last position of the workgroup belong to a run that continues in the neighbour?
            const bool head_open = pkey[0][0] >= 0;  // group 0's first run came in from the previous workgroup
            int last_g = -1;                          // the last group whose final run leaves its segment ...
            for (int g2 = GPB - 1; g2 >= 0; --g2)
                if (pkey[g2][0] >= 0 || pkey[g2][1] >= 0) {
                    last_g = g2;
                    break;
                }
            // ... leaves the WORKGROUP only if that group is the workgroup's last active one
            const int64_t wg_end = ((int64_t)blockIdx.x + 1) * GPB * RP_SEG;
            const int last_active = (int)(((n < wg_end ? n : wg_end) - (int64_t)blockIdx.x * GPB * RP_SEG + RP_SEG - 1) / RP_SEG) - 1;
            const bool tail_open = last_g >= 0 && last_g == last_active && pcont[last_g] != 0;
            int32_t cur = -1;
            bool cur_is_head = false;
            typename V::T cacc = V::zero();
            int32_t headkey = -1, tailkey = -1;
            typename V::T headv = V::zero(), tailv = V::zero();
            for (int g2 = 0; g2 < GPB; ++g2) {
#pragma unroll
                for (int slot = 0; slot < 2; ++slot) {
                    const int32_t key = pkey[g2][slot];
                    if (key < 0) continue;
                    const typename V::T pv = V::load(&piece[g2][slot][t * VEC]);
                    if (key == cur) {
                        cacc += pv;
                    } else {
                        if (cur >= 0) {
                            if (cur_is_head) {
                                headkey = cur;
                                headv = cacc;
                            } else if (col_ok) {
                                float *dst = G + (int64_t)cur * D + c;
                                V::store(dst, accumulate ? V::load(dst) + cacc : cacc);
                            }
                        }
                        cur = key;
                        cacc = pv;
                        cur_is_head = head_open && g2 == 0 && slot == 0;
                    }
                }
            }
            if (cur >= 0) {
                if (tail_open) {
                    if (cur_is_head) {  // one run covers the whole workgroup: it is both; the tail entry adds zero
                        headkey = cur;
                        headv = cacc;
                        tailkey = cur;
                    } else {
                        tailkey = cur;
                        tailv = cacc;
                    }
                } else if (cur_is_head) {
                    headkey = cur;
                    headv = cacc;
                } else if (col_ok) {
                    float *dst = G + (int64_t)cur * D + c;
                    V::store(dst, accumulate ? V::load(dst) + cacc : cacc);
                }
            }
            if (col_ok) {
                V::store(hp + c, headv);
                V::store(tp + c, tailv);
            }
            if (t == 0 && c0 == 0) {
                gkey[2 * (int64_t)blockIdx.x] = headkey;
                gkey[2 * (int64_t)blockIdx.x + 1] = tailkey;
            }
        }
        __syncthreads();
    }
}

#define EG_LD 72  // bf16 row of 64 + 8 pad: 144 B = 9 x 16 B (odd) -> conflict-free ds_read_b128 fragment reads
#define EG_CT 68  // fp32 row of the C tile
#define EGL_MAXF 32  // fields the fused gather + Linear forward keeps keys for

// ------------------------------------------------------------------------------------------------
// Gather FUSED with the Linear that consumes the gathered rows (DeepFM: embedding lookup + dense concat + FM second
// order + dnn.net.0 with its ReLU — embedding.py:59-63, utils.py:122-137, deepfm.py:57-58, interaction.py:38-44,
// deep.py:62-72 in one launch).
//
// Unfused, the gather writes x[B, F*D+ND] (453 MB at Criteo shape) and the first layer reads it straight back.  Here a
// workgroup takes 128 samples through all F fields: the field's rows go from global memory into registers once and are
// used three times — stored to x (the weight gradient still reads it), added into the FM sums, handed to the matrix core
// —, the field's 64 x 64 slice of W1 goes through LDS as bf16 pieces, and h1 = relu(x W1^T + b) accumulates over the
// fields in the MFMA accumulators.  Split-bf16, six products (fp32-faithful).
//
// Row traffic is COALESCED, as in the plain gather: 16 lanes x float4 = one 256-byte row per quarter wave, four whole
// rows per wave-instruction, for the arena loads AND for the x stores; the rows take one trip through LDS (fp32, 272-byte
// rows: conflict-free ds_write_b128 / ds_read_b128) to reach the MFMA A layout (lane = (sample, k half)).  (A first
// version loaded and stored straight in the A layout — 16-byte pieces, 64 pieces in 32 different rows per
// wave-instruction: 0.229 ms against 0.194.)  LDS: rows 34.8 KB + one W slice 27.6 KB + keys 16.9 KB: two workgroups per
// CU.  Two barriers per field (rows and W slice are single-buffered); the next field's rows and W slice are in flight in
// registers meanwhile.  All index -> arena-row keys are computed up front into LDS: an id load inside the field loop
// would be the newest vector-memory operation where its value is needed, and waiting for it (vmcnt(0)) would also wait
// for the prefetch and for every x store just issued.
// FULL: every sample of the workgroup exists and x is written — no lane masks around the loads and stores of the field
// loop, so the in-order vmcnt waits are counted.
// ------------------------------------------------------------------------------------------------
#define EGL_XLD 68  // floats per staged row (64 + 4 pad)
#define EGL_KLD 33  // keys per row in LDS (odd: the key fill writes 32 consecutive rows at a fixed field)
// STORE_X = false (with FULL): the gathered rows are NOT written to x — the first layer's weight gradient gathers them
// again itself (rp_linear_wgrad_gather) — only the dense columns go to the compact buffer xd [B, 64]
// BF16_ROWS: the arena holds bf16 rows (128 bytes at D = 64: the bf16-STORAGE inference mode, half the gather traffic);
// a lane reads 4 bf16 (8 bytes) where the fp32 form reads a float4 and widens them — exact — so everything downstream
// (FM sums in fp32, the split-bf16 products) is unchanged.
template <bool FULL, bool STORE_X = true, bool BF16_ROWS = false>
__global__ __launch_bounds__(256, 2) void embed_gather_linear_kernel(
    const float *__restrict__ arena, const int64_t *__restrict__ row_base, const int64_t *__restrict__ row_count,
    IdxPtrs idx, int F, DensePtrs dense, int ND, int64_t B, int64_t blk0, float *__restrict__ x, int64_t ldx,
    const float *__restrict__ W, int64_t ldw, const float *__restrict__ bias, float *__restrict__ h1,
    float *__restrict__ fm_out, float *__restrict__ sum_out, int32_t *__restrict__ keys_out, int32_t *__restrict__ err_flag,
    float *__restrict__ xd) {
    constexpr int D = 64;
    __shared__ __attribute__((aligned(16))) __bf16 Wt[3][64][EG_LD];
    __shared__ __attribute__((aligned(16))) float Xs[128][EGL_XLD];
    __shared__ int32_t Ks[128][EGL_KLD];
    const int t = threadIdx.x;
    // MFMA side: wave wv owns samples 32 wv .. 32 wv + 31; lane (i, h) = (sample, k half)
    const int wv = t >> 6, l = t & 63, i = l & 31, h = l >> 5;
    const int64_t blk = blk0 + blockIdx.x;
    const int64_t b = blk * 128 + 32 * wv + i;
    const bool bok = FULL || b < B;
    const int64_t bc = bok ? b : B - 1;
    // row-traffic side: 16 lanes per row (float4 each), rows g + 16 u, u = 0..7
    const int g = t >> 4, sub = t & 15;
    const int wd = t >> 2, wc = (t & 3) * 16;  // W staging: row n = wd, 16 floats from wc

    for (int f = h; f < F; f += 2) {  // arena rows of the workgroup's samples, all fields (range check as the plain gather)
        int64_t id = idx.p[f][bc];
        if (id < 0 || id >= row_count[f]) {
            if (bok) *err_flag = 1;
            id = 0;
        }
        const int64_t key = row_base[f] + id;
        Ks[32 * wv + i][f] = (int32_t)key;
        if (keys_out != nullptr && bok) keys_out[(int64_t)f * B + b] = (int32_t)key;
    }
    __syncthreads();
    // fp32 rows: 16 lanes x float4 per 256-byte row, rows g + 16 u (u < 8), v[u] = columns 4 sub .. 4 sub + 3.
    // bf16 rows: 8 lanes x 16 bytes (8 bf16) per 128-byte row — EIGHT whole rows per wave-instruction, half the load
    // instructions for half the bytes —, rows g8 + 32 u' (u' < 4); v[2 u'], v[2 u' + 1] = columns 8 s8 .. +3, +4 .. +7.
    const int g8 = t >> 3, s8 = t & 7;
    auto row_of = [&](int u) { return BF16_ROWS ? g8 + 32 * (u >> 1) : g + 16 * u; };
    auto col_of = [&](int u) { return BF16_ROWS ? 8 * s8 + 4 * (u & 1) : 4 * sub; };
    auto load_rows = [&](int f, f32x4 (&v)[8]) {
        if (BF16_ROWS) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint4 w4 = *reinterpret_cast<const uint4 *>(reinterpret_cast<const uint16_t *>(arena) +
                                                                  (int64_t)Ks[g8 + 32 * u][f] * D + 8 * s8);
                v[2 * u] = f32x4{__uint_as_float(w4.x << 16), __uint_as_float(w4.x & 0xffff0000u),
                                 __uint_as_float(w4.y << 16), __uint_as_float(w4.y & 0xffff0000u)};
                v[2 * u + 1] = f32x4{__uint_as_float(w4.z << 16), __uint_as_float(w4.z & 0xffff0000u),
                                     __uint_as_float(w4.w << 16), __uint_as_float(w4.w & 0xffff0000u)};
            }
        } else {
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4 *>(arena + (int64_t)Ks[g + 16 * u][f] * D + 4 * sub);
        }
    };
    auto load_w = [&](int col0, f32x4 (&v)[4]) {
        const float *src = W + (int64_t)wd * ldw + col0 + wc;
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const f32x4 *>(src + 4 * u);
    };
    f32x16 acc[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
    f32x4 S[8];
    float q[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        S[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        q[u] = 0.f;
    }
    f32x4 cur[8], nxt[8], wq[4], wqn[4];
    load_rows(0, cur);
    load_w(0, wq);
    for (int f = 0; f < F; ++f) {
        const int fn = f + 1 < F ? f + 1 : f;  // (the last field is simply read once more: a load behind a branch cannot be
        load_rows(fn, nxt);                    //  counted by the in-order vmcnt waits)
        load_w(fn * D, wqn);
        __syncthreads();  // everyone is past the fragment reads of field f - 1: Xs and Wt are free
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int r = row_of(u);
            *reinterpret_cast<f32x4 *>(&Xs[r][col_of(u)]) = cur[u];
            const int64_t br = blk * 128 + r;
            if (STORE_X && (FULL || (x != nullptr && br < B))) {
                if (BF16_ROWS) {  // the activation is stored as bf16 (the values ARE bf16: exact): 8 bytes per lane
                    uint2 w2;
                    w2.x = (__float_as_uint(cur[u][0]) >> 16) | (__float_as_uint(cur[u][1]) & 0xFFFF0000u);
                    w2.y = (__float_as_uint(cur[u][2]) >> 16) | (__float_as_uint(cur[u][3]) & 0xFFFF0000u);
                    *reinterpret_cast<uint2 *>(reinterpret_cast<uint16_t *>(x) + br * ldx + (int64_t)f * D + col_of(u)) = w2;
                } else {
                    *reinterpret_cast<f32x4 *>(x + br * ldx + (int64_t)f * D + col_of(u)) = cur[u];
                }
            }
            S[u] += cur[u];
#pragma unroll
            for (int e = 0; e < 4; ++e) q[u] = __builtin_fmaf(cur[u][e], cur[u][e], q[u]);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            f32x8 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] = wq[2 * u][e];
                v[4 + e] = wq[2 * u + 1][e];
            }
            bf16x8 pc[3];
            bf_split8<3>(v, pc);
#pragma unroll
            for (int qq = 0; qq < 3; ++qq) *reinterpret_cast<bf16x8 *>(&Wt[qq][wd][wc + 8 * u]) = pc[qq];
        }
        __syncthreads();  // rows and W slice of field f visible
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const f32x4 v0 = *reinterpret_cast<const f32x4 *>(&Xs[32 * wv + i][ks * 16 + 8 * h]);
            const f32x4 v1 = *reinterpret_cast<const f32x4 *>(&Xs[32 * wv + i][ks * 16 + 8 * h + 4]);
            f32x8 av;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                av[e] = v0[e];
                av[4 + e] = v1[e];
            }
            bf16x8 a[3];
            bf_split8<3>(av, a);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                bf16x8 bq[3];
#pragma unroll
                for (int qq = 0; qq < 3; ++qq) bq[qq] = *reinterpret_cast<const bf16x8 *>(&Wt[qq][nt * 32 + i][ks * 16 + 8 * h]);
#pragma unroll
                for (int pr = 0; pr < 6; ++pr) {
                    // bf16 rows ARE their first piece: the other two are zero and so are the three products they enter
                    if (BF16_ROWS && BfProd<6>::pa(pr) != 0) continue;
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[BfProd<6>::pa(pr)], bq[BfProd<6>::pb(pr)], acc[nt], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) cur[u] = nxt[u];
#pragma unroll
        for (int u = 0; u < 4; ++u) wq[u] = wqn[u];
    }
    // ---- the dense columns (K tail: ND <= 16 columns right after the F*D embedding columns): one 16-deep k-step
    if (ND > 0) {
        __syncthreads();  // the fragment reads of the last field are done
        {   // W[:, F*D + k], k < ND, zero padded to 16; guarded (the row may end right after column K - 1)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = (t & 3) * 4 + u;  // k of this element, row n = wd
                const float v = e < ND ? W[(int64_t)wd * ldw + (int64_t)F * D + e] : 0.f;
                const __bf16 hi = (__bf16)v;
                const float r1 = v - (float)hi;
                const __bf16 mid = (__bf16)r1;
                Wt[0][wd][e] = hi;
                Wt[1][wd][e] = mid;
                Wt[2][wd][e] = (__bf16)(r1 - (float)mid);
            }
        }
        f32x8 dv;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = 8 * h + e;
            dv[e] = k < ND ? dense.p[k][bc] : 0.f;
        }
        __syncthreads();
        if (x != nullptr && bok && BF16_ROWS) {  // bf16 activation: the dense columns rounded to nearest even
            uint16_t *xt = reinterpret_cast<uint16_t *>(x) + b * ldx + (int64_t)F * D;
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (8 * h + e < ND) {
                    uint32_t ub = __float_as_uint(dv[e]);
                    ub += 0x7FFFu + ((ub >> 16) & 1u);
                    xt[8 * h + e] = (uint16_t)(ub >> 16);
                }
            for (int64_t j = (int64_t)F * D + ND + h; j < ldx; j += 2) reinterpret_cast<uint16_t *>(x)[b * ldx + j] = 0;
        } else if (x != nullptr && bok) {
            float *xt = x + b * ldx + (int64_t)F * D;
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (8 * h + e < ND) xt[8 * h + e] = dv[e];
            for (int64_t j = (int64_t)F * D + ND + h; j < ldx; j += 2) x[b * ldx + j] = 0.f;  // zero padding up to ldx
        }
        if (xd != nullptr && bok) {  // the compact copy of the dense columns, zero padded to 64
            float *xt = xd + b * 64;
#pragma unroll
            for (int e = 0; e < 8; ++e) xt[8 * h + e] = (8 * h + e < ND) ? dv[e] : 0.f;
            for (int j = 16 + h; j < 64; j += 2) xt[j] = 0.f;
        }
        bf16x8 a[3];
        bf_split8<3>(dv, a);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            bf16x8 bq[3];
#pragma unroll
            for (int qq = 0; qq < 3; ++qq) bq[qq] = *reinterpret_cast<const bf16x8 *>(&Wt[qq][nt * 32 + i][8 * h]);
#pragma unroll
            for (int pr = 0; pr < 6; ++pr)
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[BfProd<6>::pa(pr)], bq[BfProd<6>::pb(pr)], acc[nt], 0, 0, 0);
        }
    } else {
        if (x != nullptr && bok && BF16_ROWS)
            for (int64_t j = (int64_t)F * D + h; j < ldx; j += 2) reinterpret_cast<uint16_t *>(x)[b * ldx + j] = 0;
        else if (x != nullptr && bok)
            for (int64_t j = (int64_t)F * D + h; j < ldx; j += 2) x[b * ldx + j] = 0.f;
        if (xd != nullptr && bok)
            for (int j = h; j < 64; j += 2) xd[b * 64 + j] = 0.f;
    }
    // ---- epilogue: h1 = relu(acc + bias) in the C layout
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int n = nt * 32 + i;
        const float bv = bias != nullptr ? bias[n] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t m = blk * 128 + 32 * wv + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (FULL || m < B) h1[m * 64 + n] = fmaxf(acc[nt][r] + bv, 0.f);
        }
    }
    // FM second order and the field sum, from the row-traffic layout (16 lanes per sample; bf16 rows: 8 lanes, two
    // accumulator quads per sample)
    if (BF16_ROWS) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t br = blk * 128 + g8 + 32 * u;
            float fmv = -(q[2 * u] + q[2 * u + 1]);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                fmv = __builtin_fmaf(S[2 * u][e], S[2 * u][e], fmv);
                fmv = __builtin_fmaf(S[2 * u + 1][e], S[2 * u + 1][e], fmv);
            }
            fmv += __shfl_xor(fmv, 1, 64);
            fmv += __shfl_xor(fmv, 2, 64);
            fmv += __shfl_xor(fmv, 4, 64);
            if ((FULL || br < B) && fm_out != nullptr && s8 == 0) fm_out[br] = 0.5f * fmv;
            if ((FULL || br < B) && sum_out != nullptr) {  // (training: the FM backward needs the field sums)
                *reinterpret_cast<f32x4 *>(sum_out + br * D + 8 * s8) = S[2 * u];
                *reinterpret_cast<f32x4 *>(sum_out + br * D + 8 * s8 + 4) = S[2 * u + 1];
            }
        }
        return;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int64_t br = blk * 128 + g + 16 * u;
        float fmv = -q[u];
#pragma unroll
        for (int e = 0; e < 4; ++e) fmv = __builtin_fmaf(S[u][e], S[u][e], fmv);
        fmv += __shfl_xor(fmv, 1, 64);
        fmv += __shfl_xor(fmv, 2, 64);
        fmv += __shfl_xor(fmv, 4, 64);
        fmv += __shfl_xor(fmv, 8, 64);
        if (FULL || br < B) {
            if (fm_out != nullptr && sub == 0) fm_out[br] = 0.5f * fmv;
            if (sum_out != nullptr) *reinterpret_cast<f32x4 *>(sum_out + br * D + 4 * sub) = S[u];
        }
    }
}

static int embed_gather_linear_launch(const float *arena, bool bf16_rows, const int64_t *row_base, const int64_t *row_count,
                                      const int64_t *const *idx_ptrs, int F, const float *const *dense_ptrs, int ND,
                                      int64_t B, int D, float *x, int64_t ldx, const float *W, int64_t ldw,
                                      const float *bias, float *h1, float *fm_out, float *sum_out, int32_t *keys_out,
                                      int32_t *err_flag, float *xd, rp_stream_t stream);

extern "C" int rp_embed_gather_linear_fits(int D, int ND, int hidden, int64_t ldx, int64_t ldw) {
    return (D == 64 && hidden == 64 && ND >= 0 && ND <= 16 && ldx % 4 == 0 && ldw % 4 == 0) ? 1 : 0;  // (and F <= 32)
}

// x[B, ldx] (NULL: not materialised), h1[B, 64] = relu(x[:, :F*64+ND] . W^T + bias), fm_out / sum_out / keys_out as
// rp_embed_gather_fwd.  W: [64, K] with ldw floats per row, K = F*64 + ND.
extern "C" int rp_embed_gather_linear_fwd(const float *arena, const int64_t *row_base, const int64_t *row_count,
                                          const int64_t *const *idx_ptrs, int F, const float *const *dense_ptrs, int ND,
                                          int64_t B, int D, float *x, int64_t ldx, const float *W, int64_t ldw,
                                          const float *bias, float *h1, float *fm_out, float *sum_out, int32_t *keys_out,
                                          int32_t *err_flag, float *xd, rp_stream_t stream) {
    return embed_gather_linear_launch(arena, false, row_base, row_count, idx_ptrs, F, dense_ptrs, ND, B, D, x, ldx, W, ldw, bias,
                                      h1, fm_out, sum_out, keys_out, err_flag, xd, stream);
}

// the same launch over a bf16 copy of the arena (rows of D bf16): bf16-STORED tables, fp32 accumulation.  Inference stores
// nothing but h1 and the FM term; the bf16-storage training mode also stores the activation (as bf16), the field sums and
// the keys.
extern "C" int rp_embed_gather_linear_fwd_bf16(const void *arena_bf16, const int64_t *row_base, const int64_t *row_count,
                                               const int64_t *const *idx_ptrs, int F, const float *const *dense_ptrs, int ND,
                                               int64_t B, int D, const float *W, int64_t ldw, const float *bias, float *h1,
                                               float *fm_out, void *x_bf16, int64_t ldx, float *sum_out, int32_t *keys_out,
                                               int32_t *err_flag, float *xd, rp_stream_t stream) {
    RP_REQUIRE(x_bf16 == nullptr || (ldx % 4 == 0 && (reinterpret_cast<uintptr_t>(x_bf16) & 7u) == 0),
               "embed_gather_linear_fwd_bf16: the bf16 activation needs ldx %% 4 == 0 and 8-byte alignment");
    return embed_gather_linear_launch(reinterpret_cast<const float *>(arena_bf16), true, row_base, row_count, idx_ptrs, F,
                                      dense_ptrs, ND, B, D, reinterpret_cast<float *>(x_bf16), ldx, W, ldw, bias, h1, fm_out,
                                      sum_out, keys_out, err_flag, xd, stream);
}

static int embed_gather_linear_launch(const float *arena, bool bf16_rows, const int64_t *row_base, const int64_t *row_count,
                                      const int64_t *const *idx_ptrs, int F, const float *const *dense_ptrs, int ND,
                                      int64_t B, int D, float *x, int64_t ldx, const float *W, int64_t ldw,
                                      const float *bias, float *h1, float *fm_out, float *sum_out, int32_t *keys_out,
                                      int32_t *err_flag, float *xd, rp_stream_t stream) {
    RP_REQUIRE(arena && row_base && row_count && idx_ptrs && W && h1 && err_flag, "embed_gather_linear_fwd: null pointer");
    RP_REQUIRE(!(x && xd), "embed_gather_linear_fwd: give x (full activation) or xd (dense columns only), not both");
    RP_REQUIRE(F >= 1 && F <= RP_MAX_FIELDS && B >= 0, "embed_gather_linear_fwd: bad F/B");
    RP_REQUIRE(ND == 0 || dense_ptrs, "embed_gather_linear_fwd: dense_ptrs is null with ND=%d", ND);
    RP_REQUIRE((int64_t)F * B < (int64_t)INT32_MAX, "embed_gather_linear_fwd: F*B overflows int32 positions");
    if (F > EGL_MAXF || !rp_embed_gather_linear_fits(D, ND, 64, x ? ldx : 4, ldw) || !rp_aligned16(arena) || !rp_aligned16(W) ||
        (x && (!rp_aligned16(x) || ldx < (int64_t)F * D + ND)) || ldw < (int64_t)F * D + ND ||
        (sum_out && !rp_aligned16(sum_out)))
        return rp_fail(RP_ERR_UNSUPPORTED, "embed_gather_linear_fwd: needs D = 64, a 64-wide layer, ND <= 16, aligned operands");
    if (B == 0) return RP_OK;
    IdxPtrs ip;
    DensePtrs dp;
    for (int f = 0; f < F; ++f) {
        RP_REQUIRE(idx_ptrs[f], "embed_gather_linear_fwd: idx_ptrs[%d] is null", f);
        ip.p[f] = idx_ptrs[f];
    }
    for (int j = 0; j < ND; ++j) {
        RP_REQUIRE(dense_ptrs[j], "embed_gather_linear_fwd: dense_ptrs[%d] is null", j);
        dp.p[j] = dense_ptrs[j];
    }
    // workgroups whose 128 samples all exist run without lane masks (x written, or no row store at all)
    const int64_t nfull = B / 128;
    const int64_t nblk = rp_cdiv(B, 128);
    hipStream_t s = (hipStream_t)stream;
    if (bf16_rows) {
        if (nfull > 0 && x != nullptr)  // training: the activation is stored (as bf16)
            hipLaunchKernelGGL((embed_gather_linear_kernel<true, true, true>), dim3((unsigned)nfull), dim3(256), 0, s, arena,
                               row_base, row_count, ip, F, dp, ND, B, (int64_t)0, x, ldx, W, ldw, bias, h1, fm_out, sum_out,
                               keys_out, err_flag, xd);
        else if (nfull > 0)
            hipLaunchKernelGGL((embed_gather_linear_kernel<true, false, true>), dim3((unsigned)nfull), dim3(256), 0, s, arena,
                               row_base, row_count, ip, F, dp, ND, B, (int64_t)0, x, ldx, W, ldw, bias, h1, fm_out, sum_out,
                               keys_out, err_flag, xd);
        if (nblk > nfull)
            hipLaunchKernelGGL((embed_gather_linear_kernel<false, true, true>), dim3((unsigned)(nblk - nfull)), dim3(256), 0, s,
                               arena, row_base, row_count, ip, F, dp, ND, B, nfull, x, ldx, W, ldw, bias, h1, fm_out, sum_out,
                               keys_out, err_flag, xd);
        RP_LAUNCH_CHECK("embed_gather_linear_fwd (bf16 rows)");
        return RP_OK;
    }
    if (nfull > 0 && x != nullptr)
        hipLaunchKernelGGL((embed_gather_linear_kernel<true, true>), dim3((unsigned)nfull), dim3(256), 0, s, arena, row_base,
                           row_count, ip, F, dp, ND, B, (int64_t)0, x, ldx, W, ldw, bias, h1, fm_out, sum_out, keys_out, err_flag,
                           xd);
    else if (nfull > 0)
        hipLaunchKernelGGL((embed_gather_linear_kernel<true, false>), dim3((unsigned)nfull), dim3(256), 0, s, arena, row_base,
                           row_count, ip, F, dp, ND, B, (int64_t)0, x, ldx, W, ldw, bias, h1, fm_out, sum_out, keys_out, err_flag,
                           xd);
    if (nblk > nfull)
        hipLaunchKernelGGL((embed_gather_linear_kernel<false, true>), dim3((unsigned)(nblk - nfull)), dim3(256), 0, s, arena,
                           row_base, row_count, ip, F, dp, ND, B, nfull, x, ldx, W, ldw, bias, h1, fm_out, sum_out, keys_out,
                           err_flag, xd);
    RP_LAUNCH_CHECK("embed_gather_linear_fwd");
    return RP_OK;
}

// ------------------------------------------------------------------------------------------------
// Gather backward FUSED with the dgrad of the Linear that consumes the gathered rows (DeepFM: dnn.net.0).
//
// Unfused, the first layer's dgrad writes dX[B, F*D] (436 MB at Criteo shape) and the segmented reduce reads it back
// one row per (sample, field) pair.  But the row of pair p = (f, b) is just  dH[b, :] . W1[:, f*D:(f+1)*D]  with
// dH = gradient w.r.t. the layer's pre-activation ([B, 64], 16.7 MB: L2/MALL resident): this kernel forms the 128
// dX rows of a workgroup's 128 sorted positions on the matrix core (A = the gathered dH rows, straight from global
// memory into MFMA fragments; B = the field's 64 x 64 slice of W1^T through LDS; split-bf16, six products: fp32-
// faithful), parks them in LDS and runs the SAME deterministic segmented reduce over them.  dX never exists.
// An optional `dx` (the sum of the gradients of x's OTHER consumers, e.g. a CIN branch) is added per pair.
// Positions are field-major, so a 128-position tile lies in one field except at the F-1 field borders, where it runs
// one matrix pass per field with the other field's rows masked out (any other order of the fields inside a tile — the
// owner borders of the row-sharded path — is handled too, by walking all fields for that tile).
// ------------------------------------------------------------------------------------------------
// FM_U (round 4): the FM part  g_fm[b] * S[b, :]  of a pair's row is put into the MFMA accumulators BEFORE the matrix
// passes (acc = g * S gathered in the accumulator layout, 32 coalesced 128-byte half-wave loads per lane, in flight
// together with the dH gathers) instead of being gathered per pair in the reduce phase, and the tile's sorted keys, sample
// indices, fields and g_fm values are staged in LDS once, so that the reduce phase starts with ONE round trip to memory
// (the table rows of the run ends) instead of three dependent ones (keys / positions -> g_fm, S rows -> table rows).
// Measured before the change (profiles/microbench/probes/probe_grad_gemm.py, Criteo shape): 0.320 ms with the FM term,
// 0.198 ms without it, 0.303 ms with perfectly sequential gathers — the launch is bound by its chain of dependent
// round trips per tile, not by the bytes the random gathers move.
// Ranges of sorted positions the launch covers (round 4: the fields whose tables rp_embed_grad_tiny handles are left out;
// positions are field-major, field f = [f B, (f + 1) B), so the kept fields are a few contiguous ranges).  Tiles are laid
// out per range: tile j of range g covers [start[g] + 128 j, min(.. + 128, end[g])).
#define EG_MAXG 34
struct GradRanges {
    int n;
    int64_t start[EG_MAXG], end[EG_MAXG];
    int tile0[EG_MAXG];  // first workgroup of the range
};

template <bool FM_U>
__global__ __launch_bounds__(256, 3) void embed_grad_gemm_kernel(
    const int32_t *__restrict__ sk, const int32_t *__restrict__ sp, int64_t n, int Bi, const float *__restrict__ dh,
    int64_t lddh, const float *__restrict__ wt, int64_t ldwt, const float *__restrict__ dx, int64_t ldx,
    const float *__restrict__ gfm, const float *__restrict__ sum_in, const float *__restrict__ arena,
    float *__restrict__ G, int accumulate, float *__restrict__ gpiece, int32_t *__restrict__ gkey, GradRanges rg) {
    typedef Vec<4> V;
    constexpr int D = 64, TPR = 16, VEC = 4, GPB = 16, W = 64;
    constexpr int SM_B = 3 * 64 * EG_LD * 2, SM_C = 128 * EG_CT * 4;
    __shared__ __attribute__((aligned(16))) char smem[SM_B > SM_C ? SM_B : SM_C];
    __shared__ __attribute__((aligned(16))) float piece[GPB][2][W];
    __shared__ int32_t pkey[GPB][2];
    __shared__ int32_t pcont[GPB];
    __shared__ int32_t tkey[130];  // sorted keys of the tile's rows, [0] = the key in front of the tile, [129] = the one behind
    __shared__ int32_t tsmp[128];  // sample index b of each tile row (0 beyond n)
    __shared__ int32_t tfld[128];  // field of each tile row
    __shared__ float tgf[128];     // g_fm[b] of each tile row (0 beyond n / without the FM term)
    typedef __bf16(*BTile)[64][EG_LD];
    BTile Bt = reinterpret_cast<BTile>(smem);                 // [3][64 d][64 n (+pad)]: W1^T slice of one field, pieces
    float(*Ct)[EG_CT] = reinterpret_cast<float(*)[EG_CT]>(smem);  // [128][64 (+pad)]: the tile's dX rows (after the MFMAs)
    const int t = threadIdx.x % TPR, grp = threadIdx.x / TPR;
    // this workgroup's tile: [wg0, min(wg0 + 128, gend)) inside one range of kept positions (gend <= n)
    int gi = 0;
    for (int g2 = 1; g2 < rg.n; ++g2)
        if ((int)blockIdx.x >= rg.tile0[g2]) gi = g2;
    const int64_t wg0 = rg.start[gi] + (int64_t)((int)blockIdx.x - rg.tile0[gi]) * GPB * RP_SEG;
    const int64_t gend = rg.end[gi];
    // ---- the tile's dX rows on the matrix core (kept register-lean: three workgroups per CU have to overlap the
    //      gather latencies of this phase with the matrix / reduce phases of the others) ----------------------------
    {
        const int wv = threadIdx.x >> 6, l = threadIdx.x & 63, i = l & 31, h = l >> 5;
        const int64_t prow = wg0 + 32 * wv + i;  // this lane's row of the tile (A operand: lane = row, 8 k per lane)
        const bool rok = prow < gend;
        const int32_t pp = rok ? sp[prow] : 0;
        const int32_t kk = rok ? sk[prow] : -1;
        // (the neighbours across a range border belong to another table: their keys differ from every key of this tile)
        if (threadIdx.x == 0) tkey[0] = wg0 > 0 ? sk[wg0 - 1] : -1;
        if (threadIdx.x == 64) tkey[129] = wg0 + 128 < n ? sk[wg0 + 128] : -1;
        const int fr = pp / Bi, br = pp - fr * Bi;
        const int64_t plast = (wg0 + 127 < gend ? wg0 + 127 : gend - 1);
        const int f_lo = __builtin_amdgcn_readfirstlane(sp[wg0] / Bi), f_hi = __builtin_amdgcn_readfirstlane(sp[plast] / Bi);
        if (h == 0) {  // one lane per tile row: the tile tables (everything the first trip to memory brought)
            tkey[1 + 32 * wv + i] = kk;
            tsmp[32 * wv + i] = br;
            tfld[32 * wv + i] = fr;
        }
        // Field-major positions: the tile's fields are f_lo..f_hi.  The row-sharded path sorts by (owner, local row):
        // field-major only INSIDE an owner, so a tile that spans an owner border holds fields {f_a..F-1} u {0..f_b} (and a
        // tiny batch any mix).  Such a tile (one per owner border) walks every field and skips the absent ones.
        const bool wrap = __syncthreads_or(rok && (fr < f_lo || fr > f_hi)) != 0;  // (also: the tile tables are visible)
        // ---- the second trip to memory, everything in flight together: g_fm of the row, its dH row (A operand), and the
        //      FM sum values of the accumulator layout.  g_fm first: the LDS write below waits for THAT load only (in-order
        //      counter), the rest stays in flight behind it
        const float gme = (gfm != nullptr && rok) ? gfm[br] : 0.f;
        f32x4 araw[4][2];
        const float *asrc = dh + (int64_t)br * lddh + 8 * h;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            araw[ks][0] = *reinterpret_cast<const f32x4 *>(asrc + ks * 16);
            araw[ks][1] = *reinterpret_cast<const f32x4 *>(asrc + ks * 16 + 4);
        }
        // acc[nt][r] is the element (row 32 wv + (r & 3) + 8 (r >> 2) + 4 h, column 32 nt + i) of the tile: it starts at
        // g_fm[b] * S[b, column].  Rows beyond n have g = 0 and read sample 0 (no branch around a load: counted waits).
        float uraw[2][16];
        if (FM_U) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float *src = sum_in + (int64_t)tsmp[32 * wv + (r & 3) + 8 * (r >> 2) + 4 * h] * D + i;
                uraw[0][r] = src[0];
                uraw[1][r] = src[32];
            }
        }
        if (h == 0) tgf[32 * wv + i] = gme;  // (read after the first staging barrier below)
        f32x16 acc[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
        const int wd = threadIdx.x >> 2, wc = (threadIdx.x & 3) * 16;  // staging: row d of the slice, 16 floats from wc
        const int f_from = wrap ? 0 : f_lo, f_to = wrap ? (int)((n - 1) / Bi) : f_hi;
        bool first_pass = true, need_init = FM_U;
        for (int f = f_from; f <= f_to; ++f) {
            if (wrap && !__syncthreads_or(rok && fr == f)) continue;
            if (!first_pass) __syncthreads();  // the previous field's fragment reads are done
            first_pass = false;
            {
                const float *src = wt + ((int64_t)f * D + wd) * ldwt + wc;
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
                    for (int q = 0; q < 3; ++q) *reinterpret_cast<bf16x8 *>(&Bt[q][wd][wc + 8 * u]) = pc[q];
                }
            }
            __syncthreads();
            if (FM_U && need_init) {  // (tgf is visible since the barrier above)
                need_init = false;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float gR = tgf[32 * wv + (r & 3) + 8 * (r >> 2) + 4 * h];
                    acc[0][r] = gR * uraw[0][r];
                    acc[1][r] = gR * uraw[1][r];
                }
            }
            // rows of another field (only in a tile that spans a field border) and rows beyond n contribute zero
            const float keep = (rok && fr == f) ? 1.f : 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                f32x8 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = araw[ks][0][e] * keep;
                    v[4 + e] = araw[ks][1][e] * keep;
                }
                bf16x8 a[3];
                bf_split8<3>(v, a);
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    bf16x8 b[3];
#pragma unroll
                    for (int q = 0; q < 3; ++q) b[q] = *reinterpret_cast<const bf16x8 *>(&Bt[q][nt * 32 + i][ks * 16 + 8 * h]);
#pragma unroll
                    for (int pr = 0; pr < 6; ++pr)
                        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[BfProd<6>::pa(pr)], b[BfProd<6>::pb(pr)], acc[nt], 0, 0, 0);
                }
            }
        }
        __syncthreads();  // every wave is done with Bt: the same LDS now takes the C tile
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) Ct[32 * wv + (r & 3) + 8 * (r >> 2) + 4 * h][nt * 32 + i] = acc[nt][r];
        __syncthreads();
    }
    // keys / samples / fields of this group's segment: from the tile tables in LDS (no trip to memory)
    const int64_t start = wg0 + (int64_t)grp * RP_SEG;
    const bool active = start < gend;
    const int cnt = active ? (int)((gend - start) < RP_SEG ? (gend - start) : RP_SEG) : 0;
    int32_t k[RP_SEG];
#pragma unroll
    for (int j = 0; j < RP_SEG; ++j) k[j] = (j < cnt) ? tkey[1 + grp * RP_SEG + j] : -1;
    const int32_t kprev = active ? tkey[grp * RP_SEG] : -1;            // ([0] = the key in front of the tile, -1 at 0)
    const int32_t knext = active ? tkey[1 + grp * RP_SEG + cnt] : -1;  // (rows beyond n hold -1, [129] = behind the tile)
    // ---- the deterministic segmented reduce of embed_grad_reduce_kernel over those rows (D = 64: one column pass) ----
    float *hp = gpiece + (int64_t)blockIdx.x * 2 * D, *tp = hp + D;
    const int c = t * VEC;
    typename V::T r[RP_SEG], w[RP_SEG];
    float gf[RP_SEG];
#pragma unroll
    for (int j = 0; j < RP_SEG; ++j) {
        r[j] = V::zero();
        w[j] = V::zero();
        gf[j] = 0.f;
        if (j < cnt) {
            r[j] = V::load(&Ct[grp * RP_SEG + j][c]);
            if (dx != nullptr)
                r[j] += V::load(dx + (int64_t)tsmp[grp * RP_SEG + j] * ldx + (int64_t)tfld[grp * RP_SEG + j] * D + c);
            if (gfm != nullptr) {
                gf[j] = tgf[grp * RP_SEG + j];
                if (!FM_U && sum_in != nullptr) r[j] += gf[j] * V::load(sum_in + (int64_t)tsmp[grp * RP_SEG + j] * D + c);
                if (j + 1 >= cnt || k[j + 1] != k[j]) w[j] = V::load(arena + (int64_t)k[j] * D + c);
            }
        }
    }
    if (t == 0) {
        pkey[grp][0] = -1;
        pkey[grp][1] = -1;
        pcont[grp] = 0;
    }
    typename V::T acc = V::zero();
    float gs = 0.f;
    bool run_head = (k[0] != kprev);
#pragma unroll
    for (int j = 0; j < RP_SEG; ++j) {
        if (j < cnt) {
            acc += r[j];
            gs += gf[j];
            const bool last_of_run = (j + 1 < cnt) ? (k[j + 1] != k[j]) : true;
            if (last_of_run) {
                const bool run_ends_here = (j + 1 < cnt) ? true : (k[j] != knext);
                if (gfm != nullptr) acc -= gs * w[j];
                if (run_head && run_ends_here) {
                    float *dst = G + (int64_t)k[j] * D + c;
                    V::store(dst, accumulate ? V::load(dst) + acc : acc);  // the only writer of this row
                } else {
                    const int slot = run_head ? 1 : 0;
                    V::store(&piece[grp][slot][t * VEC], acc);
                    if (t == 0) {
                        pkey[grp][slot] = k[j];
                        if (!run_ends_here) pcont[grp] = 1;
                    }
                }
                acc = V::zero();
                gs = 0.f;
                run_head = true;
            }
        }
    }
    __syncthreads();
    if (grp == 0) {
        const bool head_open = pkey[0][0] >= 0;
        int last_g = -1;
        for (int g2 = GPB - 1; g2 >= 0; --g2)
            if (pkey[g2][0] >= 0 || pkey[g2][1] >= 0) {
                last_g = g2;
                break;
            }
        const int64_t wg_end = wg0 + GPB * RP_SEG;
        const int last_active = (int)(((gend < wg_end ? gend : wg_end) - wg0 + RP_SEG - 1) / RP_SEG) - 1;
        const bool tail_open = last_g >= 0 && last_g == last_active && pcont[last_g] != 0;
        int32_t cur = -1;
        bool cur_is_head = false;
        typename V::T cacc = V::zero();
        int32_t headkey = -1, tailkey = -1;
        typename V::T headv = V::zero(), tailv = V::zero();
        for (int g2 = 0; g2 < GPB; ++g2) {
#pragma unroll
            for (int slot = 0; slot < 2; ++slot) {
                const int32_t key = pkey[g2][slot];
                if (key < 0) continue;
                const typename V::T pv = V::load(&piece[g2][slot][t * VEC]);
                if (key == cur) {
                    cacc += pv;
                } else {
                    if (cur >= 0) {
                        if (cur_is_head) {
                            headkey = cur;
                            headv = cacc;
                        } else {
                            float *dst = G + (int64_t)cur * D + c;
                            V::store(dst, accumulate ? V::load(dst) + cacc : cacc);
                        }
                    }
                    cur = key;
                    cacc = pv;
                    cur_is_head = head_open && g2 == 0 && slot == 0;
                }
            }
        }
        if (cur >= 0) {
            if (tail_open) {
                if (cur_is_head) {
                    headkey = cur;
                    headv = cacc;
                    tailkey = cur;
                } else {
                    tailkey = cur;
                    tailv = cacc;
                }
            } else if (cur_is_head) {
                headkey = cur;
                headv = cacc;
            } else {
                float *dst = G + (int64_t)cur * D + c;
                V::store(dst, accumulate ? V::load(dst) + cacc : cacc);
            }
        }
        V::store(hp + c, headv);
        V::store(tp + c, tailv);
        if (t == 0) {
            gkey[2 * (int64_t)blockIdx.x] = headkey;
            gkey[2 * (int64_t)blockIdx.x + 1] = tailkey;
        }
    }
}

// Chains of equal keys in the workgroup piece list [head_0, tail_0, head_1, tail_1, ...] (key -1: no piece) are runs
// that cross workgroup borders: the group that owns a chain's FIRST entry sums the chain in list order and writes the
// row — its only writer (a key's positions are contiguous in the sorted list, so it forms at most one chain).
template <int TPR, int VEC>
__global__ __launch_bounds__(256) void embed_grad_fix_kernel(const float *__restrict__ gpiece, const int32_t *__restrict__ gkey,
                                                             int64_t nent, int D, float *__restrict__ G, int accumulate) {
    typedef Vec<VEC> V;
    constexpr int GPB = 256 / TPR;
    const int t = threadIdx.x % TPR;
    const int64_t i = (int64_t)blockIdx.x * GPB + threadIdx.x / TPR;
    if (i >= nent) return;
    const int32_t key = gkey[i];
    if (key < 0 || (i > 0 && gkey[i - 1] == key)) return;
    for (int c = t * VEC; c < D; c += TPR * VEC) {
        typename V::T acc = V::load(gpiece + i * D + c);
        for (int64_t j = i + 1; j < nent && gkey[j] == key; ++j) acc += V::load(gpiece + j * D + c);
        float *dst = G + (int64_t)key * D + c;
        V::store(dst, accumulate ? V::load(dst) + acc : acc);
    }
}

template <int TPR, int VEC>
__global__ __launch_bounds__(256) void zero_rows_kernel(const int32_t *__restrict__ keys, int64_t n, int D,
                                                        float *__restrict__ G) {
    typedef Vec<VEC> V;
    constexpr int GPB = 256 / TPR;
    const int t = threadIdx.x % TPR;
    const int64_t i = (int64_t)blockIdx.x * GPB + threadIdx.x / TPR;
    if (i >= n) return;
    float *dst = G + (int64_t)keys[i] * D;
    for (int c = t * VEC; c < D; c += TPR * VEC) V::store(dst + c, V::zero());
}

static int pick_tpr(int D, int vec) {
    int need = (D + vec - 1) / vec, tpr = 1;
    while (tpr < need && tpr < 64) tpr <<= 1;
    return tpr;
}

#define RP_DISPATCH_TPR(tpr, vec, CALL)                                 \
    do {                                                                \
        if (vec == 4) {                                                 \
            switch (tpr) {                                              \
                case 1: { CALL(1, 4); } break;                          \
                case 2: { CALL(2, 4); } break;                          \
                case 4: { CALL(4, 4); } break;                          \
                case 8: { CALL(8, 4); } break;                          \
                case 16: { CALL(16, 4); } break;                        \
                case 32: { CALL(32, 4); } break;                        \
                default: { CALL(64, 4); } break;                        \
            }                                                           \
        } else {                                                        \
            switch (tpr) {                                              \
                case 1: { CALL(1, 1); } break;                          \
                case 2: { CALL(2, 1); } break;                          \
                case 4: { CALL(4, 1); } break;                          \
                case 8: { CALL(8, 1); } break;                          \
                case 16: { CALL(16, 1); } break;                        \
                case 32: { CALL(32, 1); } break;                        \
                default: { CALL(64, 1); } break;                        \
            }                                                           \
        }                                                               \
    } while (0)

extern "C" int rp_embed_gather_fwd(const float *arena, const int64_t *row_base, const int64_t *row_count,
                                   const int64_t *const *idx_ptrs, int F, const float *const *dense_ptrs, int ND,
                                   int64_t B, int D, float *x, int64_t ldx, float *fm_out, float *sum_out,
                                   int32_t *keys_out, int32_t *err_flag, rp_stream_t stream) {
    RP_REQUIRE(arena && row_base && row_count && idx_ptrs && x && err_flag, "embed_gather_fwd: null pointer");
    RP_REQUIRE(F >= 1 && F <= RP_MAX_FIELDS, "embed_gather_fwd: F=%d outside [1,%d]", F, RP_MAX_FIELDS);
    RP_REQUIRE(ND >= 0 && ND <= RP_MAX_FIELDS, "embed_gather_fwd: ND=%d outside [0,%d]", ND, RP_MAX_FIELDS);
    RP_REQUIRE(ND == 0 || dense_ptrs, "embed_gather_fwd: dense_ptrs is null with ND=%d", ND);
    RP_REQUIRE(D >= 1 && B >= 0, "embed_gather_fwd: bad D=%d B=%lld", D, (long long)B);
    RP_REQUIRE(ldx >= (int64_t)F * D + ND, "embed_gather_fwd: ldx=%lld < F*D+ND", (long long)ldx);
    RP_REQUIRE((int64_t)F * B < (int64_t)INT32_MAX, "embed_gather_fwd: F*B overflows int32 positions");
    if (B == 0) return RP_OK;
    IdxPtrs ip;
    DensePtrs dp;
    for (int f = 0; f < F; ++f) {
        RP_REQUIRE(idx_ptrs[f], "embed_gather_fwd: idx_ptrs[%d] is null", f);
        ip.p[f] = idx_ptrs[f];
    }
    for (int j = 0; j < ND; ++j) {
        RP_REQUIRE(dense_ptrs[j], "embed_gather_fwd: dense_ptrs[%d] is null", j);
        dp.p[j] = dense_ptrs[j];
    }
    const bool v4 = (D % 4 == 0) && (ldx % 4 == 0) && rp_aligned16(arena) && rp_aligned16(x) &&
                    (sum_out == nullptr || rp_aligned16(sum_out));
    const int vec = v4 ? 4 : 1;
    const int tpr = pick_tpr(D, vec);
    const unsigned grid = (unsigned)rp_cdiv(B, 256 / tpr);
    hipStream_t s = (hipStream_t)stream;
#define CALL(T, VV)                                                                                             \
    hipLaunchKernelGGL((embed_gather_fwd_kernel<T, VV>), dim3(grid), dim3(256), 0, s, arena, row_base, row_count, \
                       ip, F, dp, ND, B, D, x, ldx, fm_out, sum_out, keys_out, err_flag)
    RP_DISPATCH_TPR(tpr, vec, CALL);
#undef CALL
    RP_LAUNCH_CHECK("embed_gather_fwd");
    return RP_OK;
}

static int64_t grad_reduce_blocks(int64_t n, int D, int vec) {
    const int tpr = pick_tpr(D, vec);
    return rp_cdiv(rp_cdiv(n, RP_SEG), 256 / tpr);
}

static size_t grad_piece_bytes(int64_t nb, int D) {
    return (((size_t)nb * 2 * ((size_t)D * sizeof(float) + sizeof(int32_t))) + 255) & ~(size_t)255;
}

// workspace of rp_embed_grad_reduce: the (head, tail) piece rows and keys of every workgroup of the first pass and of
// the second pass over that list (sized for the scalar layout, the larger of the two)
extern "C" int rp_embed_grad_reduce_workspace_bytes(int64_t n, int D, size_t *bytes) {
    RP_REQUIRE(bytes && n >= 0 && D >= 1, "embed_grad_reduce_workspace_bytes: bad argument");
    const int64_t nn = n > 0 ? n : 1;
    const int64_t a = grad_reduce_blocks(nn, D, 1), b = grad_reduce_blocks(nn, D, 4);
    const int64_t nb0 = a > b ? a : b;
    const int64_t a1 = grad_reduce_blocks(2 * nb0, D, 1), b1 = grad_reduce_blocks(2 * nb0, D, 4);
    *bytes = grad_piece_bytes(nb0, D) + grad_piece_bytes(a1 > b1 ? a1 : b1, D) + 512;
    return RP_OK;
}

// second pass over the piece list of the first one (2 * nb0 entries, key -1 = none) through the same reduction, then
// the sequential walk over what is left (shared by rp_embed_grad_reduce and rp_embed_grad_gemm)
static int grad_reduce_finish(int64_t n, int D, int64_t nb0, char *wbase, float *piece0, int32_t *key0, float *grad_arena,
                              int accumulate, hipStream_t s) {
    const int64_t n1 = 2 * nb0;
    const int vec1 = (D % 4 == 0) ? 4 : 1;  // (the workspace rows are 16-byte aligned when D % 4 == 0)
    const int tpr1 = pick_tpr(D, vec1);
    const int64_t nb1 = grad_reduce_blocks(n1, D, vec1);
    const int64_t a0 = grad_reduce_blocks(n, D, 1), b0 = grad_reduce_blocks(n, D, 4);
    float *piece1 = reinterpret_cast<float *>(wbase + grad_piece_bytes(a0 > b0 ? a0 : b0, D));
    int32_t *key1 = reinterpret_cast<int32_t *>(piece1 + nb1 * 2 * D);
    const float *no_f = nullptr;
    if (n1 > 2) {
#define CALL(T, VV)                                                                                                   \
    hipLaunchKernelGGL((embed_grad_reduce_kernel<T, VV, true>), dim3((unsigned)nb1), dim3(256), 0, s, key0, key0, n1,  \
                       (int)n1, D, piece0, (int64_t)D, no_f, no_f, no_f, grad_arena, accumulate, piece1, key1)
        RP_DISPATCH_TPR(tpr1, vec1, CALL);
#undef CALL
        RP_LAUNCH_CHECK("embed_grad_reduce (piece list)");
    } else {
        piece1 = piece0;
        key1 = key0;
    }
    const int64_t nfix = (n1 > 2) ? 2 * nb1 : n1;
    const unsigned fgrid = (unsigned)rp_cdiv(nfix, 256 / tpr1);
#define CALL(T, VV)                                                                                                \
    hipLaunchKernelGGL((embed_grad_fix_kernel<T, VV>), dim3(fgrid), dim3(256), 0, s, piece1, key1, nfix, D, grad_arena, \
                       accumulate)
    RP_DISPATCH_TPR(tpr1, vec1, CALL);
#undef CALL
    RP_LAUNCH_CHECK("embed_grad_reduce (cross-workgroup runs)");
    return RP_OK;
}

extern "C" int rp_embed_grad_reduce(const int32_t *sorted_keys, const int32_t *sorted_pos, int64_t n, int64_t B,
                                    int D, const float *dx, int64_t ldx, const float *gfm, const float *sum_in,
                                    const float *arena, float *grad_arena, int accumulate, void *workspace,
                                    size_t workspace_bytes, rp_stream_t stream) {
    RP_REQUIRE(sorted_keys && sorted_pos && grad_arena && workspace, "embed_grad_reduce: null pointer");
    RP_REQUIRE(dx != nullptr || gfm != nullptr, "embed_grad_reduce: neither dx nor gfm given");
    RP_REQUIRE(gfm == nullptr || arena, "embed_grad_reduce: the FM term needs the arena (and sum_in unless folded)");
    RP_REQUIRE(B >= 1 && B < INT32_MAX && D >= 1, "embed_grad_reduce: bad B/D");
    if (n == 0) return RP_OK;
    size_t need = 0;
    rp_embed_grad_reduce_workspace_bytes(n, D, &need);
    RP_REQUIRE(workspace_bytes >= need, "embed_grad_reduce: workspace %zu < %zu bytes", workspace_bytes, need);
    char *wbase = reinterpret_cast<char *>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
    const bool v4 = (D % 4 == 0) && (dx == nullptr || ((ldx % 4 == 0) && rp_aligned16(dx))) &&
                    rp_aligned16(grad_arena) &&
                    (gfm == nullptr || ((sum_in == nullptr || rp_aligned16(sum_in)) && rp_aligned16(arena)));
    const int vec = v4 ? 4 : 1;
    const int tpr = pick_tpr(D, vec);
    const int64_t nb0 = grad_reduce_blocks(n, D, vec);
    float *piece0 = reinterpret_cast<float *>(wbase);
    int32_t *key0 = reinterpret_cast<int32_t *>(piece0 + nb0 * 2 * D);
    hipStream_t s = (hipStream_t)stream;
#define CALL(T, VV)                                                                                                 \
    hipLaunchKernelGGL((embed_grad_reduce_kernel<T, VV, false>), dim3((unsigned)nb0), dim3(256), 0, s, sorted_keys,  \
                       sorted_pos, n, (int)B, D, dx, ldx, gfm, sum_in, arena, grad_arena, accumulate, piece0, key0)
    RP_DISPATCH_TPR(tpr, vec, CALL);
#undef CALL
    RP_LAUNCH_CHECK("embed_grad_reduce");
    return grad_reduce_finish(n, D, nb0, wbase, piece0, key0, grad_arena, accumulate, s);
}

// grad_arena[key] (+)= the sum of rows[i, :] over the entries i of a KEY-SORTED list with keys[i] == key; entries with key -1
// carry nothing and are not read (round 6: the duplicate pairs of rp_embed_grad_smp — a list over ALL pairs of its fields in
// sorted order where only the pairs of runs of two and more carry a key).  The LEVEL1 form of the segmented reduce above:
// same determinism (one writer per row, fixed order), same workspace as rp_embed_grad_reduce(n, D).
extern "C" int rp_embed_grad_reduce_rows(const int32_t *keys, const float *rows, int64_t n, int D, float *grad_arena,
                                         int accumulate, void *workspace, size_t workspace_bytes, rp_stream_t stream) {
    RP_REQUIRE(keys && rows && grad_arena && workspace, "embed_grad_reduce_rows: null pointer");
    RP_REQUIRE(D >= 1 && n >= 0 && n < INT32_MAX, "embed_grad_reduce_rows: bad n / D");
    if (n == 0) return RP_OK;
    size_t need = 0;
    rp_embed_grad_reduce_workspace_bytes(n, D, &need);
    RP_REQUIRE(workspace_bytes >= need, "embed_grad_reduce_rows: workspace %zu < %zu bytes", workspace_bytes, need);
    char *wbase = reinterpret_cast<char *>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
    const bool v4 = (D % 4 == 0) && rp_aligned16(rows) && rp_aligned16(grad_arena);
    const int vec = v4 ? 4 : 1;
    const int tpr = pick_tpr(D, vec);
    const int64_t nb0 = grad_reduce_blocks(n, D, vec);
    float *piece0 = reinterpret_cast<float *>(wbase);
    int32_t *key0 = reinterpret_cast<int32_t *>(piece0 + nb0 * 2 * D);
    hipStream_t s = (hipStream_t)stream;
    const float *no_f = nullptr;
#define CALL(T, VV)                                                                                                   \
    hipLaunchKernelGGL((embed_grad_reduce_kernel<T, VV, true>), dim3((unsigned)nb0), dim3(256), 0, s, keys, keys, n, (int)n, D, \
                       rows, (int64_t)D, no_f, no_f, no_f, grad_arena, accumulate, piece0, key0)
    RP_DISPATCH_TPR(tpr, vec, CALL);
#undef CALL
    RP_LAUNCH_CHECK("embed_grad_reduce_rows");
    return grad_reduce_finish(n, D, nb0, wbase, piece0, key0, grad_arena, accumulate, s);
}

// Backward of rp_embed_gather_pool_fwd (pool.hip): grad_arena[key] (+)= sum over the ids p with that key of
// g[bag(p), :] (* scale[bag(p), :]); bag(p) = bag_of[p] (CSR bags) or p / L (dense bags of L ids).  sorted_keys /
// sorted_pos: the (arena row, flat id position) pairs sorted by row.  Same determinism, accumulate semantics and
// workspace as rp_embed_grad_reduce — a padding id shared by every bag is one long run, handled by the segmented
// reduction like any hot row.
extern "C" int rp_embed_pool_bwd(const int32_t *sorted_keys, const int32_t *sorted_pos, int64_t n, int D, const float *g,
                                 int64_t ldg, const float *scale, const int32_t *bag_of, int64_t L, float *grad_arena,
                                 int accumulate, void *workspace, size_t workspace_bytes, rp_stream_t stream) {
    RP_REQUIRE(sorted_keys && sorted_pos && g && grad_arena && workspace, "embed_pool_bwd: null pointer");
    RP_REQUIRE(D >= 1 && ldg >= D, "embed_pool_bwd: bad D / ldg");
    RP_REQUIRE(bag_of != nullptr || (L >= 1 && L < INT32_MAX), "embed_pool_bwd: dense bags need 1 <= L < 2^31");
    if (n == 0) return RP_OK;
    size_t need = 0;
    rp_embed_grad_reduce_workspace_bytes(n, D, &need);
    RP_REQUIRE(workspace_bytes >= need, "embed_pool_bwd: workspace %zu < %zu bytes", workspace_bytes, need);
    char *wbase = reinterpret_cast<char *>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
    const bool v4 = (D % 4 == 0) && (ldg % 4 == 0) && rp_aligned16(g) && rp_aligned16(grad_arena) &&
                    (scale == nullptr || rp_aligned16(scale));
    const int vec = v4 ? 4 : 1;
    const int tpr = pick_tpr(D, vec);
    const int64_t nb0 = grad_reduce_blocks(n, D, vec);
    float *piece0 = reinterpret_cast<float *>(wbase);
    int32_t *key0 = reinterpret_cast<int32_t *>(piece0 + nb0 * 2 * D);
    hipStream_t s = (hipStream_t)stream;
    const float *no_f = nullptr;
#define CALL(T, VV)                                                                                                      \
    hipLaunchKernelGGL((embed_grad_reduce_kernel<T, VV, false, true>), dim3((unsigned)nb0), dim3(256), 0, s, sorted_keys, \
                       sorted_pos, n, (int)(bag_of ? 1 : L), D, g, ldg, no_f, no_f, no_f, grad_arena, accumulate, piece0, \
                       key0, bag_of, scale)
    RP_DISPATCH_TPR(tpr, vec, CALL);
#undef CALL
    RP_LAUNCH_CHECK("embed_pool_bwd");
    return grad_reduce_finish(n, D, nb0, wbase, piece0, key0, grad_arena, accumulate, s);
}

// What the fused form needs: D == 64 rows, a 64-wide first layer, 16-byte aligned operands.
extern "C" int rp_embed_grad_gemm_fits(int D, int hidden, int64_t lddh, int64_t ldwt) {
    return (D == 64 && hidden == 64 && lddh % 4 == 0 && ldwt % 4 == 0) ? 1 : 0;
}

// grad_arena[key] (+)= sum over pairs p = (f, b) with that key of
//       dh[b, 0:64] . wt[f*64:(f+1)*64, 0:64]^T   (= the rows dX[b, f*64:(f+1)*64] of the consuming Linear's dgrad)
//     + (dx ? dx[b, f*64:(f+1)*64] : 0)  +  (gfm ? gfm[b] * (sum_in[b,:] - arena[key,:]) : 0)
// wt = W1^T, [>= F*64 rows, 64]: row f*64+d is column f*64+d of the layer's weight [64, K].  Same determinism,
// accumulate semantics and workspace as rp_embed_grad_reduce.
extern "C" int rp_embed_grad_gemm(const int32_t *sorted_keys, const int32_t *sorted_pos, int64_t n, int64_t B, int D,
                                  const float *dh, int64_t lddh, const float *wt, int64_t ldwt, const float *dx,
                                  int64_t ldx, const float *gfm, const float *sum_in, const float *arena,
                                  float *grad_arena, int accumulate, uint64_t skip_fields, void *workspace,
                                  size_t workspace_bytes, rp_stream_t stream) {
    RP_REQUIRE(sorted_keys && sorted_pos && dh && wt && grad_arena && workspace, "embed_grad_gemm: null pointer");
    RP_REQUIRE(gfm == nullptr || arena, "embed_grad_gemm: the FM term needs the arena");
    RP_REQUIRE(B >= 1 && B < INT32_MAX, "embed_grad_gemm: bad B");
    if (!rp_embed_grad_gemm_fits(D, 64, lddh, ldwt) || !rp_aligned16(dh) || !rp_aligned16(wt) || !rp_aligned16(grad_arena) ||
        (dx && (!rp_aligned16(dx) || ldx % 4 != 0)) || (sum_in && !rp_aligned16(sum_in)) || (arena && !rp_aligned16(arena)))
        return rp_fail(RP_ERR_UNSUPPORTED, "embed_grad_gemm: needs D = 64, a 64-wide layer and 16-byte aligned operands");
    if (n == 0) return RP_OK;
    size_t need = 0;
    rp_embed_grad_reduce_workspace_bytes(n, D, &need);
    RP_REQUIRE(workspace_bytes >= need, "embed_grad_gemm: workspace %zu < %zu bytes", workspace_bytes, need);
    // the ranges of positions this launch covers: everything, or — skip_fields != 0, FIELD-MAJOR positions only (position =
    // field * B + sample, n = F * B) — the fields whose bit is clear (the others' tables are rp_embed_grad_tiny's)
    GradRanges rg;
    rg.n = 0;
    int64_t tiles = 0;
    if (skip_fields == 0) {
        rg.n = 1;
        rg.start[0] = 0;
        rg.end[0] = n;
        rg.tile0[0] = 0;
        tiles = grad_reduce_blocks(n, D, 4);
    } else {
        RP_REQUIRE(n % B == 0 && n / B <= 64, "embed_grad_gemm: skip_fields needs field-major positions (n = F * B, F <= 64)");
        const int F = (int)(n / B);
        for (int f = 0; f < F; ++f) {
            if ((skip_fields >> f) & 1u) continue;
            if (rg.n > 0 && rg.end[rg.n - 1] == (int64_t)f * B) {
                rg.end[rg.n - 1] = (int64_t)(f + 1) * B;
            } else {
                RP_REQUIRE(rg.n < EG_MAXG, "embed_grad_gemm: too many field ranges");
                rg.start[rg.n] = (int64_t)f * B;
                rg.end[rg.n] = (int64_t)(f + 1) * B;
                rg.n++;
            }
        }
        if (rg.n == 0) return RP_OK;  // every field is handled elsewhere
        for (int g = 0; g < rg.n; ++g) {
            rg.tile0[g] = (int)tiles;
            tiles += rp_cdiv(rg.end[g] - rg.start[g], 128);
        }
    }
    for (int g = rg.n; g < EG_MAXG; ++g) rg.start[g] = rg.end[g] = 0, rg.tile0[g] = INT32_MAX;
    char *wbase = reinterpret_cast<char *>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
    // (tiles <= n / 128 + ranges; the piece region of the workspace is sized for the scalar layout's n / 32 workgroups)
    const int64_t nb0 = tiles;
    RP_REQUIRE(nb0 <= grad_reduce_blocks(n, D, 1), "embed_grad_gemm: %lld tiles do not fit the workspace", (long long)nb0);
    float *piece0 = reinterpret_cast<float *>(wbase);
    int32_t *key0 = reinterpret_cast<int32_t *>(piece0 + nb0 * 2 * D);
    hipStream_t s = (hipStream_t)stream;
    // RP_GRAD_GEMM_FMU=0: the FM sum rows are gathered per pair in the reduce phase (the round-3 form, kept for A/B runs)
    static const bool fmu_on = []() {
        const char *e = getenv("RP_GRAD_GEMM_FMU");
        return !(e && e[0] == '0');
    }();
    if (gfm != nullptr && sum_in != nullptr && fmu_on)
        hipLaunchKernelGGL((embed_grad_gemm_kernel<true>), dim3((unsigned)nb0), dim3(256), 0, s, sorted_keys, sorted_pos, n, (int)B,
                           dh, lddh, wt, ldwt, dx, ldx, gfm, sum_in, arena, grad_arena, accumulate, piece0, key0, rg);
    else
        hipLaunchKernelGGL((embed_grad_gemm_kernel<false>), dim3((unsigned)nb0), dim3(256), 0, s, sorted_keys, sorted_pos, n, (int)B,
                           dh, lddh, wt, ldwt, dx, ldx, gfm, sum_in, arena, grad_arena, accumulate, piece0, key0, rg);
    RP_LAUNCH_CHECK("embed_grad_gemm");
    return grad_reduce_finish(n, D, nb0, wbase, piece0, key0, grad_arena, accumulate, s);
}

// ------------------------------------------------------------------------------------------------
// rp_embed_grad_seg (round 5): the first layer's whole backward on the embedding columns, SEGMENT-SUM FIRST.
// (reference: aten::embedding_dense_backward of layers/embedding.py:61-63, the `** 2` backward of interaction.py:38-44,
//  the first Linear's dgrad AND the embedding columns of its weight gradient, deep.py:62-72 on the input of deepfm.py:57-59)
//
// rp_embed_grad_gemm forms one dX row per (sample, field) PAIR on the matrix core and reduces them per table row, while
// the layer's weight gradient is a separate GEMM over a stored activation x [B, F*64] (written by the forward for that one
// reader: 0.88 GB of round trip at Criteo shape).  Both are linear in per-run sums: for the pairs p = (f, b) of one table
// row r (a RUN of the row-sorted pair list)
//     Hs = sum_b dH[b, :]      u = sum_b g_fm[b] S[b, :]      s = sum_b g_fm[b]
//     G[r]                 = Hs . W1_f^T + u - s v_r                      (the table row's gradient)
//     dW1[:, f*64:(f+1)*64] += Hs^T (x) v_r                               (the run's share of the weight gradient)
// so a tile of 128 sorted positions is first segment-summed in registers (16 lanes x float4 own 8 positions: dH rows,
// S rows and the run's table row v_r all in flight together — ONE dependent trip to memory after the keys), the sums of
// every run PIECE (a run cut at the 8-position borders: everything downstream is linear, pieces are merged exactly as in
// rp_embed_grad_gemm) are parked in LDS next to the pieces' table rows, and two matrix-core passes follow: the dgrad over
// the piece rows (A = Hs rows, B = W1's field slice held as bf16 pieces in registers for the whole chunk) and the weight
// gradient (A = V^T, B = Hs: K = the tile's 128 rows, operands read transposed from LDS), split-bf16 x6 (fp32-faithful).
// A workgroup walks a CHUNK of consecutive tiles of ONE field with the 64 x 64 weight-gradient block in its accumulators
// and writes one partial per chunk; a small launch sums the partials in a fixed order.  The forward stores no x, the
// separate weight-gradient GEMM is gone (the 13 dense columns: a small rp_linear_wgrad over xd), v_r is read once per
// piece for both uses.  Deterministic: one writer per gradient row, fixed summation orders, no atomics.
// Field-major positions only (position = field * B + sample, n = F * B): the single-device path.
// ------------------------------------------------------------------------------------------------
#define SG_HT 68  // floats per row of the Hs / C tile (272 B: conflict-free ds_read_b128 A fragments, as EG_CT)
struct SegFields {
    int n;                    // kept fields
    unsigned char sched[64];  // chunk block -> field: the kept fields, largest table first (long chunks start first)
    unsigned char rank[64];   // field -> its ordinal among the kept fields in ascending order (tile ids = position order)
};
// SEG = sorted positions per 16-lane group: a tile is 16 * SEG positions.  SEG = 8: 128-row tiles, two workgroups per CU
// (LDS 77 KB, up to 256 registers); SEG = 4: 64-row tiles, three workgroups per CU (42 KB, 168 registers).
// LDS-only barrier: every barrier of the tile loop orders LDS traffic only (each gradient row has one writer, the piece list
// is read by later launches), and __syncthreads() would also drain the vector-memory counter — the row loads in flight across
// the piece-count barrier, the next tile's keys, the gradient-row stores
#define SEG_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
// (Round 5 A/B, removed again: ONE workgroup per CU with the whole register file — 256 VGPRs + AGPR spill space, no scratch —
//  ran 0.344 ms inside the step against 0.329 ms for this form with its 27 scratch registers, and needs an EMPTY CU to start:
//  it began 60 us after the tail backward ended, when the side streams' kernels had drained; profiles/r05 notes in DESIGN §6.)
template <bool HAS_FM, int SEG>
__global__ __launch_bounds__(256, (SEG == 8 ? 2 : 3)) void embed_grad_seg_kernel(
    const int32_t *__restrict__ sk, const int32_t *__restrict__ sp, int64_t n, int Bi, const float *__restrict__ dh,
    int64_t lddh, const float *__restrict__ w, int64_t ldw, const float *__restrict__ gfm,
    const float *__restrict__ sum_in, const float *__restrict__ arena, float *__restrict__ G, int accumulate,
    float *__restrict__ gpiece, int32_t *__restrict__ gkey, float *__restrict__ dwpart, SegFields sf, int tpf, int T,
    int cpf) {
    typedef Vec<4> V;
    constexpr int D = 64, GPB = 16, TILE = GPB * SEG, MB = TILE / 64;  // MB: 32-row blocks of the dgrad per wave
    __shared__ __attribute__((aligned(16))) float HsT[TILE][SG_HT];  // the pieces' dH sums; after the matrix passes: the C tile
    __shared__ __attribute__((aligned(16))) float VT[TILE][D];       // the pieces' table rows (zero rows elsewhere)
    __shared__ __attribute__((aligned(16))) float piece[GPB][2][D];
    __shared__ int32_t pkey[GPB][2];
    __shared__ int32_t pcont[GPB];
    __shared__ int32_t tkey[TILE + 2];  // sorted keys of the tile's rows (-1 beyond its end), [0] / [TILE + 1]: the neighbours
    __shared__ int32_t tsmp[TILE];      // sample of every tile row (0 beyond the end)
    __shared__ __attribute__((aligned(16))) int32_t gcnt[GPB];  // pieces per group (the tile's pieces are stored COMPACTED)
    const int tid = threadIdx.x, t = tid & 15, grp = tid >> 4, c = 4 * t;
    const int wv = tid >> 6, l = tid & 63, i = l & 31, h = l >> 5;
    const int chunk = (int)blockIdx.x;
    const int o = chunk / cpf;
    const int f = sf.sched[o];
    const int j0 = (chunk - o * cpf) * T;
    const int jn = (j0 + T < tpf) ? j0 + T : tpf;
    const int64_t fbase = (int64_t)f * Bi, fend = fbase + Bi;
    const int64_t tile_id0 = (int64_t)sf.rank[f] * tpf;
    const bool want_dw = dwpart != nullptr;
    // ---- the chunk's constants: this wave's B fragments of the dgrad (W1[hidden, f*64 + 32 nb + i], bf16 pieces) ----
    const int nb = wv & 1, mh = wv >> 1;  // dgrad: rows (TILE / 2) mh .. of the tile x columns 32 nb .. + 31
    bf16x8 wp[4][3];
    {
        const float *wsrc = w + (int64_t)f * D + 32 * nb + i;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            f32x8 v;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = wsrc[(int64_t)(16 * ks + 8 * h + e) * ldw];
            bf_split8<3>(v, wp[ks]);
        }
    }
    // weight gradient: this wave's block  dW1^T[d = 32 (wv >> 1) + .., hidden = 32 (wv & 1) + ..]  of the field
    f32x16 dwacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) dwacc[r] = 0.f;
    // keys / samples of a tile travel through LDS, one tile ahead: thread x < TILE loads the key of row x, thread TILE + x
    // the position (SEG = 4: the upper half of the workgroup idles here); threads 0 / 1 also load the two neighbours
    auto tile_keys = [&](int64_t wg0, int64_t gend, int32_t &a0, int32_t &a1) {
        a0 = 0;
        a1 = -1;
        const int x = tid < TILE ? tid : tid - TILE;
        const bool in = x < TILE && tid < 2 * TILE && wg0 + x < gend;
        const uint32_t q = in ? (uint32_t)(wg0 + x) : 0u;
        const int32_t raw = tid < TILE ? sk[q] : sp[q];
        a0 = in ? (tid < TILE ? raw : raw - (int32_t)fbase) : (tid < TILE ? -1 : 0);
        const bool nb_ok = (tid == 0) ? (wg0 > 0) : (wg0 + TILE < n);
        const int32_t rawn = sk[(tid < 2 && nb_ok) ? (uint32_t)(tid == 0 ? wg0 - 1 : wg0 + TILE) : 0u];
        a1 = (tid < 2 && nb_ok) ? rawn : -1;
    };
    auto tile_keys_store = [&](int32_t a0, int32_t a1) {
        if (tid < TILE) tkey[1 + tid] = a0;
        else if (tid < 2 * TILE) {
            int b = a0;
            b = b < 0 ? 0 : (b >= Bi ? Bi - 1 : b);  // (a position outside the field's range would be a caller error)
            tsmp[tid - TILE] = b;
        }
        if (tid == 0) tkey[0] = a1;
        if (tid == 1) tkey[TILE + 1] = a1;
    };
    {
        const int64_t wg0 = fbase + (int64_t)TILE * j0;
        int32_t a0, a1;
        tile_keys(wg0, (wg0 + TILE < fend) ? wg0 + TILE : fend, a0, a1);
        tile_keys_store(a0, a1);
    }
    // (rows of the two tiles beyond a tile's pieces are read by the last k-step of the weight gradient with a zero on the
    //  other side: they must hold finite data from the start)
    for (int e = tid; e < TILE * SG_HT / 4; e += 256) reinterpret_cast<f32x4 *>(&HsT[0][0])[e] = V::zero();
    for (int e = tid; e < TILE * D / 4; e += 256) reinterpret_cast<f32x4 *>(&VT[0][0])[e] = V::zero();
    __syncthreads();
    for (int j = j0; j < jn; ++j) {
        const int64_t wg0 = fbase + (int64_t)TILE * j;
        const int64_t gend = (wg0 + TILE < fend) ? wg0 + TILE : fend;
        const int64_t start = wg0 + SEG * grp;
        const int cnt = (gend - start) <= 0 ? 0 : ((gend - start) < SEG ? (int)(gend - start) : SEG);
        int32_t k[SEG + 1];
        int bs[SEG];
#pragma unroll
        for (int jj = 0; jj < SEG; ++jj) {
            k[jj] = tkey[1 + SEG * grp + jj];
            bs[jj] = tsmp[SEG * grp + jj];
        }
        k[SEG] = -1;
        const int32_t kprev = cnt > 0 ? tkey[SEG * grp] : -1;
        const int32_t knext = cnt > 0 ? tkey[1 + SEG * grp + cnt] : -1;
        // a piece ends where the key changes or the group's positions end (k = -1 beyond cnt and behind the group's
        // positions); only piece ends carry data, so they are stored COMPACTED: piece q of the tile in row q of the two LDS
        // tiles — a field of a 10 k-row table has ~30 pieces per 128 positions, and the matrix passes then cover 32 rows
        // (one m-block of the dgrad, two k-steps of the weight gradient) instead of 128
        unsigned lastm = 0;
#pragma unroll
        for (int jj = 0; jj < SEG; ++jj) lastm |= (jj < cnt && k[jj + 1] != k[jj]) ? (1u << jj) : 0u;
        if (t == 0) gcnt[grp] = __builtin_popcount(lastm);
        // ---- the one dependent trip to memory: dH rows, S rows, g_fm and the table rows, all in flight together (no
        //      branch around a load: the in-order counter stays counted) ------------------------------------------------
        f32x4 rh[SEG], rs[SEG], rv[SEG];
        float g[SEG];
        auto issue_row = [&](int jj) {
            // (dH / S: B * 64 floats < 2^32 bytes — 32-bit offsets from the scalar base)
            rh[jj] = V::load(dh + ((uint32_t)bs[jj] * (uint32_t)lddh + (uint32_t)c));
            if (HAS_FM) {
                rs[jj] = V::load(sum_in + ((uint32_t)bs[jj] * (uint32_t)D + (uint32_t)c));
                g[jj] = gfm[(uint32_t)bs[jj]];
            }
            rv[jj] = V::load(arena + (int64_t)(k[jj] < 0 ? 0 : k[jj]) * D + c);
        };
        // the group's last LATE rows are issued once its first LATE rows have been folded (their registers are free again:
        // 8 rows x 3 float4 in flight at once did not fit beside the chunk's constants — 27 spilled registers)
        constexpr int LATE = 0;  // (tried 2 at SEG = 8: the spill count did not move — the peak is in the matrix passes)
#pragma unroll
        for (int jj = 0; jj < SEG - LATE; ++jj) issue_row(jj);
        // ... and behind them the NEXT tile's keys (they land under the matrix passes and go to LDS after them)
        int32_t nk0, nk1;
        {
            const int64_t w1 = wg0 + TILE;
            const bool more = j + 1 < jn;
            tile_keys(more ? w1 : 0, more ? ((w1 + TILE < fend) ? w1 + TILE : fend) : 0, nk0, nk1);
        }
        SEG_BARRIER();  // (A0) the groups' piece counts (the row loads are in flight meanwhile)
        int pbase = 0, P = 0;
        {
            const int4 *g4 = reinterpret_cast<const int4 *>(gcnt);
#pragma unroll
            for (int q = 0; q < GPB / 4; ++q) {
                const int4 v4 = g4[q];
                const int vals[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    pbase += (4 * q + e < grp) ? vals[e] : 0;
                    P += vals[e];
                }
            }
        }
        // ---- segment sums in registers.  Masks are 0 / 1 factors, not selects: a select lets the compiler sink a row's
        //      loads into a branch, and then every wait is vmcnt(0) (rows beyond cnt read sample 0 / table row 0: real,
        //      finite data times zero) ----------------------------------------------------------------------------------
        f32x4 E[SEG];
        {
            f32x4 accH = V::zero(), accU = V::zero();
            float gs = 0.f;
            int slot = pbase;
#pragma unroll
            for (int jj = 0; jj < SEG; ++jj) {
                if (LATE > 0 && jj == LATE) {
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int u = SEG - LATE; u < SEG; ++u) issue_row(u);
                    __builtin_amdgcn_sched_barrier(0);
                }
                const float okf = (jj < cnt) ? 1.f : 0.f;
                accH += okf * rh[jj];
                if (HAS_FM) {
                    const float gj = okf * g[jj];
                    accU += gj * rs[jj];
                    gs += gj;
                }
                const bool last = (lastm >> jj) & 1u;
                const float keepf = last ? 0.f : 1.f;
                if (last) {  // (LDS stores only: no vector-memory operation sits behind this branch)
                    V::store(&HsT[slot][c], accH);
                    V::store(&VT[slot][c], rv[jj]);
                }
                slot += last ? 1 : 0;
                E[jj] = HAS_FM ? accU - gs * rv[jj] : V::zero();  // (used at piece ends only)
                accH = keepf * accH;
                if (HAS_FM) {
                    accU = keepf * accU;
                    gs = keepf * gs;
                }
            }
        }
        // the weight gradient's last k-step reads rows P .. (its multiple of 16): zero table rows there
        if (grp < ((P + 15) & ~15) - P) V::store(&VT[P + grp][c], V::zero());
        SEG_BARRIER();  // (A) the tiles are complete; wave 0 has left the previous tile's piece merge
        if (t == 0) {
            pkey[grp][0] = -1;
            pkey[grp][1] = -1;
            pcont[grp] = 0;
        }
        // ---- dgrad on the matrix core: C[row, d] = sum_hidden Hs[row, hidden] W1[hidden, f*64 + d] ------------------------
        // (m-blocks of 32 piece rows are dealt mh, mh + 2: with P <= 64 pieces both wave pairs still have one each)
        f32x16 ag[MB];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) ag[mb][r] = 0.f;
            if (32 * (mh + 2 * mb) >= P) continue;  // (workgroup-uniform)
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
        // ---- weight gradient: dW1^T[d, hidden] += sum_row V[row, d] Hs[row, hidden]  (K = the tile's rows) ----------------
        if (want_dw) {
            const int dblk = wv >> 1, hblk = wv & 1;
            const int nks = (P + 15) >> 4;
            __builtin_amdgcn_sched_barrier(0);  // (the dgrad's fragments are dead here: do not start this loop's reads early)
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
        SEG_BARRIER();  // (B) every wave is done reading the Hs tile: it becomes the C tile
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
            if (32 * (mh + 2 * mb) < P) {
#pragma unroll
                for (int r = 0; r < 16; ++r) HsT[32 * (mh + 2 * mb) + (r & 3) + 8 * (r >> 2) + 4 * h][32 * nb + i] = ag[mb][r];
            }
        tile_keys_store(nk0, nk1);  // (this tile's keys were last read in front of barrier A)
        SEG_BARRIER();  // (C)
        // ---- the pieces' gradient rows: C + u - s v; runs inside the group are stored, border pieces merged as in
        //      embed_grad_reduce_kernel (same piece list, same finish launches) -------------------------------------------
        {
            bool run_head = (k[0] != kprev);
            int slot = pbase;
#pragma unroll
            for (int jj = 0; jj < SEG; ++jj) {
                if ((lastm >> jj) & 1u) {
                    const f32x4 val = V::load(&HsT[slot++][c]) + E[jj];
                    const bool run_ends_here = (jj + 1 < cnt) ? true : (k[jj] != knext);
                    if (run_head && run_ends_here) {
                        float *dst = G + (int64_t)k[jj] * D + c;
                        V::store(dst, accumulate ? V::load(dst) + val : val);  // the only writer of this row
                    } else {
                        const int slot = run_head ? 1 : 0;
                        V::store(&piece[grp][slot][c], val);
                        if (t == 0) {
                            pkey[grp][slot] = k[jj];
                            if (!run_ends_here) pcont[grp] = 1;
                        }
                    }
                    run_head = true;
                }
            }
        }
        SEG_BARRIER();  // (D)
        if (grp == 0) {
            const int64_t tile_id = tile_id0 + j;
            float *hp = gpiece + tile_id * 2 * D, *tp = hp + D;
            const bool head_open = pkey[0][0] >= 0;
            int last_g = -1;
            for (int g2 = GPB - 1; g2 >= 0; --g2)
                if (pkey[g2][0] >= 0 || pkey[g2][1] >= 0) {
                    last_g = g2;
                    break;
                }
            const int last_active = (int)((gend - wg0 + SEG - 1) / SEG) - 1;
            const bool tail_open = last_g >= 0 && last_g == last_active && pcont[last_g] != 0;
            int32_t curk = -1;
            bool cur_is_head = false;
            f32x4 cacc = V::zero();
            int32_t headkey = -1, tailkey = -1;
            f32x4 headv = V::zero(), tailv = V::zero();
            for (int g2 = 0; g2 < GPB; ++g2) {
#pragma unroll
                for (int slot = 0; slot < 2; ++slot) {
                    const int32_t key = pkey[g2][slot];
                    if (key < 0) continue;
                    const f32x4 pv = V::load(&piece[g2][slot][c]);
                    if (key == curk) {
                        cacc += pv;
                    } else {
                        if (curk >= 0) {
                            if (cur_is_head) {
                                headkey = curk;
                                headv = cacc;
                            } else {
                                float *dst = G + (int64_t)curk * D + c;
                                V::store(dst, accumulate ? V::load(dst) + cacc : cacc);
                            }
                        }
                        curk = key;
                        cacc = pv;
                        cur_is_head = head_open && g2 == 0 && slot == 0;
                    }
                }
            }
            if (curk >= 0) {
                if (tail_open) {
                    if (cur_is_head) {
                        headkey = curk;
                        headv = cacc;
                        tailkey = curk;
                    } else {
                        tailkey = curk;
                        tailv = cacc;
                    }
                } else if (cur_is_head) {
                    headkey = curk;
                    headv = cacc;
                } else {
                    float *dst = G + (int64_t)curk * D + c;
                    V::store(dst, accumulate ? V::load(dst) + cacc : cacc);
                }
            }
            V::store(hp + c, headv);
            V::store(tp + c, tailv);
            if (t == 0) {
                gkey[2 * tile_id] = headkey;
                gkey[2 * tile_id + 1] = tailkey;
            }
        }
    }
    if (want_dw) {
        float *P = dwpart + (int64_t)chunk * (D * D);
        const int dblk = wv >> 1, hblk = wv & 1;
#pragma unroll
        for (int r = 0; r < 16; ++r) P[(32 * dblk + (r & 3) + 8 * (r >> 2) + 4 * h) * D + 32 * hblk + i] = dwacc[r];
    }
}

// dw[hidden, f*64 + d] = the fixed-order sum of the chunk partials [d][hidden] of field f.  One workgroup = 64 consecutive
// elements x four quarters of the chunk list (a wave each, eight loads in flight per lane: a first version — one thread per
// element walking all chunks four loads at a time — was latency-bound at 0.6 TB/s: 60 us for 38 MB), then the quarters are
// added in order through LDS.
__global__ __launch_bounds__(256) void embed_grad_seg_dw_kernel(const float *__restrict__ dwpart, SegFields sf, int cpf,
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

// rows per tile: 128 (two workgroups per CU) or 64 (three); RP_SEG_ROWS overrides the default
static int seg_tile_rows() {
    static const int rows = []() {
        const char *e = getenv("RP_SEG_ROWS");
        const int v = e ? atoi(e) : 0;
        return v == 64 ? 64 : 128;
    }();
    return rows;
}

// tiles per chunk: ~64 chunks per field (RP_SEG_TILES overrides)
static int seg_tiles_per_chunk(int tpf) {
    static const int forced = []() {
        const char *e = getenv("RP_SEG_TILES");
        return e ? atoi(e) : 0;
    }();
    // (measured at Criteo shape, 128-row tiles, compacted pieces: 8 tiles per chunk 0.310 ms, 4: 0.312, 2: 0.351, 16: 0.372 —
    //  with the first, latency-bound form of the partial sum; 8 halves the partials against 4)
    int T = forced > 0 ? forced : (tpf + 32) / 64;
    T = T < 1 ? 1 : (T > 64 ? 64 : T);
    return T > tpf ? tpf : T;
}

static int64_t seg_n_eff(int64_t n, int64_t B) {  // (the piece region is sized through rp_embed_grad_reduce's formula)
    const int64_t F = B > 0 ? n / B : 0;
    const int64_t tiles = F * rp_cdiv(B, seg_tile_rows());
    return n > tiles * 32 ? n : tiles * 32;
}

extern "C" int rp_embed_grad_seg_fits(int D, int hidden, int64_t lddh) {
    return (D == 64 && hidden == 64 && lddh % 4 == 0) ? 1 : 0;
}

extern "C" int rp_embed_grad_seg_workspace_bytes(int64_t n, int64_t B, int D, size_t *bytes) {
    RP_REQUIRE(bytes && n >= 0 && B >= 1 && D == 64 && n % B == 0 && n / B <= 64,
               "embed_grad_seg_workspace_bytes: needs D = 64 and field-major positions (n = F * B, F <= 64)");
    size_t red = 0;
    rp_embed_grad_reduce_workspace_bytes(seg_n_eff(n, B), D, &red);
    const int tpf = (int)rp_cdiv(B, seg_tile_rows());
    const int64_t cpf = rp_cdiv(tpf, seg_tiles_per_chunk(tpf));
    *bytes = ((red + 255) & ~(size_t)255) + (size_t)(n / B) * cpf * 4096 * sizeof(float) + 256;
    return RP_OK;
}

extern "C" int rp_embed_grad_seg(const int32_t *sorted_keys, const int32_t *sorted_pos, int64_t n, int64_t B, int D,
                                 const float *dh, int64_t lddh, const float *w, int64_t ldw, const float *gfm,
                                 const float *sum_in, const float *arena, float *grad_arena, int accumulate,
                                 uint64_t skip_fields, const int64_t *field_rows, float *dw, int64_t lddw, void *workspace,
                                 size_t workspace_bytes, rp_stream_t stream) {
    RP_REQUIRE(sorted_keys && sorted_pos && dh && w && arena && grad_arena && workspace, "embed_grad_seg: null pointer");
    RP_REQUIRE(B >= 1 && B < INT32_MAX && n >= 0 && n < INT32_MAX, "embed_grad_seg: bad B / n");
    RP_REQUIRE(n % B == 0 && n / B <= 64, "embed_grad_seg: needs field-major positions (n = F * B, F <= 64)");
    RP_REQUIRE((gfm == nullptr) == (sum_in == nullptr), "embed_grad_seg: the FM term needs both gfm and sum_in");
    const int F = (int)(n / B);
    RP_REQUIRE(ldw >= (int64_t)F * 64 && (dw == nullptr || lddw >= (int64_t)F * 64), "embed_grad_seg: weight rows shorter than F * 64");
    if (!rp_embed_grad_seg_fits(D, 64, lddh) || !rp_aligned16(dh) || !rp_aligned16(grad_arena) || !rp_aligned16(arena) ||
        (sum_in && !rp_aligned16(sum_in)))
        return rp_fail(RP_ERR_UNSUPPORTED, "embed_grad_seg: needs D = 64, a 64-wide layer and 16-byte aligned operands");
    if (n == 0) return RP_OK;
    size_t need = 0;
    if (rp_embed_grad_seg_workspace_bytes(n, B, D, &need) != RP_OK) return RP_ERR_ARG;
    RP_REQUIRE(workspace_bytes >= need, "embed_grad_seg: workspace %zu < %zu bytes", workspace_bytes, need);
    SegFields sf;
    memset(&sf, 0, sizeof(sf));
    int kept[64];
    for (int f = 0; f < F; ++f)
        if (!((skip_fields >> f) & 1u)) {
            sf.rank[f] = (unsigned char)sf.n;
            kept[sf.n++] = f;
        }
    if (sf.n == 0) return RP_OK;  // every field is handled elsewhere
    // chunk blocks in the order largest table first (insertion sort, stable: ties keep the field order)
    for (int a = 0; a < sf.n; ++a) {
        const int f = kept[a];
        int b = a;
        while (b > 0 && field_rows != nullptr && field_rows[sf.sched[b - 1]] < field_rows[f]) {
            sf.sched[b] = sf.sched[b - 1];
            --b;
        }
        sf.sched[b] = (unsigned char)f;
    }
    const int rows = seg_tile_rows();
    const int tpf = (int)rp_cdiv(B, rows);
    const int T = seg_tiles_per_chunk(tpf);
    const int cpf = (int)rp_cdiv(tpf, T);
    const int64_t n_eff = seg_n_eff(n, B);
    size_t red = 0;
    rp_embed_grad_reduce_workspace_bytes(n_eff, D, &red);
    char *wbase = reinterpret_cast<char *>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
    const int64_t nb0 = (int64_t)sf.n * tpf;
    float *piece0 = reinterpret_cast<float *>(wbase);
    int32_t *key0 = reinterpret_cast<int32_t *>(piece0 + nb0 * 2 * D);
    float *dwpart = dw ? reinterpret_cast<float *>(wbase + ((red + 255) & ~(size_t)255)) : nullptr;  // [chunks][64 d][64 hidden]
    hipStream_t s = (hipStream_t)stream;
    const unsigned grid = (unsigned)(sf.n * cpf);
#define SEG_LAUNCH(FM, SG)                                                                                                  \
    hipLaunchKernelGGL((embed_grad_seg_kernel<FM, SG>), dim3(grid), dim3(256), 0, s, sorted_keys, sorted_pos, n, (int)B, dh,  \
                       lddh, w, ldw, gfm, sum_in, arena, grad_arena, accumulate, piece0, key0, dwpart, sf, tpf, T, cpf)
    if (rows == 128) {
        if (gfm != nullptr) SEG_LAUNCH(true, 8);
        else SEG_LAUNCH(false, 8);
    } else {
        if (gfm != nullptr) SEG_LAUNCH(true, 4);
        else SEG_LAUNCH(false, 4);
    }
#undef SEG_LAUNCH
    RP_LAUNCH_CHECK("embed_grad_seg");
    if (dw != nullptr) {
        hipLaunchKernelGGL(embed_grad_seg_dw_kernel, dim3((unsigned)sf.n, 64), dim3(256), 0, s, dwpart, sf, cpf, dw, lddw);
        RP_LAUNCH_CHECK("embed_grad_seg (weight-gradient partials)");
    }
    return grad_reduce_finish(n_eff, D, nb0, wbase, piece0, key0, grad_arena, accumulate, s);
}

// (embed_ss.hip shares the piece-list finish of the segmented reduce)
int rp_int_grad_reduce_finish(int64_t n, int D, int64_t nb0, char *wbase, float *piece0, int32_t *key0, float *grad_arena,
                              int accumulate, hipStream_t s) {
    return grad_reduce_finish(n, D, nb0, wbase, piece0, key0, grad_arena, accumulate, s);
}

extern "C" int rp_zero_rows(const int32_t *keys, int64_t n, int D, float *grad_arena, rp_stream_t stream) {
    RP_REQUIRE(keys && grad_arena && D >= 1, "zero_rows: bad argument");
    if (n == 0) return RP_OK;
    const int vec = (D % 4 == 0 && rp_aligned16(grad_arena)) ? 4 : 1;
    const int tpr = pick_tpr(D, vec);
    const unsigned grid = (unsigned)rp_cdiv(n, 256 / tpr);
    hipStream_t s = (hipStream_t)stream;
#define CALL(T, VV) \
    hipLaunchKernelGGL((zero_rows_kernel<T, VV>), dim3(grid), dim3(256), 0, s, keys, n, D, grad_arena)
    RP_DISPATCH_TPR(tpr, vec, CALL);
#undef CALL
    RP_LAUNCH_CHECK("zero_rows");
    return RP_OK;
}

// keys_out[f*B + b] = row_base[f] + id_f[b] (int32 arena rows), with the same range check / flag / clamp as the
// gather.  Used when the rows must be known BEFORE the gather (lazy Adam replays the rows a batch is about to read).
__global__ __launch_bounds__(256) void embed_keys_kernel(const int64_t *__restrict__ row_base,
                                                         const int64_t *__restrict__ row_count, IdxPtrs idx, int F,
                                                         int64_t B, int32_t *__restrict__ keys_out,
                                                         int32_t *__restrict__ err_flag) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int f = blockIdx.y;
    if (b >= B || f >= F) return;
    int64_t id = idx.p[f][b];
    if (id < 0 || id >= row_count[f]) {
        *err_flag = 1;
        id = 0;
    }
    keys_out[(int64_t)f * B + b] = (int32_t)(row_base[f] + id);
}

extern "C" int rp_embed_keys(const int64_t *row_base, const int64_t *row_count, const int64_t *const *idx_ptrs, int F,
                             int64_t B, int32_t *keys_out, int32_t *err_flag, rp_stream_t stream) {
    RP_REQUIRE(row_base && row_count && idx_ptrs && keys_out && err_flag, "embed_keys: null pointer");
    RP_REQUIRE(F >= 1 && F <= RP_MAX_FIELDS && B >= 0, "embed_keys: bad F/B");
    RP_REQUIRE((int64_t)F * B < (int64_t)INT32_MAX, "embed_keys: F*B overflows int32 positions");
    if (B == 0) return RP_OK;
    IdxPtrs ip;
    for (int f = 0; f < F; ++f) {
        RP_REQUIRE(idx_ptrs[f], "embed_keys: idx_ptrs[%d] is null", f);
        ip.p[f] = idx_ptrs[f];
    }
    hipLaunchKernelGGL(embed_keys_kernel, dim3((unsigned)rp_cdiv(B, 256), (unsigned)F), dim3(256), 0,
                       (hipStream_t)stream, row_base, row_count, ip, F, B, keys_out, err_flag);
    RP_LAUNCH_CHECK("embed_keys");
    return RP_OK;
}
