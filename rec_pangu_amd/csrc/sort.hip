// (arena row, position) pair sort for the gather backward: a stable LSD radix sort over the low `end_bit` bits of the
// key (26 bits cover the 33.8 M rows of the Criteo-shape arena).  Stability keeps equal rows in ascending position
// (= ascending sample) order, which fixes the summation order downstream.
//
// Own kernels (the default): 9-bit digits, three launches per digit pass and NO memset / no host-side dispatch work.
//   sort_hist_kernel     one workgroup per 4096-pair tile: the tile's digit histogram -> hist[tile][512]
//   sort_scan_kernel     one wave per digit: exclusive prefix of that digit's counts over the tiles (in place) + its total
//   sort_scatter_kernel  the tile again: digit base (scan of the 512 totals, in LDS) + tile prefix + the stable rank of
//                        each pair inside the tile -> its place in the output
// The rank inside a tile needs no atomics and no per-thread counters: a wave walks its 1024 consecutive pairs 64 at a
// time; the lanes holding the same digit find each other with one ballot per digit bit (9), their rank is the number of
// lower peers, and the lowest peer bumps the wave's LDS counter of that digit — LDS operations of one wave execute in
// program order, so the 16 rounds chain without a barrier.  Waves then take their offsets from the waves before them.
// Positions are implicit in the first pass (no iota buffer).  26 key bits = 3 passes = 9 launches, ~35 MB of traffic
// per pass at 1.7 M pairs: latency-bound, like rocPRIM's onesweep (four 8-bit passes).  Measured (MI355X,
// profiles/microbench/probes/probe_sort_host.py): 1.7 M pairs 151 us on the device / 28 us of host time per call (rocPRIM 148 / 49);
// 213 k pairs (the per-GPU batch of a strong-scaling run) 77 us / 29 us (rocPRIM's merge sort 94 / 50); beside a step
// on the side stream the event-bracketed duration is 0.13 ms against rocPRIM's 0.33.  What the own kernels buy besides:
// NO memset nodes and no global counters carried between launches — a captured step replays at any batch size (with
// rocPRIM's onesweep, unsynchronised replays above ~1 M pairs ended in memory access faults), and the
// host-bound eager step at b = 8192 drops 1.04 -> 0.77 ms.
//
// RP_SORT=rocprim selects rocPRIM's device radix sort (onesweep above one million pairs, merge sort below) instead;
// tests/test_hip_kernels.py::test_sort_pairs_rocprim_path keeps it checked.  Wider rocPRIM digits measured slower
// (9 bits, match-based ranking: 162 us; 11 bits: 421 us).
#include "common.h"
#include <cstdlib>
#include <cstring>
#include <rocprim/rocprim.hpp>

namespace {

constexpr int SORT_RB = 9;                               // bits per digit
constexpr int SORT_BINS = 1 << SORT_RB;                  // 512
constexpr int SORT_THREADS = 256;
constexpr int SORT_WAVES = SORT_THREADS / RP_WAVE;       // 4
constexpr int SORT_ROUNDS = 16;                          // pairs per thread
constexpr int SORT_WAVE_SPAN = SORT_ROUNDS * RP_WAVE;    // 1024 consecutive pairs per wave
constexpr int SORT_TILE = SORT_THREADS * SORT_ROUNDS;    // 4096 pairs per workgroup

// the wave's digit counters: volatile (another lane's update must be re-read every round) and typed as LDS, so that the
// accesses stay ds_read / ds_write (a volatile access through a generic pointer compiles to flat_load ... sc0 sc1)
typedef __attribute__((address_space(3))) volatile int32_t lds_counter;

__device__ __forceinline__ uint32_t sort_digit(int32_t key, int shift, uint32_t mask, uint32_t flip) {
    return (((uint32_t)key ^ flip) >> shift) & mask;
}

// The wave's 16 rounds over keys k[] (already in registers): rk[r] = number of pairs of this WAVE that come before pair
// (r, lane) and hold the same digit; cnt[digit] ends as the wave's count of that digit.  cnt = this wave's 512 LDS
// counters, zeroed (and the zeroing made visible) by the caller.
template <bool KEEP>
__device__ __forceinline__ void sort_rank_wave(const int32_t (&k)[SORT_ROUNDS], int64_t first, int64_t n, int shift, int nb,
                                               uint32_t mask, uint32_t flip, lds_counter *cnt,
                                               int32_t (&rk)[SORT_ROUNDS]) {
    const int lane = threadIdx.x & (RP_WAVE - 1);
    const uint64_t lower = (1ull << lane) - 1ull;
#pragma unroll
    for (int r = 0; r < SORT_ROUNDS; ++r) {
        const bool valid = first + r * RP_WAVE + lane < n;
        const uint32_t d = sort_digit(k[r], shift, mask, flip);
        uint64_t peers = __builtin_amdgcn_ballot_w64(valid);
        for (int b = 0; b < nb; ++b) {
            const bool bit = (d >> b) & 1u;
            const uint64_t m = __builtin_amdgcn_ballot_w64(bit);
            peers &= bit ? m : ~m;
        }
        const int rank = __builtin_popcountll(peers & lower);
        int32_t base = 0;
        if (valid) {
            base = cnt[d];
            if (rank == 0) cnt[d] = base + (int32_t)__builtin_popcountll(peers);
        }
        if (KEEP) rk[r] = base + rank;
    }
}

__device__ __forceinline__ void sort_load_keys(const int32_t *__restrict__ keys, int64_t first, int64_t n,
                                               int32_t (&k)[SORT_ROUNDS]) {
    const int lane = threadIdx.x & (RP_WAVE - 1);
#pragma unroll
    for (int r = 0; r < SORT_ROUNDS; ++r) {
        const int64_t i = first + r * RP_WAVE + lane;
        k[r] = i < n ? keys[i] : 0;
    }
}

__global__ __launch_bounds__(SORT_THREADS) void sort_hist_kernel(const int32_t *__restrict__ keys, int64_t n, int shift,
                                                                 int nb, uint32_t flip, int32_t *__restrict__ hist) {
    __shared__ int32_t cnt[SORT_WAVES][SORT_BINS];
    const int w = threadIdx.x / RP_WAVE;
    for (int d = threadIdx.x; d < SORT_WAVES * SORT_BINS; d += SORT_THREADS) (&cnt[0][0])[d] = 0;
    const int64_t first = (int64_t)blockIdx.x * SORT_TILE + (int64_t)w * SORT_WAVE_SPAN;
    int32_t k[SORT_ROUNDS], rk[SORT_ROUNDS];
    sort_load_keys(keys, first, n, k);
    __syncthreads();
    sort_rank_wave<false>(k, first, n, shift, nb, (1u << nb) - 1u, flip, (lds_counter *)cnt[w], rk);
    __syncthreads();
    for (int d = threadIdx.x; d < SORT_BINS; d += SORT_THREADS)
        hist[(int64_t)blockIdx.x * SORT_BINS + d] = cnt[0][d] + cnt[1][d] + cnt[2][d] + cnt[3][d];
}

// grid = 512 / 4 workgroups of four waves; wave -> one digit
__global__ __launch_bounds__(SORT_THREADS) void sort_scan_kernel(int32_t *__restrict__ hist, int64_t ntiles,
                                                                 int32_t *__restrict__ totals) {
    const int lane = threadIdx.x & (RP_WAVE - 1);
    const int d = blockIdx.x * SORT_WAVES + threadIdx.x / RP_WAVE;
    int32_t carry = 0;
    for (int64_t t0 = 0; t0 < ntiles; t0 += RP_WAVE) {
        const int64_t t = t0 + lane;
        const int32_t v = t < ntiles ? hist[t * SORT_BINS + d] : 0;
        int32_t s = v;
#pragma unroll
        for (int o = 1; o < RP_WAVE; o <<= 1) {
            const int32_t up = __shfl_up(s, o, RP_WAVE);
            if (lane >= o) s += up;
        }
        if (t < ntiles) hist[t * SORT_BINS + d] = carry + s - v;
        carry += __shfl(s, RP_WAVE - 1, RP_WAVE);
    }
    if (lane == 0) totals[d] = carry;
}

// vin == nullptr: the values are the positions themselves (first pass)
__global__ __launch_bounds__(SORT_THREADS) void sort_scatter_kernel(const int32_t *__restrict__ kin,
                                                                    const int32_t *__restrict__ vin,
                                                                    int32_t *__restrict__ kout, int32_t *__restrict__ vout,
                                                                    int64_t n, int shift, int nb, uint32_t flip,
                                                                    const int32_t *__restrict__ hist,
                                                                    const int32_t *__restrict__ totals) {
    __shared__ int32_t cnt[SORT_WAVES][SORT_BINS];
    __shared__ int32_t dbase[SORT_BINS];
    __shared__ int32_t wsum[SORT_WAVES];
    const int tid = threadIdx.x, lane = tid & (RP_WAVE - 1), w = tid / RP_WAVE;
    const uint32_t mask = (1u << nb) - 1u;
    for (int d = tid; d < SORT_WAVES * SORT_BINS; d += SORT_THREADS) (&cnt[0][0])[d] = 0;
    const int64_t first = (int64_t)blockIdx.x * SORT_TILE + (int64_t)w * SORT_WAVE_SPAN;
    int32_t k[SORT_ROUNDS], v[SORT_ROUNDS], rk[SORT_ROUNDS];
    sort_load_keys(kin, first, n, k);
#pragma unroll
    for (int r = 0; r < SORT_ROUNDS; ++r) {
        const int64_t i = first + r * RP_WAVE + lane;
        v[r] = vin ? (i < n ? vin[i] : 0) : (int32_t)i;
    }
    // digit bases: exclusive scan of the 512 digit totals, two digits per thread
    {
        const int32_t a = totals[2 * tid], b = totals[2 * tid + 1];
        int32_t s = a + b;
#pragma unroll
        for (int o = 1; o < RP_WAVE; o <<= 1) {
            const int32_t up = __shfl_up(s, o, RP_WAVE);
            if (lane >= o) s += up;
        }
        if (lane == RP_WAVE - 1) wsum[w] = s;
        __syncthreads();  // also publishes the zeroed counters
        int32_t before = 0;
        for (int x = 0; x < w; ++x) before += wsum[x];
        const int32_t excl = before + s - (a + b);
        dbase[2 * tid] = excl;
        dbase[2 * tid + 1] = excl + a;
    }
    sort_rank_wave<true>(k, first, n, shift, nb, mask, flip, (lds_counter *)cnt[w], rk);
    __syncthreads();
    // counters -> offsets: where wave x's first pair of digit d goes
    for (int d = tid; d < SORT_BINS; d += SORT_THREADS) {
        int32_t run = dbase[d] + hist[(int64_t)blockIdx.x * SORT_BINS + d];
#pragma unroll
        for (int x = 0; x < SORT_WAVES; ++x) {
            const int32_t c = cnt[x][d];
            cnt[x][d] = run;
            run += c;
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < SORT_ROUNDS; ++r) {
        if (first + r * RP_WAVE + lane < n) {
            const int32_t dst = cnt[w][sort_digit(k[r], shift, mask, flip)] + rk[r];
            kout[dst] = k[r];
            vout[dst] = v[r];
        }
    }
}

__global__ void iota_i32_kernel(int32_t *out, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (int32_t)i;
}

size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

bool sort_use_rocprim() {
    static const bool v = [] {
        const char *e = getenv("RP_SORT");
        return e && strcmp(e, "rocprim") == 0;
    }();
    return v;
}

hipError_t sort_dispatch(void *temp, size_t &tb, const int32_t *ki, int32_t *ko, const int32_t *vi, int32_t *vo, int64_t n,
                         int end_bit, hipStream_t s) {
    return rocprim::radix_sort_pairs<rocprim::default_config>(temp, tb, ki, ko, vi, vo, (size_t)n, 0u, (unsigned)end_bit, s);
}

// rocPRIM's size query walks its config dispatch and asks the runtime for the device on every call (~0.1 ms of host time)
int sort_bytes_rocprim(int64_t n, size_t *bytes) {
    static thread_local int64_t cached_n = -1;
    static thread_local size_t cached_tb = 0;
    if (n == cached_n) {
        *bytes = cached_tb;
        return RP_OK;
    }
    size_t tb = 0;
    hipError_t e = sort_dispatch(nullptr, tb, nullptr, nullptr, nullptr, nullptr, n, 32, nullptr);
    if (e != hipSuccess) return rp_fail(RP_ERR_LAUNCH, "sort size query: %s", hipGetErrorString(e));
    cached_n = n;
    cached_tb = tb;
    *bytes = tb;
    return RP_OK;
}

// own kernels: [keys n][values n][hist tiles x 512][totals 512]
struct SortPlan {
    int64_t ntiles;
    size_t off_vals, off_hist, off_totals, bytes;
};

SortPlan sort_plan(int64_t n) {
    SortPlan p;
    p.ntiles = rp_cdiv(n, SORT_TILE);
    const size_t col = align256((size_t)n * sizeof(int32_t));
    p.off_vals = col;
    p.off_hist = 2 * col;
    p.off_totals = p.off_hist + align256((size_t)p.ntiles * SORT_BINS * sizeof(int32_t));
    p.bytes = p.off_totals + align256(SORT_BINS * sizeof(int32_t));
    return p;
}

}  // namespace

// ---- the FIELD-SEGMENTED form (round 6) --------------------------------------------------------------------------------------
// The pair list of a lookup is field-major (position = field * B + sample) and a field's keys are that field's table rows:
// sorting the whole list by arena row is F independent sorts, one per segment of B pairs, each by the row INSIDE the table —
// and a table of r rows needs ceil(log2 r) key bits, not the arena's 26: at Criteo shape 9 of the 26 fields are done after ONE
// 9-bit pass, 11 more after two, only the 6 tables of more than 2^18 rows take the third: 49 field-passes instead of 78.
// Same three kernels per pass, over the fields that still have bits left (blockIdx.y = the field's place among them); a
// field's pass reads and writes its own segment [f B, (f + 1) B) of whichever buffer its remaining pass count makes the
// source / the destination (the last pass of every field lands in the caller's buffers).  The result is the global stable sort
// by arena row, bit for bit (tables sit in the arena in field order).
struct SortSegs {
    int n;                         // fields in this pass
    unsigned char field[RP_MAX_FIELDS];
    unsigned char src[RP_MAX_FIELDS], dst[RP_MAX_FIELDS];  // 0 = the caller's input (values = positions), 1 = workspace, 2 = output
    int32_t base[RP_MAX_FIELDS];   // first arena row of the field's table
};
struct SortBufs {
    const int32_t *k[3];
    const int32_t *v[3];
};

__global__ __launch_bounds__(SORT_THREADS) void sort_seg_hist_kernel(SortBufs bufs, SortSegs sg, int Bi, int tpf, int shift,
                                                                     int32_t *__restrict__ hist) {
    __shared__ int32_t cnt[SORT_WAVES][SORT_BINS];
    const int y = blockIdx.y, f = sg.field[y], w = threadIdx.x / RP_WAVE;
    for (int d = threadIdx.x; d < SORT_WAVES * SORT_BINS; d += SORT_THREADS) (&cnt[0][0])[d] = 0;
    const int64_t seg0 = (int64_t)f * Bi, seg1 = seg0 + Bi;
    const int64_t first = seg0 + (int64_t)blockIdx.x * SORT_TILE + (int64_t)w * SORT_WAVE_SPAN;
    int32_t k[SORT_ROUNDS], rk[SORT_ROUNDS];
    sort_load_keys(bufs.k[sg.src[y]], first, seg1, k);
    const int32_t base = sg.base[y];
#pragma unroll
    for (int r = 0; r < SORT_ROUNDS; ++r) k[r] -= base;
    __syncthreads();
    sort_rank_wave<false>(k, first, seg1, shift, SORT_RB, SORT_BINS - 1, 0u, (lds_counter *)cnt[w], rk);
    __syncthreads();
    for (int d = threadIdx.x; d < SORT_BINS; d += SORT_THREADS)
        hist[((int64_t)y * tpf + blockIdx.x) * SORT_BINS + d] = cnt[0][d] + cnt[1][d] + cnt[2][d] + cnt[3][d];
}

// grid = (512 / 4, fields of the pass); wave -> one digit of one field
__global__ __launch_bounds__(SORT_THREADS) void sort_seg_scan_kernel(int32_t *__restrict__ hist, int tpf, int32_t *__restrict__ totals) {
    const int lane = threadIdx.x & (RP_WAVE - 1);
    const int d = blockIdx.x * SORT_WAVES + threadIdx.x / RP_WAVE;
    int32_t *h = hist + (int64_t)blockIdx.y * tpf * SORT_BINS;
    int32_t carry = 0;
    for (int t0 = 0; t0 < tpf; t0 += RP_WAVE) {
        const int t = t0 + lane;
        const int32_t v = t < tpf ? h[(int64_t)t * SORT_BINS + d] : 0;
        int32_t s = v;
#pragma unroll
        for (int o = 1; o < RP_WAVE; o <<= 1) {
            const int32_t up = __shfl_up(s, o, RP_WAVE);
            if (lane >= o) s += up;
        }
        if (t < tpf) h[(int64_t)t * SORT_BINS + d] = carry + s - v;
        carry += __shfl(s, RP_WAVE - 1, RP_WAVE);
    }
    if (lane == 0) totals[(int64_t)blockIdx.y * SORT_BINS + d] = carry;
}

__global__ __launch_bounds__(SORT_THREADS) void sort_seg_scatter_kernel(SortBufs bufs, int32_t *__restrict__ k1, int32_t *__restrict__ v1,
                                                                        int32_t *__restrict__ k2, int32_t *__restrict__ v2, SortSegs sg,
                                                                        int Bi, int tpf, int shift, const int32_t *__restrict__ hist,
                                                                        const int32_t *__restrict__ totals) {
    __shared__ int32_t cnt[SORT_WAVES][SORT_BINS];
    __shared__ int32_t dbase[SORT_BINS];
    __shared__ int32_t wsum[SORT_WAVES];
    const int tid = threadIdx.x, lane = tid & (RP_WAVE - 1), w = tid / RP_WAVE;
    const int y = blockIdx.y, f = sg.field[y];
    for (int d = tid; d < SORT_WAVES * SORT_BINS; d += SORT_THREADS) (&cnt[0][0])[d] = 0;
    const int64_t seg0 = (int64_t)f * Bi, seg1 = seg0 + Bi;
    const int64_t first = seg0 + (int64_t)blockIdx.x * SORT_TILE + (int64_t)w * SORT_WAVE_SPAN;
    const int32_t base = sg.base[y];
    const int32_t *vin = bufs.v[sg.src[y]];
    int32_t k[SORT_ROUNDS], v[SORT_ROUNDS], rk[SORT_ROUNDS];
    sort_load_keys(bufs.k[sg.src[y]], first, seg1, k);
#pragma unroll
    for (int r = 0; r < SORT_ROUNDS; ++r) {
        const int64_t i = first + r * RP_WAVE + lane;
        v[r] = vin ? (i < seg1 ? vin[i] : 0) : (int32_t)i;
        k[r] -= base;
    }
    {
        const int32_t *tt = totals + (int64_t)y * SORT_BINS;
        const int32_t a = tt[2 * tid], b = tt[2 * tid + 1];
        int32_t s = a + b;
#pragma unroll
        for (int o = 1; o < RP_WAVE; o <<= 1) {
            const int32_t up = __shfl_up(s, o, RP_WAVE);
            if (lane >= o) s += up;
        }
        if (lane == RP_WAVE - 1) wsum[w] = s;
        __syncthreads();  // also publishes the zeroed counters
        int32_t before = 0;
        for (int x = 0; x < w; ++x) before += wsum[x];
        const int32_t excl = before + s - (a + b);
        dbase[2 * tid] = excl;
        dbase[2 * tid + 1] = excl + a;
    }
    sort_rank_wave<true>(k, first, seg1, shift, SORT_RB, SORT_BINS - 1, 0u, (lds_counter *)cnt[w], rk);
    __syncthreads();
    for (int d = tid; d < SORT_BINS; d += SORT_THREADS) {
        int32_t run = dbase[d] + hist[((int64_t)y * tpf + blockIdx.x) * SORT_BINS + d];
#pragma unroll
        for (int x = 0; x < SORT_WAVES; ++x) {
            const int32_t c = cnt[x][d];
            cnt[x][d] = run;
            run += c;
        }
    }
    __syncthreads();
    int32_t *kout = sg.dst[y] == 2 ? k2 : k1, *vout = sg.dst[y] == 2 ? v2 : v1;
#pragma unroll
    for (int r = 0; r < SORT_ROUNDS; ++r) {
        if (first + r * RP_WAVE + lane < seg1) {
            const int64_t dst = seg0 + cnt[w][sort_digit(k[r], shift, SORT_BINS - 1, 0u)] + rk[r];
            kout[dst] = k[r] + base;
            vout[dst] = v[r];
        }
    }
}

struct SortSegPlan {
    int tpf;
    size_t off_vals, off_hist, off_totals, bytes;
};

static SortSegPlan sort_seg_plan(int64_t B, int F) {
    SortSegPlan p;
    p.tpf = (int)rp_cdiv(B, SORT_TILE);
    const size_t col = align256((size_t)B * F * sizeof(int32_t));
    p.off_vals = col;
    p.off_hist = 2 * col;
    p.off_totals = p.off_hist + align256((size_t)F * p.tpf * SORT_BINS * sizeof(int32_t));
    p.bytes = p.off_totals + align256((size_t)F * SORT_BINS * sizeof(int32_t));
    return p;
}

extern "C" int rp_sort_pairs_fields_workspace_bytes(int64_t B, int F, size_t *bytes) {
    RP_REQUIRE(bytes && B >= 1 && F >= 1 && F <= RP_MAX_FIELDS && B * F < INT32_MAX, "sort_pairs_fields_workspace_bytes: bad argument");
    *bytes = sort_seg_plan(B, F).bytes + 256;
    return RP_OK;
}

extern "C" int rp_sort_pairs_fields_i32(void *workspace, size_t workspace_bytes, const int32_t *keys_in, int32_t *keys_out,
                                        int32_t *pos_out, int64_t B, int F, const int64_t *field_base, const int64_t *field_rows,
                                        rp_stream_t stream) {
    RP_REQUIRE(workspace && keys_in && keys_out && pos_out && field_base && field_rows, "sort_pairs_fields: null pointer");
    RP_REQUIRE(B >= 1 && F >= 1 && F <= RP_MAX_FIELDS && B * F < INT32_MAX, "sort_pairs_fields: bad B / F");
    RP_REQUIRE(keys_in != keys_out && keys_in != pos_out && keys_out != pos_out, "sort_pairs_fields: buffers alias");
    size_t need = 0;
    rp_sort_pairs_fields_workspace_bytes(B, F, &need);
    RP_REQUIRE(workspace_bytes >= need, "sort_pairs_fields: workspace %zu < %zu bytes", workspace_bytes, need);
    int npass[RP_MAX_FIELDS], maxp = 1;
    for (int f = 0; f < F; ++f) {
        RP_REQUIRE(field_rows[f] >= 1 && field_base[f] >= 0 && field_base[f] + field_rows[f] <= (int64_t)INT32_MAX,
                   "sort_pairs_fields: table %d (base %lld, %lld rows) does not fit int32 keys", f, (long long)field_base[f],
                   (long long)field_rows[f]);
        RP_REQUIRE(f == 0 || field_base[f] >= field_base[f - 1] + field_rows[f - 1],
                   "sort_pairs_fields: the tables must sit in the arena in field order");
        int bits = 1;
        while (((int64_t)1 << bits) < field_rows[f]) ++bits;
        npass[f] = (bits + SORT_RB - 1) / SORT_RB;
        if (npass[f] > maxp) maxp = npass[f];
    }
    hipStream_t s = (hipStream_t)stream;
    char *base = reinterpret_cast<char *>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
    const SortSegPlan p = sort_seg_plan(B, F);
    int32_t *tmp_k = reinterpret_cast<int32_t *>(base);
    int32_t *tmp_v = reinterpret_cast<int32_t *>(base + p.off_vals);
    int32_t *hist = reinterpret_cast<int32_t *>(base + p.off_hist);
    int32_t *totals = reinterpret_cast<int32_t *>(base + p.off_totals);
    SortBufs bufs;
    bufs.k[0] = keys_in;
    bufs.v[0] = nullptr;  // positions are implicit in the first pass
    bufs.k[1] = tmp_k;
    bufs.v[1] = tmp_v;
    bufs.k[2] = keys_out;
    bufs.v[2] = pos_out;
    for (int pass = 0; pass < maxp; ++pass) {
        SortSegs sg;
        memset(&sg, 0, sizeof(sg));
        for (int f = 0; f < F; ++f) {
            if (npass[f] <= pass) continue;
            const int y = sg.n++;
            sg.field[y] = (unsigned char)f;
            sg.base[y] = (int32_t)field_base[f];
            // the field's last pass lands in the caller's buffers; the passes before it alternate backwards from there
            const bool to_out = ((npass[f] - 1 - pass) & 1) == 0;
            sg.dst[y] = to_out ? 2 : 1;
            sg.src[y] = pass == 0 ? 0 : (to_out ? 1 : 2);
        }
        const int shift = pass * SORT_RB;
        hipLaunchKernelGGL(sort_seg_hist_kernel, dim3((unsigned)p.tpf, (unsigned)sg.n), dim3(SORT_THREADS), 0, s, bufs, sg, (int)B, p.tpf,
                           shift, hist);
        hipLaunchKernelGGL(sort_seg_scan_kernel, dim3(SORT_BINS / SORT_WAVES, (unsigned)sg.n), dim3(SORT_THREADS), 0, s, hist, p.tpf, totals);
        hipLaunchKernelGGL(sort_seg_scatter_kernel, dim3((unsigned)p.tpf, (unsigned)sg.n), dim3(SORT_THREADS), 0, s, bufs, tmp_k, tmp_v,
                           keys_out, pos_out, sg, (int)B, p.tpf, shift, hist, totals);
    }
    RP_LAUNCH_CHECK("sort_pairs_fields");
    return RP_OK;
}

extern "C" int rp_sort_workspace_bytes(int64_t n, size_t *bytes) {
    RP_REQUIRE(bytes && n >= 0 && n < INT32_MAX, "sort_workspace_bytes: bad argument");
    const int64_t m = n > 0 ? n : 1;
    if (!sort_use_rocprim()) {
        *bytes = sort_plan(m).bytes + 256;
        return RP_OK;
    }
    size_t tb = 0;
    int rc = sort_bytes_rocprim(m, &tb);
    if (rc != RP_OK) return rc;
    *bytes = align256((size_t)m * sizeof(int32_t)) + align256(tb) + 256;
    return RP_OK;
}

extern "C" int rp_sort_pairs_i32(void *workspace, size_t workspace_bytes, const int32_t *keys_in, int32_t *keys_out,
                                 int32_t *pos_out, int64_t n, int end_bit, rp_stream_t stream) {
    RP_REQUIRE(workspace && keys_in && keys_out && pos_out, "sort_pairs: null pointer");
    RP_REQUIRE(n >= 0 && n < INT32_MAX && end_bit >= 1 && end_bit <= 32, "sort_pairs: bad n/end_bit");
    RP_REQUIRE(keys_in != keys_out && keys_in != pos_out && keys_out != pos_out, "sort_pairs: buffers alias");
    if (n == 0) return RP_OK;
    size_t need = 0;
    int rc = rp_sort_workspace_bytes(n, &need);
    if (rc != RP_OK) return rc;
    RP_REQUIRE(workspace_bytes >= need, "sort_pairs: workspace %zu < %zu bytes", workspace_bytes, need);
    hipStream_t s = (hipStream_t)stream;
    char *base = reinterpret_cast<char *>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);

    if (sort_use_rocprim()) {
        size_t tb = 0;
        rc = sort_bytes_rocprim(n, &tb);
        if (rc != RP_OK) return rc;
        int32_t *iota = reinterpret_cast<int32_t *>(base);
        void *temp = base + align256((size_t)n * sizeof(int32_t));
        hipLaunchKernelGGL(iota_i32_kernel, dim3((unsigned)rp_cdiv(n, 256)), dim3(256), 0, s, iota, n);
        RP_LAUNCH_CHECK("sort iota");
        hipError_t e = sort_dispatch(temp, tb, keys_in, keys_out, iota, pos_out, n, end_bit, s);
        if (e != hipSuccess) return rp_fail(RP_ERR_LAUNCH, "sort_pairs: %s", hipGetErrorString(e));
        rp_count_launch();
        return RP_OK;
    }

    const SortPlan p = sort_plan(n);
    int32_t *tmp_k = reinterpret_cast<int32_t *>(base);
    int32_t *tmp_v = reinterpret_cast<int32_t *>(base + p.off_vals);
    int32_t *hist = reinterpret_cast<int32_t *>(base + p.off_hist);
    int32_t *totals = reinterpret_cast<int32_t *>(base + p.off_totals);
    const int passes = (end_bit + SORT_RB - 1) / SORT_RB;
    // a key of all 32 bits is a SIGNED int: flipping the sign bit makes the unsigned digit order the signed order
    const uint32_t flip = end_bit == 32 ? 0x80000000u : 0u;
    const int32_t *src_k = keys_in, *src_v = nullptr;
    for (int pass = 0; pass < passes; ++pass) {
        const int shift = pass * SORT_RB;
        const int nb = end_bit - shift < SORT_RB ? end_bit - shift : SORT_RB;
        const bool to_out = ((passes - 1 - pass) & 1) == 0;  // the last pass lands in the caller's buffers
        int32_t *dst_k = to_out ? keys_out : tmp_k, *dst_v = to_out ? pos_out : tmp_v;
        hipLaunchKernelGGL(sort_hist_kernel, dim3((unsigned)p.ntiles), dim3(SORT_THREADS), 0, s, src_k, n, shift, nb, flip, hist);
        hipLaunchKernelGGL(sort_scan_kernel, dim3(SORT_BINS / SORT_WAVES), dim3(SORT_THREADS), 0, s, hist, p.ntiles, totals);
        hipLaunchKernelGGL(sort_scatter_kernel, dim3((unsigned)p.ntiles), dim3(SORT_THREADS), 0, s, src_k, src_v, dst_k, dst_v,
                           n, shift, nb, flip, hist, totals);
        src_k = dst_k;
        src_v = dst_v;
    }
    RP_LAUNCH_CHECK("sort_pairs");
    return RP_OK;
}
