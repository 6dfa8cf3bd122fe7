// Request routing for row-sharded tables (rec_pangu_amd/sharded.py: one process per GPU, arena row r lives on rank
// r % G at local row r / G).  Nothing like this exists in the reference (single device); these two entries replace the
// ~30 elementwise / unique / bincount launches the host side would otherwise spend on building one exchange:
//   rp_shard_keys   ids of a batch -> composite keys (owner << lbits | local row), range-checked like the gather
//   rp_route_build  sorted composite keys -> unique-request slots, the rows to ask each owner for, per-owner counts
#include "common.h"
#include <cstring>

struct RouteIdx {
    const int64_t *p[RP_MAX_FIELDS];
};

__global__ __launch_bounds__(256) void shard_keys_kernel(const int64_t *__restrict__ row_base,
                                                         const int64_t *__restrict__ row_count, RouteIdx idx, int F,
                                                         int64_t B, int world, int lbits, int32_t *__restrict__ keys_out,
                                                         int32_t *__restrict__ err_flag) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int f = blockIdx.y;
    if (b >= B || f >= F) return;
    int64_t id = idx.p[f][b];
    if (id < 0 || id >= row_count[f]) {  // like the gather: flag, then row 0 so the exchange stays well-formed
        *err_flag = 1;
        id = 0;
    }
    const int64_t row = row_base[f] + id;
    keys_out[(int64_t)f * B + b] = (int32_t)(((row % world) << lbits) | (row / world));
}

extern "C" int rp_shard_keys(const int64_t *row_base, const int64_t *row_count, const int64_t *const *idx_ptrs, int F,
                             int64_t B, int world, int lbits, int32_t *keys_out, int32_t *err_flag, rp_stream_t stream) {
    RP_REQUIRE(row_base && row_count && idx_ptrs && keys_out && err_flag, "shard_keys: null pointer");
    RP_REQUIRE(F >= 1 && F <= RP_MAX_FIELDS && B >= 0, "shard_keys: bad F/B");
    RP_REQUIRE(world >= 1 && lbits >= 1 && lbits <= 30, "shard_keys: bad world/lbits");
    int obits = 1;
    while ((1 << obits) < world) ++obits;
    RP_REQUIRE(lbits + obits <= 31, "shard_keys: owner (%d bits) + local row (%d bits) do not fit an int32 key", obits, lbits);
    RP_REQUIRE((int64_t)F * B < (int64_t)INT32_MAX, "shard_keys: F*B overflows int32 positions");
    if (B == 0) return RP_OK;
    RouteIdx ip;
    for (int f = 0; f < F; ++f) {
        RP_REQUIRE(idx_ptrs[f], "shard_keys: idx_ptrs[%d] is null", f);
        ip.p[f] = idx_ptrs[f];
    }
    hipLaunchKernelGGL(shard_keys_kernel, dim3((unsigned)rp_cdiv(B, 256), (unsigned)F), dim3(256), 0, (hipStream_t)stream,
                       row_base, row_count, ip, F, B, world, lbits, keys_out, err_flag);
    RP_LAUNCH_CHECK("shard_keys");
    return RP_OK;
}

__global__ __launch_bounds__(256) void route_flags_kernel(const int32_t *__restrict__ sk, int64_t n, int32_t *__restrict__ flags,
                                                          int64_t *__restrict__ starts, int world) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) flags[j] = (j == 0 || sk[j] != sk[j - 1]) ? 1 : 0;
    if (j <= world) starts[j] = -1;  // (world + 1 <= n is not required: see the launch)
}

__global__ __launch_bounds__(256) void route_scatter_kernel(const int32_t *__restrict__ sk, const int32_t *__restrict__ sp,
                                                            const int32_t *__restrict__ incl, int64_t n, int lbits,
                                                            int32_t *__restrict__ slot_sorted,
                                                            int64_t *__restrict__ slot_of_pair,
                                                            int64_t *__restrict__ uniq_rows, int64_t *__restrict__ starts) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const int32_t key = sk[j];
    const int32_t slot = incl[j] - 1;
    slot_sorted[j] = slot;
    slot_of_pair[sp[j]] = slot;
    const bool head = (j == 0) || (sk[j - 1] != key);
    if (head) {
        uniq_rows[slot] = (int64_t)(key & ((1 << lbits) - 1));
        const int owner = key >> lbits;
        if (j == 0 || (sk[j - 1] >> lbits) != owner) starts[owner] = slot;  // first unique request of this owner
    }
}

// counts[o] = unique requests owned by o (o < world), counts[world] = total; owners nobody asked have start -1
__global__ void route_counts_kernel(const int32_t *__restrict__ incl, int64_t n, int world, const int64_t *__restrict__ starts,
                                    int64_t *__restrict__ counts) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const int64_t total = incl[n - 1];
    int64_t nxt = total;
    for (int o = world - 1; o >= 0; --o) {
        const int64_t st = starts[o];
        if (st < 0) {
            counts[o] = 0;
        } else {
            counts[o] = nxt - st;
            nxt = st;
        }
    }
    counts[world] = total;
}

static size_t route_align(size_t x) { return (x + 255) & ~(size_t)255; }

// ---- inclusive scan of the head flags, own kernels (round 5: rocprim::inclusive_scan sat on the sharded step's hot path —
//      a foreign launch sequence with its own temporary storage query).  Three launches: per-block sums of 2048 flags, one
//      workgroup scanning the block sums, per-block scan + offset.  int32 counts, n < 2^31.
#define RS_PER_THREAD 8
#define RS_BLOCK (256 * RS_PER_THREAD)

__device__ __forceinline__ int32_t route_block_exclusive(int32_t v, int32_t *total) {
    // exclusive prefix of one value per thread over the 256-thread workgroup (wave shuffles + one trip through LDS)
    __shared__ int32_t wsum[4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int32_t incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int32_t up = __shfl_up(incl, o, 64);
        if (lane >= o) incl += up;
    }
    if (lane == 63) wsum[wv] = incl;
    __syncthreads();
    int32_t base = 0;
    for (int q = 0; q < wv; ++q) base += wsum[q];
    *total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    __syncthreads();  // (wsum is reused by the caller's next call)
    return base + incl - v;
}

__global__ __launch_bounds__(256) void route_scan_sums_kernel(const int32_t *__restrict__ flags, int64_t n, int32_t *__restrict__ bsum) {
    const int64_t b0 = (int64_t)blockIdx.x * RS_BLOCK + (int64_t)threadIdx.x * RS_PER_THREAD;
    int32_t s = 0;
#pragma unroll
    for (int e = 0; e < RS_PER_THREAD; ++e) s += (b0 + e < n) ? flags[b0 + e] : 0;
    int32_t total;
    (void)route_block_exclusive(s, &total);
    if (threadIdx.x == 0) bsum[blockIdx.x] = total;
}

// one workgroup: bsum[i] <- exclusive prefix of the block sums (chunks of 256 with a running carry)
__global__ __launch_bounds__(256) void route_scan_blocks_kernel(int32_t *__restrict__ bsum, int64_t nb) {
    int32_t carry = 0;
    for (int64_t c0 = 0; c0 < nb; c0 += 256) {
        const int64_t i = c0 + threadIdx.x;
        const int32_t v = i < nb ? bsum[i] : 0;
        int32_t total;
        const int32_t ex = route_block_exclusive(v, &total);
        if (i < nb) bsum[i] = carry + ex;
        carry += total;
    }
}

__global__ __launch_bounds__(256) void route_scan_final_kernel(const int32_t *__restrict__ flags, int64_t n,
                                                               const int32_t *__restrict__ boff, int32_t *__restrict__ incl) {
    const int64_t b0 = (int64_t)blockIdx.x * RS_BLOCK + (int64_t)threadIdx.x * RS_PER_THREAD;
    int32_t v[RS_PER_THREAD], s = 0;
#pragma unroll
    for (int e = 0; e < RS_PER_THREAD; ++e) {
        v[e] = (b0 + e < n) ? flags[b0 + e] : 0;
        s += v[e];
    }
    int32_t total;
    int32_t run = boff[blockIdx.x] + route_block_exclusive(s, &total);
#pragma unroll
    for (int e = 0; e < RS_PER_THREAD; ++e) {
        run += v[e];
        if (b0 + e < n) incl[b0 + e] = run;
    }
}

static size_t route_scan_bytes(int64_t n) { return route_align((size_t)rp_cdiv(n, RS_BLOCK) * sizeof(int32_t)); }

extern "C" int rp_route_workspace_bytes(int64_t n, int world, size_t *bytes) {
    RP_REQUIRE(bytes && n >= 0 && n < INT32_MAX && world >= 1, "route_workspace_bytes: bad argument");
    const size_t nn = (size_t)(n > 0 ? n : 1);
    *bytes = 2 * route_align(nn * sizeof(int32_t)) + route_align((size_t)(world + 1) * sizeof(int64_t)) + route_scan_bytes((int64_t)nn) + 256;
    return RP_OK;
}

// sorted_keys / sorted_pos: rp_sort_pairs_i32 of rp_shard_keys' output (n pairs).
//   slot_sorted  [n] int32 : unique-request slot of the j-th sorted request (ascending)
//   slot_of_pair [n] int64 : slot of request p (p = f*B + b): the row of the received block request p reads
//   uniq_rows    [n] int64 : the first counts[world] entries are the local rows to ask for, grouped by owner in
//                            ascending owner order (= all_to_all send order)
//   counts [world+1] int64 : unique requests per owner, then their total
extern "C" int rp_route_build(void *workspace, size_t workspace_bytes, const int32_t *sorted_keys, const int32_t *sorted_pos,
                              int64_t n, int world, int lbits, int32_t *slot_sorted, int64_t *slot_of_pair,
                              int64_t *uniq_rows, int64_t *counts, rp_stream_t stream) {
    RP_REQUIRE(workspace && sorted_keys && sorted_pos && slot_sorted && slot_of_pair && uniq_rows && counts,
               "route_build: null pointer");
    RP_REQUIRE(n >= 1 && n < INT32_MAX && world >= 1 && lbits >= 1 && lbits <= 30, "route_build: bad n/world/lbits");
    size_t need = 0;
    int rc = rp_route_workspace_bytes(n, world, &need);
    if (rc != RP_OK) return rc;
    RP_REQUIRE(workspace_bytes >= need, "route_build: workspace %zu < %zu bytes", workspace_bytes, need);
    hipStream_t s = (hipStream_t)stream;
    char *base = reinterpret_cast<char *>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
    int32_t *flags = reinterpret_cast<int32_t *>(base);
    int32_t *incl = reinterpret_cast<int32_t *>(base + route_align((size_t)n * sizeof(int32_t)));
    int64_t *starts = reinterpret_cast<int64_t *>(base + 2 * route_align((size_t)n * sizeof(int32_t)));
    int32_t *bsum = reinterpret_cast<int32_t *>(reinterpret_cast<char *>(starts) + route_align((size_t)(world + 1) * sizeof(int64_t)));
    const int64_t nf = n > world + 1 ? n : world + 1;  // the flag launch also clears starts[0..world]
    hipLaunchKernelGGL(route_flags_kernel, dim3((unsigned)rp_cdiv(nf, 256)), dim3(256), 0, s, sorted_keys, n, flags, starts, world);
    RP_LAUNCH_CHECK("route flags");
    const int64_t nb = rp_cdiv(n, RS_BLOCK);
    hipLaunchKernelGGL(route_scan_sums_kernel, dim3((unsigned)nb), dim3(256), 0, s, flags, n, bsum);
    hipLaunchKernelGGL(route_scan_blocks_kernel, dim3(1), dim3(256), 0, s, bsum, nb);
    hipLaunchKernelGGL(route_scan_final_kernel, dim3((unsigned)nb), dim3(256), 0, s, flags, n, bsum, incl);
    RP_LAUNCH_CHECK("route scan");
    hipLaunchKernelGGL(route_scatter_kernel, dim3((unsigned)rp_cdiv(n, 256)), dim3(256), 0, s, sorted_keys, sorted_pos, incl, n,
                       lbits, slot_sorted, slot_of_pair, uniq_rows, starts);
    RP_LAUNCH_CHECK("route scatter");
    hipLaunchKernelGGL(route_counts_kernel, dim3(1), dim3(64), 0, s, incl, n, world, starts, counts);
    RP_LAUNCH_CHECK("route counts");
    rp_count_launch();
    return RP_OK;
}


// Fixed-capacity form of one exchange (no host-side split sizes -> no host sync per step): every owner gets exactly
// `capacity` request slots.  Rewrites the compact unique-request slots of rp_route_build as
//     slot' = owner * capacity + (slot - first slot of that owner)
// in slot_sorted (in place) and slot_of_pair, and writes the local rows to ask for into rows_padded[world * capacity]
// (the caller zero-fills it: unused slots ask for local row 0, whose gradient contribution is then exactly zero).
// An owner with more than `capacity` unique requests sets bit 1 of *err_flag (the requests beyond the capacity are
// dropped: the caller must treat the step as failed and enlarge the capacity).
__global__ __launch_bounds__(256) void route_pad_kernel(const int32_t *__restrict__ sk, const int32_t *__restrict__ sp,
                                                        int64_t n, int world, int lbits, int64_t capacity,
                                                        const int64_t *__restrict__ counts,
                                                        int32_t *__restrict__ slot_sorted,
                                                        int64_t *__restrict__ slot_of_pair,
                                                        int64_t *__restrict__ rows_padded,
                                                        int32_t *__restrict__ err_flag) {
    __shared__ int64_t start[RP_MAX_FIELDS + 1];
    if (threadIdx.x == 0) {
        int64_t acc = 0;
        for (int o = 0; o < world; ++o) {
            start[o] = acc;
            if (counts[o] > capacity) *err_flag = (*err_flag) | 2;
            acc += counts[o];
        }
    }
    __syncthreads();
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const int32_t key = sk[j];
    const int owner = key >> lbits;
    int64_t within = (int64_t)slot_sorted[j] - start[owner];
    const bool fits = within < capacity;
    if (!fits) within = capacity - 1;  // dropped (flagged above): stay inside the buffer
    const int64_t slot = (int64_t)owner * capacity + within;
    slot_sorted[j] = (int32_t)slot;
    slot_of_pair[sp[j]] = slot;
    if (fits && (j == 0 || sk[j - 1] != key)) rows_padded[slot] = (int64_t)(key & ((1 << lbits) - 1));
}

extern "C" int rp_route_pad(const int32_t *sorted_keys, const int32_t *sorted_pos, int64_t n, int world, int lbits,
                            int64_t capacity, const int64_t *counts, int32_t *slot_sorted, int64_t *slot_of_pair,
                            int64_t *rows_padded, int32_t *err_flag, rp_stream_t stream) {
    RP_REQUIRE(sorted_keys && sorted_pos && counts && slot_sorted && slot_of_pair && rows_padded && err_flag,
               "route_pad: null pointer");
    RP_REQUIRE(n >= 1 && n < INT32_MAX && world >= 1 && world <= RP_MAX_FIELDS && lbits >= 1 && lbits <= 30,
               "route_pad: bad n/world/lbits");
    RP_REQUIRE(capacity >= 1 && (int64_t)world * capacity < INT32_MAX, "route_pad: bad capacity");
    hipLaunchKernelGGL(route_pad_kernel, dim3((unsigned)rp_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, sorted_keys,
                       sorted_pos, n, world, lbits, capacity, counts, slot_sorted, slot_of_pair, rows_padded, err_flag);
    RP_LAUNCH_CHECK("route_pad");
    return RP_OK;
}

// FIELD-MAJOR copy of a route's sorted request list (round 6).  The list sorted by composite key is ordered (owner, field,
// row): with more than one owner a field's requests are no longer one contiguous run, which the first layer's backward
// launches (rp_embed_grad_seg / _ss / _smp: field f's B pairs at [f B, (f + 1) B) of the list, equal keys adjacent) count on.
// Every (owner, field) segment is contiguous in both orders, so the re-ordering is a move of at most world * F segments:
//     dest(j) = f B + (requests of field f with a smaller owner) + (j - start of j's segment)
// — the order inside a field is (owner, row) = ascending slot, equal slots stay adjacent, positions inside a run keep their
// ascending order (the list came from a stable sort).  Two launches: one workgroup finds the segment starts by binary search
// over (owner, field) of the sorted entries and turns them into per-segment offsets, then one thread per entry moves it.
__global__ __launch_bounds__(1024) void route_fm_starts_kernel(const int32_t *__restrict__ sk, const int32_t *__restrict__ sp,
                                                               int64_t n, int64_t B, int F, int world, int lbits,
                                                               int64_t *__restrict__ delta) {
    __shared__ int64_t start[RP_MAX_FIELDS * RP_MAX_FIELDS / 4 + 1];  // world * F + 1 <= 1025 entries
    const int S = world * F;
    for (int s = threadIdx.x; s <= S; s += blockDim.x) {
        // first j with (owner(j), field(j)) >= (s / F, s % F); s == S: n
        int64_t lo = 0, hi = n;
        if (s < S) {
            const int o = s / F, f = s % F;
            while (lo < hi) {
                const int64_t mid = (lo + hi) >> 1;
                const int om = sk[mid] >> lbits;
                const int fm = (int)(sp[mid] / B);
                if (om < o || (om == o && fm < f)) lo = mid + 1;
                else hi = mid;
            }
        } else {
            lo = n;
        }
        start[s] = lo;
    }
    __syncthreads();
    for (int s = threadIdx.x; s < S; s += blockDim.x) {
        const int o = s / F, f = s % F;
        int64_t before = 0;
        for (int q = 0; q < o; ++q) before += start[q * F + f + 1] - start[q * F + f];
        delta[s] = (int64_t)f * B + before - start[s];
    }
}

__global__ __launch_bounds__(256) void route_fm_move_kernel(const int32_t *__restrict__ sk, const int32_t *__restrict__ sp,
                                                            const int32_t *__restrict__ slot_sorted, int64_t n, int64_t B, int F,
                                                            int lbits, const int64_t *__restrict__ delta,
                                                            int32_t *__restrict__ slot_fm, int32_t *__restrict__ pos_fm) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const int32_t p = sp[j];
    const int o = sk[j] >> lbits, f = (int)(p / B);
    const int64_t d = j + delta[o * F + f];
    slot_fm[d] = slot_sorted[j];
    pos_fm[d] = p;
}

extern "C" int rp_route_field_major(const int32_t *sorted_keys, const int32_t *sorted_pos, const int32_t *slot_sorted, int64_t n,
                                    int64_t B, int world, int lbits, int32_t *slot_fm, int32_t *pos_fm, int64_t *delta,
                                    rp_stream_t stream) {
    RP_REQUIRE(sorted_keys && sorted_pos && slot_sorted && slot_fm && pos_fm && delta, "route_field_major: null pointer");
    RP_REQUIRE(n >= 1 && n < INT32_MAX && B >= 1 && n % B == 0 && n / B <= RP_MAX_FIELDS, "route_field_major: needs n = F * B pairs, F <= %d", RP_MAX_FIELDS);
    RP_REQUIRE(world >= 1 && lbits >= 1 && lbits <= 30, "route_field_major: bad world/lbits");
    const int F = (int)(n / B);
    RP_REQUIRE((int64_t)world * F <= RP_MAX_FIELDS * RP_MAX_FIELDS / 4, "route_field_major: world * F = %d segments (at most %d)", world * F,
               RP_MAX_FIELDS * RP_MAX_FIELDS / 4);
    RP_REQUIRE(slot_fm != slot_sorted && pos_fm != sorted_pos, "route_field_major: the outputs must not alias the inputs");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(route_fm_starts_kernel, dim3(1), dim3(1024), 0, s, sorted_keys, sorted_pos, n, B, F, world, lbits, delta);
    RP_LAUNCH_CHECK("route_field_major (segment starts)");
    hipLaunchKernelGGL(route_fm_move_kernel, dim3((unsigned)rp_cdiv(n, 256)), dim3(256), 0, s, sorted_keys, sorted_pos, slot_sorted,
                       n, B, F, lbits, delta, slot_fm, pos_fm);
    RP_LAUNCH_CHECK("route_field_major (move)");
    rp_count_launch();
    return RP_OK;
}

// dst[0 .. n_words) = value (32-bit words) — the zero fill of a buffer inside a recorded step (an ATen fill kernel or a memset
// node there would keep the step from replaying as a launch plan): the padded request list of rp_route_pad, the
// requester's per-slot gradient rows
__global__ __launch_bounds__(256) void fill_words_kernel(uint32_t *__restrict__ dst, int64_t n_words, uint32_t value) {
    const int64_t n4 = n_words >> 2;
    const uint4 v4 = {value, value, value, value};
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) reinterpret_cast<uint4 *>(dst)[i] = v4;
    for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_words; i += (int64_t)gridDim.x * 256) dst[i] = value;
}

extern "C" int rp_fill_words(void *dst, int64_t n_words, uint32_t value, rp_stream_t stream) {
    RP_REQUIRE(n_words >= 0 && (n_words == 0 || dst != nullptr), "fill_words: null pointer");
    RP_REQUIRE((reinterpret_cast<uintptr_t>(dst) & 15u) == 0, "fill_words: the buffer must be 16-byte aligned");
    if (n_words == 0) return RP_OK;
    int64_t g = rp_cdiv(n_words, 256 * 4 * 4);
    g = g < 1 ? 1 : (g > 2048 ? 2048 : g);
    hipLaunchKernelGGL(fill_words_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<uint32_t *>(dst),
                       n_words, value);
    RP_LAUNCH_CHECK("fill_words");
    return RP_OK;
}
