// (arena row, position) pair sort for the gather backward: a stable LSD radix sort over the low
// `end_bit` bits of the key (26 bits cover the 33.8 M rows of the Criteo-shape arena), on rocPRIM's
// device radix sort called directly (onesweep above one million pairs, merge sort below).
// Stability keeps equal rows in ascending position (= ascending sample) order, which fixes the
// summation order downstream.
// rocPRIM's tuned configuration for 32-bit pairs on this part sorts 8 bits per pass: four digit passes over 26 key
// bits, each a ~26 us launch at 1.7 M pairs (latency-, not bandwidth-bound at this size): 154 us.  Wider digits mean
// fewer passes but measured slower with the configurations rocPRIM accepts (9 bits, match-based ranking: 162 us;
// 11 bits: 421 us; 9+ bits with the default ranking exceed the LDS): the tuned default stays.
#include "common.h"
#include <cstring>
#include <rocprim/rocprim.hpp>

__global__ void iota_i32_kernel(int32_t *out, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (int32_t)i;
}

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

static hipError_t sort_dispatch(void *temp, size_t &tb, const int32_t *ki, int32_t *ko, const int32_t *vi, int32_t *vo,
                                int64_t n, int end_bit, hipStream_t s) {
    return rocprim::radix_sort_pairs<rocprim::default_config>(temp, tb, ki, ko, vi, vo, (size_t)n, 0u, (unsigned)end_bit, s);
}

// rocPRIM's size query walks its config dispatch and asks the runtime for the device on every call (~0.1 ms of host time,
// three of them per sort before this cache: more than the host spends on the rest of a DeepFM step's forward)
static int sort_bytes(int64_t n, size_t *bytes) {
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

extern "C" int rp_sort_workspace_bytes(int64_t n, size_t *bytes) {
    RP_REQUIRE(bytes && n >= 0 && n < INT32_MAX, "sort_workspace_bytes: bad argument");
    size_t tb = 0;
    int rc = sort_bytes(n > 0 ? n : 1, &tb);
    if (rc != RP_OK) return rc;
    *bytes = align256((size_t)(n > 0 ? n : 1) * sizeof(int32_t)) + align256(tb) + 256;
    return RP_OK;
}

extern "C" int rp_sort_pairs_i32(void *workspace, size_t workspace_bytes, const int32_t *keys_in, int32_t *keys_out,
                                 int32_t *pos_out, int64_t n, int end_bit, rp_stream_t stream) {
    RP_REQUIRE(workspace && keys_in && keys_out && pos_out, "sort_pairs: null pointer");
    RP_REQUIRE(n >= 0 && n < INT32_MAX && end_bit >= 1 && end_bit <= 32, "sort_pairs: bad n/end_bit");
    if (n == 0) return RP_OK;
    size_t need = 0, tb = 0;
    int rc = rp_sort_workspace_bytes(n, &need);
    if (rc != RP_OK) return rc;
    RP_REQUIRE(workspace_bytes >= need, "sort_pairs: workspace %zu < %zu bytes", workspace_bytes, need);
    rc = sort_bytes(n, &tb);
    if (rc != RP_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    char *base = reinterpret_cast<char *>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
    int32_t *iota = reinterpret_cast<int32_t *>(base);
    void *temp = base + align256((size_t)n * sizeof(int32_t));
    hipLaunchKernelGGL(iota_i32_kernel, dim3((unsigned)rp_cdiv(n, 256)), dim3(256), 0, s, iota, n);
    RP_LAUNCH_CHECK("sort iota");
    hipError_t e = sort_dispatch(temp, tb, keys_in, keys_out, iota, pos_out, n, end_bit, s);
    if (e != hipSuccess) return rp_fail(RP_ERR_LAUNCH, "sort_pairs: %s", hipGetErrorString(e));
    rp_count_launch();
    return RP_OK;
}
