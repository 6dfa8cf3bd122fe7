// logit sum + sigmoid + binary cross entropy (mean), forward and backward, gfx950.
// One pass over [B]: HBM-trivial (a few bytes per sample); the point is to keep the
// reference's five tiny ATen launches (add, sigmoid, squeeze, BCELoss, mean) as one launch plus a
// one-block finish, with a fixed summation order (no float atomics -> bit-stable loss).
#include "common.h"

#define LOSS_BLOCK 256
#define LOSS_MAX_BLOCKS 1024

struct ZPtrs {
    const float *p[4];
};

__device__ __forceinline__ float block_sum_256(float v, float *sh) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    float r = 0.f;
    if (threadIdx.x == 0) r = (sh[0] + sh[1]) + (sh[2] + sh[3]);
    __syncthreads();
    return r;  // valid on thread 0
}

__global__ __launch_bounds__(LOSS_BLOCK) void sigmoid_bce_fwd_kernel(ZPtrs z, int n_add, int apply_sigmoid,
                                                                      const float *__restrict__ label, int64_t B,
                                                                      float p_eps, float *__restrict__ pred,
                                                                      float *__restrict__ partial) {
    __shared__ float sh[4];
    float acc = 0.f;
    for (int64_t b = (int64_t)blockIdx.x * LOSS_BLOCK + threadIdx.x; b < B; b += (int64_t)gridDim.x * LOSS_BLOCK) {
        float s = z.p[0][b];
        for (int a = 1; a < n_add; ++a) s += z.p[a][b];
        const float p = apply_sigmoid ? 1.f / (1.f + expf(-s)) : s;
        if (pred != nullptr) pred[b] = p;
        if (label != nullptr) {
            const float pe = p + p_eps, y = label[b];
            const float lp = fmaxf(logf(pe), -100.f), l1p = fmaxf(log1pf(-pe), -100.f);
            acc -= y * lp + (1.f - y) * l1p;
        }
    }
    const float tot = block_sum_256(acc, sh);
    if (threadIdx.x == 0 && partial != nullptr) partial[blockIdx.x] = tot;
}

// accumulate: loss[0] += (a multi-task loss: sum of the tasks' weighted means, multi_task/mmoe.py:127 — each task's launch adds
// its term, in task order: the same fp32 additions as the reference's python sum)
__global__ __launch_bounds__(LOSS_BLOCK) void loss_finish_kernel(const float *__restrict__ partial, int n, float scale,
                                                                 float *__restrict__ loss, int accumulate) {
    __shared__ float sh[4];
    float acc = 0.f;
    for (int i = threadIdx.x; i < n; i += LOSS_BLOCK) acc += partial[i];
    const float tot = block_sum_256(acc, sh);
    if (threadIdx.x == 0) {
        const float term = __fmul_rn(tot, scale);  // (no contraction into an fma: the python sum rounds the product first)
        loss[0] = accumulate ? __fadd_rn(loss[0], term) : term;
    }
}

__global__ __launch_bounds__(LOSS_BLOCK) void sigmoid_bce_bwd_kernel(const float *__restrict__ pred,
                                                                      const float *__restrict__ label,
                                                                      const float *__restrict__ gloss, int64_t B,
                                                                      float p_eps, float scale, int apply_sigmoid,
                                                                      float *__restrict__ dz) {
    const float g = gloss[0] * scale;
    for (int64_t b = (int64_t)blockIdx.x * LOSS_BLOCK + threadIdx.x; b < B; b += (int64_t)gridDim.x * LOSS_BLOCK) {
        const float p = pred[b], y = label[b], pe = p + p_eps;
        // ATen binary_cross_entropy_backward: (x - y) / max((1 - x) * x, 1e-12) * grad
        float d = (pe - y) / fmaxf((1.f - pe) * pe, 1e-12f) * g;
        if (apply_sigmoid) d *= p * (1.f - p);
        dz[b] = d;
    }
}

static int loss_blocks(int64_t B) {
    int64_t nb = rp_cdiv(B > 0 ? B : 1, LOSS_BLOCK);
    return (int)(nb > LOSS_MAX_BLOCKS ? LOSS_MAX_BLOCKS : nb);
}

extern "C" int rp_loss_partials(int64_t B) { return loss_blocks(B); }

static int sigmoid_bce_fwd_launch(const float *const *z_ptrs, int n_addends, int apply_sigmoid, const float *label,
                                  int64_t B, float p_eps, float weight, float *pred, float *partial, float *loss,
                                  int accumulate, rp_stream_t stream) {
    RP_REQUIRE(z_ptrs && n_addends >= 1 && n_addends <= 4, "sigmoid_bce_fwd: 1..4 logit addends");
    RP_REQUIRE(B >= 1, "sigmoid_bce_fwd: empty batch");
    RP_REQUIRE(loss == nullptr || (label && partial), "sigmoid_bce_fwd: loss needs label and partial");
    ZPtrs z{};
    for (int a = 0; a < n_addends; ++a) {
        RP_REQUIRE(z_ptrs[a], "sigmoid_bce_fwd: null addend %d", a);
        z.p[a] = z_ptrs[a];
    }
    const int nb = loss_blocks(B);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(sigmoid_bce_fwd_kernel, dim3(nb), dim3(LOSS_BLOCK), 0, s, z, n_addends, apply_sigmoid,
                       loss ? label : nullptr, B, p_eps, pred, loss ? partial : nullptr);
    RP_LAUNCH_CHECK("sigmoid_bce_fwd");
    if (loss != nullptr) {
        hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(LOSS_BLOCK), 0, s, partial, nb, weight / (float)B, loss, accumulate);
        RP_LAUNCH_CHECK("loss_finish");
    }
    return RP_OK;
}

extern "C" int rp_sigmoid_bce_fwd(const float *const *z_ptrs, int n_addends, int apply_sigmoid, const float *label,
                                  int64_t B, float p_eps, float weight, float *pred, float *partial, float *loss,
                                  rp_stream_t stream) {
    return sigmoid_bce_fwd_launch(z_ptrs, n_addends, apply_sigmoid, label, B, p_eps, weight, pred, partial, loss, 0, stream);
}

// the same with loss[0] += instead of = (the second .. last task of a multi-task loss)
extern "C" int rp_sigmoid_bce_fwd_accum(const float *const *z_ptrs, int n_addends, int apply_sigmoid, const float *label,
                                        int64_t B, float p_eps, float weight, float *pred, float *partial, float *loss,
                                        rp_stream_t stream) {
    RP_REQUIRE(loss != nullptr, "sigmoid_bce_fwd_accum: null loss");
    return sigmoid_bce_fwd_launch(z_ptrs, n_addends, apply_sigmoid, label, B, p_eps, weight, pred, partial, loss, 1, stream);
}

// loss[0] = scale * sum(partial[0..n)) in a fixed order (the second stage of rp_sigmoid_bce_fwd as an entry of its own: the
// partials of rp_mlp_tail_fwd_bce end here; nobody on the device needs the scalar, so a recorded step issues this launch on
// its side section)
extern "C" int rp_loss_finish(const float *partial, int n, float scale, float *loss, rp_stream_t stream) {
    RP_REQUIRE(partial && loss && n >= 1, "loss_finish: bad argument");
    hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(LOSS_BLOCK), 0, (hipStream_t)stream, partial, n, scale, loss, 0);
    RP_LAUNCH_CHECK("loss_finish");
    return RP_OK;
}

extern "C" int rp_sigmoid_bce_bwd(const float *pred, const float *label, const float *gloss, int64_t B, float p_eps,
                                  float weight, int apply_sigmoid, float *dz, rp_stream_t stream) {
    RP_REQUIRE(pred && label && gloss && dz && B >= 1, "sigmoid_bce_bwd: bad argument");
    hipLaunchKernelGGL(sigmoid_bce_bwd_kernel, dim3(loss_blocks(B)), dim3(LOSS_BLOCK), 0, (hipStream_t)stream, pred,
                       label, gloss, B, p_eps, weight / (float)B, apply_sigmoid, dz);
    RP_LAUNCH_CHECK("sigmoid_bce_bwd");
    return RP_OK;
}
