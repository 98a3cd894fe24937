// Volume-weighted KL cell loss of the training step in one launch each way (reference learning/runModel.py:171-209):
//
//     cell_k = sum_c kl_div(log_softmax(logits_k)_c, gt_kc)           kl_div(x, t) = t * (log t - x), 0 where t == 0
//     w_k    = vol_k | log(1 + vol_k) | sqrt(vol_k)                     (regularization.cell_norm)
//     loss   = sum_k cell_k * w_k / sum_k w_k
//     OA    += #{ k : [gt_k0 > gt_k1] == argmax_c logits_kc }           (the reference's overall-accuracy counter, :178-180)
//
// The reference runs this as ~20 elementwise / reduction launches forward and as many backward on a [2048, 2] batch -- pure
// launch overhead.  Row terms are evaluated in fp32 like the reference's, the three sums accumulate in fp64 in a fixed order
// (deterministic), so the loss agrees with the reference's fp32 reduction to its last few ulps.
#include "common.h"

namespace {

constexpr int LOSS_THREADS = 1024;
constexpr int LOSS_MAX_BLOCKS = 256;

__device__ __forceinline__ float weight_of(float vol, int norm) { return norm == 1 ? logf(1.f + vol) : (norm == 2 ? sqrtf(vol) : vol); }

__device__ __forceinline__ void block_sum3(double (&v)[3], double* red /*[3][16]*/) {
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        double x = v[q];
        for (int o = 32; o > 0; o >>= 1) x += __shfl_down(x, o, 64);
        if ((threadIdx.x & 63) == 0) red[q * 16 + (threadIdx.x >> 6)] = x;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            double x = 0;
            for (int w = 0; w < (int)(blockDim.x >> 6); ++w) x += red[q * 16 + w];
            v[q] = x;
        }
    }
}

// partials[b] = (sum cell*w, sum w, OA count) of block b's rows; a single block also finalises
__global__ void __launch_bounds__(LOSS_THREADS) k_kl_loss_fwd(const float* __restrict__ logits, int64_t ldl, const float* __restrict__ gt, int64_t ldg,
                                                              const float* __restrict__ vol, int64_t ldv, int norm, int64_t n,
                                                              double* __restrict__ partials, double* __restrict__ sums, float* __restrict__ loss) {
    __shared__ double red[3 * 16];
    double v[3] = {0, 0, 0};
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x) {
        const float l0 = logits[k * ldl], l1 = logits[k * ldl + 1];
        const float t0 = gt[k * ldg], t1 = gt[k * ldg + 1];
        const float m = fmaxf(l0, l1);
        const float lg = logf(expf(l0 - m) + expf(l1 - m));   // log_softmax = (x - max) - log(sum exp(x - max)), torch's order
        const float c0 = (t0 > 0.f ? t0 * logf(t0) : 0.f) - t0 * ((l0 - m) - lg);
        const float c1 = (t1 > 0.f ? t1 * logf(t1) : 0.f) - t1 * ((l1 - m) - lg);
        const float w = weight_of(vol[k * ldv], norm);
        v[0] += (double)((c0 + c1) * w);
        v[1] += (double)w;
        v[2] += ((t0 > t1 ? 1 : 0) == (l1 > l0 ? 1 : 0)) ? 1.0 : 0.0;
    }
    block_sum3(v, red);
    if (threadIdx.x == 0) {
        if (gridDim.x == 1) {
            sums[0] = v[0], sums[1] = v[1], sums[2] = v[2];
            *loss = (float)v[0] / (float)v[1];   // the reference divides two fp32 sums
        } else {
            partials[blockIdx.x * 3 + 0] = v[0], partials[blockIdx.x * 3 + 1] = v[1], partials[blockIdx.x * 3 + 2] = v[2];
        }
    }
}

__global__ void k_kl_loss_finalize(const double* __restrict__ partials, int nb, double* __restrict__ sums, float* __restrict__ loss) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double v[3] = {0, 0, 0};
        for (int b = 0; b < nb; ++b)
            for (int q = 0; q < 3; ++q) v[q] += partials[b * 3 + q];
        sums[0] = v[0], sums[1] = v[1], sums[2] = v[2];
        *loss = (float)v[0] / (float)v[1];
    }
}

// dlogits_kc = g * w_k / W * (softmax_kc * (t_k0 + t_k1) - t_kc)
__global__ void k_kl_loss_bwd(const float* __restrict__ logits, int64_t ldl, const float* __restrict__ gt, int64_t ldg, const float* __restrict__ vol,
                              int64_t ldv, int norm, int64_t n, const double* __restrict__ sums, const float* __restrict__ g,
                              float* __restrict__ dlogits, int64_t ldd) {
    const float scale = *g / (float)sums[1];
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x) {
        const float l0 = logits[k * ldl], l1 = logits[k * ldl + 1];
        const float t0 = gt[k * ldg], t1 = gt[k * ldg + 1];
        const float m = fmaxf(l0, l1);
        const float e0 = expf(l0 - m), e1 = expf(l1 - m);
        const float inv = 1.f / (e0 + e1);
        const float s = scale * weight_of(vol[k * ldv], norm);
        dlogits[k * ldd] = s * (e0 * inv * (t0 + t1) - t0);
        dlogits[k * ldd + 1] = s * (e1 * inv * (t0 + t1) - t1);
    }
}

// Forward, finalisation, the running metric sums and the backward of a batch in ONE launch of one workgroup (round 6: the training step's five
// launches around the loss -- forward partials, finalise, the metric accumulator's add, backward, plus their boundaries -- were 25 us of GPU time and
// as much host time for 2048 rows).  The workgroup plays the nb = ceil(n / 1024) blocks of k_kl_loss_fwd one after the other (same rows per thread,
// same reduction tree per block, block sums added in block order as k_kl_loss_finalize does): sums, loss and dlogits are bit-identical to the
// three-launch path.  running (optional): fp64 [3] += sums -- the Trainer's packed metric accumulator (learning/runModel.py Metrics.addPacked).
__global__ void __launch_bounds__(LOSS_THREADS) k_kl_loss_step(const float* __restrict__ logits, int64_t ldl, const float* __restrict__ gt, int64_t ldg,
                                                               const float* __restrict__ vol, int64_t ldv, int norm, int64_t n, int nb,
                                                               const float* __restrict__ g, double* __restrict__ sums, float* __restrict__ loss,
                                                               double* __restrict__ running, float* __restrict__ dlogits, int64_t ldd) {
    __shared__ double red[3 * 16];
    __shared__ float scale_s;
    double tot[3] = {0, 0, 0};
    for (int b = 0; b < nb; ++b) {
        double v[3] = {0, 0, 0};
        for (int64_t k = (int64_t)b * LOSS_THREADS + threadIdx.x; k < n; k += (int64_t)nb * LOSS_THREADS) {
            const float l0 = logits[k * ldl], l1 = logits[k * ldl + 1];
            const float t0 = gt[k * ldg], t1 = gt[k * ldg + 1];
            const float m = fmaxf(l0, l1);
            const float lg = logf(expf(l0 - m) + expf(l1 - m));
            const float c0 = (t0 > 0.f ? t0 * logf(t0) : 0.f) - t0 * ((l0 - m) - lg);
            const float c1 = (t1 > 0.f ? t1 * logf(t1) : 0.f) - t1 * ((l1 - m) - lg);
            const float w = weight_of(vol[k * ldv], norm);
            v[0] += (double)((c0 + c1) * w);
            v[1] += (double)w;
            v[2] += ((t0 > t1 ? 1 : 0) == (l1 > l0 ? 1 : 0)) ? 1.0 : 0.0;
        }
        block_sum3(v, red);
        if (threadIdx.x == 0) {
            if (nb == 1) {
                tot[0] = v[0], tot[1] = v[1], tot[2] = v[2];
            } else {
#pragma unroll
                for (int q = 0; q < 3; ++q) tot[q] += v[q];
            }
        }
        __syncthreads();      // `red` is rewritten by the next block's sums
    }
    if (threadIdx.x == 0) {
        sums[0] = tot[0], sums[1] = tot[1], sums[2] = tot[2];
        *loss = (float)tot[0] / (float)tot[1];
        if (running) {
#pragma unroll
            for (int q = 0; q < 3; ++q) running[q] += tot[q];
        }
        scale_s = (g ? *g : 1.f) / (float)tot[1];
    }
    __syncthreads();
    if (!dlogits) return;
    const float scale = scale_s;
    for (int64_t k = threadIdx.x; k < n; k += LOSS_THREADS) {
        const float l0 = logits[k * ldl], l1 = logits[k * ldl + 1];
        const float t0 = gt[k * ldg], t1 = gt[k * ldg + 1];
        const float m = fmaxf(l0, l1);
        const float e0 = expf(l0 - m), e1 = expf(l1 - m);
        const float inv = 1.f / (e0 + e1);
        const float s = scale * weight_of(vol[k * ldv], norm);
        dlogits[k * ldd] = s * (e0 * inv * (t0 + t1) - t0);
        dlogits[k * ldd + 1] = s * (e1 * inv * (t0 + t1) - t1);
    }
}

}  // namespace

// dgnn_kl_cell_loss_fwd + (running += sums) + dgnn_kl_cell_loss_bwd as ONE launch (learning/runModel.py:171-209 and its gradient): sums fp64 [3], loss,
// dlogits [n][>= 2] (NULL: forward only), grad_loss (NULL: 1), running fp64 [3] accumulator (NULL: none).  Same bits as the separate entry points.
// DGNN_E_UNSUPPORTED (nothing launched) beyond DGNN_KL_STEP_MAX_ROWS rows: the caller then takes the separate entry points.
extern "C" int dgnn_kl_cell_loss_step(const float* logits, int64_t ldl, const float* gt, int64_t ldg, const float* vol, int64_t ldv, int norm, int64_t n,
                                      const float* grad_loss, double* sums, float* loss, double* running, float* dlogits, int64_t ldd, void* stream_) {
    DGNN_REQUIRE(n > 0 && logits && gt && vol && sums && loss && ldl >= 2 && ldg >= 2 && norm >= 0 && norm <= 2 && (!dlogits || ldd >= 2), DGNN_E_INVALID,
                 "kl_cell_loss_step: bad args (two-class logits and targets, at least one row)");
    if (n > 65536) return DGNN_E_UNSUPPORTED;
    const int nb = (int)dgnn_cdiv(n, LOSS_THREADS);      // <= 64 <= LOSS_MAX_BLOCKS: the block count dgnn_kl_cell_loss_fwd would launch
    hipLaunchKernelGGL(k_kl_loss_step, dim3(1), dim3(LOSS_THREADS), 0, (hipStream_t)stream_, logits, ldl, gt, ldg, vol, ldv, norm, n, nb, grad_loss, sums, loss,
                       running, dlogits, ldd);
    return dgnn_check_launch("kl_cell_loss_step");
}

extern "C" int64_t dgnn_kl_cell_loss_scratch_doubles(int64_t n) {
    const int64_t nb = dgnn_cdiv(n > 0 ? n : 1, LOSS_THREADS);
    return 3 * (nb < LOSS_MAX_BLOCKS ? nb : LOSS_MAX_BLOCKS) + 3;
}

extern "C" int dgnn_kl_cell_loss_fwd(const float* logits, int64_t ldl, const float* gt, int64_t ldg, const float* vol, int64_t ldv, int norm, int64_t n,
                                     double* sums, float* loss, double* scratch, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DGNN_REQUIRE(n > 0 && logits && gt && vol && sums && loss && scratch && ldl >= 2 && ldg >= 2 && norm >= 0 && norm <= 2, DGNN_E_INVALID,
                 "kl_cell_loss_fwd: bad args (two-class logits and targets, at least one row)");
    int64_t nb = dgnn_cdiv(n, LOSS_THREADS);
    if (nb > LOSS_MAX_BLOCKS) nb = LOSS_MAX_BLOCKS;
    hipLaunchKernelGGL(k_kl_loss_fwd, dim3((unsigned)nb), dim3(LOSS_THREADS), 0, stream, logits, ldl, gt, ldg, vol, ldv, norm, n, scratch, sums, loss);
    if (nb > 1) hipLaunchKernelGGL(k_kl_loss_finalize, dim3(1), dim3(64), 0, stream, scratch, (int)nb, sums, loss);
    return dgnn_check_launch("kl_cell_loss_fwd");
}

extern "C" int dgnn_kl_cell_loss_bwd(const float* logits, int64_t ldl, const float* gt, int64_t ldg, const float* vol, int64_t ldv, int norm, int64_t n,
                                     const double* sums, const float* grad_loss, float* dlogits, int64_t ldd, void* stream_) {
    DGNN_REQUIRE(n > 0 && logits && gt && vol && sums && grad_loss && dlogits && ldd >= 2, DGNN_E_INVALID, "kl_cell_loss_bwd: bad args");
    hipLaunchKernelGGL(k_kl_loss_bwd, dim3(dgnn_grid_cap(dgnn_cdiv(n, 256))), dim3(256), 0, (hipStream_t)stream_, logits, ldl, gt, ldg, vol, ldv, norm, n,
                       sums, grad_loss, dlogits, ldd);
    return dgnn_check_launch("kl_cell_loss_bwd");
}
