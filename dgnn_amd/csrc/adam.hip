// Adam step over all parameters of a model in ONE launch (reference learning/runModel.py:290 `torch.optim.Adam(model.parameters(), lr)`, stepped at
// :282).  torch's own fused Adam is one or two launches as well, but its Python side (parameter grouping, state initialisation checks, step-tensor
// updates) costs ~0.1 ms per step on the thread that issues a 0.9 ms training step.  Here the host passes the tensors' addresses as a kernel
// argument (no device-side table to maintain) and the step count as a number.
//
//   m = m + (g - m) * (1 - beta1)            v = beta2 * v + (1 - beta2) * g * g
//   p = p - (lr / (1 - beta1^t)) * m / (sqrt(v) / sqrt(1 - beta2^t) + eps)
// (torch.optim.Adam without amsgrad / weight decay / maximize; bias corrections computed on the host in double precision.)
#include "common.h"

namespace {

constexpr int ADAM_MAX = 48;       // tensors per launch: the argument block stays below 2 KB
constexpr int ADAM_CHUNK = 2048;   // elements per block

struct AdamArgs {
    float* p[ADAM_MAX];
    const float* g[ADAM_MAX];
    float* m[ADAM_MAX];
    float* v[ADAM_MAX];
    int n[ADAM_MAX];
    int block_end[ADAM_MAX];       // blocks of tensors 0 .. t
    int nt;
};

__global__ void __launch_bounds__(256) k_adam(AdamArgs a, float lr_over_bc1, float inv_bc2_sqrt, float beta1, float beta2, float eps) {
    int t = 0;
    while ((int)blockIdx.x >= a.block_end[t]) ++t;
    const int b0 = t ? a.block_end[t - 1] : 0;
    const int beg = ((int)blockIdx.x - b0) * ADAM_CHUNK, end = min(a.n[t], beg + ADAM_CHUNK);
    float* __restrict__ p = a.p[t];
    const float* __restrict__ g = a.g[t];
    float* __restrict__ m = a.m[t];
    float* __restrict__ v = a.v[t];
    const float w1 = 1.f - beta1, w2 = 1.f - beta2;
    for (int i = beg + threadIdx.x; i < end; i += 256) {
        const float gi = g[i];
        const float mi = m[i] + (gi - m[i]) * w1;
        const float vi = beta2 * v[i] + w2 * gi * gi;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) * inv_bc2_sqrt + eps;
        p[i] = p[i] - lr_over_bc1 * (mi / denom);
    }
}

}  // namespace

extern "C" int dgnn_adam_step(int n_tensors, float* const* p, const float* const* g, float* const* m, float* const* v, const int64_t* numel, float lr,
                              float beta1, float beta2, float eps, int64_t step, void* stream) {
    DGNN_REQUIRE(n_tensors >= 0 && step >= 1 && (n_tensors == 0 || (p && g && m && v && numel)), DGNN_E_INVALID, "adam_step: bad arguments");
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    const float lr_over_bc1 = (float)((double)lr / bc1), inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
    for (int t0 = 0; t0 < n_tensors; t0 += ADAM_MAX) {
        AdamArgs a;
        a.nt = n_tensors - t0 < ADAM_MAX ? n_tensors - t0 : ADAM_MAX;
        int blocks = 0;
        for (int t = 0; t < a.nt; ++t) {
            DGNN_REQUIRE(p[t0 + t] && g[t0 + t] && m[t0 + t] && v[t0 + t] && numel[t0 + t] >= 0 && numel[t0 + t] < ((int64_t)1 << 31), DGNN_E_INVALID,
                         "adam_step: tensor %d", t0 + t);
            a.p[t] = p[t0 + t], a.g[t] = g[t0 + t], a.m[t] = m[t0 + t], a.v[t] = v[t0 + t], a.n[t] = (int)numel[t0 + t];
            blocks += (int)dgnn_cdiv(numel[t0 + t], ADAM_CHUNK);
            a.block_end[t] = blocks;
        }
        for (int t = a.nt; t < ADAM_MAX; ++t) a.block_end[t] = blocks;
        if (blocks > 0)
            hipLaunchKernelGGL(k_adam, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a, lr_over_bc1, inv_bc2_sqrt, beta1, beta2, eps);
    }
    return dgnn_check_launch("adam_step");
}
