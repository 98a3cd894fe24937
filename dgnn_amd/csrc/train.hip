// Training-mode conv layer as ONE call each way (SurfaceNet.forward :214-219 + autograd at runModel.py:279).
//
// The training step runs on 4-hop blocks of ~10^5 cells: every kernel is short (10-100 us) and the step was bound by the ~250
// launches issued one by one from Python, not by a kernel.  These two entry points issue a layer's whole launch chain from
// C++ -- the same kernels as the separate entry points, in the same order, so results are bit-identical to calling them one by
// one (tests/test_gpu_train.py) -- into caller-provided buffers; nothing allocates or synchronises.
//
//   forward : a = mean_j x_j * (We.A + be)          dgnn_sage_aggregate_fwd   (rowptr == NULL: a = x, a plain Linear + BN block)
//             z = a.Wj^T + x_dst.Wi^T + bj           dgnn_linear_fwd / _x3
//             mean, var (+ running statistics)       dgnn_bn_batch_stats
//             scale, shift                           dgnn_bn_fold
//             y = relu(z * scale + shift)            dgnn_scale_shift_act
//   backward: dz, dgamma, dbeta                      dgnn_bn_relu_bwd
//             dWj = dz^T a, dWi = dz^T x_dst, dbj    dgnn_linear_wgrad / _x3, dgnn_colsum
//             da = dz.Wj                              dgnn_linear_fwd on Wj^T (transposed here)
//             dx_src, dWe, dbe                        dgnn_sage_aggregate_bwd   (dx_src == NULL: first layer, x is data)
//             dx_src[:n_dst] += dz.Wi                 dgnn_linear_fwd with DGNN_LINEAR_ACCUMULATE
#include "common.h"

namespace {

__global__ void k_transpose(const float* __restrict__ in, int rows, int cols, float* __restrict__ out) {
    const int n = rows * cols;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int r = i / cols, c = i - r * cols;
        out[c * rows + r] = in[i];
    }
}

inline int64_t align4(int64_t v) { return (v + 3) & ~(int64_t)3; }

}  // namespace

#define TRY(call)                 \
    do {                          \
        const int rc_ = (call);   \
        if (rc_ != DGNN_OK) return rc_; \
    } while (0)

extern "C" int64_t dgnn_sage_layer_train_scratch_elems(int64_t n_src, int64_t n_dst, int c_in, int c_out, int f_e) {
    if (n_src < 0 || n_dst < 0 || c_in <= 0 || c_out <= 0) return 16;
    const int64_t stats = dgnn_colstats_scratch_elems(n_dst, c_out > c_in ? c_out : c_in);
    const int64_t wg = dgnn_linear_wgrad_scratch_elems(n_dst, c_out, c_in);
    const int64_t ab = dgnn_sage_aggregate_bwd_scratch_elems(n_src, c_in, f_e > 0 ? f_e : 1);
    // backward: dz [n_dst,c_out] | da [n_dst,c_in] | WjT, WiT [c_in,c_out] each | max(stats, wgrad, agg-bwd partials)
    int64_t big = stats > wg ? stats : wg;
    if (ab > big) big = ab;
    return align4(n_dst * c_out) + align4(n_dst * c_in) + 2 * align4((int64_t)c_in * c_out) + align4(big) + 64;
}

extern "C" int dgnn_sage_layer_train_fwd(const int32_t* rowptr, const int32_t* src, const int32_t* eid, int64_t n_dst, const float* x,
                                         int64_t ldx, int c_in, const float* edge_attr, int64_t lde, int f_e, const float* We,
                                         const float* be, const float* Wj, const float* bj, const float* Wi, int c_out, const float* gamma,
                                         const float* beta, float* running_mean, float* running_var, float momentum, float eps, int relu,
                                         float* a, float* z, float* mean, float* var, float* scale, float* shift, float* y, float* scratch,
                                         int gemm_mode, void* stream) {
    DGNN_REQUIRE(n_dst > 0 && c_in > 0 && c_out > 0, DGNN_E_INVALID, "sage_layer_train_fwd: bad sizes (BatchNorm needs at least one row)");
    DGNN_REQUIRE(x && Wj && z && mean && var && scale && shift && y && scratch, DGNN_E_INVALID, "sage_layer_train_fwd: null pointer");
    const float* A1 = x;
    int64_t lda1 = ldx;
    if (rowptr) {
        DGNN_REQUIRE(a && src, DGNN_E_INVALID, "sage_layer_train_fwd: the aggregate needs src and a");
        TRY(dgnn_sage_aggregate_fwd(rowptr, src, eid, n_dst, x, ldx, c_in, edge_attr, lde, f_e, We, be, nullptr, 0, nullptr, 0, a, c_in, stream));
        A1 = a;
        lda1 = c_in;
    }
    const float* A2 = (rowptr && Wi) ? x : nullptr;   // x_dst = x[:n_dst] (reference :217)
    if (gemm_mode == DGNN_GEMM_F32)
        TRY(dgnn_linear_fwd(A1, lda1, c_in, Wj, c_in, A2, ldx, A2 ? c_in : 0, A2 ? Wi : nullptr, c_in, bj, nullptr, nullptr, 0, n_dst, c_out, z, c_out,
                            stream));
    else
        TRY(dgnn_linear_fwd_x3(A1, lda1, c_in, Wj, c_in, A2, ldx, A2 ? c_in : 0, A2 ? Wi : nullptr, c_in, bj, nullptr, nullptr, 0, n_dst, c_out, z,
                               c_out, stream));
    TRY(dgnn_bn_batch_stats(z, c_out, n_dst, c_out, mean, var, running_mean, running_var, momentum, scratch, stream));
    TRY(dgnn_bn_fold(gamma, beta, mean, var, eps, c_out, scale, shift, stream));
    TRY(dgnn_scale_shift_act(z, c_out, scale, shift, relu, n_dst, c_out, y, c_out, stream));
    return DGNN_OK;
}

extern "C" int dgnn_sage_layer_train_bwd(const int32_t* t_rowptr, const int32_t* t_dst, const int32_t* t_eid, const int32_t* rowptr_dst,
                                         int64_t n_src, int64_t n_dst, const float* x, int64_t ldx, int c_in, const float* edge_attr,
                                         int64_t lde, int f_e, const float* We, const float* be, const float* Wj, const float* Wi, int c_out,
                                         const float* gamma, const float* mean, const float* var, float eps, int relu, const float* a,
                                         const float* z, const float* y, const float* dy, float* dx, float* dWe, float* dbe, float* dWj,
                                         float* dbj, float* dWi, float* dgamma, float* dbeta, float* scratch, int gemm_mode, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DGNN_REQUIRE(n_dst > 0 && n_src >= n_dst && c_in > 0 && c_out > 0, DGNN_E_INVALID, "sage_layer_train_bwd: bad sizes");
    DGNN_REQUIRE(x && Wj && z && y && dy && mean && var && dWj && dgamma && dbeta && scratch, DGNN_E_INVALID, "sage_layer_train_bwd: null pointer");
    const bool agg = t_rowptr != nullptr;
    float* dz = scratch;
    float* da = dz + align4(n_dst * c_out);
    float* WjT = da + align4(n_dst * c_in);
    float* WiT = WjT + align4((int64_t)c_in * c_out);
    float* tmp = WiT + align4((int64_t)c_in * c_out);
    const bool x3 = gemm_mode != DGNN_GEMM_F32;
    auto gemm = [&](const float* A, int64_t lda, int k, const float* W, int64_t ldw, int flags, int64_t M, int n, float* out, int64_t ldo) {
        return x3 ? dgnn_linear_fwd_x3(A, lda, k, W, ldw, nullptr, 0, 0, nullptr, 0, nullptr, nullptr, nullptr, flags, M, n, out, ldo, stream_)
                  : dgnn_linear_fwd(A, lda, k, W, ldw, nullptr, 0, 0, nullptr, 0, nullptr, nullptr, nullptr, flags, M, n, out, ldo, stream_);
    };
    auto wgrad = [&](const float* A, int64_t lda, int na, const float* B, int64_t ldb, int nb, float* dW) {
        return x3 ? dgnn_linear_wgrad_x3(A, lda, na, B, ldb, nb, n_dst, dW, nb, 0, tmp, stream_)
                  : dgnn_linear_wgrad(A, lda, na, B, ldb, nb, n_dst, dW, nb, 0, tmp, stream_);
    };
    // BatchNorm (batch statistics) + ReLU backward: dz, dgamma, dbeta
    TRY(dgnn_bn_relu_bwd(z, c_out, y, c_out, dy, c_out, gamma, mean, var, eps, 1, relu, n_dst, c_out, dz, c_out, dgamma, dbeta, tmp, stream_));
    const float* A1 = agg ? a : x;
    const int64_t lda1 = agg ? c_in : ldx;
    TRY(wgrad(dz, c_out, c_out, A1, lda1, c_in, dWj));
    if (dbj) TRY(dgnn_colsum(dz, c_out, n_dst, c_out, dbj, 0, tmp, stream_));
    if (agg && Wi && dWi) TRY(wgrad(dz, c_out, c_out, x, ldx, c_in, dWi));
    const bool need_dx = dx != nullptr;
    const bool need_da = agg ? (need_dx || We != nullptr) : need_dx;
    if (need_da) {
        hipLaunchKernelGGL(k_transpose, dim3(dgnn_grid_cap(dgnn_cdiv((int64_t)c_in * c_out, 256))), dim3(256), 0, stream, Wj, c_out, c_in, WjT);
        // plain Linear block: the gradient of the input is da itself
        TRY(gemm(dz, c_out, c_out, WjT, c_out, 0, n_dst, c_in, agg ? da : dx, c_in));
    }
    if (agg) {
        if (We) {
            DGNN_REQUIRE(dWe && dbe, DGNN_E_INVALID, "sage_layer_train_bwd: dWe / dbe missing");
            if (dbe == dWe + (size_t)c_in * f_e) {   // one buffer (the Python binding's layout): one fill
                (void)hipMemsetAsync(dWe, 0, sizeof(float) * (size_t)c_in * (f_e + 1), stream);
            } else {
                (void)hipMemsetAsync(dWe, 0, sizeof(float) * (size_t)c_in * f_e, stream);
                (void)hipMemsetAsync(dbe, 0, sizeof(float) * (size_t)c_in, stream);
            }
        }
        if (need_da)
            TRY(dgnn_sage_aggregate_bwd(t_rowptr, t_dst, t_eid, n_src, rowptr_dst, x, ldx, c_in, edge_attr, lde, f_e, We, be, nullptr, 0, da, c_in, dx,
                                        c_in, dWe, dbe, nullptr, 0, tmp, stream_));
        if (need_dx && Wi) {
            hipLaunchKernelGGL(k_transpose, dim3(dgnn_grid_cap(dgnn_cdiv((int64_t)c_in * c_out, 256))), dim3(256), 0, stream, Wi, c_out, c_in, WiT);
            TRY(gemm(dz, c_out, c_out, WiT, c_out, DGNN_LINEAR_ACCUMULATE, n_dst, c_in, dx, c_in));
        }
    }
    return dgnn_check_launch("sage_layer_train_bwd");
}
