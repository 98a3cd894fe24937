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

// =====================================================================================================================
// Updated variant (surfaceNetUpdatedEdgeFilters.py:147-170 and its autograd): one conv layer per call each way, fp32 or bf16
// storage (activations / phi bf16, parameters and their gradients fp32).
//   forward : phi = ea.We^T + be  [E, c_in]   ->   a = mean_j x_j * phi   ->   y = relu?(a.Wl^T + x[:n_dst].Wr^T + bl)
//   backward: dz = dy * [y > 0]; dWl, dbl, dWr; da = dz.Wl; (dx, dphi) = aggregate backward; dx[:n_dst] += dz.Wr;
//             dphi += dphi_ext (the next layer's use of phi as its edge input); dWe = dphi^T ea, dbe, d_ea = dphi.We
// The same kernels in the same order as the separate entry points (bit-identical results).
// =====================================================================================================================
namespace {

template <typename T>
__global__ void k_add_inplace(T* __restrict__ a, const T* __restrict__ b, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        dgnn_st(a + i, dgnn_ld(a + i) + dgnn_ld(b + i));
}

// the two storage types behind one set of names
struct F32 {
    typedef float T;
    static int linear(const T* A1, int64_t lda1, int k1, const float* W1, int64_t ldw1, const T* A2, int64_t lda2, int k2, const float* W2, int64_t ldw2,
                      const float* bias, int flags, int64_t M, int n, T* out, int64_t ldo, int mode, void* st) {
        return mode == DGNN_GEMM_F32 ? dgnn_linear_fwd(A1, lda1, k1, W1, ldw1, A2, lda2, k2, W2, ldw2, bias, nullptr, nullptr, flags, M, n, out, ldo, st)
                                     : dgnn_linear_fwd_x3(A1, lda1, k1, W1, ldw1, A2, lda2, k2, W2, ldw2, bias, nullptr, nullptr, flags, M, n, out, ldo, st);
    }
    static int wgrad(const T* A, int64_t lda, int na, const T* B, int64_t ldb, int nb, int64_t M, float* dW, float* tmp, int mode, void* st) {
        return mode == DGNN_GEMM_F32 ? dgnn_linear_wgrad(A, lda, na, B, ldb, nb, M, dW, nb, 0, tmp, st)
                                     : dgnn_linear_wgrad_x3(A, lda, na, B, ldb, nb, M, dW, nb, 0, tmp, st);
    }
    static int colsum(const T* x, int64_t ld, int64_t M, int c, float* out, float* tmp, void* st) { return dgnn_colsum(x, ld, M, c, out, 0, tmp, st); }
    static int relu_bwd(const T* y, const T* g, int64_t n, T* out, void* st) { return dgnn_relu_bwd(y, g, n, out, st); }
    static int agg_fwd(const int32_t* rp, const int32_t* src, const int32_t* eid, int64_t n_dst, const T* x, int64_t ldx, int c, const T* phi, T* a, void* st) {
        return dgnn_sage_aggregate_fwd(rp, src, eid, n_dst, x, ldx, c, nullptr, 0, 0, nullptr, nullptr, phi, c, nullptr, 0, a, c, st);
    }
    static int agg_bwd(const int32_t* trp, const int32_t* td, const int32_t* te, int64_t n_src, const int32_t* rpd, const T* x, int64_t ldx, int c, const T* phi,
                       const T* da, T* dx, T* dphi, void* st) {
        return dgnn_sage_aggregate_bwd(trp, td, te, n_src, rpd, x, ldx, c, nullptr, 0, 0, nullptr, nullptr, phi, c, da, c, dx, c, nullptr, nullptr, dphi, c, nullptr, st);
    }
};
struct BF16 {
    typedef uint16_t T;
    static int linear(const T* A1, int64_t lda1, int k1, const float* W1, int64_t ldw1, const T* A2, int64_t lda2, int k2, const float* W2, int64_t ldw2,
                      const float* bias, int flags, int64_t M, int n, T* out, int64_t ldo, int, void* st) {
        return dgnn_linear_fwd_bf16(A1, lda1, k1, W1, ldw1, A2, lda2, k2, W2, ldw2, bias, nullptr, nullptr, flags, M, n, out, ldo, 0, st);
    }
    static int wgrad(const T* A, int64_t lda, int na, const T* B, int64_t ldb, int nb, int64_t M, float* dW, float* tmp, int, void* st) {
        return dgnn_linear_wgrad_bf16(A, 0, lda, na, B, 0, ldb, nb, M, dW, nb, 0, tmp, st);
    }
    static int colsum(const T* x, int64_t ld, int64_t M, int c, float* out, float* tmp, void* st) { return dgnn_colsum_bf16(x, ld, M, c, out, 0, tmp, st); }
    static int relu_bwd(const T* y, const T* g, int64_t n, T* out, void* st) { return dgnn_relu_bwd_bf16(y, g, n, out, st); }
    static int agg_fwd(const int32_t* rp, const int32_t* src, const int32_t* eid, int64_t n_dst, const T* x, int64_t ldx, int c, const T* phi, T* a, void* st) {
        return dgnn_sage_aggregate_fwd_bf16(rp, src, eid, n_dst, x, ldx, c, nullptr, 0, 0, nullptr, nullptr, phi, c, nullptr, 0, a, c, st);
    }
    static int agg_bwd(const int32_t* trp, const int32_t* td, const int32_t* te, int64_t n_src, const int32_t* rpd, const T* x, int64_t ldx, int c, const T* phi,
                       const T* da, T* dx, T* dphi, void* st) {
        return dgnn_sage_aggregate_bwd_bf16(trp, td, te, n_src, rpd, x, ldx, c, nullptr, 0, 0, nullptr, nullptr, phi, c, da, c, dx, c, nullptr, nullptr, dphi, c, nullptr,
                                            st);
    }
};

inline void transpose_to(const float* W, int rows, int cols, float* out, hipStream_t stream) {
    hipLaunchKernelGGL(k_transpose, dim3(dgnn_grid_cap(dgnn_cdiv((int64_t)rows * cols, 256))), dim3(256), 0, stream, W, rows, cols, out);
}

template <typename K>
int updated_fwd(const int32_t* rowptr, const int32_t* src, const int32_t* eid, int64_t n_dst, const void* x_, int64_t ldx, int c_in, const void* ea_,
                int64_t lde, int k_e, int64_t E, const float* We, const float* be, const float* Wl, const float* bl, const float* Wr, int c_out, int relu,
                void* phi_, void* a_, void* y_, int mode, void* st) {
    typedef typename K::T T;
    const T *x = (const T*)x_, *ea = (const T*)ea_;
    T *phi = (T*)phi_, *a = (T*)a_, *y = (T*)y_;
    if (E > 0) TRY(K::linear(ea, lde, k_e, We, k_e, nullptr, 0, 0, nullptr, 0, be, 0, E, c_in, phi, c_in, mode, st));                     // :156
    TRY(K::agg_fwd(rowptr, src, eid, n_dst, x, ldx, c_in, phi, a, st));                                                                   // :158
    TRY(K::linear(a, c_in, c_in, Wl, c_in, Wr ? x : nullptr, ldx, Wr ? c_in : 0, Wr, c_in, bl, relu ? 1 : 0, n_dst, c_out, y, c_out, mode, st));   // :159-165
    return DGNN_OK;
}

template <typename K>
int updated_bwd(const int32_t* t_rowptr, const int32_t* t_dst, const int32_t* t_eid, const int32_t* rowptr_dst, int64_t n_src, int64_t n_dst, int64_t E,
                const void* x_, int64_t ldx, int c_in, const void* ea_, int64_t lde, int k_e, const float* We, const float* Wl, const float* Wr, int c_out,
                int relu, const void* phi_, const void* a_, const void* y_, const void* dy_, const void* dphi_ext_, void* dx_, void* d_ea_, float* dWe,
                float* dbe, float* dWl, float* dbl, float* dWr, void* dz_, void* da_, void* dphi_, float* scratch, int mode, void* st) {
    typedef typename K::T T;
    hipStream_t stream = (hipStream_t)st;
    const T *x = (const T*)x_, *ea = (const T*)ea_, *phi = (const T*)phi_, *a = (const T*)a_, *y = (const T*)y_, *dy = (const T*)dy_,
            *dphi_ext = (const T*)dphi_ext_;
    T *dx = (T*)dx_, *d_ea = (T*)d_ea_, *dz = (T*)dz_, *da = (T*)da_, *dphi = (T*)dphi_;
    float* WlT = scratch;
    float* WrT = WlT + align4((int64_t)c_in * c_out);
    float* WeT = WrT + align4((int64_t)c_in * c_out);
    float* tmp = WeT + align4((int64_t)c_in * k_e);
    const T* g = dy;
    if (relu) {
        TRY(K::relu_bwd(y, dy, n_dst * c_out, dz, st));
        g = dz;
    }
    TRY(K::wgrad(g, c_out, c_out, a, c_in, c_in, n_dst, dWl, tmp, mode, st));
    if (dbl) TRY(K::colsum(g, c_out, n_dst, c_out, dbl, tmp, st));
    if (Wr && dWr) TRY(K::wgrad(g, c_out, c_out, x, ldx, c_in, n_dst, dWr, tmp, mode, st));
    transpose_to(Wl, c_out, c_in, WlT, stream);
    TRY(K::linear(g, c_out, c_out, WlT, c_out, nullptr, 0, 0, nullptr, 0, nullptr, 0, n_dst, c_in, da, c_in, mode, st));
    TRY(K::agg_bwd(t_rowptr, t_dst, t_eid, n_src, rowptr_dst, x, ldx, c_in, phi, da, dx, dphi, st));
    if (dx && Wr) {
        transpose_to(Wr, c_out, c_in, WrT, stream);
        TRY(K::linear(g, c_out, c_out, WrT, c_out, nullptr, 0, 0, nullptr, 0, nullptr, DGNN_LINEAR_ACCUMULATE, n_dst, c_in, dx, c_in, mode, st));
    }
    if (E > 0) {
        if (dphi_ext)
            hipLaunchKernelGGL((k_add_inplace<T>), dim3(dgnn_grid_cap(dgnn_cdiv(E * c_in, 256))), dim3(256), 0, stream, dphi, dphi_ext, E * c_in);
        TRY(K::wgrad(dphi, c_in, c_in, ea, lde, k_e, E, dWe, tmp, mode, st));
        TRY(K::colsum(dphi, c_in, E, c_in, dbe, tmp, st));
        if (d_ea) {
            transpose_to(We, c_in, k_e, WeT, stream);
            TRY(K::linear(dphi, c_in, c_in, WeT, c_in, nullptr, 0, 0, nullptr, 0, nullptr, 0, E, k_e, d_ea, k_e, mode, st));
        }
    } else {
        (void)hipMemsetAsync(dWe, 0, sizeof(float) * (size_t)c_in * k_e, stream);
        (void)hipMemsetAsync(dbe, 0, sizeof(float) * (size_t)c_in, stream);
    }
    return dgnn_check_launch("sage_updated_train_bwd");
}

}  // namespace

extern "C" int64_t dgnn_sage_updated_train_scratch_elems(int64_t n_dst, int64_t E, int c_in, int c_out, int k_e) {
    if (n_dst < 0 || E < 0 || c_in <= 0 || c_out <= 0 || k_e <= 0) return 16;
    int64_t big = dgnn_colstats_scratch_elems(n_dst > E ? n_dst : E, c_in > c_out ? c_in : c_out);
    const int64_t w1 = dgnn_linear_wgrad_scratch_elems(n_dst, c_out, c_in), w2 = dgnn_linear_wgrad_scratch_elems(E, c_in, k_e);
    if (w1 > big) big = w1;
    if (w2 > big) big = w2;
    return 2 * align4((int64_t)c_in * c_out) + align4((int64_t)c_in * k_e) + align4(big) + 64;
}

extern "C" int dgnn_sage_updated_train_fwd(const int32_t* rowptr, const int32_t* src, const int32_t* eid, int64_t n_dst, const void* x, int64_t ldx,
                                           int c_in, const void* ea, int64_t lde, int k_e, int64_t E, const float* We, const float* be,
                                           const float* Wl, const float* bl, const float* Wr, int c_out, int relu, void* phi, void* a, void* y,
                                           int bf16, int gemm_mode, void* stream) {
    DGNN_REQUIRE(n_dst > 0 && E >= 0 && c_in > 0 && c_out > 0 && k_e > 0, DGNN_E_INVALID, "sage_updated_train_fwd: bad sizes");
    DGNN_REQUIRE(rowptr && src && x && We && be && Wl && phi && a && y && (E == 0 || ea), DGNN_E_INVALID, "sage_updated_train_fwd: null pointer");
    return bf16 ? updated_fwd<BF16>(rowptr, src, eid, n_dst, x, ldx, c_in, ea, lde, k_e, E, We, be, Wl, bl, Wr, c_out, relu, phi, a, y, gemm_mode, stream)
                : updated_fwd<F32>(rowptr, src, eid, n_dst, x, ldx, c_in, ea, lde, k_e, E, We, be, Wl, bl, Wr, c_out, relu, phi, a, y, gemm_mode, stream);
}

extern "C" int dgnn_sage_updated_train_bwd(const int32_t* t_rowptr, const int32_t* t_dst, const int32_t* t_eid, const int32_t* rowptr_dst,
                                           int64_t n_src, int64_t n_dst, int64_t E, const void* x, int64_t ldx, int c_in, const void* ea, int64_t lde,
                                           int k_e, const float* We, const float* Wl, const float* Wr, int c_out, int relu, const void* phi,
                                           const void* a, const void* y, const void* dy, const void* dphi_ext, void* dx, void* d_ea, float* dWe,
                                           float* dbe, float* dWl, float* dbl, float* dWr, void* dz, void* da, void* dphi, float* scratch, int bf16,
                                           int gemm_mode, void* stream) {
    DGNN_REQUIRE(n_dst > 0 && n_src >= n_dst && E >= 0 && c_in > 0 && c_out > 0 && k_e > 0, DGNN_E_INVALID, "sage_updated_train_bwd: bad sizes");
    DGNN_REQUIRE(t_rowptr && t_dst && t_eid && rowptr_dst && x && We && Wl && phi && a && dy && dWe && dbe && dWl && da && dphi && scratch &&
                     (!relu || (y && dz)) && (E == 0 || ea),
                 DGNN_E_INVALID, "sage_updated_train_bwd: null pointer");
    return bf16 ? updated_bwd<BF16>(t_rowptr, t_dst, t_eid, rowptr_dst, n_src, n_dst, E, x, ldx, c_in, ea, lde, k_e, We, Wl, Wr, c_out, relu, phi, a, y, dy,
                                    dphi_ext, dx, d_ea, dWe, dbe, dWl, dbl, dWr, dz, da, dphi, scratch, gemm_mode, stream)
                : updated_bwd<F32>(t_rowptr, t_dst, t_eid, rowptr_dst, n_src, n_dst, E, x, ldx, c_in, ea, lde, k_e, We, Wl, Wr, c_out, relu, phi, a, y, dy,
                                   dphi_ext, dx, d_ea, dWe, dbe, dWl, dbl, dWr, dz, da, dphi, scratch, gemm_mode, stream);
}
