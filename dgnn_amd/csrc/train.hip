// Training-mode conv layer as ONE call each way (SurfaceNet.forward :214-219 + autograd at runModel.py:279).
//
// The training step runs on 4-hop blocks of ~10^5 cells: every kernel is short (10-100 us) and the step was bound by the ~250
// launches issued one by one from Python, not by a kernel.  These two entry points issue a layer's whole launch chain from
// C++ -- the same kernels as the separate entry points, in the same order, so results are bit-identical to calling them one by
// one (tests/test_gpu_train.py) -- into caller-provided buffers; nothing allocates or synchronises.
//
//   forward : a = mean_j x_j * (We.A + be)          dgnn_sage_aggregate_fwd   (rowptr == NULL: a = x, a plain Linear + BN block)
//             z = a.Wj^T + x_dst.Wi^T + bj           dgnn_linear_fwd / _x3
//             mean, var (+ running statistics),      dgnn_bn_batch_stats_fold (= dgnn_bn_batch_stats + dgnn_bn_fold, the fold computed
//             scale, shift                             by the finalising kernel on the values it has just stored)
//             y = relu(z * scale + shift)            dgnn_scale_shift_act
//   backward: dz, dgamma, dbeta                      dgnn_bn_relu_bwd
//             dWj = dz^T a, dWi = dz^T x_dst, dbj    dgnn_linear_wgrad / _x3, dgnn_colsum
//             da = dz.Wj                              dgnn_linear_fwd on Wj^T (transposed here)
//             dx_src, dWe, dbe                        dgnn_sage_aggregate_bwd   (dx_src == NULL: first layer, x is data)
//             dx_src[:n_dst] += dz.Wi                 dgnn_linear_fwd with DGNN_LINEAR_ACCUMULATE
#include <cstdlib>
#include <mutex>

#include "common.h"
#include "reduce_common.h"

namespace {

__global__ void k_transpose(const float* __restrict__ in, int rows, int cols, float* __restrict__ out) {
    const int n = rows * cols;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int r = i / cols, c = i - r * cols;
        out[c * rows + r] = in[i];
    }
}
// two matrices of one shape in one launch (lin_j and lin_i of a layer)
__global__ void k_transpose2(const float* __restrict__ in0, const float* __restrict__ in1, int rows, int cols, float* __restrict__ out0,
                             float* __restrict__ out1) {
    const int n = rows * cols;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < 2 * n; i += gridDim.x * blockDim.x) {
        const int j = i < n ? i : i - n;
        const int r = j / cols, c = j - r * cols;
        (i < n ? out0 : out1)[c * rows + r] = (i < n ? in0 : in1)[j];
    }
}

inline int64_t align4(int64_t v) { return (v + 3) & ~(int64_t)3; }

// every layer's lin_j / lin_i transposed by ONE launch at the start of a backward pass (was one or two launches per layer)
struct TrJobs {
    const float* in[16];
    float* out[16];
    int rows[16], cols[16], end[16];   // end[j] = elements of jobs 0 .. j
    int n;
};
__global__ void k_transpose_many(TrJobs jobs) {
    const int total = jobs.end[jobs.n - 1];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        int j = 0;
        while (i >= jobs.end[j]) ++j;
        const int e = i - (j ? jobs.end[j - 1] : 0);
        const int r = e / jobs.cols[j], c = e - r * jobs.cols[j];
        jobs.out[j][c * jobs.rows[j] + r] = jobs.in[j][e];
    }
}

// both reductions of a conv layer's backward in one launch of 1024-thread blocks: blocks [0, ns) sum the aggregate backward's slabs
// (dWe, dbe), every later block runs four 256-thread blocks of the weight-gradient reduction (dWj, dWi, dbj)
__global__ void __launch_bounds__(1024) k_reduce_layer(SlabReduceDesc sd, WgradReduceDesc wd, int ns) {
    __shared__ float red[RS_SLICES][17];
    __shared__ float redf[4][16][17];
    __shared__ double redd[4][16][17];
    if ((int)blockIdx.x < ns) {
        reduce_slabs_block(sd, blockIdx.x, red);
        return;
    }
    const int sub = threadIdx.x >> 8, t = threadIdx.x & 255;
    const int blk = ((int)blockIdx.x - ns) * 4 + sub;
    const bool on = blk < wgrad_reduce_blocks(wd);
    if (on) wgrad_reduce_cat_phase(wd, blk, t, redf[sub], redd[sub], 0);
    __syncthreads();
    if (on) wgrad_reduce_cat_phase(wd, blk, t, redf[sub], redd[sub], 1);
}

// DGNN_TRAIN_FUSED=0: the launch chain of the separate entry points (one launch per weight gradient, bias sum, transpose and
// input-gradient GEMM).  Default: dWj / dWi / dbj from one launch pair (dgnn_linear_wgrad_x3_cat), da and dz.Wi from one GEMM against
// the stacked [Wj^T ; Wi^T], the latter added where the aggregate backward stores dx (dgnn_sage_aggregate_bwd_add), all transposes of a
// backward pass in one launch.  Same arithmetic per element; only dbj is summed in another (fp64) order.
int g_fused_on = -1;   // bit 0: the backward chain, bit 1: batch statistics from the forward GEMM's epilogue
int fused_mask() {
    int v = __atomic_load_n(&g_fused_on, __ATOMIC_ACQUIRE);
    if (v < 0) {
        const char* e = getenv("DGNN_TRAIN_FUSED");
        v = e ? atoi(e) & 3 : 3;
        if (getenv("DGNN_AGG_CHUNKED") && getenv("DGNN_AGG_CHUNKED")[0] == '0') v &= ~1;   // the addend form lives in the chunked kernel
        __atomic_store_n(&g_fused_on, v, __ATOMIC_RELEASE);
    }
    return v;
}
bool fused_enabled() { return (fused_mask() & 1) != 0; }
bool fused_stats_enabled() { return (fused_mask() & 2) != 0; }

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
    // backward: dz [n_dst,c_out] | da [n_dst,c_in] | WjT, WiT [c_in,c_out] each | max(stats, wgrad, agg-bwd partials), twice (the weight
    // gradients run on a second stream with their own partials)
    int64_t big = stats > wg ? stats : wg;
    if (ab > big) big = ab;
    const int64_t wc = dgnn_linear_wgrad_cat_scratch_elems(n_dst, c_out, c_in, c_in);
    return align4(n_dst * c_out) + 2 * align4(n_dst * c_in) + 2 * align4((int64_t)c_in * c_out) + align4(big) + align4(big > wc ? big : wc) + 64;
}

// library-internal (csrc/norm.hip): dgnn_bn_relu_bwd with the ReLU mask from x and the forward's (scale, shift)
int dgnn_bn_relu_bwd_zmask(const float* x, int64_t ldx, const float* y, int64_t ldy, const float* dy, int64_t lddy, const float* gamma, const float* mean,
                           const float* var, float eps, int train, int relu, int64_t M, int c, float* dx, int64_t lddx, float* dgamma, float* dbeta,
                           float* scratch, const float* zscale, const float* zshift, void* stream);
// library-internal (csrc/norm.hip): dgnn_bn_stats_finalize_fold that also counts the batch in *nbt
int dgnn_bn_stats_finalize_fold_nbt(const double* colstats, int64_t nblk, int64_t M, int c, float* mean, float* var, float* running_mean, float* running_var,
                                    float momentum, const float* gamma, const float* beta, float eps, float* scale, float* shift, int64_t* nbt, void* stream);

// nbt / counted: the BatchNorm's num_batches_tracked and whether this call has counted the batch in it (the finalising launch of the statistics does
// when they come out of the GEMM's epilogue -- round 6; otherwise the caller still has to)
static int layer_train_fwd(const int32_t* rowptr, const int32_t* src, const int32_t* eid, int64_t n_dst, const float* x, int64_t ldx, int c_in,
                           const float* edge_attr, int64_t lde, int f_e, const float* We, const float* be, const float* Wj, const float* bj, const float* Wi,
                           int c_out, const float* gamma, const float* beta, float* running_mean, float* running_var, float momentum, float eps, int relu,
                           float* a, float* z, float* mean, float* var, float* scale, float* shift, float* y, float* scratch, int gemm_mode, void* stream,
                           int64_t* nbt, bool* counted);

extern "C" int dgnn_sage_layer_train_fwd(const int32_t* rowptr, const int32_t* src, const int32_t* eid, int64_t n_dst, const float* x,
                                         int64_t ldx, int c_in, const float* edge_attr, int64_t lde, int f_e, const float* We,
                                         const float* be, const float* Wj, const float* bj, const float* Wi, int c_out, const float* gamma,
                                         const float* beta, float* running_mean, float* running_var, float momentum, float eps, int relu,
                                         float* a, float* z, float* mean, float* var, float* scale, float* shift, float* y, float* scratch,
                                         int gemm_mode, void* stream) {
    return layer_train_fwd(rowptr, src, eid, n_dst, x, ldx, c_in, edge_attr, lde, f_e, We, be, Wj, bj, Wi, c_out, gamma, beta, running_mean, running_var, momentum,
                           eps, relu, a, z, mean, var, scale, shift, y, scratch, gemm_mode, stream, nullptr, nullptr);
}

static int layer_train_fwd(const int32_t* rowptr, const int32_t* src, const int32_t* eid, int64_t n_dst, const float* x, int64_t ldx, int c_in,
                           const float* edge_attr, int64_t lde, int f_e, const float* We, const float* be, const float* Wj, const float* bj, const float* Wi,
                           int c_out, const float* gamma, const float* beta, float* running_mean, float* running_var, float momentum, float eps, int relu,
                           float* a, float* z, float* mean, float* var, float* scale, float* shift, float* y, float* scratch, int gemm_mode, void* stream,
                           int64_t* nbt, bool* counted) {
    if (counted) *counted = false;
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
    if (gemm_mode != DGNN_GEMM_F32 && fused_stats_enabled()) {
        // the GEMM's epilogue leaves the column sums of z per block of 32 rows: no launch that reads z back for the batch statistics
        double* cs = reinterpret_cast<double*>(((uintptr_t)scratch + 7) & ~(uintptr_t)7);
        const int rc = dgnn_linear_fwd_x3_stats(A1, lda1, c_in, Wj, c_in, A2, ldx, A2 ? c_in : 0, A2 ? Wi : nullptr, c_in, bj, n_dst, c_out, z, c_out, cs, stream);
        if (rc == DGNN_OK) {
            TRY(dgnn_bn_stats_finalize_fold_nbt(cs, (n_dst + 31) / 32, n_dst, c_out, mean, var, running_mean, running_var, momentum, gamma, beta, eps, scale,
                                                shift, nbt, stream));
            if (counted) *counted = nbt != nullptr;
            TRY(dgnn_scale_shift_act(z, c_out, scale, shift, relu, n_dst, c_out, y, c_out, stream));
            return DGNN_OK;
        }
        if (rc != DGNN_E_UNSUPPORTED) return rc;
    }
    if (gemm_mode == DGNN_GEMM_F32)
        TRY(dgnn_linear_fwd(A1, lda1, c_in, Wj, c_in, A2, ldx, A2 ? c_in : 0, A2 ? Wi : nullptr, c_in, bj, nullptr, nullptr, 0, n_dst, c_out, z, c_out,
                            stream));
    else
        TRY(dgnn_linear_fwd_x3(A1, lda1, c_in, Wj, c_in, A2, ldx, A2 ? c_in : 0, A2 ? Wi : nullptr, c_in, bj, nullptr, nullptr, 0, n_dst, c_out, z,
                               c_out, stream));
    TRY(dgnn_bn_batch_stats_fold(z, c_out, n_dst, c_out, mean, var, running_mean, running_var, momentum, gamma, beta, eps, scale, shift, scratch, stream));
    TRY(dgnn_scale_shift_act(z, c_out, scale, shift, relu, n_dst, c_out, y, c_out, stream));
    return DGNN_OK;
}

namespace {

// A second stream for the weight gradients (per device, created on first use): dWj, dbj and dWi depend only on dz, nothing on the
// dx chain (da -> aggregate backward -> dx) depends on them, and at ~10 us per kernel plus the queue's per-kernel turnaround the
// chain is what the backward pass takes; on their own queue they run beside it.  (Measured: no gain, see aux_enabled.)
struct Aux {
    hipStream_t stream = nullptr;
    hipEvent_t ev[32] = {};
    int next = 0;
    bool ok = false;
};
Aux* aux_of_current_device() {
    static Aux table[DGNN_MAX_DEVICES];
    static std::mutex m;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= DGNN_MAX_DEVICES) return nullptr;
    std::lock_guard<std::mutex> lock(m);
    Aux& a = table[dev];
    if (!a.stream) {
        a.ok = hipStreamCreateWithFlags(&a.stream, hipStreamNonBlocking) == hipSuccess;
        for (int i = 0; a.ok && i < 32; ++i) a.ok = hipEventCreateWithFlags(&a.ev[i], hipEventDisableTiming) == hipSuccess;
        (void)hipGetLastError();
    }
    return a.ok ? &a : nullptr;
}
hipEvent_t next_event(Aux* a) {
    hipEvent_t e = a->ev[a->next];
    a->next = (a->next + 1) & 31;
    return e;
}
// OFF by default: interleaved A/B runs (tools/ab_train.py aux=1 aux=0) put the two-stream backward 2-8 % BEHIND the one-stream one --
// the event hand-offs cost what the overlap of ~10 us kernels buys.  Kept as an option (DGNN_TRAIN_AUX_STREAM=1,
// dgnn_train_set_aux_stream) because the balance shifts with block size.
int g_aux_on = -1;   // read / written with atomics: the first call may race between the caller's thread and the block-builder thread
bool aux_enabled() {
    int v = __atomic_load_n(&g_aux_on, __ATOMIC_ACQUIRE);
    if (v < 0) {
        v = (getenv("DGNN_TRAIN_AUX_STREAM") && getenv("DGNN_TRAIN_AUX_STREAM")[0] == '1') ? 1 : 0;
        __atomic_store_n(&g_aux_on, v, __ATOMIC_RELEASE);
    }
    return v != 0;
}
// partial-sum scratch of one layer's backward on the main stream (column reductions, aggregate backward slabs; also the weight
// gradients when there is no second stream)
int64_t layer_tmp_elems(int64_t n_src, int64_t n_dst, int c_in, int c_out, int f_e) {
    const int64_t stats = dgnn_colstats_scratch_elems(n_dst, c_out > c_in ? c_out : c_in);
    const int64_t wg = dgnn_linear_wgrad_scratch_elems(n_dst, c_out, c_in);
    const int64_t ab = dgnn_sage_aggregate_bwd_scratch_elems(n_src, c_in, f_e > 0 ? f_e : 1);
    int64_t big = stats > wg ? stats : wg;
    if (ab > big) big = ab;
    return align4(big);
}

// One layer's backward.  `aux` == nullptr: everything on `stream`, in the order of the separate entry points.  Otherwise the
// weight gradients go to aux->stream (scratch tmp_w, ordered after dz by an event) and *done receives the event that marks them
// finished -- the caller must make `stream` wait for it before dz / tmp_w are reused and before the gradients are consumed.
int layer_bwd(const int32_t* t_rowptr, const int32_t* t_dst, const int32_t* t_eid, const int32_t* rowptr_dst, int64_t n_src, int64_t n_dst, const float* x,
              int64_t ldx, int c_in, const float* edge_attr, int64_t lde, int f_e, const float* We, const float* be, const float* Wj, const float* Wi, int c_out,
              const float* gamma, const float* mean, const float* var, float eps, int relu, const float* a, const float* z, const float* y, const float* dy,
              float* dx, float* dWe, float* dbe, float* dWj, float* dbj, float* dWi, float* dgamma, float* dbeta, float* dz_buf, float* da, float* WjT, float* WiT,
              float* tmp, float* tmp_w, int gemm_mode, hipStream_t stream, Aux* aux, hipEvent_t* done, bool pre_t = false, bool has_bn = true,
              const float* bn_scale = nullptr) {
    void* stream_ = (void*)stream;
    const bool agg = t_rowptr != nullptr;
    const bool x3 = gemm_mode != DGNN_GEMM_F32;
    // fused chain (see fused_enabled): `da` holds [n_dst, 2 c_in] floats, WiT == WjT + c_in * c_out (the stacked transposes)
    const bool fused = x3 && !aux && fused_enabled();
    auto gemm = [&](const float* A, int64_t lda, int k, const float* W, int64_t ldw, int flags, int64_t M, int n, float* out, int64_t ldo) {
        return x3 ? dgnn_linear_fwd_x3(A, lda, k, W, ldw, nullptr, 0, 0, nullptr, 0, nullptr, nullptr, nullptr, flags, M, n, out, ldo, stream_)
                  : dgnn_linear_fwd(A, lda, k, W, ldw, nullptr, 0, 0, nullptr, 0, nullptr, nullptr, nullptr, flags, M, n, out, ldo, stream_);
    };
    void* wstream = aux ? (void*)aux->stream : stream_;
    float* wtmp = tmp_w;   // always the region that both callers size for the weight-gradient partials (`tmp` is sized for the main chain's only)
    auto wgrad = [&](const float* A, int64_t lda, int na, const float* B, int64_t ldb, int nb, float* dW) {
        return x3 ? dgnn_linear_wgrad_x3(A, lda, na, B, ldb, nb, n_dst, dW, nb, 0, wtmp, wstream)
                  : dgnn_linear_wgrad(A, lda, na, B, ldb, nb, n_dst, dW, nb, 0, wtmp, wstream);
    };
    // BatchNorm (batch statistics) + ReLU backward: dz, dgamma, dbeta
    // (has_bn == false: a plain Linear -- the decoder's output layer -- whose dz is dy itself)
    // (round 6: the whole-model call hands the forward's scale / shift on -- they sit behind mean / var in a layer's stats block -- and the ReLU mask is
    // taken from z: y is not read by the two passes; DGNN_BN_ZMASK=0: the mask from y)
    if (has_bn) {
        static const bool zmask_on = !(getenv("DGNN_BN_ZMASK") && getenv("DGNN_BN_ZMASK")[0] == '0');
        if (zmask_on && bn_scale && relu)
            TRY(dgnn_bn_relu_bwd_zmask(z, c_out, y, c_out, dy, c_out, gamma, mean, var, eps, 1, relu, n_dst, c_out, dz_buf, c_out, dgamma, dbeta, tmp, bn_scale,
                                       bn_scale + c_out, stream_));
        else
            TRY(dgnn_bn_relu_bwd(z, c_out, y, c_out, dy, c_out, gamma, mean, var, eps, 1, relu, n_dst, c_out, dz_buf, c_out, dgamma, dbeta, tmp, stream_));
    }
    const float* const dz = has_bn ? dz_buf : dy;
    if (aux) {
        hipEvent_t e = next_event(aux);
        (void)hipEventRecord(e, stream);
        (void)hipStreamWaitEvent(aux->stream, e, 0);
    }
    const float* A1 = agg ? a : x;
    const int64_t lda1 = agg ? c_in : ldx;
    const bool need_dx = dx != nullptr;
    const bool need_da = agg ? (need_dx || We != nullptr) : need_dx;
    const bool both = need_da && agg && need_dx && Wi;
    const bool one_gemm = fused && both && We && f_e == 20;   // [da | dz.Wi] from one GEMM, the layer's two reductions from one launch
    WgradReduceDesc wdesc;
    if (fused) {
        const float* B2 = (agg && Wi && dWi) ? x : nullptr;
        TRY(dgnn_linear_wgrad_x3_cat_deferred(dz, c_out, c_out, A1, lda1, c_in, B2, ldx, B2 ? c_in : 0, n_dst, dWj, dWi, dbj, wtmp, stream_,
                                              one_gemm ? &wdesc : nullptr));
    } else {
        TRY(wgrad(dz, c_out, c_out, A1, lda1, c_in, dWj));
        if (dbj) TRY(dgnn_colsum(dz, c_out, n_dst, c_out, dbj, 0, wtmp, wstream));
        if (agg && Wi && dWi) TRY(wgrad(dz, c_out, c_out, x, ldx, c_in, dWi));
    }
    if (aux) {
        *done = next_event(aux);
        (void)hipEventRecord(*done, aux->stream);
    }
    if (both && !pre_t)
        hipLaunchKernelGGL(k_transpose2, dim3(dgnn_grid_cap(dgnn_cdiv((int64_t)2 * c_in * c_out, 256))), dim3(256), 0, stream, Wj, Wi, c_out, c_in, WjT, WiT);
    if (one_gemm) {
        // [da | dz.Wi] = dz . [Wj^T ; Wi^T]^T in one GEMM (every output column is the separate GEMMs' own dot product), the second half added to
        // the aggregate's sums where dx is stored
        DGNN_REQUIRE(dWe && dbe, DGNN_E_INVALID, "sage_layer_train_bwd: dWe / dbe missing");
        TRY(gemm(dz, c_out, c_out, WjT, c_out, 0, n_dst, 2 * c_in, da, 2 * c_in));
        SlabReduceDesc sdesc;
        TRY(dgnn_sage_aggregate_bwd_add_deferred(t_rowptr, t_dst, t_eid, n_src, rowptr_dst, x, ldx, c_in, edge_attr, lde, f_e, We, be, da, 2 * c_in, dx, c_in,
                                                 da + c_in, 2 * c_in, n_dst, dWe, dbe, tmp, stream_, &sdesc));
        const int ns = slab_reduce_blocks(sdesc), nw = (wgrad_reduce_blocks(wdesc) + 3) / 4;
        hipLaunchKernelGGL(k_reduce_layer, dim3((unsigned)(ns + nw)), dim3(1024), 0, stream, sdesc, wdesc, ns);
        return dgnn_check_launch("sage_layer_train_bwd");
    }
    if (need_da) {
        if (!both && !pre_t) hipLaunchKernelGGL(k_transpose, dim3(dgnn_grid_cap(dgnn_cdiv((int64_t)c_in * c_out, 256))), dim3(256), 0, stream, Wj, c_out, c_in, WjT);
        // plain Linear block: the gradient of the input is da itself
        TRY(gemm(dz, c_out, c_out, WjT, c_out, 0, n_dst, c_in, agg ? da : dx, c_in));
    }
    if (agg) {
        if (We) DGNN_REQUIRE(dWe && dbe, DGNN_E_INVALID, "sage_layer_train_bwd: dWe / dbe missing");
        if (need_da)   // dWe / dbe are written (not accumulated) by the slab reduction: no fill
            TRY(dgnn_sage_aggregate_bwd(t_rowptr, t_dst, t_eid, n_src, rowptr_dst, x, ldx, c_in, edge_attr, lde, f_e, We, be, nullptr, 0, da, c_in, dx,
                                        c_in, dWe, dbe, nullptr, 0, tmp, stream_));
        if (need_dx && Wi) {
            if (!both && !pre_t) hipLaunchKernelGGL(k_transpose, dim3(dgnn_grid_cap(dgnn_cdiv((int64_t)c_in * c_out, 256))), dim3(256), 0, stream, Wi, c_out, c_in, WiT);
            TRY(gemm(dz, c_out, c_out, WiT, c_out, DGNN_LINEAR_ACCUMULATE, n_dst, c_in, dx, c_in));
        }
    }
    return dgnn_check_launch("sage_layer_train_bwd");
}

}  // namespace

extern "C" int dgnn_sage_layer_train_bwd(const int32_t* t_rowptr, const int32_t* t_dst, const int32_t* t_eid, const int32_t* rowptr_dst,
                                         int64_t n_src, int64_t n_dst, const float* x, int64_t ldx, int c_in, const float* edge_attr,
                                         int64_t lde, int f_e, const float* We, const float* be, const float* Wj, const float* Wi, int c_out,
                                         const float* gamma, const float* mean, const float* var, float eps, int relu, const float* a,
                                         const float* z, const float* y, const float* dy, float* dx, float* dWe, float* dbe, float* dWj,
                                         float* dbj, float* dWi, float* dgamma, float* dbeta, float* scratch, int gemm_mode, void* stream_) {
    DGNN_REQUIRE(n_dst > 0 && n_src >= n_dst && c_in > 0 && c_out > 0, DGNN_E_INVALID, "sage_layer_train_bwd: bad sizes");
    DGNN_REQUIRE(x && Wj && z && y && dy && mean && var && dWj && dgamma && dbeta && scratch, DGNN_E_INVALID, "sage_layer_train_bwd: null pointer");
    hipStream_t stream = (hipStream_t)stream_;
    float* dz = scratch;
    float* da = dz + align4(n_dst * c_out);
    float* WjT = da + 2 * align4(n_dst * c_in);
    float* WiT = WjT + (int64_t)c_in * c_out;            // stacked: [Wj^T ; Wi^T] is one [2 c_in, c_out] matrix
    float* tmp = WjT + 2 * align4((int64_t)c_in * c_out);
    float* tmp_w = tmp + layer_tmp_elems(n_src, n_dst, c_in, c_out, f_e);
    Aux* aux = aux_enabled() ? aux_of_current_device() : nullptr;
    hipEvent_t done = nullptr;
    if (aux) {   // the scratch may still be in use by the previous call's kernels on `stream`: the second stream starts behind them
        hipEvent_t e = next_event(aux);
        (void)hipEventRecord(e, stream);
        (void)hipStreamWaitEvent(aux->stream, e, 0);
    }
    const int rc = layer_bwd(t_rowptr, t_dst, t_eid, rowptr_dst, n_src, n_dst, x, ldx, c_in, edge_attr, lde, f_e, We, be, Wj, Wi, c_out, gamma, mean, var, eps,
                             relu, a, z, y, dy, dx, dWe, dbe, dWj, dbj, dWi, dgamma, dbeta, dz, da, WjT, WiT, tmp, tmp_w, gemm_mode, stream, aux, &done);
    if (aux && done) (void)hipStreamWaitEvent(stream, done, 0);   // join: the caller consumes the gradients on `stream`
    return rc;
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
    static bool can_fuse(int mode) { return mode != DGNN_GEMM_F32; }
    static int wgrad_cat(const T* A, int64_t lda, int na, const T* B1, int64_t ldb1, int nb1, const T* B2, int64_t ldb2, int nb2, int64_t M, float* dW1,
                         float* dW2, float* dbias, float* tmp, void* st) {
        return dgnn_linear_wgrad_x3_cat(A, lda, na, B1, ldb1, nb1, B2, ldb2, nb2, M, dW1, dW2, dbias, tmp, st);
    }
    static constexpr int kBf16 = 0;
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
    static bool can_fuse(int) { return true; }
    static int wgrad_cat(const T* A, int64_t lda, int na, const T* B1, int64_t ldb1, int nb1, const T* B2, int64_t ldb2, int nb2, int64_t M, float* dW1,
                         float* dW2, float* dbias, float* tmp, void* st) {
        return dgnn_linear_wgrad_bf16_cat(A, 0, lda, na, B1, ldb1, nb1, B2, ldb2, nb2, 0, M, dW1, dW2, dbias, tmp, st);
    }
    static constexpr int kBf16 = 1;
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

}  // namespace
// library-internal (csrc/aggregate.hip)
bool dgnn_agg_bwd_can_mask();
int dgnn_sage_aggregate_bwd_phi_add_masked(const int32_t* t_rowptr, const int32_t* t_dst, const int32_t* t_eid, int64_t n_src, const int32_t* rowptr_dst,
                                           const void* x_src, int64_t ldx, int c_in, const void* phi, int64_t ldphi, const void* da, int64_t ldda,
                                           void* dx_src, int64_t lddx, const void* add, int64_t ldadd, int64_t n_add, void* dphi_out, int64_t lddphi,
                                           const void* dphi_ext, int bf16, int mask_dx, void* stream);
namespace {

// dy_is_dz: the caller's dy already carries this layer's ReLU mask (the layer above stored its dx masked): no k_relu_bwd launch.
// want_mask / *masked: store dx with the mask of the layer BELOW (dx * [x > 0]; x is that layer's post-ReLU output) -- done in the fused chain only.
template <typename K>
int updated_bwd(const int32_t* t_rowptr, const int32_t* t_dst, const int32_t* t_eid, const int32_t* rowptr_dst, int64_t n_src, int64_t n_dst, int64_t E,
                const void* x_, int64_t ldx, int c_in, const void* ea_, int64_t lde, int k_e, const float* We, const float* Wl, const float* Wr, int c_out,
                int relu, const void* phi_, const void* a_, const void* y_, const void* dy_, const void* dphi_ext_, void* dx_, void* d_ea_, float* dWe,
                float* dbe, float* dWl, float* dbl, float* dWr, void* dz_, void* da_, void* dphi_, float* scratch, int mode, void* st,
                bool dy_is_dz = false, bool want_mask = false, bool* masked = nullptr) {
    typedef typename K::T T;
    hipStream_t stream = (hipStream_t)st;
    if (masked) *masked = false;
    const T *x = (const T*)x_, *ea = (const T*)ea_, *phi = (const T*)phi_, *a = (const T*)a_, *y = (const T*)y_, *dy = (const T*)dy_,
            *dphi_ext = (const T*)dphi_ext_;
    T *dx = (T*)dx_, *d_ea = (T*)d_ea_, *dz = (T*)dz_, *da = (T*)da_, *dphi = (T*)dphi_;
    float* WlT = scratch;
    float* WrT = WlT + align4((int64_t)c_in * c_out);
    float* WeT = WrT + align4((int64_t)c_in * c_out);
    float* tmp = WeT + align4((int64_t)c_in * k_e);
    // weight gradients and bias sums (the only kernels here that use partial-sum scratch) go to the second stream: dWl / dbl / dWr once
    // dz exists, dWe / dbe once dphi is complete; the dx / d_ea chain stays on `stream`, which joins them at the end
    Aux* aux = aux_enabled() ? aux_of_current_device() : nullptr;
    void* ws = aux ? (void*)aux->stream : st;
    auto fork = [&]() {
        if (!aux) return;
        hipEvent_t e = next_event(aux);
        (void)hipEventRecord(e, stream);
        (void)hipStreamWaitEvent(aux->stream, e, 0);
    };
    const T* g = dy;
    if (relu && !dy_is_dz) {
        TRY(K::relu_bwd(y, dy, n_dst * c_out, dz, st));
        g = dz;
    }
    if (!aux && fused_enabled() && K::can_fuse(mode)) {
        const int mask_dx = (want_mask && dx && dgnn_agg_bwd_can_mask()) ? 1 : 0;
        if (masked) *masked = mask_dx != 0;
        // The launch chain of the Static layer's fused backward (layer_bwd) for this variant: dWl / dWr / dbl from one launch pair, the three
        // transposes from one launch, [da | dz.Wr] from one GEMM against the stacked [Wl^T ; Wr^T] with the second half added where the
        // aggregate backward stores dx and dphi_ext added where it stores dphi, dWe / dbe from one launch pair.  `da` holds [n_dst, 2 c_in].
        const bool both = dx && Wr;
        TrJobs jobs;
        jobs.n = 0;
        int total = 0;
        auto job = [&](const float* in, float* out, int rows, int cols) {
            const int j = jobs.n++;
            jobs.in[j] = in, jobs.out[j] = out, jobs.rows[j] = rows, jobs.cols[j] = cols;
            total += rows * cols;
            jobs.end[j] = total;
        };
        float* WrS = WlT + (int64_t)c_in * c_out;      // stacked right under Wl^T (the region of WrT begins at or after it)
        job(Wl, WlT, c_out, c_in);
        if (both) job(Wr, WrS, c_out, c_in);
        if (E > 0 && d_ea) job(We, WeT, c_in, k_e);
        hipLaunchKernelGGL(k_transpose_many, dim3(dgnn_grid_cap(dgnn_cdiv(total, 256))), dim3(256), 0, stream, jobs);
        const T* B2 = (Wr && dWr) ? x : nullptr;
        TRY(K::wgrad_cat(g, c_out, c_out, a, c_in, c_in, B2, ldx, B2 ? c_in : 0, n_dst, dWl, dWr, dbl, tmp, st));
        // (dphi += dphi_ext stays a launch of its own: folded into the aggregate backward's dphi store -- dgnn_sage_aggregate_bwd_phi_add can do
        // it -- the fourth row load per edge cost the kernel 60 % (72 -> 115 us on the outermost block) against the 7-9 us of k_add_inplace)
        if (both) {
            TRY(K::linear(g, c_out, c_out, WlT, c_out, nullptr, 0, 0, nullptr, 0, nullptr, 0, n_dst, 2 * c_in, da, 2 * c_in, mode, st));
            TRY(dgnn_sage_aggregate_bwd_phi_add_masked(t_rowptr, t_dst, t_eid, n_src, rowptr_dst, x, ldx, c_in, phi, c_in, da, 2 * c_in, dx, c_in, da + c_in,
                                                       2 * c_in, n_dst, dphi, c_in, nullptr, K::kBf16, mask_dx, st));
        } else {
            TRY(K::linear(g, c_out, c_out, WlT, c_out, nullptr, 0, 0, nullptr, 0, nullptr, 0, n_dst, c_in, da, c_in, mode, st));
            TRY(dgnn_sage_aggregate_bwd_phi_add_masked(t_rowptr, t_dst, t_eid, n_src, rowptr_dst, x, ldx, c_in, phi, c_in, da, c_in, dx, c_in, nullptr, 0, 0, dphi,
                                                       c_in, nullptr, K::kBf16, mask_dx, st));
        }
        if (E > 0 && dphi_ext)
            hipLaunchKernelGGL((k_add_inplace<T>), dim3(dgnn_grid_cap(dgnn_cdiv(E * c_in, 256))), dim3(256), 0, stream, dphi, dphi_ext, E * c_in);
        if (E > 0) {
            TRY(K::wgrad_cat(dphi, c_in, c_in, ea, lde, k_e, nullptr, 0, 0, E, dWe, nullptr, dbe, tmp, st));
            if (d_ea) TRY(K::linear(dphi, c_in, c_in, WeT, c_in, nullptr, 0, 0, nullptr, 0, nullptr, 0, E, k_e, d_ea, k_e, mode, st));
        } else {
            (void)hipMemsetAsync(dWe, 0, sizeof(float) * (size_t)c_in * k_e, stream);
            (void)hipMemsetAsync(dbe, 0, sizeof(float) * (size_t)c_in, stream);
        }
        return dgnn_check_launch("sage_updated_train_bwd");
    }
    fork();   // also orders the second stream behind whatever used `scratch` before on `stream`
    TRY(K::wgrad(g, c_out, c_out, a, c_in, c_in, n_dst, dWl, tmp, mode, ws));
    if (dbl) TRY(K::colsum(g, c_out, n_dst, c_out, dbl, tmp, ws));
    if (Wr && dWr) TRY(K::wgrad(g, c_out, c_out, x, ldx, c_in, n_dst, dWr, tmp, mode, ws));
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
        fork();
        TRY(K::wgrad(dphi, c_in, c_in, ea, lde, k_e, E, dWe, tmp, mode, ws));
        TRY(K::colsum(dphi, c_in, E, c_in, dbe, tmp, ws));
        if (d_ea) {
            transpose_to(We, c_in, k_e, WeT, stream);
            TRY(K::linear(dphi, c_in, c_in, WeT, c_in, nullptr, 0, 0, nullptr, 0, nullptr, 0, E, k_e, d_ea, k_e, mode, st));
        }
    } else {
        (void)hipMemsetAsync(dWe, 0, sizeof(float) * (size_t)c_in * k_e, stream);
        (void)hipMemsetAsync(dbe, 0, sizeof(float) * (size_t)c_in, stream);
    }
    if (aux) {   // join
        hipEvent_t e = next_event(aux);
        (void)hipEventRecord(e, aux->stream);
        (void)hipStreamWaitEvent(stream, e, 0);
    }
    return dgnn_check_launch("sage_updated_train_bwd");
}

}  // namespace

extern "C" int64_t dgnn_sage_updated_train_scratch_elems(int64_t n_dst, int64_t E, int c_in, int c_out, int k_e) {
    if (n_dst < 0 || E < 0 || c_in <= 0 || c_out <= 0 || k_e <= 0) return 16;
    int64_t big = dgnn_colstats_scratch_elems(n_dst > E ? n_dst : E, c_in > c_out ? c_in : c_out);
    const int64_t w1 = dgnn_linear_wgrad_cat_scratch_elems(n_dst, c_out, c_in, c_in), w2 = dgnn_linear_wgrad_cat_scratch_elems(E, c_in, k_e, 0);
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

// =====================================================================================================================
// Static model, training mode, ALL layers per call: the chain of dgnn_sage_layer_train_fwd / _bwd calls (conv layers, then the
// decoder's Linear + BatchNorm + ReLU block as a layer with rowptr[l] == NULL) issued from one entry point each way.  Layer l
// reads the previous layer's y (layer 0: x0); the blocks nest (the destinations of layer l are the sources of layer l+1), so
// layer l's output gradient IS layer l+1's dx.  Per-layer arrays are HOST arrays of device pointers / sizes.
// =====================================================================================================================
namespace {
struct Ptr8 {
    int64_t* p[8];
};
__global__ void k_inc_i64(Ptr8 ps, int n) {
    if (threadIdx.x < n && ps.p[threadIdx.x]) *ps.p[threadIdx.x] += 1;
}
}  // namespace

extern "C" int dgnn_static_train_fwd(int n_layers, const int32_t* const* rowptr, const int32_t* const* src, const int32_t* const* eid,
                                     const int64_t* n_dst, const float* x0, int64_t ldx0, const int32_t* widths, const float* const* edge_attr,
                                     const int64_t* lde, int f_e, const float* const* We, const float* const* be, const float* const* Wj,
                                     const float* const* bj, const float* const* Wi, const float* const* gamma, const float* const* beta,
                                     float* const* running_mean, float* const* running_var, int64_t* const* num_batches_tracked,
                                     const float* momentum, const float* eps, float* const* a, float* const* z, float* const* stats,
                                     float* const* y, float* scratch, int gemm_mode, void* stream) {
    DGNN_REQUIRE(n_layers >= 1 && n_layers <= 8 && rowptr && src && eid && n_dst && x0 && widths && edge_attr && lde && We && be && Wj && bj && Wi && gamma &&
                     beta && running_mean && running_var && momentum && eps && a && z && stats && y && scratch,
                 DGNN_E_INVALID, "static_train_fwd: bad args (at most 8 layers)");
    const float* x = x0;
    int64_t ldx = ldx0;
    unsigned counted_mask = 0;
    for (int l = 0; l < n_layers; ++l) {
        const int c_in = widths[l], c_out = widths[l + 1];
        float* st = stats[l];
        if (!st) {   // a plain Linear (the decoder's output layer, :187): y = x . Wj^T + bj, no BatchNorm, no ReLU
            DGNN_REQUIRE(!rowptr[l] && Wj[l] && y[l], DGNN_E_INVALID, "static_train_fwd: a layer without statistics is a plain Linear");
            if (gemm_mode == DGNN_GEMM_F32)
                TRY(dgnn_linear_fwd(x, ldx, c_in, Wj[l], c_in, nullptr, 0, 0, nullptr, 0, bj[l], nullptr, nullptr, 0, n_dst[l], c_out, y[l], c_out, stream));
            else
                TRY(dgnn_linear_fwd_x3(x, ldx, c_in, Wj[l], c_in, nullptr, 0, 0, nullptr, 0, bj[l], nullptr, nullptr, 0, n_dst[l], c_out, y[l], c_out, stream));
            x = y[l];
            ldx = c_out;
            continue;
        }
        bool counted = false;
        TRY(layer_train_fwd(rowptr[l], src[l], eid[l], n_dst[l], x, ldx, c_in, edge_attr[l], lde[l], We[l] ? f_e : 0, We[l], be[l], Wj[l], bj[l],
                            Wi[l], c_out, gamma[l], beta[l], running_mean[l], running_var[l], momentum[l], eps[l], 1, a[l], z[l], st, st + c_out,
                            st + 2 * c_out, st + 3 * c_out, y[l], scratch, gemm_mode, stream, num_batches_tracked ? num_batches_tracked[l] : nullptr, &counted));
        if (counted) counted_mask |= 1u << l;
        x = y[l];
        ldx = c_out;
    }
    if (num_batches_tracked) {
        // the counters the statistics' finalising launches have not already stepped (round 6: in the default arithmetic every BatchNorm's has been)
        Ptr8 ps;
        int left = 0;
        for (int l = 0; l < 8; ++l) {
            ps.p[l] = (l < n_layers && !((counted_mask >> l) & 1u)) ? num_batches_tracked[l] : nullptr;
            left += ps.p[l] != nullptr;
        }
        if (left) hipLaunchKernelGGL(k_inc_i64, dim3(1), dim3(64), 0, (hipStream_t)stream, ps, n_layers);
    }
    return dgnn_check_launch("static_train_fwd");
}

// dy: gradient of the last layer's y.  dx_buf[0], dx_buf[1]: two work buffers of max_l n_src[l] * widths[l] floats (layer l writes
// its dx into dx_buf[l & 1], layer l-1 reads it as dy); layer 0's input is data (no dx).  Parameter gradients per layer.
namespace {
// scratch of dgnn_static_train_bwd: dz ping-pong | da ([n_dst, 2 c_in]: da next to dz.Wi) | every layer's [Wj^T ; Wi^T] | main-stream partials |
// weight-gradient partials
struct StaticScratch {
    int64_t dz = 0, da = 0, wt_total = 0, tmp = 0, tw = 0, wt_off[8] = {};
};
StaticScratch static_scratch(int n_layers, const int64_t* n_src, const int64_t* n_dst, const int32_t* widths, int f_e) {
    StaticScratch r;
    for (int l = 0; l < n_layers; ++l) {
        const int ci = widths[l], co = widths[l + 1];
        const int64_t a1 = align4(n_dst[l] * co), a2 = 2 * align4(n_dst[l] * ci), a3 = 2 * align4((int64_t)ci * co);
        const int64_t stats = dgnn_colstats_scratch_elems(n_dst[l], co > ci ? co : ci), wg = dgnn_linear_wgrad_scratch_elems(n_dst[l], co, ci),
                      wc = dgnn_linear_wgrad_cat_scratch_elems(n_dst[l], co, ci, ci), ab = dgnn_sage_aggregate_bwd_scratch_elems(n_src[l], ci, f_e > 0 ? f_e : 1);
        if (a1 > r.dz) r.dz = a1;
        if (a2 > r.da) r.da = a2;
        r.wt_off[l] = r.wt_total;
        r.wt_total += a3;
        const int64_t t1 = align4(stats > ab ? stats : ab);
        int64_t t2 = stats > wg ? stats : wg;
        if (wc > t2) t2 = wc;
        t2 = align4(t2);
        if (t1 > r.tmp) r.tmp = t1;
        if (t2 > r.tw) r.tw = t2;
    }
    return r;
}
}  // namespace

extern "C" int64_t dgnn_static_train_scratch_elems(int n_layers, const int64_t* n_src, const int64_t* n_dst, const int32_t* widths, int f_e) {
    if (n_layers < 1 || n_layers > 8 || !n_src || !n_dst || !widths) return 16;
    const StaticScratch r = static_scratch(n_layers, n_src, n_dst, widths, f_e);
    return 2 * r.dz + r.da + r.wt_total + r.tmp + r.tw + 64;
}

// dy: gradient of the last layer's y.  dx_buf[0], dx_buf[1]: two work buffers of max_l n_src[l] * widths[l] floats (layer l writes
// its dx into dx_buf[l & 1], layer l-1 reads it as dy); layer 0's input is data (no dx).  Parameter gradients per layer.  scratch:
// dgnn_static_train_scratch_elems floats.  The weight gradients (dWj, dbj, dWi) run on a library-owned second stream beside the dx
// chain (DGNN_TRAIN_AUX_STREAM=0: everything on `stream`); `stream` has waited for all of them when the call returns.
extern "C" int dgnn_static_train_bwd(int n_layers, const int32_t* const* t_rowptr, const int32_t* const* t_dst, const int32_t* const* t_eid,
                                     const int32_t* const* rowptr_dst, const int64_t* n_src, const int64_t* n_dst, const float* x0, int64_t ldx0,
                                     const int32_t* widths, const float* const* edge_attr, const int64_t* lde, int f_e, const float* const* We,
                                     const float* const* be, const float* const* Wj, const float* const* Wi, const float* const* gamma,
                                     const float* const* stats, const float* eps, const float* const* a, const float* const* z,
                                     const float* const* y, const float* dy, float* const* dWe, float* const* dbe, float* const* dWj,
                                     float* const* dbj, float* const* dWi, float* const* dgamma, float* const* dbeta, float* const* dx_buf,
                                     float* scratch, int gemm_mode, void* stream_) {
    DGNN_REQUIRE(n_layers >= 1 && n_layers <= 8 && t_rowptr && t_dst && t_eid && rowptr_dst && n_src && n_dst && x0 && widths && edge_attr && lde && We && be &&
                     Wj && Wi && gamma && stats && eps && a && z && y && dy && dWe && dbe && dWj && dbj && dWi && dgamma && dbeta && dx_buf && scratch,
                 DGNN_E_INVALID, "static_train_bwd: bad args");
    hipStream_t stream = (hipStream_t)stream_;
    Aux* aux = aux_enabled() ? aux_of_current_device() : nullptr;
    const StaticScratch lay = static_scratch(n_layers, n_src, n_dst, widths, f_e);
    float* dzb[2] = {scratch, scratch + lay.dz};
    float* da = scratch + 2 * lay.dz;
    float* wt = da + lay.da;
    float* tmp = wt + lay.wt_total;
    float* tmp_w = tmp + lay.tmp;
    // all transposes of the pass in one launch (fused chain), into per-layer regions; otherwise every layer transposes into the first region
    const bool pre_t = gemm_mode != DGNN_GEMM_F32 && !aux && fused_enabled();
    if (pre_t) {
        TrJobs jobs;
        jobs.n = 0;
        int total = 0;
        for (int l = 0; l < n_layers; ++l) {
            const int ci = widths[l], co = widths[l + 1];
            const bool agg = t_rowptr[l] != nullptr, need_dx = l > 0;
            const bool need_da = agg ? (need_dx || We[l] != nullptr) : need_dx;
            if (!need_da) continue;
            const bool both = agg && need_dx && Wi[l];
            for (int k = 0; k < (both ? 2 : 1); ++k) {
                const int j = jobs.n++;
                jobs.in[j] = k == 0 ? Wj[l] : Wi[l];
                jobs.out[j] = wt + lay.wt_off[l] + (int64_t)k * ci * co;
                jobs.rows[j] = co, jobs.cols[j] = ci;
                total += ci * co;
                jobs.end[j] = total;
            }
        }
        if (jobs.n) hipLaunchKernelGGL(k_transpose_many, dim3(dgnn_grid_cap(dgnn_cdiv(total, 256))), dim3(256), 0, stream, jobs);
    }
    hipEvent_t done[8] = {};
    if (aux) {   // the second stream starts after everything already queued on `stream` (its inputs, and last step's use of the scratch)
        hipEvent_t e = next_event(aux);
        (void)hipEventRecord(e, stream);
        (void)hipStreamWaitEvent(aux->stream, e, 0);
    }
    const float* g = dy;
    int rc = DGNN_OK;
    for (int l = n_layers - 1; l >= 0 && rc == DGNN_OK; --l) {
        const int c_in = widths[l], c_out = widths[l + 1];
        const float* x = l == 0 ? x0 : y[l - 1];
        const int64_t ldx = l == 0 ? ldx0 : c_in;
        float* dx = l == 0 ? nullptr : dx_buf[l & 1];
        const float* st = stats[l];
        if (aux && l + 2 < n_layers && done[l + 2]) (void)hipStreamWaitEvent(stream, done[l + 2], 0);   // dz[l & 1] is still read by layer l+2's weight gradients
        float* WjT = wt + (pre_t ? lay.wt_off[l] : 0);
        rc = layer_bwd(t_rowptr[l], t_dst[l], t_eid[l], rowptr_dst[l], n_src[l], n_dst[l], x, ldx, c_in, edge_attr[l], lde[l], We[l] ? f_e : 0, We[l], be[l], Wj[l],
                       Wi[l], c_out, gamma[l], st, st ? st + c_out : nullptr, eps[l], 1, a[l], z[l], y[l], g, dx, dWe[l], dbe[l], dWj[l], dbj[l], dWi[l], dgamma[l],
                       dbeta[l], dzb[l & 1], da, WjT, WjT + (int64_t)c_in * c_out, tmp, tmp_w, gemm_mode, stream, aux, aux ? &done[l] : nullptr, pre_t, st != nullptr,
                       st ? st + 2 * c_out : nullptr);
        g = dx;
    }
    if (aux)   // the gradients are consumed on `stream` (optimizer step): join.  In-order on the second stream: the last event covers all.
        for (int l = 0; l < n_layers; ++l)
            if (done[l]) (void)hipStreamWaitEvent(stream, done[l], 0);
    return rc;
}

// =====================================================================================================================
// All conv layers of the Updated variant per call (surfaceNetUpdatedEdgeFilters.py:229-243 and its autograd): the per-layer composite calls
// above and the edge chaining between them (chain.hip) issued back to back from C++ -- one autograd node for the stack instead of three per
// layer (edge rows, cast, conv): the step was bound by the ~16 Python-level nodes each way.  The same kernels in the same order as the
// per-layer path (bit-identical results).
//   ea_0 = edge_attr_all[rows0, :edge_in_0]                        (:237; bf16 storage: cast once)
//   ea_l = relu(zeros[E_all, C]; [e_id_{l-1}] = phi_{l-1})[e_id_l, :edge_in_l]      (:233-241, only the rows that are read)
//   (y_l, phi_l) = conv_l(x_l, ea_l), x_{l+1} = relu?(y_l)
// Backward: layer l's d_ea goes through the chaining's backward into dphi_ext of layer l-1.
// =====================================================================================================================
extern "C" int dgnn_updated_stack_fwd(int n_layers, const int32_t* const* rowptr, const int32_t* const* src, const int32_t* const* eid,
                                      const int64_t* const* e_id, const int32_t* rows0, const int64_t* n_dst, const int64_t* E, const void* x0,
                                      int64_t ldx0, const int32_t* widths, const int32_t* edge_in, const float* edge_attr_all, int64_t lde_all,
                                      int64_t E_all, int32_t* pos, const float* const* We, const float* const* be, const float* const* Wl,
                                      const float* const* bl, const float* const* Wr, const int32_t* relu, void* const* ea, const int64_t* ld_ea,
                                      float* ea0_f32, void* const* phi, void* const* a, void* const* y, int32_t* const* inv, int bf16, int gemm_mode,
                                      void* stream) {
    DGNN_REQUIRE(n_layers >= 1 && n_layers <= 8 && rowptr && src && eid && e_id && rows0 && n_dst && E && x0 && widths && edge_in && edge_attr_all && pos &&
                     We && be && Wl && bl && Wr && relu && ea && ld_ea && phi && a && y && inv && (!bf16 || ea0_f32),
                 DGNN_E_INVALID, "updated_stack_fwd: bad args (at most 8 layers)");
    const void* x = x0;
    int64_t ldx = ldx0;
    for (int l = 0; l < n_layers; ++l) {
        const int c_in = widths[l], c_out = widths[l + 1], k = edge_in[l];
        if (l == 0) {
            if (E[0] > 0) {
                if (bf16) {
                    TRY(dgnn_gather_rows_f32(edge_attr_all, lde_all, rows0, E[0], k, ea0_f32, k, stream));
                    TRY(dgnn_cast_f32_to_bf16(ea0_f32, k, E[0], k, (int)ld_ea[0], (uint16_t*)ea[0], ld_ea[0], stream));
                } else {
                    TRY(dgnn_gather_rows_f32(edge_attr_all, lde_all, rows0, E[0], k, (float*)ea[0], ld_ea[0], stream));
                }
            }
        } else if (E[l] > 0 || E[l - 1] > 0) {
            DGNN_REQUIRE(inv[l], DGNN_E_INVALID, "updated_stack_fwd: inv[%d] missing", l);
            if (bf16)
                TRY(dgnn_edge_chain_fwd_bf16((const uint16_t*)phi[l - 1], widths[l - 1], k, e_id[l - 1], E[l - 1], e_id[l], E[l], E_all, pos, 1, (uint16_t*)ea[l],
                                             ld_ea[l], inv[l], stream));
            else
                TRY(dgnn_edge_chain_fwd((const float*)phi[l - 1], widths[l - 1], k, e_id[l - 1], E[l - 1], e_id[l], E[l], E_all, pos, 1, (float*)ea[l], ld_ea[l],
                                        inv[l], stream));
        }
        TRY(dgnn_sage_updated_train_fwd(rowptr[l], src[l], eid[l], n_dst[l], x, ldx, c_in, ea[l], ld_ea[l], k, E[l], We[l], be[l], Wl[l], bl[l], Wr[l], c_out,
                                        relu[l], phi[l], a[l], y[l], bf16, gemm_mode, stream));
        x = y[l];
        ldx = c_out;
    }
    return dgnn_check_launch("updated_stack_fwd");
}

// scratch: max over the layers of dgnn_sage_updated_train_scratch_elems.  Work buffers (storage type): dx_buf[0], dx_buf[1] (max n_src * c_in
// over the layers l >= 1), d_ea (max E_l * edge_in_l, l >= 1), dphi_ext (max E_l * c_in_l, l < n_layers - 1), dz (max n_dst * c_out), da
// (max 2 * n_dst * c_in), dphi (max E_l * c_in_l).
extern "C" int dgnn_updated_stack_bwd(int n_layers, const int32_t* const* t_rowptr, const int32_t* const* t_dst, const int32_t* const* t_eid,
                                      const int32_t* const* rowptr_dst, const int64_t* n_src, const int64_t* n_dst, const int64_t* E, const void* x0,
                                      int64_t ldx0, const int32_t* widths, const int32_t* edge_in, const float* const* We, const float* const* Wl,
                                      const float* const* Wr, const int32_t* relu, const void* const* ea, const int64_t* ld_ea, const void* const* phi,
                                      const void* const* a, const void* const* y, const int32_t* const* inv, const void* dy, float* const* dWe,
                                      float* const* dbe, float* const* dWl, float* const* dbl, float* const* dWr, void* const* dx_buf, void* d_ea,
                                      void* dphi_ext, void* dz, void* da, void* dphi, float* scratch, int bf16, int gemm_mode, void* stream) {
    DGNN_REQUIRE(n_layers >= 1 && n_layers <= 8 && t_rowptr && t_dst && t_eid && rowptr_dst && n_src && n_dst && E && x0 && widths && edge_in && We && Wl && Wr &&
                     relu && ea && ld_ea && phi && a && y && inv && dy && dWe && dbe && dWl && dbl && dWr && dx_buf && dz && da && dphi && scratch &&
                     (n_layers == 1 || (d_ea && dphi_ext && dx_buf[0] && dx_buf[1])),
                 DGNN_E_INVALID, "updated_stack_bwd: bad args");
    const void* g = dy;
    bool have_ext = false;
    // round 6: a layer stores its dx already masked by the ReLU of the layer below (its own input x is that layer's post-ReLU output), so the layer below
    // takes it as dz: one k_relu_bwd launch per inner layer less.  DGNN_UPDATED_MASK_DX=0: the launch chain of rounds 4-5.
    static const bool mask_on = !(getenv("DGNN_UPDATED_MASK_DX") && getenv("DGNN_UPDATED_MASK_DX")[0] == '0');
    bool g_masked = false;
    for (int l = n_layers - 1; l >= 0; --l) {
        const int c_in = widths[l], c_out = widths[l + 1], k = edge_in[l];
        const void* x = l == 0 ? x0 : y[l - 1];
        const int64_t ldx = l == 0 ? ldx0 : c_in;
        void* dx = l == 0 ? nullptr : dx_buf[l & 1];
        DGNN_REQUIRE(n_dst[l] > 0 && n_src[l] >= n_dst[l] && E[l] >= 0 && c_in > 0 && c_out > 0 && k > 0, DGNN_E_INVALID, "updated_stack_bwd: bad sizes");
        const bool want_mask = mask_on && l > 0 && relu[l - 1] != 0;
        bool masked = false;
        const int rc = bf16 ? updated_bwd<BF16>(t_rowptr[l], t_dst[l], t_eid[l], rowptr_dst[l], n_src[l], n_dst[l], E[l], x, ldx, c_in, ea[l], ld_ea[l], k, We[l], Wl[l],
                                                Wr[l], c_out, relu[l], phi[l], a[l], y[l], g, have_ext ? dphi_ext : nullptr, dx, l > 0 ? d_ea : nullptr, dWe[l],
                                                dbe[l], dWl[l], dbl[l], dWr[l], dz, da, dphi, scratch, gemm_mode, stream, g_masked, want_mask, &masked)
                            : updated_bwd<F32>(t_rowptr[l], t_dst[l], t_eid[l], rowptr_dst[l], n_src[l], n_dst[l], E[l], x, ldx, c_in, ea[l], ld_ea[l], k, We[l], Wl[l],
                                               Wr[l], c_out, relu[l], phi[l], a[l], y[l], g, have_ext ? dphi_ext : nullptr, dx, l > 0 ? d_ea : nullptr, dWe[l],
                                               dbe[l], dWl[l], dbl[l], dWr[l], dz, da, dphi, scratch, gemm_mode, stream, g_masked, want_mask, &masked);
        if (rc != DGNN_OK) return rc;
        g_masked = masked;
        have_ext = false;
        if (l > 0 && E[l - 1] > 0) {   // layer l's edge rows came out of phi_{l-1}: their gradient is what layer l-1 adds to its dphi
            if (bf16)
                TRY(dgnn_edge_chain_bwd_bf16((const uint16_t*)d_ea, k, (const uint16_t*)phi[l - 1], widths[l - 1], inv[l], E[l - 1], k, widths[l - 1], 1,
                                             (uint16_t*)dphi_ext, widths[l - 1], stream));
            else
                TRY(dgnn_edge_chain_bwd((const float*)d_ea, k, (const float*)phi[l - 1], widths[l - 1], inv[l], E[l - 1], k, widths[l - 1], 1, (float*)dphi_ext,
                                        widths[l - 1], stream));
            have_ext = true;
        }
        g = dx;
    }
    return dgnn_check_launch("updated_stack_bwd");
}

// The Updated model's output network behind the conv stack ("sage+": out_net = ReLU, Linear(C, H), ReLU, Linear(H, n_out), reference
// surfaceNetUpdatedEdgeFilters.py:210, 245-247; the first ReLU is the one after the last conv) as one call each way, for the same autograd node:
//   h = relu(x . W1^T + b1)   (storage type)        logits = h . W3^T + b3   (fp32)
// backward: dW3 / db3 and dW1 / db1 from one launch pair each, dh = g . W3, dz1 = dh * [h > 0], dx = dz1 . W1; one transpose launch.
// scratch (floats): 2 * (C*H + H*n_out) + dgnn_linear_wgrad_cat_scratch_elems(n, H, C, 0) + n * n_out + 64.  dh: [n, H] work buffer (storage type).
extern "C" int dgnn_updated_tail_fwd(int64_t n, const void* x, int64_t ldx, int c, const float* W1, const float* b1, int hdim, const float* W3, const float* b3,
                                     int n_out, void* h, float* logits, int bf16, int gemm_mode, void* stream) {
    DGNN_REQUIRE(n > 0 && c > 0 && hdim > 0 && n_out > 0 && x && W1 && W3 && h && logits, DGNN_E_INVALID, "updated_tail_fwd: bad arguments");
    if (bf16) {
        TRY(dgnn_linear_fwd_bf16((const uint16_t*)x, ldx, c, W1, c, nullptr, 0, 0, nullptr, 0, b1, nullptr, nullptr, 1, n, hdim, h, hdim, 0, stream));
        TRY(dgnn_linear_fwd_bf16((const uint16_t*)h, hdim, hdim, W3, hdim, nullptr, 0, 0, nullptr, 0, b3, nullptr, nullptr, 0, n, n_out, logits, n_out, 1, stream));
    } else if (gemm_mode == DGNN_GEMM_F32) {
        TRY(dgnn_linear_fwd((const float*)x, ldx, c, W1, c, nullptr, 0, 0, nullptr, 0, b1, nullptr, nullptr, 1, n, hdim, (float*)h, hdim, stream));
        TRY(dgnn_linear_fwd((const float*)h, hdim, hdim, W3, hdim, nullptr, 0, 0, nullptr, 0, b3, nullptr, nullptr, 0, n, n_out, logits, n_out, stream));
    } else {
        TRY(dgnn_linear_fwd_x3((const float*)x, ldx, c, W1, c, nullptr, 0, 0, nullptr, 0, b1, nullptr, nullptr, 1, n, hdim, (float*)h, hdim, stream));
        TRY(dgnn_linear_fwd_x3((const float*)h, hdim, hdim, W3, hdim, nullptr, 0, 0, nullptr, 0, b3, nullptr, nullptr, 0, n, n_out, logits, n_out, stream));
    }
    return DGNN_OK;
}

extern "C" int64_t dgnn_updated_tail_scratch_elems(int64_t n, int c, int hdim, int n_out) {
    if (n < 0 || c <= 0 || hdim <= 0 || n_out <= 0) return 64;
    const int64_t w1 = dgnn_linear_wgrad_cat_scratch_elems(n, hdim, c, 0), w3 = dgnn_linear_wgrad_cat_scratch_elems(n, n_out, hdim, 0);
    const int64_t cs = dgnn_colstats_scratch_elems(n, hdim > n_out ? hdim : n_out), wg = dgnn_linear_wgrad_scratch_elems(n, hdim, c);
    int64_t big = w1 > w3 ? w1 : w3;
    if (cs > big) big = cs;
    if (wg > big) big = wg;
    return 2 * (align4((int64_t)c * hdim) + align4((int64_t)hdim * n_out)) + align4(big) + align4(n * n_out) + 64;
}

extern "C" int dgnn_updated_tail_bwd(int64_t n, const void* x, int64_t ldx, int c, const float* W1, int hdim, const float* W3, int n_out, const void* h,
                                     const float* g, float* dW1, float* db1, float* dW3, float* db3, void* dx, void* dh, float* scratch, int bf16,
                                     int gemm_mode, void* stream_) {
    DGNN_REQUIRE(n > 0 && c > 0 && hdim > 0 && n_out > 0 && x && W1 && W3 && h && g && dW1 && dW3 && dx && dh && scratch, DGNN_E_INVALID,
                 "updated_tail_bwd: bad arguments");
    hipStream_t stream = (hipStream_t)stream_;
    float* W3T = scratch;                                        // [hdim, n_out]
    float* W1T = W3T + align4((int64_t)hdim * n_out);            // [c, hdim]
    float* tmp = W1T + align4((int64_t)c * hdim);
    const int64_t w1 = dgnn_linear_wgrad_cat_scratch_elems(n, hdim, c, 0), w3 = dgnn_linear_wgrad_cat_scratch_elems(n, n_out, hdim, 0);
    const int64_t cs = dgnn_colstats_scratch_elems(n, hdim > n_out ? hdim : n_out), wg = dgnn_linear_wgrad_scratch_elems(n, hdim, c);
    int64_t big = w1 > w3 ? w1 : w3;
    if (cs > big) big = cs;
    if (wg > big) big = wg;
    uint16_t* gb = reinterpret_cast<uint16_t*>(tmp + align4(big));   // bf16 copy of the logits' gradient (bf16 storage)
    TrJobs jobs;
    jobs.n = 2;
    jobs.in[0] = W3, jobs.out[0] = W3T, jobs.rows[0] = n_out, jobs.cols[0] = hdim, jobs.end[0] = n_out * hdim;
    jobs.in[1] = W1, jobs.out[1] = W1T, jobs.rows[1] = hdim, jobs.cols[1] = c, jobs.end[1] = jobs.end[0] + hdim * c;
    hipLaunchKernelGGL(k_transpose_many, dim3(dgnn_grid_cap(dgnn_cdiv(jobs.end[1], 256))), dim3(256), 0, stream, jobs);
    const bool fused = fused_enabled();
    if (bf16) {
        const int gp = (n_out + 1) / 2 * 2;
        TRY(dgnn_cast_f32_to_bf16(g, n_out, n, n_out, gp, gb, gp, stream_));
        if (fused) {
            TRY(dgnn_linear_wgrad_bf16_cat(g, 1, n_out, n_out, h, hdim, hdim, nullptr, 0, 0, 0, n, dW3, nullptr, db3, tmp, stream_));
        } else {
            TRY(dgnn_linear_wgrad_bf16(g, 1, n_out, n_out, h, 0, hdim, hdim, n, dW3, hdim, 0, tmp, stream_));
            if (db3) TRY(dgnn_colsum(g, n_out, n, n_out, db3, 0, tmp, stream_));
        }
        TRY(dgnn_linear_fwd_bf16(gb, gp, n_out, W3T, n_out, nullptr, 0, 0, nullptr, 0, nullptr, nullptr, nullptr, 0, n, hdim, dh, hdim, 0, stream_));
        TRY(dgnn_relu_bwd_bf16((const uint16_t*)h, (const uint16_t*)dh, n * hdim, (uint16_t*)dh, stream_));
        if (fused) {
            TRY(dgnn_linear_wgrad_bf16_cat(dh, 0, hdim, hdim, x, ldx, c, nullptr, 0, 0, 0, n, dW1, nullptr, db1, tmp, stream_));
        } else {
            TRY(dgnn_linear_wgrad_bf16(dh, 0, hdim, hdim, x, 0, ldx, c, n, dW1, c, 0, tmp, stream_));
            if (db1) TRY(dgnn_colsum_bf16((const uint16_t*)dh, hdim, n, hdim, db1, 0, tmp, stream_));
        }
        TRY(dgnn_linear_fwd_bf16((const uint16_t*)dh, hdim, hdim, W1T, hdim, nullptr, 0, 0, nullptr, 0, nullptr, nullptr, nullptr, 0, n, c, dx, c, 0, stream_));
        return dgnn_check_launch("updated_tail_bwd");
    }
    const bool x3 = gemm_mode != DGNN_GEMM_F32;
    auto gemm = [&](const float* A, int64_t lda, int k, const float* W, int64_t ldw, int64_t M, int no, float* out, int64_t ldo) {
        return x3 ? dgnn_linear_fwd_x3(A, lda, k, W, ldw, nullptr, 0, 0, nullptr, 0, nullptr, nullptr, nullptr, 0, M, no, out, ldo, stream_)
                  : dgnn_linear_fwd(A, lda, k, W, ldw, nullptr, 0, 0, nullptr, 0, nullptr, nullptr, nullptr, 0, M, no, out, ldo, stream_);
    };
    auto wgrad = [&](const float* A, int64_t lda, int na, const float* B, int64_t ldb, int nb, float* dW, float* dbias) {
        if (x3 && fused) return dgnn_linear_wgrad_x3_cat(A, lda, na, B, ldb, nb, nullptr, 0, 0, n, dW, nullptr, dbias, tmp, stream_);
        int rc = x3 ? dgnn_linear_wgrad_x3(A, lda, na, B, ldb, nb, n, dW, nb, 0, tmp, stream_) : dgnn_linear_wgrad(A, lda, na, B, ldb, nb, n, dW, nb, 0, tmp, stream_);
        if (rc == DGNN_OK && dbias) rc = dgnn_colsum(A, lda, n, na, dbias, 0, tmp, stream_);
        return rc;
    };
    TRY(wgrad(g, n_out, n_out, (const float*)h, hdim, hdim, dW3, db3));
    TRY(gemm(g, n_out, n_out, W3T, n_out, n, hdim, (float*)dh, hdim));
    TRY(dgnn_relu_bwd((const float*)h, (const float*)dh, n * hdim, (float*)dh, stream_));
    TRY(wgrad((const float*)dh, hdim, hdim, (const float*)x, ldx, c, dW1, db1));
    TRY(gemm((const float*)dh, hdim, hdim, W1T, hdim, n, c, (float*)dx, c));
    return dgnn_check_launch("updated_tail_bwd");
}

// Which of the training step's fused launch chains run (bit 0: backward chain, bit 1: batch statistics from the forward GEMM's epilogue;
// default 3, DGNN_TRAIN_FUSED in the environment).  Returns the previous mask.
extern "C" int dgnn_train_set_fused(int mask) {
    const int was = fused_mask();
    int v = mask & 3;
    if (getenv("DGNN_AGG_CHUNKED") && getenv("DGNN_AGG_CHUNKED")[0] == '0') v &= ~1;
    __atomic_store_n(&g_fused_on, v, __ATOMIC_RELEASE);
    return was;
}

// Whether the composite backward entry points run the weight gradients on the library's second stream (default: no).
extern "C" int dgnn_train_set_aux_stream(int on) {
    const int was = aux_enabled() ? 1 : 0;
    __atomic_store_n(&g_aux_on, on ? 1 : 0, __ATOMIC_RELEASE);
    return was;
}


// =====================================================================================================================
// Static conv layer in training mode with bf16 STORAGE (activations bf16, parameters / statistics / gradients of parameters fp32):
// the launch chain of dgnn_sage_layer_train_fwd / _bwd over the *_bf16 entry points.  Scratch sizes as for the fp32 functions.
// =====================================================================================================================
extern "C" int dgnn_sage_layer_train_fwd_bf16(const int32_t* rowptr, const int32_t* src, const int32_t* eid, int64_t n_dst, const uint16_t* x,
                                              int64_t ldx, int c_in, const float* edge_attr, int64_t lde, int f_e, const float* We,
                                              const float* be, const float* Wj, const float* bj, const float* Wi, int c_out,
                                              const float* gamma, const float* beta, float* running_mean, float* running_var, float momentum,
                                              float eps, int relu, uint16_t* a, uint16_t* z, float* mean, float* var, float* scale, float* shift,
                                              uint16_t* y, float* scratch, void* stream) {
    DGNN_REQUIRE(n_dst > 0 && c_in > 0 && c_out > 0, DGNN_E_INVALID, "sage_layer_train_fwd_bf16: bad sizes (BatchNorm needs at least one row)");
    DGNN_REQUIRE(x && Wj && z && mean && var && scale && shift && y && scratch, DGNN_E_INVALID, "sage_layer_train_fwd_bf16: null pointer");
    const uint16_t* A1 = x;
    int64_t lda1 = ldx;
    if (rowptr) {
        DGNN_REQUIRE(a && src, DGNN_E_INVALID, "sage_layer_train_fwd_bf16: the aggregate needs src and a");
        TRY(dgnn_sage_aggregate_fwd_bf16(rowptr, src, eid, n_dst, x, ldx, c_in, edge_attr, lde, f_e, We, be, nullptr, 0, nullptr, 0, a, c_in, stream));
        A1 = a;
        lda1 = c_in;
    }
    const uint16_t* A2 = (rowptr && Wi) ? x : nullptr;
    TRY(dgnn_linear_fwd_bf16(A1, lda1, c_in, Wj, c_in, A2, ldx, A2 ? c_in : 0, A2 ? Wi : nullptr, c_in, bj, nullptr, nullptr, 0, n_dst, c_out, z, c_out, 0,
                             stream));
    TRY(dgnn_bn_batch_stats_bf16(z, c_out, n_dst, c_out, mean, var, running_mean, running_var, momentum, scratch, stream));
    TRY(dgnn_bn_fold(gamma, beta, mean, var, eps, c_out, scale, shift, stream));
    TRY(dgnn_scale_shift_act_bf16(z, c_out, scale, shift, relu, n_dst, c_out, y, c_out, stream));
    return DGNN_OK;
}

// dz / da: work buffers [n_dst, c_out] / [n_dst, c_in] of bf16; scratch (floats): dgnn_sage_layer_train_scratch_elems
extern "C" int dgnn_sage_layer_train_bwd_bf16(const int32_t* t_rowptr, const int32_t* t_dst, const int32_t* t_eid, const int32_t* rowptr_dst,
                                              int64_t n_src, int64_t n_dst, const uint16_t* x, int64_t ldx, int c_in, const float* edge_attr,
                                              int64_t lde, int f_e, const float* We, const float* be, const float* Wj, const float* Wi, int c_out,
                                              const float* gamma, const float* mean, const float* var, float eps, int relu, const uint16_t* a,
                                              const uint16_t* z, const uint16_t* y, const uint16_t* dy, uint16_t* dx, float* dWe, float* dbe,
                                              float* dWj, float* dbj, float* dWi, float* dgamma, float* dbeta, uint16_t* dz, uint16_t* da,
                                              float* scratch, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DGNN_REQUIRE(n_dst > 0 && n_src >= n_dst && c_in > 0 && c_out > 0, DGNN_E_INVALID, "sage_layer_train_bwd_bf16: bad sizes");
    DGNN_REQUIRE(x && Wj && z && y && dy && mean && var && dWj && dgamma && dbeta && dz && scratch, DGNN_E_INVALID, "sage_layer_train_bwd_bf16: null pointer");
    const bool agg = t_rowptr != nullptr;
    float* WjT = scratch;
    float* WiT = WjT + align4((int64_t)c_in * c_out);
    float* tmp = WiT + align4((int64_t)c_in * c_out);
    TRY(dgnn_bn_relu_bwd_bf16(z, c_out, y, c_out, dy, c_out, gamma, mean, var, eps, 1, relu, n_dst, c_out, dz, c_out, dgamma, dbeta, tmp, stream_));
    const uint16_t* A1 = agg ? a : x;
    const int64_t lda1 = agg ? c_in : ldx;
    TRY(dgnn_linear_wgrad_bf16(dz, 0, c_out, c_out, A1, 0, lda1, c_in, n_dst, dWj, c_in, 0, tmp, stream_));
    if (dbj) TRY(dgnn_colsum_bf16(dz, c_out, n_dst, c_out, dbj, 0, tmp, stream_));
    if (agg && Wi && dWi) TRY(dgnn_linear_wgrad_bf16(dz, 0, c_out, c_out, x, 0, ldx, c_in, n_dst, dWi, c_in, 0, tmp, stream_));
    const bool need_dx = dx != nullptr;
    const bool need_da = agg ? (need_dx || We != nullptr) : need_dx;
    const bool both = need_da && agg && need_dx && Wi;
    if (both)
        hipLaunchKernelGGL(k_transpose2, dim3(dgnn_grid_cap(dgnn_cdiv((int64_t)2 * c_in * c_out, 256))), dim3(256), 0, stream, Wj, Wi, c_out, c_in, WjT, WiT);
    if (need_da) {
        DGNN_REQUIRE(!agg || da, DGNN_E_INVALID, "sage_layer_train_bwd_bf16: da buffer missing");
        if (!both) hipLaunchKernelGGL(k_transpose, dim3(dgnn_grid_cap(dgnn_cdiv((int64_t)c_in * c_out, 256))), dim3(256), 0, stream, Wj, c_out, c_in, WjT);
        TRY(dgnn_linear_fwd_bf16(dz, c_out, c_out, WjT, c_out, nullptr, 0, 0, nullptr, 0, nullptr, nullptr, nullptr, 0, n_dst, c_in, agg ? da : dx, c_in, 0, stream_));
    }
    if (agg) {
        if (We) DGNN_REQUIRE(dWe && dbe, DGNN_E_INVALID, "sage_layer_train_bwd_bf16: dWe / dbe missing");
        if (need_da)
            TRY(dgnn_sage_aggregate_bwd_bf16(t_rowptr, t_dst, t_eid, n_src, rowptr_dst, x, ldx, c_in, edge_attr, lde, f_e, We, be, nullptr, 0, da, c_in, dx, c_in,
                                             dWe, dbe, nullptr, 0, tmp, stream_));
        if (need_dx && Wi) {
            if (!both) hipLaunchKernelGGL(k_transpose, dim3(dgnn_grid_cap(dgnn_cdiv((int64_t)c_in * c_out, 256))), dim3(256), 0, stream, Wi, c_out, c_in, WiT);
            TRY(dgnn_linear_fwd_bf16(dz, c_out, c_out, WiT, c_out, nullptr, 0, 0, nullptr, 0, nullptr, nullptr, nullptr, DGNN_LINEAR_ACCUMULATE, n_dst, c_in, dx, c_in, 0,
                                     stream_));
        }
    }
    return dgnn_check_launch("sage_layer_train_bwd_bf16");
}
