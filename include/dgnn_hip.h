/*
 * dgnn_hip.h -- C ABI of libdgnn_hip.so: the MI355X (gfx950) kernels behind the hot path of
 * raphaelsulzer/dgnn (edge-filtered GraphSAGE over the tetrahedron-adjacency graph).
 *
 * The reference has no FFI / operator registry: its seam is the Python module API
 * (learning/surfaceNetStaticEdgeFilters.py, learning/surfaceNetUpdatedEdgeFilters.py).  The Python
 * mirror in dgnn_amd/ keeps that API and binds these entry points with ctypes;
 * each entry point below cites the reference lines (relative to the reference root) it replaces.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer (e.g. torch.Tensor.data_ptr()); sizes are element counts;
 *    `ld*` are row strides in elements; all float data is fp32 row-major, all indices int32 unless
 *    stated (the reference's edge_index is int64 [2,E] and is consumed as such by the plan builder);
 *  - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream);
 *  - nothing here allocates, frees or synchronises: the caller owns every buffer, including the
 *    plan arrays and scratch; calls are asynchronous on `stream`;
 *  - return value: 0 = ok, negative = DGNN_E_*; dgnn_last_error_string() describes the last
 *    failure on the calling thread.  Functions never throw.
 *  - thread safety: no global mutable state except the thread-local error string; calls on
 *    distinct streams may run concurrently.
 */
#ifndef DGNN_HIP_H
#define DGNN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DGNN_VERSION 100 /* major*10000 + minor*100 + patch */

#define DGNN_OK 0
#define DGNN_E_INVALID -1     /* bad argument (null pointer, negative size, unsupported width) */
#define DGNN_E_UNSUPPORTED -2 /* shape outside what the kernel family handles */
#define DGNN_E_LAUNCH -3      /* hipGetLastError() after launch */
#define DGNN_E_INDEX -4       /* a kernel met an index outside its range (reported asynchronously, see dgnn_poll_async_error) */

int dgnn_version(void);
const char* dgnn_last_error_string(void);
/* Data-dependent errors a launch-time check cannot see (an edge_index entry outside [0, n): torch's scatter raises an index
 * error there, surfaceNetStaticEdgeFilters.py:80) are reported asynchronously, like HIP's own: the kernel stays memory-safe
 * (skips the edge), sets a bit in a pinned host word, and this call -- after any later stream synchronisation, or simply at
 * the next entry point -- returns DGNN_E_INDEX once and clears it (0 = nothing pending). */
int dgnn_poll_async_error(void);

/* ------------------------------------------------------------------------------------------------
 * Graph plan: destination-sorted CSR of an edge list, STABLE in edge position.
 *
 * Replaces the implicit per-call work of PyG `MessagePassing.propagate` + torch_scatter
 * `scatter(reduce='mean')` (call site surfaceNetStaticEdgeFilters.py:80): there the aggregation
 * index is edge_index[1] and CPU accumulation runs in ascending edge position per destination.  The
 * plan makes that order explicit: for destination i, sorted edges rowptr[i]..rowptr[i+1]-1 are its
 * in-edges in ascending original position; src[k] = edge_index[0][eid[k]].
 *
 * `by` selects the sort key row of edge_index: 1 = by destination (forward), 0 = by source
 * (the transposed plan used by the backward pass; then `src` receives the OTHER endpoint,
 * i.e. the destination, and n_key = number of sources).
 *
 *   edge_index : int64 [2,E] view, row0=src row1=dst: element (r,k) at edge_index[r*stride_row + k*stride_col].
 *                The reference passes torch.transpose(adjacencies,1,0) (processing/data.py:434-438), i.e.
 *                strides (1,2) over the [E,2] array; a contiguous [2,E] tensor has (E,1).  Read in place.
 *   n_other    : number of nodes on the OTHER side (sources for by=1), for range checking; 0 = unknown, unchecked.
 *                Endpoints outside [0,n_key) / [0,n_other) never index memory: such edges are left out of the plan (key)
 *                or point at node 0 (other) and DGNN_E_INDEX is reported through dgnn_poll_async_error().
 *   rowptr     : int32 [n_key+1]   out
 *   other      : int32 [E]         out  (src for by=1, dst for by=0)
 *   eid        : int32 [E]         out  original edge position of the k-th sorted edge
 *   scratch    : int32 [dgnn_plan_scratch_elems(E,n_key)]
 * Two verified single-pass builders run before the generic count/scan/fill/sort kernels:
 *   GROUPED   - the edge list is already ascending in the key row (k-hop sampled blocks, a partition's local
 *               list, the by-source plan of the reference layout): the sort is the identity;
 *   REFERENCE - by destination, E == 4*n_key, row 4t+r = r-th neighbour of cell t, symmetric relation (what
 *               processing/data.py hands the model): in-edges are the reversed out-edges, no atomics/scan/sort.
 * `hint` only selects which of them is attempted (AUTO: both).  Each checks its precondition on the device;
 * when it does not hold, the generic builder queued behind it -- ONE persistent launch that returns at once when a fast path
 * produced the plan -- rebuilds it.  Same result in every case.  (REFERENCE with the (src, dst) pairs interleaved in memory,
 * stride_row 1 / stride_col 2 and a 16-byte aligned base: four lanes per cell, 16-byte loads.)
 * ---------------------------------------------------------------------------------------------- */
#define DGNN_PLAN_HINT_AUTO 0
#define DGNN_PLAN_HINT_GROUPED 1
#define DGNN_PLAN_HINT_REFERENCE 2
#define DGNN_PLAN_HINT_GENERIC 3 /* attempt neither: straight to the generic kernels */
#define DGNN_PLAN_HINT_GROUPED_TRUSTED 4 /* GROUPED without the device-side check and without the generic builder standing by: ONE launch.  For lists
                                           * the caller laid out itself (a ring part); a list that is not ascending in range gives a wrong plan (never an
                                           * out-of-bounds write) */
int64_t dgnn_plan_scratch_elems(int64_t E, int64_t n_key);
int dgnn_plan_build(const int64_t* edge_index, int64_t stride_row, int64_t stride_col, int64_t E, int64_t n_key,
                    int64_t n_other, int by, int hint, int32_t* rowptr, int32_t* other, int32_t* eid, int32_t* scratch,
                    void* stream);

/* out[k, 0:cols] = in[idx[k], 0:cols]  -- stages edge_attr rows into plan order once per scene
 * (the reference gathers them implicitly: `xe[e_id]` at surfaceNetStaticEdgeFilters.py:262,304). */
int dgnn_gather_rows_f32(const float* in, int64_t ld_in, const int32_t* idx, int64_t n, int cols, float* out,
                         int64_t ld_out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Edge-filtered mean aggregation (forward).  Replaces surfaceNetStaticEdgeFilters.py:75-80 + :89-96
 * (lin_e -> index_select -> x_j * phi -> scatter mean) and surfaceNetUpdatedEdgeFilters.py:156-158.
 *
 *   a[i,:] = ( sum_{k in rowptr[i]..rowptr[i+1]} x_src[src[k],:] * phi_k ) / max(deg_i,1)
 *   phi_k  = We . edge_attr[e,:] + be          (mode "fused": We != NULL, F_e <= 32)
 *          = phi[e,:]                           (mode "given": We == NULL, phi != NULL)
 *          = 1                                  (both NULL: plain mean, lin_e is None at :77-78)
 *   e      = eid ? eid[k] : k                   (eid == NULL: edge rows already in plan order)
 *
 * The sum runs in plan order with separately rounded multiply and add, i.e. the CPU order of
 * torch_scatter.  phi_out (optional, fused mode) receives phi rows at index e: the Updated variant
 * returns them (surfaceNetUpdatedEdgeFilters.py:170).
 * ---------------------------------------------------------------------------------------------- */
int dgnn_sage_aggregate_fwd(const int32_t* rowptr, const int32_t* src, const int32_t* eid, int64_t n_dst,
                            const float* x_src, int64_t ldx, int c_in, const float* edge_attr, int64_t lde, int f_e,
                            const float* We, const float* be, const float* phi, int64_t ldphi, float* phi_out,
                            int64_t ldphi_out, float* a, int64_t lda, void* stream);

/* ------------------------------------------------------------------------------------------------
 * out = act( (A1 . W1^T + A2 . W2^T + bias) * scale + shift )      [M, n_out]
 *
 * Replaces lin_j(out) + lin_i(x_r) (surfaceNetStaticEdgeFilters.py:81-86; lin_l/lin_r at
 * surfaceNetUpdatedEdgeFilters.py:159-165), eval-mode BatchNorm folded to scale/shift (:218) and
 * ReLU (:219); also the decoder Linears (:180-187) and lin_e of the Updated variant when its input
 * is wide.  W1 [n_out,K1], W2 [n_out,K2] are torch.nn.Linear weights (row-major).  A2/W2, bias,
 * scale/shift may be NULL.  `relu`: bit 0 applies max(0,.); bit 1 (DGNN_LINEAR_ACCUMULATE) adds the result to what `out`
 * holds (dx_dst += G . Wi in the backward pass).  fp32 MFMA (v_mfma_f32_32x32x2_f32).
 * ---------------------------------------------------------------------------------------------- */
#define DGNN_LINEAR_ACCUMULATE 2
int dgnn_linear_fwd(const float* A1, int64_t lda1, int k1, const float* W1, int64_t ldw1, const float* A2,
                    int64_t lda2, int k2, const float* W2, int64_t ldw2, const float* bias, const float* scale,
                    const float* shift, int relu, int64_t M, int n_out, float* out, int64_t ldo, void* stream);

/* dgnn_linear_fwd / dgnn_linear_wgrad with the fused layers' arithmetic (DGNN_GEMM_BF16X3): both fp32 operands are split exactly
 * into 3 bf16 parts while staged, 6 partial products on v_mfma_f32_32x32x16_bf16, fp32 accumulate -- fp32-class accuracy
 * (dropped terms <= 2^-25 relative) at 6/16 of the fp32-MFMA time.  Same arguments and scratch sizes. */
int dgnn_linear_fwd_x3(const float* A1, int64_t lda1, int k1, const float* W1, int64_t ldw1, const float* A2, int64_t lda2, int k2,
                       const float* W2, int64_t ldw2, const float* bias, const float* scale, const float* shift, int relu, int64_t M,
                       int n_out, float* out, int64_t ldo, void* stream);
/* The same product in the fp16 form of DGNN_GEMM_F16X2 (below): every row of [A1 | A2] and of [W1 | W2] scaled by its own power of two,
 * split into 2 fp16 parts, 3 products per fp32 product on v_mfma_f32_32x32x16_f16, fp32 accumulate (dropped terms <= 2^-22 relative).
 * Covers M >= 8192 with n_out > 128 (the wide conv layers); other shapes return DGNN_E_UNSUPPORTED -- use dgnn_linear_fwd_x3.
 * scratch: dgnn_linear_fwd_x2h_scratch_elems(M, n_out) floats (the row scales, written by a first pass over the operands). */
int64_t dgnn_linear_fwd_x2h_scratch_elems(int64_t M, int n_out);
int dgnn_linear_fwd_x2h(const float* A1, int64_t lda1, int k1, const float* W1, int64_t ldw1, const float* A2, int64_t lda2, int k2,
                        const float* W2, int64_t ldw2, const float* bias, const float* scale, const float* shift, int relu,
                        int64_t M, int n_out, float* out, int64_t ldo, float* scratch, void* stream);
/* dgnn_linear_fwd_x2h with both operands split ONCE by a pass of their own (row scale, (hi, lo) fp16 in the GEMM's staging format) and staged
 * global -> LDS by DMA: same arguments, results bit-identical to dgnn_linear_fwd_x2h, no split arithmetic between the matrix instructions.
 * scratch: dgnn_linear_fwd_x2hp_scratch_elems(M, n_out, k1, k2) floats, 16-byte aligned (row scales + the pre-split operands, 4 bytes per
 * element of [A1 | A2] and [W1 | W2] with k1, k2 rounded up to multiples of 32). */
int64_t dgnn_linear_fwd_x2hp_scratch_elems(int64_t M, int n_out, int k1, int k2);
int dgnn_linear_fwd_x2hp(const float* A1, int64_t lda1, int k1, const float* W1, int64_t ldw1, const float* A2, int64_t lda2, int k2,
                         const float* W2, int64_t ldw2, const float* bias, const float* scale, const float* shift, int relu,
                         int64_t M, int n_out, float* out, int64_t ldo, float* scratch, void* stream);

int dgnn_linear_wgrad_x3(const float* A, int64_t lda, int n_a, const float* B, int64_t ldb, int n_b, int64_t M, float* dW, int64_t lddw,
                         int accumulate, float* partials, void* stream);

/* Training forward of `lin_j(a) + lin_i(x_dst)` with the BatchNorm that follows (:217-218): dgnn_linear_fwd_x3 (no scale / shift / relu) whose
 * epilogue also leaves the fp64 column sums and sums of squares of `out` per block of 32 rows in colstats[ceil(M / 32)][2][n_out]
 * (8-byte aligned; dgnn_colstats_scratch_elems floats hold it); dgnn_bn_stats_finalize_fold turns them into mean / var / running
 * statistics / folded scale and shift like dgnn_bn_batch_stats_fold does from its own partial sums.  DGNN_E_UNSUPPORTED (nothing
 * launched) where the GEMM takes its 256 x 256 tile (n_out > 128 and at least 192 such tiles). */
int dgnn_linear_fwd_x3_stats(const float* A1, int64_t lda1, int k1, const float* W1, int64_t ldw1, const float* A2, int64_t lda2, int k2,
                             const float* W2, int64_t ldw2, const float* bias, int64_t M, int n_out, float* out, int64_t ldo,
                             double* colstats, void* stream);
int dgnn_bn_stats_finalize_fold(const double* colstats, int64_t nblk, int64_t M, int c, float* mean, float* var, float* running_mean,
                                float* running_var, float momentum, const float* gamma, const float* beta, float eps, float* scale,
                                float* shift, void* stream);

/* Both weight gradients of a conv layer and its bias gradient in one launch pair (autograd of :81-86 for lin_j, lin_i and lin_j.bias):
 * dW1[n_a, n_b1] = A^T . B1, dW2[n_a, n_b2] = A^T . B2 (B2 NULL / n_b2 0: none), dbias[n_a] = column sums of A (NULL: none), all
 * WRITTEN, contiguous.  dW1 / dW2 are bit-identical to two dgnn_linear_wgrad_x3 calls (same row splits, same products); dbias is summed
 * in fp64 like dgnn_colsum, in another order.  scratch: dgnn_linear_wgrad_cat_scratch_elems floats. */
int64_t dgnn_linear_wgrad_cat_scratch_elems(int64_t M, int n_a, int n_b1, int n_b2);
int dgnn_linear_wgrad_x3_cat(const float* A, int64_t lda, int n_a, const float* B1, int64_t ldb1, int n_b1, const float* B2, int64_t ldb2,
                             int n_b2, int64_t M, float* dW1, float* dW2, float* dbias, float* scratch, void* stream);

/* dW[n_a, n_b] (+)= A^T . B over M rows (weight gradients of the Linears above: autograd of :81-86).
 * Deterministic two-stage reduction; `partials` holds dgnn_linear_wgrad_scratch_elems floats. */
int64_t dgnn_linear_wgrad_scratch_elems(int64_t M, int n_a, int n_b);
int dgnn_linear_wgrad(const float* A, int64_t lda, int n_a, const float* B, int64_t ldb, int n_b, int64_t M,
                      float* dW, int64_t lddw, int accumulate, float* partials, void* stream);

/* ------------------------------------------------------------------------------------------------
 * BatchNorm1d pieces (torch_geometric BatchNorm -> torch.nn.BatchNorm1d, eps 1e-5, momentum 0.1;
 * used at surfaceNetStaticEdgeFilters.py:165,173,185; applied :218,263,305,345).
 * ---------------------------------------------------------------------------------------------- */
/* eval fold: scale = gamma/sqrt(var+eps), shift = beta - mean*scale */
int dgnn_bn_fold(const float* gamma, const float* beta, const float* mean, const float* var, float eps, int c,
                 float* scale, float* shift, void* stream);
/* column sums and sums of squares (fp64 accumulation inside) -> mean[c], biased var[c]; when
 * running_* != NULL updates them with momentum and the UNBIASED variance, as BatchNorm1d.train(). */
int64_t dgnn_colstats_scratch_elems(int64_t M, int c);
int dgnn_bn_batch_stats(const float* x, int64_t ldx, int64_t M, int c, float* mean, float* var, float* running_mean,
                        float* running_var, float momentum, float* scratch, void* stream);
/* dgnn_bn_batch_stats followed by dgnn_bn_fold on its result, in the same two launches as dgnn_bn_batch_stats alone */
int dgnn_bn_batch_stats_fold(const float* x, int64_t ldx, int64_t M, int c, float* mean, float* var, float* running_mean,
                             float* running_var, float momentum, const float* gamma, const float* beta, float eps, float* scale,
                             float* shift, float* scratch, void* stream);
/* y = act(x*scale + shift) elementwise over [M,c] (train-mode BN apply + ReLU; in place allowed) */
int dgnn_scale_shift_act(const float* x, int64_t ldx, const float* scale, const float* shift, int relu, int64_t M,
                         int c, float* y, int64_t ldy, void* stream);
/* backward of y = relu(bn(x)):  given y, dy (and x_hat recomputed from x, mean, invstd) produce
 * dx, dgamma, dbeta.  train != 0 uses the batch-statistics formula, else dx = dy*mask*scale. */
int dgnn_bn_relu_bwd(const float* x, int64_t ldx, const float* y, int64_t ldy, const float* dy, int64_t lddy,
                     const float* gamma, const float* mean, const float* var, float eps, int train, int relu,
                     int64_t M, int c, float* dx, int64_t lddx, float* dgamma, float* dbeta, float* scratch,
                     void* stream);
/* out[c] (+)= sum over rows (bias gradients) */
/* dgnn_bn_relu_bwd in two halves, for batch statistics that span several ranks (a scene cut across GPUs; SURVEY 8e: the [2 C] all-reduce per layer):
 * _sums: sums[0..c) = sum g, sums[c..2c) = sum g * x_hat over the M local rows (g = dy behind the ReLU mask; also this rank's dbeta / dgamma terms);
 * _apply: dx from sums taken over `count` rows (all ranks').  fp32 rows; scratch as dgnn_bn_relu_bwd. */
int dgnn_bn_relu_bwd_sums(const float* x, int64_t ldx, const float* y, int64_t ldy, const float* dy, int64_t lddy, const float* mean,
                          const float* var, float eps, int relu, int64_t M, int c, float* sums, float* scratch, void* stream);
int dgnn_bn_relu_bwd_apply(const float* x, int64_t ldx, const float* y, int64_t ldy, const float* dy, int64_t lddy, const float* gamma,
                           const float* mean, const float* var, float eps, int relu, int64_t M, int c, const float* sums, double count,
                           float* dx, int64_t lddx, void* stream);
int dgnn_colsum(const float* x, int64_t ldx, int64_t M, int c, float* out, int accumulate, float* scratch, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Aggregation backward (autograd of :75-80,:89-96, invoked at learning/runModel.py:279).
 * Runs over the TRANSPOSED plan (edges grouped by source) so dx_src needs no atomics:
 *   dm_e   = da[dst_e,:] / max(deg_dst,1)
 *   dphi_e = dm_e * x_src[s,:]          dx_src[s,:] = sum_e dm_e * phi_e
 *   fused mode (We != NULL): phi recomputed; dWe = sum_e dphi_e (x) edge_attr[e]; dbe = sum_e dphi_e
 *   given mode (phi != NULL): dphi written to dphi_out[e,:]
 *   t_rowptr/t_dst/t_eid : transposed plan (dgnn_plan_build with by=0); deg_dst from rowptr_dst.
 * dWe [c_in,f_e], dbe [c_in] are WRITTEN (no zero fill needed), summed deterministically via `partials`.
 * ---------------------------------------------------------------------------------------------- */
int64_t dgnn_sage_aggregate_bwd_scratch_elems(int64_t n_src, int c_in, int f_e);
int dgnn_sage_aggregate_bwd(const int32_t* t_rowptr, const int32_t* t_dst, const int32_t* t_eid, int64_t n_src,
                            const int32_t* rowptr_dst, const float* x_src, int64_t ldx, int c_in,
                            const float* edge_attr, int64_t lde, int f_e, const float* We, const float* be,
                            const float* phi, int64_t ldphi, const float* da, int64_t ldda, float* dx_src,
                            int64_t lddx, float* dWe, float* dbe, float* dphi_out, int64_t lddphi, float* partials,
                            void* stream);

/* bf16-storage form of dgnn_linear_wgrad_x3_cat (A, B1, B2 bf16, or fp32 rounded to bf16 when staged: a_f32 / b_f32 = 1): one bf16 product per
 * element, dW1 / dW2 bit-identical to dgnn_linear_wgrad_bf16; same scratch size function. */
int dgnn_linear_wgrad_bf16_cat(const void* A, int a_f32, int64_t lda, int n_a, const void* B1, int64_t ldb1, int n_b1, const void* B2, int64_t ldb2,
                               int n_b2, int b_f32, int64_t M, float* dW1, float* dW2, float* dbias, float* scratch, void* stream);

/* Given-phi form of the aggregate backward (the Updated variant, surfaceNetUpdatedEdgeFilters.py:147-170) with the two additions of its conv
 * layer's backward folded into the stores: dx_src[row] += add[row] for row < n_add (add NULL: none) and dphi_out[e] = dphi_e + dphi_ext[e]
 * (dphi_ext NULL: none; :233-241: the next layer takes this layer's phi as its edge input and sends a gradient back to it).  bf16 = 1: bf16
 * storage of x, phi, da, dx, add, dphi. */
int dgnn_sage_aggregate_bwd_phi_add(const int32_t* t_rowptr, const int32_t* t_dst, const int32_t* t_eid, int64_t n_src, const int32_t* rowptr_dst,
                                    const void* x_src, int64_t ldx, int c_in, const void* phi, int64_t ldphi, const void* da, int64_t ldda,
                                    void* dx_src, int64_t lddx, const void* add, int64_t ldadd, int64_t n_add, void* dphi_out, int64_t lddphi,
                                    const void* dphi_ext, int bf16, void* stream);

/* dgnn_sage_aggregate_bwd in fused mode (f_e = 20, fp32) with `dx_src[row] += add[row]` for row < n_add folded into the store of dx_src:
 * the x_dst = x[:n_dst] branch of a conv layer (:217, lin_i) sends its gradient dz . Wi to the first n_dst rows of dx; the addend is
 * added to the finished sum in one fp32 addition, like an accumulating GEMM epilogue after the aggregate would. */
int dgnn_sage_aggregate_bwd_add(const int32_t* t_rowptr, const int32_t* t_dst, const int32_t* t_eid, int64_t n_src,
                                const int32_t* rowptr_dst, const float* x_src, int64_t ldx, int c_in, const float* edge_attr,
                                int64_t lde, int f_e, const float* We, const float* be, const float* da, int64_t ldda, float* dx_src,
                                int64_t lddx, const float* add, int64_t ldadd, int64_t n_add, float* dWe, float* dbe, float* partials,
                                void* stream);

/* ------------------------------------------------------------------------------------------------
 * Fused inference layer (the timed path, surfaceNetStaticEdgeFilters.py:343-346, one call per
 * layer): aggregate + lin_j + lin_i + BN(eval) + ReLU in one persistent launch; the aggregate
 * never leaves the CU.  Requires c_in <= 128, c_out in {64,128}, f_e == 20 with packed rows.
 * Edge row of plan position k = edge_attr[eid ? eid[k] : k]: pass the plan's eid to read the
 * caller's edge_attr in place (each 80-byte row is DMA-gathered into LDS; no staging copy), or
 * eid == NULL with rows already in plan order (dgnn_gather_rows_f32).  Returns DGNN_E_UNSUPPORTED
 * for other shapes (callers fall back to the aggregate + linear pair above).
 * x_dst: own rows of the n_dst destinations (row stride ldx), the second element of the reference's
 * (x_src, x_dst) pair (:66-72); NULL = x_src (x_dst is x_src[:n_dst] at every reference call site).
 * A sub-range [b, b+n) of the destinations is one call with rowptr+b, x_dst = x_src + b*ldx, out + b*ldo
 * (the partitioned forward runs interior and boundary cells as two such launches).  gemm_mode selects how the dense part runs on the matrix cores.
 * ---------------------------------------------------------------------------------------------- */
#define DGNN_GEMM_F32 0    /* v_mfma_f32_32x32x2_f32: bit-faithful fp32 fmaf chains */
#define DGNN_GEMM_BF16X3 1 /* operands split exactly into 3 bf16 parts, 6 partial products on v_mfma_f32_32x32x16_bf16,
                              fp32 accumulate: fp32-class accuracy (dropped terms <= 2^-25 relative) at 6/16 of the time */
#define DGNN_GEMM_BF16X3_FILTER 2 /* as BF16X3, and the 20-tap filter MLP runs on v_mfma_f32_16x16x32_bf16 too (same exact
                                     3-part split, 6 products, fp32 accumulate); falls back to BF16X3 for shapes it
                                     does not cover (c_in not a multiple of c_in_pad/16, unaligned wide rows) */
#define DGNN_GEMM_F16X2_DENSE 3 /* dense product in the fp16 two-part form: every [mean | own] row of a tet and the [Wj | Wi] pair are multiplied
                                  by a power of two (exact) that puts their largest magnitude into [2^14, 2^15), split into (hi, lo) fp16
                                  parts (22 significand bits), 3 products hi.hi + hi.lo + lo.hi on v_mfma_f32_32x32x16_f16 with fp32
                                  accumulation (lo.lo dropped: <= 2^-22 relative -- the "3xTF32" scheme, TF32 and fp16 both carry 11
                                  significant bits), inverse scales applied to the accumulator; filter MLP as BF16X3_FILTER */
#define DGNN_GEMM_F16X2 4       /* (library default of the Python host) as F16X2_DENSE, and the filter MLP on v_mfma_f32_16x16x32_f16 in the
                                  same form: one scale per EDGE (its 20 attributes and the constant 1 of the bias column) and one for
                                  [We | be].  Half the matrix instructions and a third of the split instructions of BF16X3_FILTER; measured
                                  error against fp64 at the level of the other modes, also with rows / edges / weights spread over 60
                                  binades (tests/test_gpu_parity.py).  Falls back like BF16X3_FILTER.  The training entry points and
                                  dgnn_linear_*_x3 treat both F16X2 values as BF16X3; dgnn_linear_fwd_x2h is the GEMM in this form */
int dgnn_sage_layer_fused_fwd(const int32_t* rowptr, const int32_t* src, const int32_t* eid, int64_t n_dst, const float* x_src,
                              const float* x_dst, int64_t ldx, int c_in, const float* edge_attr, int64_t lde, int f_e,
                              const float* We, const float* be, const float* Wj, const float* bj, const float* Wi,
                              const float* scale, const float* shift, int relu, int c_out, float* out, int64_t ldo,
                              int gemm_mode, void* stream);
/* 1: the 128 -> 128 layers in DGNN_GEMM_F16X2 (plain and decoder-carrying) run as the WAVE-SPECIALISED kernel (csrc/fused_ws.hip: workgroups of 8
 * producer wavefronts -- gather, filter MLP, mean, row split -- and 8 consumer wavefronts -- the dense product against register-resident weights, the
 * epilogue, the decoder -- around a ring of 32-tet tiles in LDS; same arithmetic form and error level as the two-phase kernel, not the same bits);
 * 0: the two-phase kernel of rounds 2-4 (environment DGNN_WS=0, read once per process).  What a benchmark line names its dominant kernel by.
 * The same kernel takes the 64 -> 128 layer (DGNN_WS_64=0: not) and, in the bf16-storage chain, the plain 64 -> 128 and 128 -> 128 layers on UNSIGNED
 * 16-bit rows (dgnn_sage_layer_fused_fwd_bf16 with DGNN_BF16_COMPENSATED | _ROWS_IN_UNSIGNED | _ROWS_OUT_UNSIGNED; DGNN_WS_16=0: not): rows decoded
 * exactly, the fp16 two-part arithmetic of the fp32-I/O kernel in between (tighter than the compensated bf16 products), rows encoded in the epilogue. */
int dgnn_wave_specialised_enabled(void);

/* The LAST conv layer + BatchNorm(eval) + ReLU of SurfaceNet.inference_layer (learning/surfaceNetStaticEdgeFilters.py:343-347) together with the
 * decoder Linear(c_out -> c_hidden) - BatchNorm(eval, folded into scale1 / shift1, NULL = none) - ReLU - Linear(c_hidden -> n_logits)
 * (:180-187, applied :350-351) in ONE launch: a finished tile of 32 tets stays in the compute unit, goes through the decoder there and
 * only logits [n_dst, n_logits] (row stride n_logits) are written -- 8 bytes per tet instead of 512 out and 512 back in.
 * Layer arguments as dgnn_sage_layer_fused_fwd.  Covers the shipped shape: 64 < c_in <= 128 (c_in % 8 == 0), c_out = 128, f_e = 20 (packed rows),
 * c_hidden = 64, n_logits = 2, rows 16-byte aligned; arithmetic = DGNN_GEMM_F16X2 throughout (the decoder's first product: one power-of-two
 * scale per tet row and 16-column slab, applied to the fp32 sum).  Any other shape: DGNN_E_UNSUPPORTED, nothing launched -- the caller then
 * runs dgnn_sage_layer_fused_fwd and dgnn_decoder_fused_fwd.  A tet's logits depend on its own inputs only (destination sub-ranges of a
 * partitioned scene give bit-identical results). */
int dgnn_sage_layer_fused_decoder_fwd(const int32_t* rowptr, const int32_t* src, const int32_t* eid, int64_t n_dst, const float* x_src,
                                      const float* x_dst, int64_t ldx, int c_in, const float* edge_attr, int64_t lde, int f_e,
                                      const float* We, const float* be, const float* Wj, const float* bj, const float* Wi,
                                      const float* scale, const float* shift, int relu, int c_out, const float* W0, const float* b0,
                                      const float* scale1, const float* shift1, int c_hidden, const float* W3, const float* b3, int n_logits,
                                      float* logits, void* stream);

/* Prepared parameters of the fused layers (fp16 two-part arithmetic, DGNN_GEMM_F16X2).  What a fused launch derives from the layer's parameters
 * before its first tile -- two power-of-two scales, the split filter operand [We | be], every wavefront's resident fragments of [Wj | Wi], the
 * decoder's W0 fragments and folded constants -- costs each of the 256 workgroups a column-wise read of the weights and their split: 15-20 us per
 * launch.  dgnn_sage_layer_prepare runs that prologue once and parks the result in `prepared` (dgnn_sage_layer_prepared_bytes(c_in, c_out,
 * with_decoder) bytes, 16-byte aligned; 0 = shape not covered); dgnn_sage_layer_fused_fwd_p / dgnn_sage_layer_fused_decoder_fwd_p are the two
 * entry points above reading it back (coalesced 16-byte loads).  The values are the same: results are bit-identical to the unprepared calls.
 * The buffer belongs to the parameter VALUES it was made from (W0 .. b3: the decoder's, NULL for a plain layer); prepare again when they change.
 * BatchNorm(eval) scale / shift and bj are still passed per call.  Shapes outside the all-matrix-core kernel: DGNN_E_UNSUPPORTED. */
int64_t dgnn_sage_layer_prepared_bytes(int c_in, int c_out, int with_decoder);
int dgnn_sage_layer_prepare(int c_in, int c_out, const float* We, const float* be, const float* Wj, const float* Wi, const float* W0,
                            const float* b0, const float* scale1, const float* shift1, const float* W3, const float* b3, void* prepared,
                            void* stream);
int dgnn_sage_layer_fused_fwd_p(const int32_t* rowptr, const int32_t* src, const int32_t* eid, int64_t n_dst, const float* x_src,
                                const float* x_dst, int64_t ldx, int c_in, const float* edge_attr, int64_t lde, int f_e, const float* We,
                                const float* be, const float* Wj, const float* bj, const float* Wi, const float* scale, const float* shift,
                                int relu, int c_out, float* out, int64_t ldo, const void* prepared, void* stream);
int dgnn_sage_layer_fused_decoder_fwd_p(const int32_t* rowptr, const int32_t* src, const int32_t* eid, int64_t n_dst, const float* x_src,
                                        const float* x_dst, int64_t ldx, int c_in, const float* edge_attr, int64_t lde, int f_e,
                                        const float* We, const float* be, const float* Wj, const float* bj, const float* Wi,
                                        const float* scale, const float* shift, int relu, int c_out, const float* W0, const float* b0,
                                        const float* scale1, const float* shift1, int c_hidden, const float* W3, const float* b3, int n_logits,
                                        float* logits, const void* prepared, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Whole-scene inference, ONE call (SurfaceNet.inference_layer, learning/surfaceNetStaticEdgeFilters.py:323-355: every conv -> norm -> relu
 * over the whole graph, then the decoder :350-351).
 *   plan     edge_index != NULL: the destination-sorted plan of the scene is built first (dgnn_plan_build with `plan_hint`) INTO the caller's
 *            rowptr [n+1] / src [E] / eid [E] / plan_scratch [dgnn_plan_scratch_elems(E, n)] -- they stay valid for later calls on the same
 *            graph; edge_index == NULL: rowptr / src / eid ARE the plan (eid NULL = edge_attr already in plan order).  edge_attr rows follow
 *            edge_index and are gathered through eid inside the layer launches.
 *   layers   n_layers fused conv layers (dgnn_sage_layer_fused_fwd[_p]; BatchNorm(eval) folded into scale[l] / shift[l], NULL = none; ReLU)
 *   decoder  Linear(widths[L] -> c_hidden) - BatchNorm(eval, scale1 / shift1) - ReLU - Linear(-> n_logits): fuse_decoder != 0 -- inside the last
 *            layer's launch where its shape allows (dgnn_sage_layer_fused_decoder_fwd[_p]; prepared[L-1], if given, must then have been made
 *            WITH the decoder); otherwise dgnn_decoder_fused_fwd behind a plain last layer (prepared[L-1] a plain layer's).  W0 == NULL: no
 *            decoder, `logits` receives the last layer's rows [n, widths[L]]
 * Per-layer arguments are HOST arrays [n_layers] of device pointers (widths [n_layers + 1]); prepared (may be NULL, entries may be NULL):
 * dgnn_sage_layer_prepare buffers, the last one made WITH the decoder.  Issues exactly the launches of the per-layer entry points, in their
 * order: bit-identical results; nothing allocates or synchronises.  Shapes outside the fused kernels (see dgnn_sage_layer_fused_fwd):
 * DGNN_E_UNSUPPORTED before anything is launched.  workspace: dgnn_static_infer_workspace_bytes(n, n_layers, widths) bytes, 16-byte aligned.
 * ---------------------------------------------------------------------------------------------- */
int64_t dgnn_static_infer_workspace_bytes(int64_t n, int n_layers, const int32_t* widths);
int dgnn_static_infer_fwd(const int64_t* edge_index, int64_t stride_row, int64_t stride_col, int64_t E, int plan_hint, int32_t* rowptr,
                          int32_t* src, int32_t* eid, int32_t* plan_scratch, int64_t n, const float* x, int64_t ldx, const float* edge_attr,
                          int64_t lde, int f_e, int n_layers, const int32_t* widths, const float* const* We, const float* const* be,
                          const float* const* Wj, const float* const* bj, const float* const* Wi, const float* const* scale,
                          const float* const* shift, const void* const* prepared, const float* W0, const float* b0, const float* scale1,
                          const float* shift1, int c_hidden, const float* W3, const float* b3, int n_logits, int fuse_decoder, int gemm_mode,
                          void* workspace, float* logits, void* stream);

/* ------------------------------------------------------------------------------------------------
 * One rank's part of a scene cut across GPUs, ONE call per step and NO data-path exchange.  Behind its owned cells the part holds the
 * rings of cells 1 .. L hops away (L = n_layers; their input rows are static and resident), and layer l is computed for the owned cells and the
 * rings up to L-1-l hops out -- the reference's own k-hop recomputation (learning/surfaceNetStaticEdgeFilters.py:232-275, one sampled batch at a
 * time) applied to a whole part.  Local cell order: owned cells, ring 1, ring 2, ...; the local edge list holds the in-edges of every cell that some
 * layer computes (owned + rings 1 .. L-1), sources < n_loc.
 *   n_dst [n_layers]  destinations of layer l = the first n_dst[l] local cells (non-increasing; n_dst[L-1] = the owned cells = rows of `logits`);
 *                     the plan (built when edge_index != NULL) covers n_dst[0] destinations, n_loc = all local cells incl. the outermost ring
 *   attr_in_plan_order != 0: edge_attr rows are grouped by destination already (a part's local list is): read in place, not through eid
 * Everything else as dgnn_static_infer_fwd (which is this call with every n_dst[l] = n).  On a reference-layout scene (4 in-edges per cell) a cell's
 * result does not depend on which other cells share its launch: the union of the ranks' logits is bit-identical to the whole scene's (a ragged graph:
 * equal to fp32 rounding).  workspace: dgnn_static_infer_workspace_bytes(n_dst[0], ...).
 * ---------------------------------------------------------------------------------------------- */
int dgnn_static_infer_rings_fwd(const int64_t* edge_index, int64_t stride_row, int64_t stride_col, int64_t E, int plan_hint, int32_t* rowptr,
                                int32_t* src, int32_t* eid, int32_t* plan_scratch, int attr_in_plan_order, int64_t n_loc, const int64_t* n_dst,
                                const float* x, int64_t ldx, const float* edge_attr, int64_t lde, int f_e, int n_layers, const int32_t* widths,
                                const float* const* We, const float* const* be, const float* const* Wj, const float* const* bj,
                                const float* const* Wi, const float* const* scale, const float* const* shift, const void* const* prepared,
                                const float* W0, const float* b0, const float* scale1, const float* shift1, int c_hidden, const float* W3,
                                const float* b3, int n_logits, int fuse_decoder, int gemm_mode, void* workspace, float* logits, void* stream);

/* The same in bf16 STORAGE (BASELINE config 3; whole scene: every n_dst[l] = n_loc): layer 0 reads the caller's fp32 rows in place (widths[0] <= 32) and
 * starts the 16-bit rows -- unsigned when `mode` carries DGNN_BF16_ROWS_OUT_UNSIGNED --, the middle layers keep the format, the last layer's launch
 * carries the decoder (dgnn_sage_layer_fused_decoder_fwd_bf16) and writes fp32 logits.  mode = DGNN_BF16_COMPENSATED [| DGNN_BF16_ROWS_OUT_UNSIGNED];
 * the decoder is required (128 -> 64 -> 2 behind a 128-wide last layer).  Anything else: DGNN_E_UNSUPPORTED, nothing launched.  Bit-identical to the
 * per-layer bf16 entry points.  workspace as dgnn_static_infer_rings_fwd. */
int dgnn_static_infer_rings_fwd_bf16(const int64_t* edge_index, int64_t stride_row, int64_t stride_col, int64_t E, int plan_hint, int32_t* rowptr,
                                     int32_t* src, int32_t* eid, int32_t* plan_scratch, int attr_in_plan_order, int64_t n_loc, const int64_t* n_dst,
                                     const float* x, int64_t ldx, const float* edge_attr, int64_t lde, int f_e, int n_layers, const int32_t* widths,
                                     const float* const* We, const float* const* be, const float* const* Wj, const float* const* bj,
                                     const float* const* Wi, const float* const* scale, const float* const* shift, const float* W0, const float* b0,
                                     const float* scale1, const float* shift1, int c_hidden, const float* W3, const float* b3, int n_logits, int mode,
                                     void* workspace, float* logits, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Training-mode conv layer, one call each way (SurfaceNet.forward :214-219 and its autograd, learning/runModel.py:279):
 *   forward : a = aggregate(x)  ->  z = a.Wj^T + x[:n_dst].Wi^T + bj  ->  BatchNorm1d with batch statistics (running buffers
 *             updated when given)  ->  y = relu(.)       rowptr == NULL: a plain Linear + BatchNorm (+ReLU) block, z = x.Wj^T + bj
 *   backward: dy -> dx (NULL: x is data), dWe, dbe, dWj, dbj (NULL: no bias), dWi, dgamma, dbeta
 * Both issue the launch chain of the separate entry points above, in the same order, on `stream` (results are bit-identical
 * to calling them one by one); nothing allocates or synchronises.  Buffers: a [n_dst,c_in], z / y [n_dst,c_out], mean / var /
 * scale / shift [c_out]; forward scratch = dgnn_colstats_scratch_elems(n_dst, c_out) floats, backward scratch =
 * dgnn_sage_layer_train_scratch_elems floats.  Transposed plan (t_*) and rowptr_dst as dgnn_sage_aggregate_bwd.
 * ---------------------------------------------------------------------------------------------- */
int64_t dgnn_sage_layer_train_scratch_elems(int64_t n_src, int64_t n_dst, int c_in, int c_out, int f_e);
int dgnn_sage_layer_train_fwd(const int32_t* rowptr, const int32_t* src, const int32_t* eid, int64_t n_dst, const float* x, int64_t ldx,
                              int c_in, const float* edge_attr, int64_t lde, int f_e, const float* We, const float* be, const float* Wj,
                              const float* bj, const float* Wi, int c_out, const float* gamma, const float* beta, float* running_mean,
                              float* running_var, float momentum, float eps, int relu, float* a, float* z, float* mean, float* var,
                              float* scale, float* shift, float* y, float* scratch, int gemm_mode, void* stream);
int dgnn_sage_layer_train_bwd(const int32_t* t_rowptr, const int32_t* t_dst, const int32_t* t_eid, const int32_t* rowptr_dst,
                              int64_t n_src, int64_t n_dst, const float* x, int64_t ldx, int c_in, const float* edge_attr, int64_t lde,
                              int f_e, const float* We, const float* be, const float* Wj, const float* Wi, int c_out, const float* gamma,
                              const float* mean, const float* var, float eps, int relu, const float* a, const float* z, const float* y,
                              const float* dy, float* dx, float* dWe, float* dbe, float* dWj, float* dbj, float* dWi, float* dgamma,
                              float* dbeta, float* scratch, int gemm_mode, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Edge-embedding chaining of the Updated variant (surfaceNetUpdatedEdgeFilters.py:233-241): the next layer's edge rows
 *   out[k, :c] = relu?( phi[r, :c] )  where e_id_cur[r] == e_id_next[k],   0 when this layer did not produce that edge
 * i.e. relu(zeros[E_all, C]; [e_id_cur] = phi)[e_id_next, :c] without materialising the [E_all, C] tensor.  `pos` is an
 * [n_edges] int32 table that must hold -1 everywhere on entry and does again on exit (only e_id_cur entries are touched);
 * `inv` [n_cur] receives the inverse map (row of `out` that read phi row r, or -1) for the backward pass:
 *   dphi[r, j] = (inv[r] >= 0 && j < c) ? g[inv[r], j] * [phi[r, j] > 0] : 0        (dphi [n_cur, c_tot], fully written)
 * Edge ids are unique within a block (k-hop blocks: every graph edge appears once).  Ids outside [0, n_edges) are skipped
 * and reported through dgnn_poll_async_error.
 * ---------------------------------------------------------------------------------------------- */
int dgnn_edge_chain_fwd(const float* phi, int64_t ldphi, int c, const int64_t* e_id_cur, int64_t n_cur, const int64_t* e_id_next,
                        int64_t n_next, int64_t n_edges, int32_t* pos, int relu, float* out, int64_t ldo, int32_t* inv, void* stream);
int dgnn_edge_chain_fwd_bf16(const uint16_t* phi, int64_t ldphi, int c, const int64_t* e_id_cur, int64_t n_cur, const int64_t* e_id_next,
                             int64_t n_next, int64_t n_edges, int32_t* pos, int relu, uint16_t* out, int64_t ldo, int32_t* inv, void* stream);
int dgnn_edge_chain_bwd(const float* g, int64_t ldg, const float* phi, int64_t ldphi, const int32_t* inv, int64_t n_cur, int c, int c_tot,
                        int relu, float* dphi, int64_t lddphi, void* stream);
int dgnn_edge_chain_bwd_bf16(const uint16_t* g, int64_t ldg, const uint16_t* phi, int64_t ldphi, const int32_t* inv, int64_t n_cur, int c,
                             int c_tot, int relu, uint16_t* dphi, int64_t lddphi, void* stream);

/* dgnn_sage_layer_train_fwd / _bwd with bf16 STORAGE: x / a / z / y / dy / dx (and the work buffers dz [n_dst,c_out], da [n_dst,c_in]) are bf16
 * bit patterns, parameters, statistics and parameter gradients fp32; the chain of the *_bf16 entry points.  Scratch (floats) as for the
 * fp32 functions. */
int dgnn_sage_layer_train_fwd_bf16(const int32_t* rowptr, const int32_t* src, const int32_t* eid, int64_t n_dst, const uint16_t* x, int64_t ldx,
                                   int c_in, const float* edge_attr, int64_t lde, int f_e, const float* We, const float* be, const float* Wj,
                                   const float* bj, const float* Wi, int c_out, const float* gamma, const float* beta, float* running_mean,
                                   float* running_var, float momentum, float eps, int relu, uint16_t* a, uint16_t* z, float* mean, float* var,
                                   float* scale, float* shift, uint16_t* y, float* scratch, void* stream);
int dgnn_sage_layer_train_bwd_bf16(const int32_t* t_rowptr, const int32_t* t_dst, const int32_t* t_eid, const int32_t* rowptr_dst, int64_t n_src,
                                   int64_t n_dst, const uint16_t* x, int64_t ldx, int c_in, const float* edge_attr, int64_t lde, int f_e,
                                   const float* We, const float* be, const float* Wj, const float* Wi, int c_out, const float* gamma,
                                   const float* mean, const float* var, float eps, int relu, const uint16_t* a, const uint16_t* z,
                                   const uint16_t* y, const uint16_t* dy, uint16_t* dx, float* dWe, float* dbe, float* dWj, float* dbj,
                                   float* dWi, float* dgamma, float* dbeta, uint16_t* dz, uint16_t* da, float* scratch, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Static model in training mode, ALL layers per call (SurfaceNet.forward :196-227 up to the decoder's last Linear, and its
 * autograd): the chain of dgnn_sage_layer_train_fwd / _bwd calls issued from one entry point each way.  Per-layer arguments are
 * HOST arrays [n_layers] of device pointers / sizes (widths [n_layers+1]); rowptr[l] == NULL marks a plain Linear + BatchNorm +
 * ReLU block (the decoder's).  Layer l reads y[l-1] (layer 0: x0); stats[l] = [4, widths[l+1]] (mean, var, scale, shift).
 * Forward scratch: max_l dgnn_colstats_scratch_elems(n_dst[l], widths[l+1]); backward scratch: dgnn_static_train_scratch_elems
 * (host arrays); dx_buf[0..1]: max_l n_src[l] * widths[l] floats each.  The backward runs the weight gradients (dWj, dbj, dWi: they
 * depend on dz only) on a second, library-owned stream beside the dx chain when dgnn_train_set_aux_stream(1) / DGNN_TRAIN_AUX_STREAM=1
 * is in effect, and makes `stream` wait for them before it returns; default: one stream.  Results are the same either way.
 * num_batches_tracked (may be NULL): device int64 counters of the BatchNorm modules, incremented once.
 * ---------------------------------------------------------------------------------------------- */
int dgnn_static_train_fwd(int n_layers, const int32_t* const* rowptr, const int32_t* const* src, const int32_t* const* eid, const int64_t* n_dst,
                          const float* x0, int64_t ldx0, const int32_t* widths, const float* const* edge_attr, const int64_t* lde, int f_e,
                          const float* const* We, const float* const* be, const float* const* Wj, const float* const* bj, const float* const* Wi,
                          const float* const* gamma, const float* const* beta, float* const* running_mean, float* const* running_var,
                          int64_t* const* num_batches_tracked, const float* momentum, const float* eps, float* const* a, float* const* z,
                          float* const* stats, float* const* y, float* scratch, int gemm_mode, void* stream);
/* Whether the composite backward entry points (dgnn_sage_layer_train_bwd, dgnn_sage_updated_train_bwd, dgnn_static_train_bwd) run the
 * weight gradients on the library's second stream (default 0 -- measured 2-8 % slower than one stream at the reference's block sizes;
 * environment DGNN_TRAIN_AUX_STREAM=1 starts with 1).  Returns the previous setting.  Results do not depend on it. */
int dgnn_train_set_aux_stream(int on);
/* The training step's fused launch chains: bit 0 = backward (dgnn_linear_wgrad_x3_cat, the stacked input-gradient GEMM with
 * dgnn_sage_aggregate_bwd_add, one transpose launch per pass), bit 1 = BatchNorm statistics from the forward GEMM's epilogue
 * (dgnn_linear_fwd_x3_stats).  Default 3 (environment: DGNN_TRAIN_FUSED=<mask>); 0 = the launch chain of the separate entry points.
 * Returns the previous mask. */
int dgnn_train_set_fused(int mask);

/* One Adam step over n_tensors fp32 parameter tensors in one launch (reference learning/runModel.py:290 torch.optim.Adam(model.parameters(), lr),
 * stepped at :282; no amsgrad / weight decay): p, g, m (exp_avg), v (exp_avg_sq) are host arrays of device pointers, numel the element counts,
 * step the 1-based step count (bias corrections are computed on the host in double precision). */
int dgnn_adam_step(int n_tensors, float* const* p, const float* const* g, float* const* m, float* const* v, const int64_t* numel, float lr,
                   float beta1, float beta2, float eps, int64_t step, void* stream);

/* All conv layers of the Updated variant (learning/surfaceNetUpdatedEdgeFilters.py:229-243) and the edge chaining between them per call: the
 * per-layer composite calls (dgnn_sage_updated_train_fwd / _bwd) and dgnn_edge_chain_fwd / _bwd issued back to back, results bit-identical to
 * calling them one by one.  Per layer l: plan (rowptr, src, eid), e_id[l] = the block edges' rows in the scene's edge tensor (int64; rows0 = e_id[0]
 * as int32), n_dst, E; widths[0 .. n_layers], edge_in[l] = columns of the layer's edge input (layer 0: of edge_attr_all, layer l: of phi_{l-1});
 * pos: the int32 [E_all] table of dgnn_edge_chain_fwd (all -1 between calls); relu[l]: the ReLU that follows conv l.  Saved for the backward:
 * ea[l] [E_l, ld_ea[l]], phi[l] [E_l, widths[l]], a[l] [n_dst_l, widths[l]], y[l] [n_dst_l, widths[l+1]], inv[l] [E_{l-1}] (l >= 1).
 * bf16 = 1: bf16 storage of x0 and of every saved tensor (ea0_f32: an [E_0, edge_in_0] fp32 work buffer for the cast).  Work buffers of the backward
 * (storage type): dx_buf[0 / 1] [max_l>=1 n_src_l * widths[l]], d_ea [max_l>=1 E_l * edge_in_l], dphi_ext and dphi [max_l E_l * widths[l]],
 * dz [max n_dst_l * widths[l+1]], da [max 2 * n_dst_l * widths[l]]; scratch: the largest dgnn_sage_updated_train_scratch_elems of the layers. */
int dgnn_updated_stack_fwd(int n_layers, const int32_t* const* rowptr, const int32_t* const* src, const int32_t* const* eid, const int64_t* const* e_id,
                           const int32_t* rows0, const int64_t* n_dst, const int64_t* E, const void* x0, int64_t ldx0, const int32_t* widths,
                           const int32_t* edge_in, const float* edge_attr_all, int64_t lde_all, int64_t E_all, int32_t* pos, const float* const* We,
                           const float* const* be, const float* const* Wl, const float* const* bl, const float* const* Wr, const int32_t* relu,
                           void* const* ea, const int64_t* ld_ea, float* ea0_f32, void* const* phi, void* const* a, void* const* y, int32_t* const* inv,
                           int bf16, int gemm_mode, void* stream);
int dgnn_updated_stack_bwd(int n_layers, const int32_t* const* t_rowptr, const int32_t* const* t_dst, const int32_t* const* t_eid,
                           const int32_t* const* rowptr_dst, const int64_t* n_src, const int64_t* n_dst, const int64_t* E, const void* x0, int64_t ldx0,
                           const int32_t* widths, const int32_t* edge_in, const float* const* We, const float* const* Wl, const float* const* Wr,
                           const int32_t* relu, const void* const* ea, const int64_t* ld_ea, const void* const* phi, const void* const* a,
                           const void* const* y, const int32_t* const* inv, const void* dy, float* const* dWe, float* const* dbe, float* const* dWl,
                           float* const* dbl, float* const* dWr, void* const* dx_buf, void* d_ea, void* dphi_ext, void* dz, void* da, void* dphi,
                           float* scratch, int bf16, int gemm_mode, void* stream);
/* The Updated model's output network behind the conv stack ("sage+": out_net, surfaceNetUpdatedEdgeFilters.py:210, 245-247), one call each way:
 * h = relu(x . W1^T + b1) in the storage type, logits = h . W3^T + b3 in fp32.  Backward from g = d logits: dW3 / db3 and dW1 / db1 (one launch
 * pair each), dx [n, c] in the storage type.  dh: an [n, hdim] work buffer (storage type); scratch: dgnn_updated_tail_scratch_elems floats. */
int dgnn_updated_tail_fwd(int64_t n, const void* x, int64_t ldx, int c, const float* W1, const float* b1, int hdim, const float* W3, const float* b3,
                          int n_out, void* h, float* logits, int bf16, int gemm_mode, void* stream);
int64_t dgnn_updated_tail_scratch_elems(int64_t n, int c, int hdim, int n_out);
int dgnn_updated_tail_bwd(int64_t n, const void* x, int64_t ldx, int c, const float* W1, int hdim, const float* W3, int n_out, const void* h,
                          const float* g, float* dW1, float* db1, float* dW3, float* db3, void* dx, void* dh, float* scratch, int bf16,
                          int gemm_mode, void* stream);

int64_t dgnn_static_train_scratch_elems(int n_layers, const int64_t* n_src, const int64_t* n_dst, const int32_t* widths, int f_e);
int dgnn_static_train_bwd(int n_layers, const int32_t* const* t_rowptr, const int32_t* const* t_dst, const int32_t* const* t_eid,
                          const int32_t* const* rowptr_dst, const int64_t* n_src, const int64_t* n_dst, const float* x0, int64_t ldx0,
                          const int32_t* widths, const float* const* edge_attr, const int64_t* lde, int f_e, const float* const* We,
                          const float* const* be, const float* const* Wj, const float* const* Wi, const float* const* gamma,
                          const float* const* stats, const float* eps, const float* const* a, const float* const* z, const float* const* y,
                          const float* dy, float* const* dWe, float* const* dbe, float* const* dWj, float* const* dbj, float* const* dWi,
                          float* const* dgamma, float* const* dbeta, float* const* dx_buf, float* scratch, int gemm_mode, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Updated variant, one conv layer per call each way (surfaceNetUpdatedEdgeFilters.py:147-170 and its autograd); `bf16` != 0:
 * x / ea / phi / a / y / dy and the gradients of activations are bf16 (uint16_t), parameters and their gradients fp32.
 *   forward : phi [E,c_in] = ea.We^T + be;  a [n_dst,c_in] = mean_j x_j * phi;  y [n_dst,c_out] = relu?(a.Wl^T + x[:n_dst].Wr^T + bl)
 *   backward: dy (+ dphi_ext [E,c_in], the gradient reaching phi through the next layer's edge input, or NULL) ->
 *             dx [n_src,c_in] (NULL: not needed), d_ea [E,k_e] (NULL: not needed), dWe, dbe, dWl, dbl (NULL: no bias), dWr;
 *             dz [n_dst,c_out] (relu only), da [n_dst,c_in], dphi [E,c_in] are work buffers of the activations' type,
 *             scratch = dgnn_sage_updated_train_scratch_elems floats.  Plans as dgnn_sage_aggregate_fwd / _bwd.
 * The launch chain of the separate entry points in the same order; nothing allocates or synchronises.
 * ---------------------------------------------------------------------------------------------- */
int64_t dgnn_sage_updated_train_scratch_elems(int64_t n_dst, int64_t E, int c_in, int c_out, int k_e);
int dgnn_sage_updated_train_fwd(const int32_t* rowptr, const int32_t* src, const int32_t* eid, int64_t n_dst, const void* x, int64_t ldx, int c_in,
                                const void* ea, int64_t lde, int k_e, int64_t E, const float* We, const float* be, const float* Wl,
                                const float* bl, const float* Wr, int c_out, int relu, void* phi, void* a, void* y, int bf16, int gemm_mode,
                                void* stream);
int dgnn_sage_updated_train_bwd(const int32_t* t_rowptr, const int32_t* t_dst, const int32_t* t_eid, const int32_t* rowptr_dst, int64_t n_src,
                                int64_t n_dst, int64_t E, const void* x, int64_t ldx, int c_in, const void* ea, int64_t lde, int k_e,
                                const float* We, const float* Wl, const float* Wr, int c_out, int relu, const void* phi, const void* a,
                                const void* y, const void* dy, const void* dphi_ext, void* dx, void* d_ea, float* dWe, float* dbe, float* dWl,
                                float* dbl, float* dWr, void* dz, void* da, void* dphi, float* scratch, int bf16, int gemm_mode, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Volume-weighted KL cell loss of the training step (learning/runModel.py:171-209), one launch each way:
 *   cell_k = sum_c kl_div(log_softmax(logits_k)_c, gt_kc);  w_k = vol_k | log(1+vol_k) | sqrt(vol_k)  (norm 0 | 1 | 2)
 *   loss = sum cell_k w_k / sum w_k;   sums[3] (fp64) = sum cell_k w_k, sum w_k, #{k: [gt_k0 > gt_k1] == argmax logits_k}
 *   backward: dlogits_kc = grad_loss * w_k / sums[1] * (softmax_kc (gt_k0 + gt_k1) - gt_kc)
 * logits / gt: two leading columns of rows with strides ldl / ldg; vol: element stride ldv.  scratch:
 * dgnn_kl_cell_loss_scratch_doubles(n) doubles.  grad_loss: device pointer to the upstream scalar.
 * ---------------------------------------------------------------------------------------------- */
int64_t dgnn_kl_cell_loss_scratch_doubles(int64_t n);
int dgnn_kl_cell_loss_fwd(const float* logits, int64_t ldl, const float* gt, int64_t ldg, const float* vol, int64_t ldv, int norm, int64_t n,
                          double* sums, float* loss, double* scratch, void* stream);
int dgnn_kl_cell_loss_bwd(const float* logits, int64_t ldl, const float* gt, int64_t ldg, const float* vol, int64_t ldv, int norm, int64_t n,
                          const double* sums, const float* grad_loss, float* dlogits, int64_t ldd, void* stream);
/* forward + (running += sums) + backward of one batch as ONE launch (round 6; the step around learning/runModel.py:171-209 and its gradient, with the
 * metric sums of :48-80 accumulated on the device): the same bits as dgnn_kl_cell_loss_fwd followed by dgnn_kl_cell_loss_bwd.  grad_loss NULL = 1;
 * running (fp64 [3]) and dlogits may be NULL.  DGNN_E_UNSUPPORTED (nothing launched) for n > 65536: use the two entry points above. */
int dgnn_kl_cell_loss_step(const float* logits, int64_t ldl, const float* gt, int64_t ldg, const float* vol, int64_t ldv, int norm, int64_t n,
                           const float* grad_loss, double* sums, float* loss, double* running, float* dlogits, int64_t ldd, void* stream);

/* Fused decoder, eval mode (reference :180-187 applied at :350-351):
 *   logits = W3 . relu((W0 . y + b0) * scale + shift) + b3,   y [M,k] -> out [M,n_out]
 * Supports k == 128, hidden == 64, n_out in {1,2}; DGNN_E_UNSUPPORTED otherwise (use dgnn_linear_fwd twice). */
int dgnn_decoder_fused_fwd(const float* y, int64_t ldy, int64_t M, int k, const float* W0, const float* b0,
                           const float* scale, const float* shift, int hidden, const float* W3, const float* b3, int n_out,
                           float* out, int64_t ldo, void* stream);

/* ------------------------------------------------------------------------------------------------
 * bf16 STORAGE path (BASELINE config 3): activations (and, for the Updated variant, the edge
 * embeddings phi) live in HBM as bf16 (uint16_t bit patterns here, torch.bfloat16 on the Python side), every matrix
 * product runs once on the bf16 matrix cores with fp32 accumulation, parameters stay fp32 (master weights, rounded to
 * bf16 when staged).  Stated tolerance: |dlogit| <= 5e-2 * max(1, |logit|/8), arg-max agreement >= 99.9 % vs fp32.
 * Same argument meaning as the fp32 entry points of the same name; rows must be aligned as each comment says.
 * ---------------------------------------------------------------------------------------------- */
#define DGNN_BF16_SINGLE 0      /* every operand rounded to bf16 once, one MFMA per product */
#define DGNN_BF16_COMPENSATED 1 /* only what is STORED is bf16: the fp32 mean, attributes and parameters enter the matrix cores as
                                   (hi, lo) bf16 pairs (16 bits; filter 3 products, a.Wj 3, x_i.Wi 2) -- the default */
/* Row-format flags OR-ed into `mode` of dgnn_sage_layer_fused_fwd_bf16 / dgnn_sage_layer_fused_decoder_fwd_bf16 (round 4, compensated arithmetic only):
 * UNSIGNED rows.  A row written behind a ReLU has no negative entries, so its 16 bits can hold the fp32 bits [30:15] -- bf16's 8 exponent bits and
 * EIGHT explicit mantissa bits (9 significant bits; value = bits << 15) -- instead of sign + 7: same bytes, half the storage rounding (2^-10 of a
 * value).  _OUT: this launch writes such rows (needs relu != 0); _IN: x_src / x_dst are such rows.  A layer on 16-bit rows keeps the format (IN and OUT
 * together); the first layer, reading fp32 rows, may start it (OUT only); the decoder-carrying launch takes IN.  With the three stored layers of the
 * shipped model unsigned: max |dlogit| on the 1M-tet graph 5.9e-2 -> 2.1e-2.  dgnn_rows_unsigned_to_bf16 converts to plain bf16. */
#define DGNN_BF16_ROWS_IN_UNSIGNED 16
#define DGNN_BF16_ROWS_OUT_UNSIGNED 32
int dgnn_rows_unsigned_to_bf16(const uint16_t* in, int64_t ld_in, int64_t n, int cols, uint16_t* out, int64_t ld_out, void* stream);
/* out[r, 0:cols] = bf16(in[r, 0:cols]), out[r, cols:cols_pad] = 0   (cols_pad even, ld_out >= cols_pad, ld_out even) */
int dgnn_cast_f32_to_bf16(const float* in, int64_t ld_in, int64_t n, int cols, int cols_pad, uint16_t* out, int64_t ld_out, void* stream);
int dgnn_cast_bf16_to_f32(const uint16_t* in, int64_t ld_in, int64_t n, int cols, float* out, int64_t ld_out, void* stream);
/* dgnn_sage_layer_fused_fwd with bf16 x_src / x_dst / out (edge_attr and all parameters fp32).  c_in <= 128, c_out in {64,128};
 * with nb = 2/4/8 for c_in <= 32/64/128: c_in % nb == 0, ldx % nb == 0, rows 2*nb-byte aligned; ldo even, out 4-byte aligned.
 * x_f32 != 0 (c_in <= 32, the first layer): x_src / x_dst are the caller's fp32 rows (any 4-byte aligned stride), read in
 * place -- the input features are never rounded to bf16; the output is bf16 as always. */
int dgnn_sage_layer_fused_fwd_bf16(const int32_t* rowptr, const int32_t* src, const int32_t* eid, int64_t n_dst, const void* x_src, int x_f32,
                                   const void* x_dst, int64_t ldx, int c_in, const float* edge_attr, int64_t lde, int f_e,
                                   const float* We, const float* be, const float* Wj, const float* bj, const float* Wi,
                                   const float* scale, const float* shift, int relu, int c_out, uint16_t* out, int64_t ldo, int mode,
                                   void* stream);
/* dgnn_decoder_fused_fwd on bf16 rows (16-byte aligned, ldy % 8 == 0); logits stay fp32 */
/* The LAST conv layer in bf16 storage with the decoder inside its launch (round 4; SurfaceNet.inference_layer :343-351 in the bf16 storage path):
 * dgnn_sage_layer_fused_fwd_bf16 followed by dgnn_decoder_fused_fwd_bf16 without the bf16 round trip of the layer's output -- the finished tile
 * reaches the decoder's matrix products as (hi, lo) bf16 pairs (16 significant bits; it is NEVER rounded to bf16: the storage rounding next to
 * the logits is gone) and only fp32 logits [n_dst, 2] are written.  Arguments as the two entry points.  Covers the shipped shape in the
 * compensated arithmetic: 64 < c_in <= 128 (c_in % 8 == 0), c_out = 128, f_e = 20 packed rows, decoder 128 -> 64 -> 2, bf16 rows 16-byte aligned
 * (ldx % 8 == 0), mode = DGNN_BF16_COMPENSATED; anything else: DGNN_E_UNSUPPORTED, nothing launched.  A tet's logits depend on its own inputs only
 * (destination sub-ranges give bit-identical results). */
int dgnn_sage_layer_fused_decoder_fwd_bf16(const int32_t* rowptr, const int32_t* src, const int32_t* eid, int64_t n_dst, const uint16_t* x_src,
                                           const uint16_t* x_dst, int64_t ldx, int c_in, const float* edge_attr, int64_t lde, int f_e,
                                           const float* We, const float* be, const float* Wj, const float* bj, const float* Wi,
                                           const float* scale, const float* shift, int relu, int c_out, const float* W0, const float* b0,
                                           const float* scale1, const float* shift1, int c_hidden, const float* W3, const float* b3, int n_logits,
                                           float* logits, int mode, void* stream);
int dgnn_decoder_fused_fwd_bf16(const uint16_t* y, int64_t ldy, int64_t M, int k, const float* W0, const float* b0, const float* scale,
                                const float* shift, int hidden, const float* W3, const float* b3, int n_out, float* out, int64_t ldo,
                                int mode, void* stream);

/* bf16-storage twins of the generic (training / any-width) entry points above: x / phi / a / gradients of activations are
 * bf16, parameters and their gradients fp32, all arithmetic fp32 between a widening load and a rounding store; the GEMMs run
 * on v_mfma_f32_32x32x16_bf16.  dgnn_linear_fwd_bf16: out is bf16 (out_f32 == 0) or fp32 (logits).  dgnn_linear_wgrad_bf16:
 * A / B are bf16 (x_f32 == 0) or fp32 (rounded to bf16 when staged); scratch sizes as for the fp32 functions. */
int dgnn_sage_aggregate_fwd_bf16(const int32_t* rowptr, const int32_t* src, const int32_t* eid, int64_t n_dst, const uint16_t* x_src,
                                 int64_t ldx, int c_in, const float* edge_attr, int64_t lde, int f_e, const float* We, const float* be,
                                 const uint16_t* phi, int64_t ldphi, uint16_t* phi_out, int64_t ldphi_out, uint16_t* a, int64_t lda,
                                 void* stream);
int dgnn_sage_aggregate_bwd_bf16(const int32_t* t_rowptr, const int32_t* t_dst, const int32_t* t_eid, int64_t n_src,
                                 const int32_t* rowptr_dst, const uint16_t* x_src, int64_t ldx, int c_in, const float* edge_attr,
                                 int64_t lde, int f_e, const float* We, const float* be, const uint16_t* phi, int64_t ldphi,
                                 const uint16_t* da, int64_t ldda, uint16_t* dx_src, int64_t lddx, float* dWe, float* dbe,
                                 uint16_t* dphi_out, int64_t lddphi, float* partials, void* stream);
int dgnn_linear_fwd_bf16(const uint16_t* A1, int64_t lda1, int k1, const float* W1, int64_t ldw1, const uint16_t* A2, int64_t lda2, int k2,
                         const float* W2, int64_t ldw2, const float* bias, const float* scale, const float* shift, int relu, int64_t M,
                         int n_out, void* out, int64_t ldo, int out_f32, void* stream);
int dgnn_linear_wgrad_bf16(const void* A, int a_f32, int64_t lda, int n_a, const void* B, int b_f32, int64_t ldb, int n_b, int64_t M,
                           float* dW, int64_t lddw, int accumulate, float* partials, void* stream);
int dgnn_bn_batch_stats_bf16(const uint16_t* x, int64_t ldx, int64_t M, int c, float* mean, float* var, float* running_mean,
                             float* running_var, float momentum, float* scratch, void* stream);
int dgnn_scale_shift_act_bf16(const uint16_t* x, int64_t ldx, const float* scale, const float* shift, int relu, int64_t M, int c,
                              uint16_t* y, int64_t ldy, void* stream);
int dgnn_bn_relu_bwd_bf16(const uint16_t* x, int64_t ldx, const uint16_t* y, int64_t ldy, const uint16_t* dy, int64_t lddy,
                          const float* gamma, const float* mean, const float* var, float eps, int train, int relu, int64_t M, int c,
                          uint16_t* dx, int64_t lddx, float* dgamma, float* dbeta, float* scratch, void* stream);
int dgnn_colsum_bf16(const uint16_t* x, int64_t ldx, int64_t M, int c, float* out, int accumulate, float* scratch, void* stream);
int dgnn_relu_bf16(const uint16_t* x, int64_t n, uint16_t* y, void* stream);
int dgnn_relu_bwd_bf16(const uint16_t* y, const uint16_t* g, int64_t n, uint16_t* out, void* stream);

/* Debug only: register a device buffer of n int64 slots; workgroup 0 of the fused kernel then stamps
 * wall_clock64() at phase boundaries (slot = (tile_iter*12 + wave)*8 + phase).  NULL disables. */
int dgnn_debug_trace_buffer(int64_t* dev_buf, int64_t n);

/* ------------------------------------------------------------------------------------------------
 * k-hop full-neighbour block builder: GPU replacement of the CPU
 * torch_geometric NeighborSampler(edge_index, sizes=[-1]*k) the reference builds at run.py:72-74,221-223.
 * One hop: dgnn_khop_count -> host reads off[n_t] (edge total) -> dgnn_khop_expand -> host reads *n_new ->
 * dgnn_khop_commit; dgnn_khop_reset after the last hop of a batch.  `pos` int32 [n_nodes] all -1 and `first`
 * int32 [n_nodes] all INT32_MAX between batches (dgnn_fill_i32).  rowptr/src/eid: the by-destination plan.
 * Output of a hop: local edge list e_src/e_dst int64 [n_edges] (targets keep ids 0..n_t-1, new sources appended in
 * first-appearance order), e_id int64 [n_edges] rows of edge_attr, n_id_out int64 [n_t + n_new].
 * ---------------------------------------------------------------------------------------------- */
int dgnn_fill_i32(int32_t* p, int64_t n, int32_t value, void* stream);
int64_t dgnn_khop_scratch_elems(int64_t n_t, int64_t max_edges);
int dgnn_khop_count(const int32_t* rowptr, const int64_t* n_id, int64_t n_t, int first_hop, int32_t* pos, int32_t* off,
                    int32_t* scratch, void* stream);
int dgnn_khop_expand(const int32_t* rowptr, const int32_t* src, const int32_t* eid, const int64_t* n_id, int64_t n_t,
                     const int32_t* off, int64_t n_edges, int32_t* pos, int32_t* first, int64_t* e_src, int64_t* e_dst,
                     int64_t* e_id, int64_t* n_id_out, int32_t* n_new_out, int32_t* scratch, void* stream);
int dgnn_khop_commit(const int64_t* n_id_out, int64_t n_t, int64_t n_all, int32_t* pos, int32_t* first, void* stream);
int dgnn_khop_reset(const int64_t* n_id, int64_t n, int32_t* pos, void* stream);
/* All hops of one batch in one call for graphs with exactly `deg` in-edges per node (Delaunay scenes: 4): edge counts are known
 * without a read-back, the one 4-byte read per hop (new-node count) is waited for inside the call.  Hop h writes into
 * caller-allocated buffers of capacity cap_t[h] targets / cap_e[h] edges: ei[h] int64 [2, cap_e[h]], e_id[h] int64 [cap_e[h]],
 * src32[h] / e_id32[h] int32 [cap_e[h]] (int32 copies of ei[h] row 0 and of e_id[h]), off[h] int32 [cap_t[h]+1], n_id_out[h] int64 [cap_t[h]+cap_e[h]]; scratch =
 * max_h dgnn_khop_scratch_elems(cap_t[h], cap_e[h]) int32; n_new_dev one device int32; counts_out HOST int64 [hops+1] = targets
 * of every hop, then the node count of the outermost block.  t_rowptr != NULL also builds every hop's transposed plan (by source:
 * t_rowptr[h] int32 [cap_all[h]+1], t_dst[h] / t_eid[h] int32 [cap_e[h]]) and t_rows[h] = e_id32[h][t_eid[h]], with plan_scratch =
 * max_h dgnn_plan_scratch_elems(cap_e[h], cap_all[h]) int32.  DGNN_E_INVALID if a capacity is too small (pos is left all -1).
 * The per-hop count travels through pinned host memory the GPU writes to, not through a stream synchronize, so a host thread
 * running this call does not contend with another one that is launching kernels. */
int dgnn_khop_blocks_regular(const int32_t* rowptr, const int32_t* src, const int32_t* eid, int deg, const int64_t* batch, int64_t n_batch,
                             int hops, int32_t* pos, int32_t* first, int64_t* const* ei, int64_t* const* e_id, int32_t* const* src32,
                             int32_t* const* e_id32, int32_t* const* off, int64_t* const* n_id_out, const int64_t* cap_t, const int64_t* cap_e, int32_t* scratch,
                             int32_t* n_new_dev, int32_t* const* t_rowptr, int32_t* const* t_dst, int32_t* const* t_eid, int32_t* const* t_rows,
                             const int64_t* cap_all, int32_t* plan_scratch, int64_t* counts_out, void* stream);
/* The same call on a library-owned host thread: start() returns a job handle at once (NULL + error text on bad arguments), a
 * std::thread issues the launches on `stream` and waits for the per-hop counts, wait() joins it, fills counts_out [hops+1] and
 * frees the job.  Between the two the caller may enqueue on OTHER streams only and must leave pos / first / the buffers alone. */
void* dgnn_khop_blocks_regular_start(const int32_t* rowptr, const int32_t* src, const int32_t* eid, int deg, const int64_t* batch,
                                     int64_t n_batch, int hops, int32_t* pos, int32_t* first, int64_t* const* ei, int64_t* const* e_id,
                                     int32_t* const* src32, int32_t* const* e_id32, int32_t* const* off, int64_t* const* n_id_out,
                                     const int64_t* cap_t, const int64_t* cap_e, int32_t* scratch, int32_t* n_new_dev, int32_t* const* t_rowptr,
                                     int32_t* const* t_dst, int32_t* const* t_eid, int32_t* const* t_rows, const int64_t* cap_all,
                                     int32_t* plan_scratch, void* stream);
/* ... and up to 4 row gathers behind the block, on the builder's stream (the head-of-step x_all[n_id, 1:] / x_all[ids] / y_all[ids] of the reference's
 * training loop, learning/surfaceNetStaticEdgeFilters.py:206 and learning/runModel.py:273-274): r_out[i][k, 0:r_cols[i]] = r_src[i][id_k * r_ld[i] + 0:r_cols[i]]
 * for id_k = the nodes of the outermost block (r_which[i] = 0; n_id_out[hops-1][:counts[hops]]) or the batch's targets (r_which[i] = 1).  r_src[i] is
 * fp32 and already points at the first wanted column; r_out[i] holds capacity x r_cols[i] packed rows.  Valid after dgnn_khop_blocks_regular_wait. */
void* dgnn_khop_blocks_regular_start_rows(const int32_t* rowptr, const int32_t* src, const int32_t* eid, int deg, const int64_t* batch,
                                          int64_t n_batch, int hops, int32_t* pos, int32_t* first, int64_t* const* ei, int64_t* const* e_id,
                                          int32_t* const* src32, int32_t* const* e_id32, int32_t* const* off, int64_t* const* n_id_out,
                                          const int64_t* cap_t, const int64_t* cap_e, int32_t* scratch, int32_t* n_new_dev,
                                          int32_t* const* t_rowptr, int32_t* const* t_dst, int32_t* const* t_eid, int32_t* const* t_rows,
                                          const int64_t* cap_all, int32_t* plan_scratch, int n_rows, const float* const* r_src, const int64_t* r_ld,
                                          const int32_t* r_cols, const int32_t* r_which, float* const* r_out, void* stream);
int dgnn_khop_blocks_regular_wait(void* job, int hops, int64_t* counts_out);

/* ------------------------------------------------------------------------------------------------
 * Logits -> labels -> interface facets (reference processing/generate_mesh.py:75 and :93-105).
 *   dgnn_argmax_rows     labels[i] = argmax_c logits[i,c]             (log_softmax(...).argmax(1), ties -> 0)
 *   dgnn_compact_i32     order-preserving stream compaction (finite-cell labels; interface facet ids)
 *   dgnn_interface_flags flags[f] = label(nfacets[f,0]) != label(nfacets[f,1]), cell -1 = outside
 * ---------------------------------------------------------------------------------------------- */
int dgnn_argmax_rows(const float* logits, int64_t ld, int64_t n, int c, int32_t* labels, void* stream);
int64_t dgnn_compact_scratch_elems(int64_t n);
int dgnn_compact_i32(const int32_t* values, const int32_t* keep, int invert, int64_t n, int32_t* out, int32_t* count_out,
                     int32_t* scratch, void* stream);
int dgnn_interface_flags(const int32_t* nfacets, const int32_t* labels_finite, int64_t n_facets, int32_t* flags, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Per-scene standardisation (reference processing/data.py:444-506 sklearn StandardScaler + :512-519
 * float32 cast): out[i,c] = float((x[i,c] - mean_c) / std_c) for c >= c_first, plain cast for c < c_first;
 * fp64 statistics (population variance, zero scale -> 1).  scratch: dgnn_standardize_scratch_doubles(c) doubles.
 * ---------------------------------------------------------------------------------------------- */
int64_t dgnn_standardize_scratch_doubles(int c);
int dgnn_standardize_f64(const double* x, int64_t ld, int64_t n, int c, int c_first, float* out, int64_t ldo, double* scratch,
                         void* stream);

/* ------------------------------------------------------------------------------------------------
 * Ingest-time cell locality order (sits where the reference builds edge_index, processing/data.py:434-438).
 * CGAL writes the cells in insertion order (the 4 neighbours of a cell are tens of thousands of rows apart); the loader relabels them once
 * per scene so that the conv layers' neighbour gathers hit L2, and keeps the permutation for the places where per-cell results leave
 * (dataLoader.exportScore, generate_mesh.generate :75-81).  All int32 index work, deterministic:
 *   dgnn_cell_centroids_3dt  centroids [n,3] from <scene>_3dt.npz: `vertices` fp32 [n_vertices,3], `tetrahedra` int32 [n_finite,4] = the
 *                            FINITE cells in file order (generate_mesh.py:78-81 relies on the same correspondence); `infinite` int32 [n]
 *                            (labels.npz, :202-208); an infinite cell takes its finite neighbour's centroid (edge_index rows 4i..4i+3).
 *                            scratch: dgnn_cell_centroids_scratch_elems(n) int32
 *   dgnn_cell_order_morton   order[i] = old id of new cell i (cells sorted, stable, by the 48-bit Morton code of their centroid: 16 bits per
 *                            axis over the bounding box), rank = its inverse.  scratch (8-byte aligned): dgnn_cell_order_morton_scratch_elems(n) int32
 *   dgnn_cell_order_bfs      the same from the adjacency alone (no coordinates): breadth-first order, every level in the order a serial
 *                            queue produces (first discoverer wins, neighbours in slot order), components started at their lowest cell id.
 *                            Reference layout required (row 4t+k leaves cell t).  SYNCHRONISES `stream` once per level (ingest-time call).
 *                            scratch: dgnn_cell_order_bfs_scratch_elems(n) int32
 *   dgnn_reorder_edges_ref   relabelled adjacency: pairs_out int64 [4n,2] = (new src, new dst) of new edge row e = old row 4*order[e/4] + e%4,
 *                            i.e. the reference layout again (its transposed [2,4n] view with strides (1,2) is what dgnn_plan_build's
 *                            REFERENCE fast path takes); edge_rows_out int32 [4n] = that old row (gathers edge_attr with dgnn_gather_rows_f32).
 *                            A source that is not the reference layout raises the asynchronous index error.
 * ---------------------------------------------------------------------------------------------- */
int64_t dgnn_cell_centroids_scratch_elems(int64_t n);
int dgnn_cell_centroids_3dt(const float* vertices, int64_t n_vertices, const int32_t* tetrahedra, int64_t n_finite, const int32_t* infinite,
                            const int64_t* edge_index, int64_t stride_row, int64_t stride_col, int64_t n, float* centroids, int32_t* scratch,
                            void* stream);
int64_t dgnn_cell_order_morton_scratch_elems(int64_t n);
int dgnn_cell_order_morton(const float* centroids, int64_t n, int32_t* order, int32_t* rank, int32_t* scratch, void* stream);
int64_t dgnn_cell_order_bfs_scratch_elems(int64_t n);
int dgnn_cell_order_bfs(const int64_t* edge_index, int64_t stride_row, int64_t stride_col, int64_t n, int32_t* order, int32_t* rank,
                        int32_t* scratch, void* stream);
int dgnn_reorder_edges_ref(const int64_t* edge_index, int64_t stride_row, int64_t stride_col, int64_t n, const int32_t* order,
                           const int32_t* rank, int64_t* pairs_out, int32_t* edge_rows_out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Partitioned scene: per-layer halo exchange over RCCL (what replaces the reference's k-hop recomputation for scenes that do not
 * fit one pass, learning/surfaceNetStaticEdgeFilters.py:232-275 / run.py:221-223).  A rank's activation buffer is [n_own + n_halo, C]: owned
 * rows first, then the halo rows grouped by owner rank.  Before conv layer l >= 1:
 *   dgnn_halo_exchange_start  packs x[send_idx] on `stream`, then -- on the plan's own side stream, ordered behind the pack by an event -- ONE
 *                             group of ncclRecv (straight into the buffer's tail) and ncclSend (every GPU pair of an MI355X node has a direct xGMI
 *                             link: single-hop neighbour exchange, no ring)
 *   dgnn_halo_exchange_wait   makes `stream` wait for the received rows
 * Launches queued on `stream` between the two (the interior cells) overlap the transfer.  Rows are `elem_bytes` (2 or 4) x C bytes, whole 4-byte
 * words; send_buf: dgnn_halo_send_rows(plan) * C * elem_bytes bytes, caller-owned, reusable after the wait.  send_idx (device int32, caller-owned,
 * must outlive the plan): local ids of the owned rows to send, grouped by destination rank; send_counts / recv_counts: host arrays [world].
 * `comm`: an ncclComm_t -- the caller's, or one made by dgnn_comm_create from the 128-byte id of dgnn_comm_unique_id (rank 0 asks, the host
 * broadcasts it over its own channel, every rank creates on its current device).  RCCL is resolved at run time (DGNN_RCCL_LIB, an already
 * loaded librccl, the loader's path): dgnn_rccl_available() == 0 and DGNN_E_UNSUPPORTED from these entry points where there is none.
 * What goes over a link is ONE message of rows * C * elem_bytes per peer and direction, whatever either side's row stride: with ld == C it lands in
 * the tail directly and nothing allocates or synchronises the host; with ld != C it lands in a receive staging area owned by the plan (allocated on
 * first use, grown on demand, freed by dgnn_halo_plan_destroy) and is spread into the strided tail by a copy kernel on the side stream.
 * A rank may be its own peer (RCCL allows self send / recv inside a group).  dgnn_comm_count: ncclCommCount of `comm` (the number of ranks the
 * communicator spans: what a benchmark line reports as the RCCL world size), negative error code on failure.
 * ---------------------------------------------------------------------------------------------- */
typedef struct dgnn_halo_plan dgnn_halo_plan;
int dgnn_rccl_available(void);
int dgnn_comm_unique_id(void* id128);
int dgnn_comm_create(const void* id128, int rank, int world, void** comm_out);
int dgnn_comm_destroy(void* comm);
int dgnn_comm_count(void* comm);
int dgnn_halo_plan_create(int rank, int world, int64_t n_own, const int32_t* send_idx, const int64_t* send_counts, const int64_t* recv_counts,
                          dgnn_halo_plan** out);
int dgnn_halo_plan_destroy(dgnn_halo_plan* plan);
int64_t dgnn_halo_send_rows(const dgnn_halo_plan* plan);
int64_t dgnn_halo_recv_rows(const dgnn_halo_plan* plan);
int dgnn_halo_exchange_start(dgnn_halo_plan* plan, void* comm, void* x, int64_t ld, int C, int elem_bytes, void* send_buf, void* stream);
int dgnn_halo_exchange_wait(dgnn_halo_plan* plan, void* stream);

/* ------------------------------------------------------------------------------------------------
 * One rank's part of a scene cut across GPUs, ONE call per step (what replaces the reference's k-hop recomputation at
 * learning/surfaceNetStaticEdgeFilters.py:232-275 for a scene larger than -- or, strong scaling, spread over more than -- one GPU).
 * dgnn_static_infer_fwd for the LOCAL bipartite graph of a part: n_own owned cells (the first n_interior of them neither read a halo row nor are
 * sent to a peer) + n_halo halo rows behind them; x [n_own + n_halo, widths[0]] with the halo's input rows resident; the local edge list has
 * destinations < n_own and sources < n_own + n_halo.  Launch chain (dgnn_amd/partition.py run_partitioned_layers, now inside the library):
 *   plan of the local graph (edge_index != NULL; n_key = n_own, n_other = n_own + n_halo) -> layer 0 over the owned cells ->
 *   per layer l >= 1: interior cells [0, n_interior) | dgnn_halo_exchange_wait | boundary cells [n_interior, n_own)
 *   behind every layer but the last: dgnn_halo_exchange_start on its output rows [n_own + n_halo, widths[l+1]] (fp32, packed)
 *   last layer: both launches carry the decoder where dgnn_static_infer_fwd's would; logits [n_own, n_logits]
 * attr_in_plan_order != 0: edge_attr rows are already grouped by destination (a part's local list is): the layers read them in place, eid is
 * only written by the plan build.  halo == NULL (n_halo must be 0): a single-rank part, no exchange.  send_buf: dgnn_halo_send_rows(halo) *
 * max(widths[1..L-1]) * 4 bytes.  workspace: dgnn_static_infer_workspace_bytes(n_own + n_halo, n_layers, widths).  Same kernels, ranges and
 * order as the per-layer calls: the union of the ranks' logits is bit-identical to dgnn_static_infer_fwd on the whole scene.
 * Nothing allocates or synchronises; DGNN_E_UNSUPPORTED before anything is launched for shapes outside the fused kernels.
 * ---------------------------------------------------------------------------------------------- */
int dgnn_static_infer_partitioned_fwd(const int64_t* edge_index, int64_t stride_row, int64_t stride_col, int64_t E, int plan_hint, int32_t* rowptr,
                                      int32_t* src, int32_t* eid, int32_t* plan_scratch, int attr_in_plan_order, int64_t n_own, int64_t n_interior,
                                      int64_t n_halo, const float* x, int64_t ldx, const float* edge_attr, int64_t lde, int f_e, int n_layers,
                                      const int32_t* widths, const float* const* We, const float* const* be, const float* const* Wj,
                                      const float* const* bj, const float* const* Wi, const float* const* scale, const float* const* shift,
                                      const void* const* prepared, const float* W0, const float* b0, const float* scale1, const float* shift1,
                                      int c_hidden, const float* W3, const float* b3, int n_logits, int fuse_decoder, int gemm_mode,
                                      dgnn_halo_plan* halo, void* comm, void* send_buf, void* workspace, float* logits, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Wide conv layers on SPLIT ROWS (round 5; csrc/wide.hip).  SAGEConv.forward (learning/surfaceNetStaticEdgeFilters.py:66-96) at the widths the
 * reference's real configs use (configs/eth.yaml:56, aerial.yaml:57: [64,128,256,512]; configs/modelnet.yaml:56: [128,256,512,1024]): layers
 * with C_in in {128, 256, 512} and C_out a multiple of 256 -- outside dgnn_sage_layer_fused_fwd's shapes -- as
 *     dgnn_sage_aggregate_sr   a = mean_j x_j * lin_e(edge_attr_j)                       (:75-80, :89-96)
 *     dgnn_linear_sr           act(([a | x_i] . [Wj | Wi]^T + bj) * scale + shift)       (:81-86 + BatchNorm(eval) + ReLU; also the decoder's first Linear)
 * with every wide activation stored in HBM as SPLIT ROWS: a row of C channels (C % 32 == 0) is C / 32 chunks of 128 bytes; chunk q holds, for its 32
 * positions p, hi[p] (fp16, bytes 2p) and lo[p] (fp16, bytes 64 + 2p) of x * s for channel 32 q + PI(p), PI(p) = (p & 3) | (p >> 4) << 2 |
 * ((p >> 2) & 3) << 3, hi = RN16(x s), lo = RN16(x s - hi) (22 significand bits in 4 bytes); s is a power of two per row and GROUP of 256 channels
 * (`scales` [rows][ceil(C / 256)] fp32) that puts the group's largest magnitude into [2^14, 2^15); a group below 2^-112 is stored as zeros with
 * s = 2^127.  dgnn_sr_row_bytes(C) = C / 32 * 128.  The format is what the fp16 two-part matrix products consume (3 products per fp32 product,
 * fp32 accumulation: fp32-class results, tests/test_gpu_wide.py), written by the kernel that PRODUCES the activation: no row-scale pass, no
 * split instructions and no fp32 staging in the consumer.  dgnn_sr_pack / dgnn_sr_unpack convert from / to fp32 rows (weights at prepare time,
 * boundaries, tests).  All pointers 16-byte aligned; nothing allocates or synchronises.
 * ---------------------------------------------------------------------------------------------- */
int64_t dgnn_sr_row_bytes(int C);
/* fp32 rows [rows][k1 (+ k2)] (k1, k2 multiples of 32) -> split rows dst [rows][dst_row_bytes] + scales [rows][ng]; gch = chunks per scale group:
 * 8 for activations (ng = ceil(chunks / 8)), 0 = ONE group per row (ng = 1): the weights of dgnn_linear_sr are [Wj | Wi] packed with gch = 0 */
int dgnn_sr_pack(const float* A1, int64_t ld1, int k1, const float* A2, int64_t ld2, int k2, int64_t rows, int gch, void* dst,
                 int64_t dst_row_bytes, float* scales, int ng, void* stream);
int dgnn_sr_unpack(const void* src, int64_t row_bytes, const float* scales, int ng, int gch, int C, int64_t rows, float* out, int64_t ldo,
                   void* stream);
/* the filter operand [We^T ; be] of dgnn_sage_aggregate_sr, prepared once per set of weights (C in {128, 256, 512}; 0 bytes = unsupported width) */
int64_t dgnn_sr_filter_prepared_bytes(int C);
int dgnn_sr_prepare_filter(const float* We, const float* be, int C, void* buf, void* stream);
/* a_out / a_scales: split rows of a [n_dst][C].  x: the source rows, split rows (x_is_sr != 0, xs their scales) or fp32 rows (row stride ldx floats);
 * with fp32 rows and x_out != NULL the destinations' own rows x[:n_dst] are also written as split rows (x_out, x_scales).  edge_attr: fp32 packed
 * [E][20] rows, gathered by eid (NULL: plan order).  Any in-degree (4-regular groups take the matrix-core path).  DGNN_E_UNSUPPORTED before anything
 * is launched for other widths / layouts. */
int dgnn_sage_aggregate_sr(const int32_t* rowptr, const int32_t* src, const int32_t* eid, int64_t n_dst, const void* x, int x_is_sr, int64_t ldx,
                           const float* xs, int C, const float* edge_attr, int64_t lde, const float* We, const float* be, const void* prep, void* a_out,
                           float* a_scales, void* x_out, float* x_scales, void* stream);
/* A1 (C1 channels) and A2 (C2 channels; NULL / 0: one operand) split rows with their scales; Wp / sw = dgnn_sr_pack([W1 | W2] fp32 [n_out][C1 + C2],
 * gch = 0); n_out a multiple of 256.  Exactly ONE output: split rows (out_sr, out_row_bytes, out_scales [M][n_out / 256]); fp32 rows out_f32
 * [M][n_out] with row stride ldo; or logits [M][n_proj] = act(...) . W3^T + b3 (W3 fp32 [n_proj][n_out], n_proj 1 or 2: the decoder's output
 * Linear applied to the finished hidden rows inside the launch, reference :180-187; n_out > 256: logits must be ZERO on entry, the column tiles
 * add into them -- two addends per logit at n_out = 512, an order-independent sum; n_out > 512 is declined).  DGNN_E_UNSUPPORTED for other shapes. */
int dgnn_linear_sr(const void* A1, int64_t row_bytes1, const float* scales1, int C1, const void* A2, int64_t row_bytes2, const float* scales2, int C2,
                   const void* Wp, const float* sw, const float* bias, const float* scale, const float* shift, int relu, int64_t M, int n_out, void* out_sr,
                   int64_t out_row_bytes, float* out_scales, float* out_f32, int64_t ldo, const float* W3, const float* b3, int n_proj, float* logits,
                   void* stream);

/* elementwise helpers used by the Updated variant (F.relu at surfaceNetUpdatedEdgeFilters.py:239-241
 * and the scatter of phi rows into the zero [E_all,C] buffer at :236-237) */
int dgnn_relu(const float* x, int64_t n, float* y, void* stream);
int dgnn_relu_bwd(const float* y, const float* g, int64_t n, float* out, void* stream); /* out = g * [y > 0] */
int dgnn_scatter_rows_f32(const float* in, int64_t ld_in, const int64_t* idx, int64_t n, int cols, float* out,
                          int64_t ld_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DGNN_HIP_H */
