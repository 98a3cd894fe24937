// The two "sum the partial slabs" routines of a conv layer's backward -- filter-weight gradients (one slab per block of the aggregate
// backward) and lin_j / lin_i / bias gradients (one partial matrix per row split of the merged weight-gradient launch) -- as device
// functions with descriptors, so that the training step can run both from ONE launch per layer (train.hip: k_reduce_layer) while the
// separate entry points keep their own kernels.  Fixed summation orders: results do not depend on which launch runs them.
#pragma once
#include "common.h"

struct SlabReduceDesc {      // dWe [c_in, fe], dbe [c_in] = sums over `nblocks` slabs of [nchunks][per = 64 cpl (fe + 1)]
    const float* slabs;
    int nblocks, nchunks, per, c_in, fe, cpl;
    float *dWe, *dbe;
};
struct WgradReduceDesc {     // dW1 [na, nb1], dW2 [na, nb2] = sums over `splits` partials of [na][nb1 + nb2]; dbias [na] from fp64 partials [splits][na]
    const float* partials;
    const double* bias_partials;
    int splits, na, nb1, nb2;
    float *dW1, *dW2, *dbias;
};

constexpr int RS_SLICES = 64;   // reduce_slabs_block: 16 outputs x 64 slices = 1024 threads
__host__ __device__ inline int slab_reduce_blocks(const SlabReduceDesc& d) { return (d.nchunks * d.per + 15) / 16; }
// WIDE form of the matrix part (round 6): with few row splits and a large gradient (the reference's training widths: [1024, 1024 + 1024] elements,
// 8-32 splits) the 16-outputs-per-256-threads form above launched 65 536 workgroups in which 16 x splits threads made one 4-byte load each (k_reduce_layer
// 52 us for 33 MB).  Here a thread owns 4 consecutive outputs (one 16-byte load per split), 1024 outputs per 256 threads; the sum per output runs in
// the SAME order (16 slices z = sl, sl + 16, ... each in ascending z from 0.f, the slice sums added in slice order from 0.f): bit-identical results.
__host__ __device__ inline bool wgrad_reduce_wide(const WgradReduceDesc& d) {
    return d.splits <= 64 && (int64_t)d.na * (d.nb1 + d.nb2) >= 32768 && d.nb1 % 4 == 0 && d.nb2 % 4 == 0 && (((uintptr_t)d.partials | (uintptr_t)d.dW1 | (uintptr_t)d.dW2) & 15) == 0;
}
__host__ __device__ inline int wgrad_reduce_mat_blocks(const WgradReduceDesc& d) {
    const int nmat = d.na * (d.nb1 + d.nb2);
    return wgrad_reduce_wide(d) ? (nmat + 1023) / 1024 : (nmat + 15) / 16;
}
__host__ __device__ inline int wgrad_reduce_blocks(const WgradReduceDesc& d) {   // blocks of 256 threads
    return wgrad_reduce_mat_blocks(d) + (d.dbias ? (d.na + 15) / 16 : 0);
}

// one block of 1024 threads; `blk` = which 16 outputs.  Slice s adds slabs s, s + 64, ... in order, the 64 slice sums are added in slice order.
__device__ __forceinline__ void reduce_slabs_block(const SlabReduceDesc& d, int blk, float (*red)[17]) {
    const int o = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int i = blk * 16 + o;
    const bool live = i < d.nchunks * d.per;
    const int chunk = live ? i / d.per : 0, r = live ? i - chunk * d.per : 0;
    float p = 0.f;
    if (live) {      // four slabs requested at a time, added in ascending order (same sums; a load -> add chain per slab otherwise)
        int b = sl;
        for (; b + 3 * RS_SLICES < d.nblocks; b += 4 * RS_SLICES) {
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = d.slabs[((int64_t)(b + u * RS_SLICES) * d.nchunks + chunk) * d.per + r];
#pragma unroll
            for (int u = 0; u < 4; ++u) p += v[u];
        }
        for (; b < d.nblocks; b += RS_SLICES) p += d.slabs[((int64_t)b * d.nchunks + chunk) * d.per + r];
    }
    red[sl][o] = p;
    __syncthreads();
    if (sl != 0 || !live) return;
    const int c = chunk * 64 * d.cpl + r / (d.fe + 1), f = r % (d.fe + 1);
    if (c >= d.c_in) return;
    float s = 0.f;
    for (int k = 0; k < RS_SLICES; ++k) s += red[k][o];
    if (f < d.fe)
        d.dWe[(int64_t)c * d.fe + f] = s;
    else
        d.dbe[c] = s;
}

// 256 consecutive threads (`t` = index among them, `blk` = which 16 outputs); redf / redd: [16][17] each, private to these 256 threads.
// The caller synchronises: phase 0 fills the scratch, phase 1 (after a barrier) writes the results.
__device__ __forceinline__ void wgrad_reduce_cat_phase(const WgradReduceDesc& d, int blk, int t, float (*redf)[17], double (*redd)[17], int phase) {
    const int o = t & 15, sl = t >> 4;
    const int nbt = d.nb1 + d.nb2, nmat = d.na * nbt, nblk_mat = wgrad_reduce_mat_blocks(d);
    if (blk < nblk_mat && wgrad_reduce_wide(d)) {
        if (phase == 0) return;
        const int i = (blk * 256 + t) * 4;
        if (i >= nmat) return;
        typedef float f4 __attribute__((ext_vector_type(4)));
        f4 s = {0.f, 0.f, 0.f, 0.f};
        for (int slc = 0; slc < 16 && slc < d.splits; ++slc) {
            f4 p_ = {0.f, 0.f, 0.f, 0.f};
            for (int z = slc; z < d.splits; z += 16) p_ += *reinterpret_cast<const f4*>(d.partials + (int64_t)z * nmat + i);
            s += p_;
        }
        const int row = i / nbt, col = i - row * nbt;     // (nb1, nb2 multiples of 4: the four outputs sit in one matrix)
        float* out = col < d.nb1 ? d.dW1 + (int64_t)row * d.nb1 + col : d.dW2 + (int64_t)row * d.nb2 + (col - d.nb1);
        *reinterpret_cast<f4*>(out) = s;
        return;
    }
    if (blk < nblk_mat) {
        const int i = blk * 16 + o;
        const bool live = i < nmat;
        if (phase == 0) {
            float p = 0.f;
            if (live) {
                int z = sl;
                for (; z + 48 < d.splits; z += 64) {
                    float v[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) v[u] = d.partials[(int64_t)(z + 16 * u) * nmat + i];
#pragma unroll
                    for (int u = 0; u < 4; ++u) p += v[u];
                }
                for (; z < d.splits; z += 16) p += d.partials[(int64_t)z * nmat + i];
            }
            redf[sl][o] = p;
            return;
        }
        if (sl != 0 || !live) return;
        float s = 0.f;
        for (int k = 0; k < 16; ++k) s += redf[k][o];
        const int row = i / nbt, col = i - row * nbt;
        if (col < d.nb1)
            d.dW1[(int64_t)row * d.nb1 + col] = s;
        else
            d.dW2[(int64_t)row * d.nb2 + (col - d.nb1)] = s;
    } else {
        const int i = (blk - nblk_mat) * 16 + o;
        const bool live = i < d.na && d.dbias != nullptr;
        if (phase == 0) {
            double p = 0.0;
            if (live)
                for (int z = sl; z < d.splits; z += 16) p += d.bias_partials[(int64_t)z * d.na + i];
            redd[sl][o] = p;
            return;
        }
        if (sl != 0 || !live) return;
        double s = 0.0;
        for (int k = 0; k < 16; ++k) s += redd[k][o];
        d.dbias[i] = (float)s;
    }
}

// deferred forms (train.hip): the first launch of the pair only, the reduction described in *desc for the caller to run
int dgnn_linear_wgrad_x3_cat_deferred(const float* A, int64_t lda, int n_a, const float* B1, int64_t ldb1, int n_b1, const float* B2, int64_t ldb2, int n_b2,
                                      int64_t M, float* dW1, float* dW2, float* dbias, float* scratch, void* stream, WgradReduceDesc* desc);
int dgnn_linear_wgrad_bf16_cat_deferred(const void* A, int a_f32, int64_t lda, int n_a, const void* B1, int64_t ldb1, int n_b1, const void* B2, int64_t ldb2,
                                        int n_b2, int b_f32, int64_t M, float* dW1, float* dW2, float* dbias, float* scratch, void* stream, WgradReduceDesc* desc);
int dgnn_sage_aggregate_bwd_add_deferred(const int32_t* t_rowptr, const int32_t* t_dst, const int32_t* t_eid, int64_t n_src, const int32_t* rowptr_dst,
                                         const float* x_src, int64_t ldx, int c_in, const float* edge_attr, int64_t lde, int f_e, const float* We,
                                         const float* be, const float* da, int64_t ldda, float* dx_src, int64_t lddx, const float* add, int64_t ldadd,
                                         int64_t n_add, float* dWe, float* dbe, float* partials, void* stream, SlabReduceDesc* desc);
