// Edge-embedding chaining of the Updated variant (reference surfaceNetUpdatedEdgeFilters.py:233-241):
//
//     new_edge_attr = zeros[E_all, C];  new_edge_attr[e_id_cur] = phi;  edge_attr = relu(new_edge_attr)
//     next layer reads edge_attr[e_id_next, :c]
//
// The reference materialises the [E_all, C] tensor of the WHOLE scene for every layer of every batch (zero fill, scatter,
// ReLU over all of it forward; the same three passes backward) although only the rows e_id_next are ever read.  Here the
// next layer's rows are produced directly from this layer's phi:
//
//     out[k, :] = relu(phi[row(k), :c])   if e_id_next[k] == e_id_cur[row(k)] for some row(k),   else 0
//
// through an [E_all] int32 position table that is all -1 between calls (only the e_id_cur entries are touched and reset),
// so the work is O(E_cur + E_next) rows instead of O(E_all).  `inv` (the inverse of row()) is kept for the backward pass,
// which then writes every element of dphi once, coalesced, with no zero fill.  Same values as the reference, element for
// element (a copy, a compare and a select: nothing is rounded).
#include "common.h"

namespace {

template <typename T, int V>
struct alignas(sizeof(T) * V) Pk {
    T v[V];
};

__device__ __forceinline__ bool positive(float x) { return x > 0.f; }

__global__ void k_chain_map(const int64_t* __restrict__ e_cur, int64_t n_cur, int64_t n_edges, int32_t* __restrict__ pos, int32_t value_is_index,
                            int32_t* aflag, int32_t* __restrict__ inv) {
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n_cur; k += (int64_t)gridDim.x * blockDim.x) {
        if (inv) inv[k] = -1;      // the inverse map starts empty (was a memset launch of its own)
        const int64_t e = e_cur[k];
        if (e < 0 || e >= n_edges) {
            dgnn_raise_async(aflag, DGNN_ASYNC_OTHER_RANGE);
            continue;
        }
        // PRECONDITION: the ids of a block's e_id are unique (PyG's NeighborSampler and dgnn_khop_* emit every edge once).  A duplicate would keep
        // one position and drop the other's gradient in the backward pass -- it is reported through the asynchronous error word instead.
        if (value_is_index) {
            if (atomicExch(&pos[e], (int32_t)k) >= 0) dgnn_raise_async(aflag, DGNN_ASYNC_DUPLICATE);
        } else {
            pos[e] = -1;
        }
    }
}

// one thread per V-element piece of an output row
template <typename T, int V>
__global__ void k_chain_gather(const T* __restrict__ phi, int64_t ldphi, int c, const int64_t* __restrict__ e_next, int64_t n_next,
                               int64_t n_edges, const int32_t* __restrict__ pos, int relu, T* __restrict__ out, int64_t ldo,
                               int32_t* __restrict__ inv, int32_t* aflag) {
    const int cpv = c / V;
    const int64_t items = n_next * cpv;
    for (int64_t it = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; it < items; it += (int64_t)gridDim.x * blockDim.x) {
        const int64_t k = it / cpv;
        const int j = (int)(it - k * cpv);
        const int64_t e = e_next[k];
        int32_t row = -1;
        if (e < 0 || e >= n_edges)
            dgnn_raise_async(aflag, DGNN_ASYNC_OTHER_RANGE);
        else
            row = pos[e];
        Pk<T, V> o;
        if (row >= 0) {
            o = *reinterpret_cast<const Pk<T, V>*>(phi + (int64_t)row * ldphi + (int64_t)j * V);
            if (relu) {
#pragma unroll
                for (int q = 0; q < V; ++q)
                    if (!positive(dgnn_ld(&o.v[q]))) o.v[q] = T(0);
            }
            if (j == 0 && atomicExch(&inv[row], (int32_t)k) >= 0) dgnn_raise_async(aflag, DGNN_ASYNC_DUPLICATE);   // e_id_next names this edge twice
        } else {
#pragma unroll
            for (int q = 0; q < V; ++q) o.v[q] = T(0);
        }
        *reinterpret_cast<Pk<T, V>*>(out + k * ldo + (int64_t)j * V) = o;
    }
}

// dphi[r, j] = (inv[r] >= 0 && j < c) ? g[inv[r], j] * [phi[r, j] > 0] : 0      -- every element written once
template <typename T, int V>
__global__ void k_chain_bwd(const T* __restrict__ g, int64_t ldg, const T* __restrict__ phi, int64_t ldphi, const int32_t* __restrict__ inv,
                            int64_t n_cur, int c, int c_tot, int relu, T* __restrict__ dphi, int64_t lddphi) {
    const int cpv = c_tot / V;
    const int64_t items = n_cur * cpv;
    for (int64_t it = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; it < items; it += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = it / cpv;
        const int j = (int)(it - r * cpv);
        const int32_t k = inv[r];
        Pk<T, V> o;
        if (k >= 0 && j * V < c) {   // c is a multiple of V whenever V > 1 (launcher)
            o = *reinterpret_cast<const Pk<T, V>*>(g + (int64_t)k * ldg + (int64_t)j * V);
            if (relu) {
                const Pk<T, V> p = *reinterpret_cast<const Pk<T, V>*>(phi + r * ldphi + (int64_t)j * V);
#pragma unroll
                for (int q = 0; q < V; ++q)
                    if (!positive(dgnn_ld(&p.v[q]))) o.v[q] = T(0);
            }
        } else {
#pragma unroll
            for (int q = 0; q < V; ++q) o.v[q] = T(0);
        }
        *reinterpret_cast<Pk<T, V>*>(dphi + r * lddphi + (int64_t)j * V) = o;
    }
}

inline bool aligned_to(const void* p, size_t a) { return ((uintptr_t)p % a) == 0; }

template <typename T>
int chain_fwd(const T* phi, int64_t ldphi, int c, const int64_t* e_cur, int64_t n_cur, const int64_t* e_next, int64_t n_next, int64_t n_edges,
              int32_t* pos, int relu, T* out, int64_t ldo, int32_t* inv, hipStream_t stream) {
    DGNN_REQUIRE(n_cur >= 0 && n_next >= 0 && n_edges >= 0 && c > 0, DGNN_E_INVALID, "edge_chain_fwd: bad sizes");
    DGNN_REQUIRE(n_cur < (int64_t)1 << 31 && n_next < (int64_t)1 << 31, DGNN_E_INVALID, "edge_chain_fwd: more than 2^31 edges in one block");
    if (n_next == 0) return DGNN_OK;
    DGNN_REQUIRE(out && e_next && pos && (n_cur == 0 || (phi && e_cur && inv)), DGNN_E_INVALID, "edge_chain_fwd: null pointer");
    int32_t* const aflag = dgnn_async_flag_dev();
    constexpr int VMAX = 16 / sizeof(T);
    if (n_cur > 0) {
        hipLaunchKernelGGL(k_chain_map, dim3(dgnn_grid_cap(dgnn_cdiv(n_cur, 256))), dim3(256), 0, stream, e_cur, n_cur, n_edges, pos, 1, aflag, inv);
    }
    const bool vec = c % VMAX == 0 && ldphi % VMAX == 0 && ldo % VMAX == 0 && aligned_to(phi, 16) && aligned_to(out, 16);
    if (vec)
        hipLaunchKernelGGL((k_chain_gather<T, VMAX>), dim3(dgnn_grid_cap(dgnn_cdiv(n_next * (c / VMAX), 256))), dim3(256), 0, stream, phi, ldphi, c,
                           e_next, n_next, n_edges, pos, relu, out, ldo, inv, aflag);
    else
        hipLaunchKernelGGL((k_chain_gather<T, 1>), dim3(dgnn_grid_cap(dgnn_cdiv(n_next * c, 256))), dim3(256), 0, stream, phi, ldphi, c, e_next, n_next,
                           n_edges, pos, relu, out, ldo, inv, aflag);
    if (n_cur > 0)
        hipLaunchKernelGGL(k_chain_map, dim3(dgnn_grid_cap(dgnn_cdiv(n_cur, 256))), dim3(256), 0, stream, e_cur, n_cur, n_edges, pos, 0, aflag, (int32_t*)nullptr);
    return dgnn_check_launch("edge_chain_fwd");
}

template <typename T>
int chain_bwd(const T* g, int64_t ldg, const T* phi, int64_t ldphi, const int32_t* inv, int64_t n_cur, int c, int c_tot, int relu, T* dphi,
              int64_t lddphi, hipStream_t stream) {
    DGNN_REQUIRE(n_cur >= 0 && c > 0 && c_tot >= c, DGNN_E_INVALID, "edge_chain_bwd: bad sizes");
    if (n_cur == 0) return DGNN_OK;
    DGNN_REQUIRE(phi && inv && dphi, DGNN_E_INVALID, "edge_chain_bwd: null pointer");
    constexpr int VMAX = 16 / sizeof(T);
    const bool vec = c % VMAX == 0 && c_tot % VMAX == 0 && ldg % VMAX == 0 && ldphi % VMAX == 0 && lddphi % VMAX == 0 && aligned_to(g, 16) &&
                     aligned_to(phi, 16) && aligned_to(dphi, 16);
    if (vec)
        hipLaunchKernelGGL((k_chain_bwd<T, VMAX>), dim3(dgnn_grid_cap(dgnn_cdiv(n_cur * (c_tot / VMAX), 256))), dim3(256), 0, stream, g, ldg, phi, ldphi,
                           inv, n_cur, c, c_tot, relu, dphi, lddphi);
    else
        hipLaunchKernelGGL((k_chain_bwd<T, 1>), dim3(dgnn_grid_cap(dgnn_cdiv(n_cur * c_tot, 256))), dim3(256), 0, stream, g, ldg, phi, ldphi, inv, n_cur,
                           c, c_tot, relu, dphi, lddphi);
    return dgnn_check_launch("edge_chain_bwd");
}

}  // namespace

extern "C" int dgnn_edge_chain_fwd(const float* phi, int64_t ldphi, int c, const int64_t* e_id_cur, int64_t n_cur, const int64_t* e_id_next,
                                   int64_t n_next, int64_t n_edges, int32_t* pos, int relu, float* out, int64_t ldo, int32_t* inv, void* stream) {
    return chain_fwd<float>(phi, ldphi, c, e_id_cur, n_cur, e_id_next, n_next, n_edges, pos, relu, out, ldo, inv, (hipStream_t)stream);
}
extern "C" int dgnn_edge_chain_fwd_bf16(const uint16_t* phi, int64_t ldphi, int c, const int64_t* e_id_cur, int64_t n_cur, const int64_t* e_id_next,
                                        int64_t n_next, int64_t n_edges, int32_t* pos, int relu, uint16_t* out, int64_t ldo, int32_t* inv,
                                        void* stream) {
    return chain_fwd<uint16_t>(phi, ldphi, c, e_id_cur, n_cur, e_id_next, n_next, n_edges, pos, relu, out, ldo, inv, (hipStream_t)stream);
}
extern "C" int dgnn_edge_chain_bwd(const float* g, int64_t ldg, const float* phi, int64_t ldphi, const int32_t* inv, int64_t n_cur, int c, int c_tot,
                                   int relu, float* dphi, int64_t lddphi, void* stream) {
    return chain_bwd<float>(g, ldg, phi, ldphi, inv, n_cur, c, c_tot, relu, dphi, lddphi, (hipStream_t)stream);
}
extern "C" int dgnn_edge_chain_bwd_bf16(const uint16_t* g, int64_t ldg, const uint16_t* phi, int64_t ldphi, const int32_t* inv, int64_t n_cur, int c,
                                        int c_tot, int relu, uint16_t* dphi, int64_t lddphi, void* stream) {
    return chain_bwd<uint16_t>(g, ldg, phi, ldphi, inv, n_cur, c, c_tot, relu, dphi, lddphi, (hipStream_t)stream);
}
