// k-hop full-neighbour block builder on the GPU: the replacement for
// torch_geometric NeighborSampler(edge_index, sizes=[-1]*k, ...) that the reference runs on the CPU
// (run.py:72-74, 221-223; consumed at surfaceNetStaticEdgeFilters.py:214-217, 258-264).
//
// One hop = dgnn_khop_count + (host reads the edge total to size the outputs) + dgnn_khop_expand.
// Semantics (PyG 2.0.2 sample_adj with size -1, see oracle/pyg_semantics.py): for the current target list
// n_id[0..n_t), emit all in-edges in (target order, plan order) -- plan order = ascending edge position, which
// for the reference's adjacency layout equals ascending source id, PyG's order -- relabelled to local ids where
// targets keep their positions and newly seen sources are appended in order of first appearance.
// `pos` (int32 [n_nodes], all -1 between batches) carries the global->local map across the hops of a batch.
#include "common.h"

namespace {

__global__ void k_khop_deg(const int32_t* __restrict__ rowptr, const int64_t* __restrict__ n_id, int64_t n_t, int first_hop,
                           int32_t* __restrict__ deg, int32_t* __restrict__ pos) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n_t; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t g = n_id[i];
        deg[i] = rowptr[g + 1] - rowptr[g];
        if (first_hop) pos[g] = (int32_t)i;  // later hops: already set when the node was discovered
    }
}

// first[g] = smallest slot at which an unseen source g appears
__global__ void k_khop_first(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ src, const int64_t* __restrict__ n_id,
                             int64_t n_t, const int32_t* __restrict__ off, const int32_t* __restrict__ pos,
                             int32_t* __restrict__ first) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n_t; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t g = n_id[i];
        const int b = rowptr[g], e = rowptr[g + 1], o = off[i];
        for (int k = b; k < e; ++k) {
            const int s = src[k];
            if (pos[s] < 0) atomicMin(&first[s], o + (k - b));
        }
    }
}

__global__ void k_khop_flags(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ src, const int64_t* __restrict__ n_id,
                             int64_t n_t, const int32_t* __restrict__ off, const int32_t* __restrict__ pos,
                             const int32_t* __restrict__ first, int32_t* __restrict__ flags) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n_t; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t g = n_id[i];
        const int b = rowptr[g], e = rowptr[g + 1], o = off[i];
        for (int k = b; k < e; ++k) {
            const int s = src[k], q = o + (k - b);
            flags[q] = (pos[s] < 0 && first[s] == q) ? 1 : 0;
        }
    }
}

__global__ void k_khop_emit(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ src, const int32_t* __restrict__ eid,
                            const int64_t* __restrict__ n_id, int64_t n_t, const int32_t* __restrict__ off,
                            const int32_t* __restrict__ pos, const int32_t* __restrict__ first, const int32_t* __restrict__ rank,
                            int64_t* __restrict__ e_src, int64_t* __restrict__ e_dst, int64_t* __restrict__ e_id,
                            int64_t* __restrict__ n_id_out) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n_t; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t g = n_id[i];
        const int b = rowptr[g], e = rowptr[g + 1], o = off[i];
        n_id_out[i] = g;
        for (int k = b; k < e; ++k) {
            const int s = src[k], q = o + (k - b);
            const int p = pos[s];
            int64_t local;
            if (p >= 0) {
                local = p;
            } else {
                const int fq = first[s];
                local = n_t + rank[fq];
                if (fq == q) n_id_out[local] = s;
            }
            e_src[q] = local;
            e_dst[q] = i;
            e_id[q] = eid[k];
        }
    }
}

// after emit: newly discovered nodes get their local id in `pos`, `first` goes back to INT_MAX
__global__ void k_khop_commit(const int64_t* __restrict__ n_id_out, int64_t n_t, int64_t n_all, int32_t* __restrict__ pos,
                              int32_t* __restrict__ first) {
    for (int64_t i = n_t + blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n_all; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t g = n_id_out[i];
        pos[g] = (int32_t)i;
        first[g] = INT32_MAX;
    }
}

__global__ void k_khop_reset(const int64_t* __restrict__ n_id, int64_t n, int32_t* __restrict__ pos) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) pos[n_id[i]] = -1;
}

__global__ void k_fill_i32(int32_t* p, int64_t n, int32_t v) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = v;
}

}  // namespace

int dgnn_exclusive_scan_i32(const int32_t* in, int64_t n, int32_t* out /*[n+1]*/, int32_t* sums_scratch, hipStream_t stream);  // plan.hip

extern "C" int dgnn_fill_i32(int32_t* p, int64_t n, int32_t value, void* stream) {
    DGNN_REQUIRE(n >= 0 && (n == 0 || p), DGNN_E_INVALID, "fill_i32: bad args");
    if (n == 0) return DGNN_OK;
    hipLaunchKernelGGL(k_fill_i32, dim3(dgnn_grid_cap(dgnn_cdiv(n, 256))), dim3(256), 0, (hipStream_t)stream, p, n, value);
    return dgnn_check_launch("fill_i32");
}

extern "C" int64_t dgnn_khop_scratch_elems(int64_t n_t, int64_t max_edges) {
    return (n_t + 1) + (max_edges + 1) + (max_edges + 1) + 2 * (dgnn_cdiv(n_t > max_edges ? n_t : max_edges, 2048) + 4);
}

// step 1: off[i] = exclusive scan of in-degrees of the targets, off[n_t] = number of block edges (device)
extern "C" int dgnn_khop_count(const int32_t* rowptr, const int64_t* n_id, int64_t n_t, int first_hop, int32_t* pos, int32_t* off,
                               int32_t* scratch, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DGNN_REQUIRE(n_t >= 0 && rowptr && pos && off && scratch && (n_t == 0 || n_id), DGNN_E_INVALID, "khop_count: bad args");
    int32_t* deg = scratch;
    if (n_t > 0)
        hipLaunchKernelGGL(k_khop_deg, dim3(dgnn_grid_cap(dgnn_cdiv(n_t, 256))), dim3(256), 0, stream, rowptr, n_id, n_t, first_hop, deg, pos);
    const int rc = dgnn_exclusive_scan_i32(deg, n_t, off, scratch + n_t + 1, stream);
    if (rc) return rc;
    return dgnn_check_launch("khop_count");
}

// step 2: emit the block.  n_edges = off[n_t] as read by the host; `first` must be all INT32_MAX, `pos` as described above.
extern "C" int dgnn_khop_expand(const int32_t* rowptr, const int32_t* src, const int32_t* eid, const int64_t* n_id, int64_t n_t,
                                const int32_t* off, int64_t n_edges, int32_t* pos, int32_t* first, int64_t* e_src, int64_t* e_dst,
                                int64_t* e_id, int64_t* n_id_out, int32_t* n_new_out, int32_t* scratch, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DGNN_REQUIRE(n_t >= 0 && n_edges >= 0 && rowptr && src && eid && off && pos && first && n_id_out && n_new_out && scratch,
                 DGNN_E_INVALID, "khop_expand: bad args");
    int32_t* flags = scratch;                 // [n_edges]
    int32_t* rank = flags + n_edges + 1;      // [n_edges + 1]
    int32_t* sums = rank + n_edges + 1;
    const dim3 grid(dgnn_grid_cap(dgnn_cdiv(n_t > 0 ? n_t : 1, 256))), block(256);
    if (n_t > 0 && n_edges > 0) {
        hipLaunchKernelGGL(k_khop_first, grid, block, 0, stream, rowptr, src, n_id, n_t, off, pos, first);
        hipLaunchKernelGGL(k_khop_flags, grid, block, 0, stream, rowptr, src, n_id, n_t, off, pos, first, flags);
    }
    const int rc = dgnn_exclusive_scan_i32(flags, n_edges, rank, sums, stream);
    if (rc) return rc;
    (void)hipMemcpyAsync(n_new_out, rank + n_edges, sizeof(int32_t), hipMemcpyDeviceToDevice, stream);
    if (n_t > 0)
        hipLaunchKernelGGL(k_khop_emit, grid, block, 0, stream, rowptr, src, eid, n_id, n_t, off, pos, first, rank, e_src, e_dst, e_id,
                           n_id_out);
    return dgnn_check_launch("khop_expand");
}

// step 3 (after the host read n_new): record the new nodes' local ids for the next hop
extern "C" int dgnn_khop_commit(const int64_t* n_id_out, int64_t n_t, int64_t n_all, int32_t* pos, int32_t* first, void* stream) {
    DGNN_REQUIRE(n_all >= n_t && n_t >= 0 && pos && first && (n_all == 0 || n_id_out), DGNN_E_INVALID, "khop_commit: bad args");
    if (n_all > n_t)
        hipLaunchKernelGGL(k_khop_commit, dim3(dgnn_grid_cap(dgnn_cdiv(n_all - n_t, 256))), dim3(256), 0, (hipStream_t)stream, n_id_out,
                           n_t, n_all, pos, first);
    return dgnn_check_launch("khop_commit");
}

// end of batch: pos[n_id[*]] = -1
extern "C" int dgnn_khop_reset(const int64_t* n_id, int64_t n, int32_t* pos, void* stream) {
    DGNN_REQUIRE(n >= 0 && pos && (n == 0 || n_id), DGNN_E_INVALID, "khop_reset: bad args");
    if (n > 0) hipLaunchKernelGGL(k_khop_reset, dim3(dgnn_grid_cap(dgnn_cdiv(n, 256))), dim3(256), 0, (hipStream_t)stream, n_id, n, pos);
    return dgnn_check_launch("khop_reset");
}
