// Logits -> labels -> interface facets on the device: the step right after the hot path,
// reference processing/generate_mesh.py:75 (labels of the finite cells) and :93-105 (two interpreter loops over
// all facets: infinite neighbour = one extra OUTSIDE cell; a facet is on the surface iff its two cells differ).
// The optional integer graph cut between the two (:84-91, third-party gco) stays on the CPU.
#include "common.h"

int dgnn_exclusive_scan_i32(const int32_t* in, int64_t n, int32_t* out, int32_t* sums_scratch, hipStream_t stream);  // plan.hip

namespace {

// label = argmax over the class scores of a row (log_softmax is monotone, so argmax of logits; ties -> class 0)
__global__ void k_argmax_rows(const float* __restrict__ logits, int64_t ld, int64_t n, int c, int32_t* __restrict__ labels) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float* r = logits + i * ld;
        int best = 0;
        float bv = r[0];
        for (int k = 1; k < c; ++k)
            if (r[k] > bv) { bv = r[k]; best = k; }
        labels[i] = best;
    }
}

__global__ void k_compact(const int32_t* __restrict__ values, const int32_t* __restrict__ keep, const int32_t* __restrict__ rank,
                          int64_t n, int32_t* __restrict__ out) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        if (keep[i]) out[rank[i]] = values ? values[i] : (int32_t)i;
}

__global__ void k_not(const int32_t* __restrict__ in, int64_t n, int32_t* __restrict__ out) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = in[i] == 0;
}

__global__ void k_interface_flags(const int32_t* __restrict__ nfacets, const int32_t* __restrict__ labels_finite, int64_t n_facets,
                                  int32_t* __restrict__ flags) {
    for (int64_t f = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; f < n_facets; f += (int64_t)gridDim.x * blockDim.x) {
        const int c0 = nfacets[2 * f], c1 = nfacets[2 * f + 1];
        const int l0 = c0 < 0 ? 1 : labels_finite[c0], l1 = c1 < 0 ? 1 : labels_finite[c1];  // -1 = the infinite cell = outside
        flags[f] = l0 != l1;
    }
}

}  // namespace

extern "C" int dgnn_argmax_rows(const float* logits, int64_t ld, int64_t n, int c, int32_t* labels, void* stream) {
    DGNN_REQUIRE(n >= 0 && c > 0 && (n == 0 || (logits && labels)), DGNN_E_INVALID, "argmax_rows: bad args");
    if (n == 0) return DGNN_OK;
    hipLaunchKernelGGL(k_argmax_rows, dim3(dgnn_grid_cap(dgnn_cdiv(n, 256))), dim3(256), 0, (hipStream_t)stream, logits, ld, n, c, labels);
    return dgnn_check_launch("argmax_rows");
}

extern "C" int64_t dgnn_compact_scratch_elems(int64_t n) { return (n + 1) + (n + 1) + dgnn_cdiv(n, 2048) + 4; }

// out = values[keep != 0] (values == NULL: the indices themselves), order preserved; *count_out = number kept (device).
// invert != 0 keeps where keep == 0 (finite cells = infinite flag clear).
extern "C" int dgnn_compact_i32(const int32_t* values, const int32_t* keep, int invert, int64_t n, int32_t* out, int32_t* count_out,
                                int32_t* scratch, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DGNN_REQUIRE(n >= 0 && count_out && scratch && (n == 0 || (keep && out)), DGNN_E_INVALID, "compact_i32: bad args");
    int32_t* k = scratch;            // [n+1] normalised keep flags
    int32_t* rank = k + n + 1;       // [n+1]
    int32_t* sums = rank + n + 1;
    const dim3 grid(dgnn_grid_cap(dgnn_cdiv(n > 0 ? n : 1, 256))), block(256);
    if (n > 0) {
        if (invert) hipLaunchKernelGGL(k_not, grid, block, 0, stream, keep, n, k);
        else (void)hipMemcpyAsync(k, keep, sizeof(int32_t) * n, hipMemcpyDeviceToDevice, stream);
    }
    const int rc = dgnn_exclusive_scan_i32(k, n, rank, sums, stream);
    if (rc) return rc;
    (void)hipMemcpyAsync(count_out, rank + n, sizeof(int32_t), hipMemcpyDeviceToDevice, stream);
    if (n > 0) hipLaunchKernelGGL(k_compact, grid, block, 0, stream, values, k, rank, n, out);
    return dgnn_check_launch("compact_i32");
}

extern "C" int dgnn_interface_flags(const int32_t* nfacets, const int32_t* labels_finite, int64_t n_facets, int32_t* flags,
                                    void* stream) {
    DGNN_REQUIRE(n_facets >= 0 && (n_facets == 0 || (nfacets && labels_finite && flags)), DGNN_E_INVALID, "interface_flags: bad args");
    if (n_facets == 0) return DGNN_OK;
    hipLaunchKernelGGL(k_interface_flags, dim3(dgnn_grid_cap(dgnn_cdiv(n_facets, 256))), dim3(256), 0, (hipStream_t)stream, nfacets,
                       labels_finite, n_facets, flags);
    return dgnn_check_launch("interface_flags");
}
