// Ingest-time cell locality order (SURVEY 7 step 2 "optional locality permutation ... with its inverse applied on output").
//
// A scene leaves the reference's CGAL front end in insertion order: the 4 neighbours of a cell are tens of thousands of rows away
// (Ignatius), so every gathered feature row of the conv layers misses L2 (bench `random_cell_order`: -19 %).  The loader
// (dgnn_amd/processing/data.py, where the reference builds edge_index: processing/data.py:434-438) therefore relabels the cells once per scene:
//   * dgnn_cell_centroids_3dt   centroid of every cell from <scene>_3dt.npz (vertices, tetrahedra of the FINITE cells in file order --
//                               generate_mesh.py:78-81 relies on the same correspondence); an infinite cell sits on its finite neighbour
//   * dgnn_cell_order_morton    order = cells sorted by the 48-bit Morton code of their centroid (16 bits per axis over the bounding box),
//                               stable LSD radix sort written here (8-bit digits: per-tile histograms, one scan, ranked scatter)
//   * dgnn_cell_order_bfs       no coordinates: breadth-first (Cuthill-McKee style) order of the adjacency itself, level by level, every level
//                               in the order a serial queue would produce (first discoverer wins, neighbours in slot order): deterministic
//   * dgnn_reorder_edges_ref    the relabelled adjacency in the reference layout (4 rows per cell, interleaved (src, dst) int64 pairs = the
//                               transposed view the plan builder's four-lanes-per-cell pass takes) + the old row of every new edge row
// Index work only: results are checked bit for bit against numpy (tests/test_gpu_reorder.py).
#include "common.h"

int dgnn_exclusive_scan_i32(const int32_t* in, int64_t n, int32_t* out, int32_t* sums_scratch, hipStream_t stream);   // plan.hip

namespace {

constexpr int RX_THREADS = 256;
constexpr int RX_ITEMS = 8;
constexpr int RX_TILE = RX_THREADS * RX_ITEMS;   // elements per block and pass
constexpr int BB_BLOCKS = 256;

// ---- centroids ------------------------------------------------------------------------------------------------------------
__global__ void k_finite_flags(const int32_t* __restrict__ infinite, int64_t n, int32_t* __restrict__ flag) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) flag[i] = infinite[i] == 0;
}

__device__ __forceinline__ void tet_centroid(const float* __restrict__ v, int64_t nv, const int32_t* __restrict__ tet, float* c, bool& bad) {
    float s[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int64_t p = tet[k];
        if (p < 0 || p >= nv) { bad = true; p = 0; }
#pragma unroll
        for (int a = 0; a < 3; ++a) s[a] += v[3 * p + a];
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) c[a] = 0.25f * s[a];
}

// finite cell i <-> tetrahedra[fin_rank[i]]; an infinite cell takes the centroid of the first finite cell among its 4 neighbours (rows 4i..4i+3
// of the adjacency; CGAL wires every infinite cell to exactly one finite cell), the origin if it has none
__global__ void k_centroids_3dt(const float* __restrict__ verts, int64_t nv, const int32_t* __restrict__ tets, int64_t n_fin,
                                const int32_t* __restrict__ infinite, const int32_t* __restrict__ fin_rank, const int64_t* __restrict__ dst,
                                int64_t sc, int64_t n, float* __restrict__ cent, int32_t* aflag) {
    bool bad = false;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t cell = i;
        bool have = infinite[i] == 0;
        if (!have) {
            for (int k = 0; k < 4 && !have; ++k) {
                const int64_t d = dst[(4 * i + k) * sc];
                if (d >= 0 && d < n && infinite[d] == 0) { cell = d; have = true; }
                bad |= d < 0 || d >= n;
            }
        }
        float c[3] = {0.f, 0.f, 0.f};
        if (have) {
            const int64_t r = fin_rank[cell];
            if (r < n_fin) tet_centroid(verts, nv, tets + 4 * r, c, bad); else bad = true;
        }
        cent[3 * i] = c[0]; cent[3 * i + 1] = c[1]; cent[3 * i + 2] = c[2];
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) dgnn_raise_async(aflag, DGNN_ASYNC_OTHER_RANGE);
}

// ---- Morton keys ----------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_bbox_partial(const float* __restrict__ cent, int64_t n, float* __restrict__ partial) {
    __shared__ float red[6][256];
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float v = cent[3 * i + a];
            if (v == v) { lo[a] = fminf(lo[a], v); hi[a] = fmaxf(hi[a], v); }   // NaN coordinates do not take part
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) { red[a][threadIdx.x] = lo[a]; red[3 + a][threadIdx.x] = hi[a]; }
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                red[a][threadIdx.x] = fminf(red[a][threadIdx.x], red[a][threadIdx.x + s]);
                red[3 + a][threadIdx.x] = fmaxf(red[3 + a][threadIdx.x], red[3 + a][threadIdx.x + s]);
            }
        }
        __syncthreads();
    }
    if (threadIdx.x < 6) partial[blockIdx.x * 6 + threadIdx.x] = red[threadIdx.x][0];
}

__global__ void k_bbox_final(const float* __restrict__ partial, int nblk, float* __restrict__ bbox) {
    const int a = threadIdx.x;
    if (a >= 6) return;
    float v = partial[a];
    for (int b = 1; b < nblk; ++b) v = a < 3 ? fminf(v, partial[b * 6 + a]) : fmaxf(v, partial[b * 6 + a]);
    bbox[a] = v;
}

__device__ __forceinline__ uint64_t spread16(uint32_t v) {   // bit i of the low 16 -> bit 3i
    uint64_t x = v & 0xFFFFu;
    x = (x | (x << 16)) & 0x0000FF0000FFull;
    x = (x | (x << 8)) & 0x00F00F00F00Full;
    x = (x | (x << 4)) & 0x0C30C30C30C3ull;
    x = (x | (x << 2)) & 0x249249249249ull;
    return x;
}

__global__ void k_morton_keys(const float* __restrict__ cent, int64_t n, const float* __restrict__ bbox, uint64_t* __restrict__ keys,
                              int32_t* __restrict__ vals) {
    float lo[3], inv[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        lo[a] = bbox[a];
        const float ext = bbox[3 + a] - bbox[a];
        inv[a] = ext > 0.f ? 65535.0f / ext : 0.f;
    }
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        uint64_t k = 0;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float v = cent[3 * i + a];
            float q = (v - lo[a]) * inv[a];
            q = q == q ? fminf(fmaxf(q, 0.f), 65535.f) : 65535.f;    // NaN -> the far corner
            k |= spread16((uint32_t)q) << a;
        }
        keys[i] = k;
        vals[i] = (int32_t)i;
    }
}

// ---- stable LSD radix sort of (u64 key, i32 value) pairs, 8-bit digits --------------------------------------------------------
// hist[d * nblk + b] = number of elements of tile b whose digit is d  (digit-major: one exclusive scan over the whole array yields, for
// every (digit, tile), the first output position of that tile's elements with that digit)
__global__ void __launch_bounds__(RX_THREADS) k_radix_hist(const uint64_t* __restrict__ keys, int64_t n, int shift, int nblk,
                                                           int32_t* __restrict__ hist) {
    __shared__ int32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * RX_TILE;
#pragma unroll
    for (int it = 0; it < RX_ITEMS; ++it) {
        const int64_t e = base + it * RX_THREADS + threadIdx.x;
        if (e < n) atomicAdd(&h[(int)((keys[e] >> shift) & 255u)], 1);
    }
    __syncthreads();
    hist[(int64_t)threadIdx.x * nblk + blockIdx.x] = h[threadIdx.x];
}

// Every element's output position = scanned[(digit, tile)] + its rank among the tile's elements with the same digit, ranks taken in element
// order (item, wavefront, lane): the pass is stable.  Inside a wavefront the rank comes from the mask of lanes holding the same digit.
__global__ void __launch_bounds__(RX_THREADS) k_radix_scatter(const uint64_t* __restrict__ keys, const int32_t* __restrict__ vals, int64_t n,
                                                              int shift, int nblk, const int32_t* __restrict__ scanned,
                                                              uint64_t* __restrict__ keys_out, int32_t* __restrict__ vals_out) {
    __shared__ int32_t run[256];                       // next free position per digit
    __shared__ int32_t cnt[RX_THREADS / 64][256];      // this item's count per wavefront and digit
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    run[threadIdx.x] = scanned[(int64_t)threadIdx.x * nblk + blockIdx.x];
#pragma unroll
    for (int i = 0; i < RX_THREADS / 64; ++i) cnt[i][threadIdx.x] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * RX_TILE;
    for (int it = 0; it < RX_ITEMS; ++it) {
        const int64_t e = base + it * RX_THREADS + threadIdx.x;
        const bool live = e < n;
        const uint64_t key = live ? keys[e] : 0;
        const int d = live ? (int)((key >> shift) & 255u) : 256;   // 256: matches no live lane
        uint64_t same = __ballot(live);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const uint64_t m = __ballot((d >> b) & 1);
            same &= ((d >> b) & 1) ? m : ~m;
        }
        const int below = __popcll(same & ((1ull << lane) - 1ull));
        if (live && below == 0) cnt[w][d] = __popcll(same);      // the group's first lane publishes the group size
        __syncthreads();
        if (live) {
            int pos = run[d] + below;
            for (int i = 0; i < w; ++i) pos += cnt[i][d];
            keys_out[pos] = key;
            vals_out[pos] = vals[e];
        }
        __syncthreads();
        {
            int add = 0;
#pragma unroll
            for (int i = 0; i < RX_THREADS / 64; ++i) { add += cnt[i][threadIdx.x]; cnt[i][threadIdx.x] = 0; }
            run[threadIdx.x] += add;
        }
        __syncthreads();
    }
}

__global__ void k_order_to_rank(const int32_t* __restrict__ order, int64_t n, int32_t* __restrict__ order_out, int32_t* __restrict__ rank) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int32_t o = order[i];
        if (order_out) order_out[i] = o;
        rank[o] = (int32_t)i;
    }
}

// ---- breadth-first order ------------------------------------------------------------------------------------------------
// state (device int32): [0] head, [1] tail (order[head:tail] = current frontier), [2] next seed candidate, [3] frontier size handed to the host
// claim[v] = 4p + k of the first (queue order) frontier slot that reaches the unvisited cell v; INT32_MAX = unclaimed; -1 = already in `order`
__global__ void k_bfs_init(int32_t* __restrict__ claim, int64_t n, int32_t* __restrict__ state) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) claim[i] = INT32_MAX;
    if (blockIdx.x == 0 && threadIdx.x == 0) { state[0] = 0; state[1] = 0; state[2] = 0; state[3] = 0; }
}

// starts a new component: the lowest-numbered cell not yet ordered becomes a frontier of one (one block walks the seed cursor 256 cells at a time;
// O(n / 256) steps over a whole run)
__global__ void __launch_bounds__(256) k_bfs_seed(int32_t* __restrict__ claim, int64_t n, int32_t* __restrict__ order, int32_t* __restrict__ state,
                                                  volatile int32_t* host_word) {
    __shared__ int32_t found;
    int64_t s = state[2];
    if (threadIdx.x == 0) found = INT32_MAX;
    __syncthreads();
    while (s < n) {
        const int64_t i = s + threadIdx.x;
        if (i < n && claim[i] != -1) atomicMin(&found, (int32_t)i);
        __syncthreads();
        if (found != INT32_MAX) break;
        s += 256;
    }
    if (threadIdx.x != 0) return;
    int32_t fs = 0;
    if (found != INT32_MAX) {
        const int32_t t = state[1];
        order[t] = found;
        claim[found] = -1;
        state[0] = t;
        state[1] = t + 1;
        state[2] = found + 1;
        fs = 1;
    } else {
        state[2] = (int32_t)n;
    }
    state[3] = fs;
    if (host_word) { host_word[0] = fs; host_word[1] = state[1]; }
}

__global__ void k_bfs_claim(const int64_t* __restrict__ dst, int64_t sc, int64_t n, const int32_t* __restrict__ order, const int32_t* __restrict__ state,
                            int32_t* __restrict__ claim, int32_t* aflag) {
    const int32_t head = state[0], fs = state[1] - state[0];
    bool bad = false;
    for (int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; c < 4 * (int64_t)fs; c += (int64_t)gridDim.x * blockDim.x) {
        const int64_t u = order[head + (c >> 2)];
        const int64_t v = dst[(4 * u + (c & 3)) * sc];
        if (v < 0 || v >= n) { bad = true; continue; }
        if (claim[v] > 0) atomicMin(&claim[v], (int32_t)c + 1);   // stored +1 so that candidate 0 differs from nothing; -1 stays -1 (min keeps it)
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) dgnn_raise_async(aflag, DGNN_ASYNC_OTHER_RANGE);
}

// flag[c] = 1 where candidate c won its cell
__global__ void k_bfs_flags(const int64_t* __restrict__ dst, int64_t sc, int64_t n, const int32_t* __restrict__ order, const int32_t* __restrict__ state,
                            const int32_t* __restrict__ claim, int32_t* __restrict__ flag) {
    const int32_t head = state[0], fs = state[1] - state[0];
    for (int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; c < 4 * (int64_t)fs; c += (int64_t)gridDim.x * blockDim.x) {
        const int64_t u = order[head + (c >> 2)];
        const int64_t v = dst[(4 * u + (c & 3)) * sc];
        flag[c] = (v >= 0 && v < n && claim[v] == (int32_t)c + 1) ? 1 : 0;
    }
}

__global__ void k_bfs_append(const int64_t* __restrict__ dst, int64_t sc, const int32_t* __restrict__ flag, const int32_t* __restrict__ pos,
                             int32_t* __restrict__ order, const int32_t* __restrict__ state, int32_t* __restrict__ claim) {
    const int32_t head = state[0], tail = state[1], fs = tail - head;
    for (int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; c < 4 * (int64_t)fs; c += (int64_t)gridDim.x * blockDim.x) {
        if (!flag[c]) continue;
        const int64_t u = order[head + (c >> 2)];
        const int32_t v = (int32_t)dst[(4 * u + (c & 3)) * sc];
        order[tail + pos[c]] = v;
        claim[v] = -1;
    }
}

// head <- tail, tail <- tail + number of cells appended (pos[4 fs] = the scan's total); the new frontier size goes to the host word
__global__ void k_bfs_advance(int32_t* __restrict__ state, const int32_t* __restrict__ pos, volatile int32_t* host_word) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const int32_t head = state[0], tail = state[1];
    const int32_t added = pos[4 * (tail - head)];
    state[0] = tail;
    state[1] = tail + added;
    state[3] = added;
    if (host_word) { host_word[0] = added; host_word[1] = tail + added; }
}

// ---- relabelled adjacency -------------------------------------------------------------------------------------------------
__global__ void k_reorder_edges_ref(const int64_t* __restrict__ src, const int64_t* __restrict__ dst, int64_t sc, int64_t n,
                                    const int32_t* __restrict__ order, const int32_t* __restrict__ rank, int64_t* __restrict__ pairs,
                                    int32_t* __restrict__ edge_rows, int32_t* aflag) {
    bool bad = false;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < 4 * n; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = e >> 2, k = e & 3;
        const int64_t o = order[i];
        const int64_t eo = 4 * o + k;
        const int64_t s = src[eo * sc], d = dst[eo * sc];
        bad |= s != o || d < 0 || d >= n;          // not the reference layout (4 rows per cell, row 4t+k leaves cell t)
        pairs[2 * e] = i;
        pairs[2 * e + 1] = (d >= 0 && d < n) ? rank[d] : 0;
        edge_rows[e] = (int32_t)eo;
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) dgnn_raise_async(aflag, DGNN_ASYNC_OTHER_RANGE);
}

struct HostWord {   // pinned, device-visible: the frontier size of the BFS level just finished
    int32_t* host = nullptr;
    int32_t* dev = nullptr;
};

}  // namespace

// scratch (int32): flag[n] | fin_rank[n + 1] | scan sums
extern "C" int64_t dgnn_cell_centroids_scratch_elems(int64_t n) { return n < 0 ? 0 : 2 * n + 1 + dgnn_cdiv(n, 2048) + 4; }

extern "C" int dgnn_cell_centroids_3dt(const float* vertices, int64_t n_vertices, const int32_t* tetrahedra, int64_t n_finite, const int32_t* infinite,
                                       const int64_t* edge_index, int64_t stride_row, int64_t stride_col, int64_t n, float* centroids, int32_t* scratch,
                                       void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DGNN_REQUIRE(n >= 0 && n_vertices >= 0 && n_finite >= 0 && 4 * n < INT32_MAX, DGNN_E_INVALID, "cell_centroids_3dt: bad sizes");
    if (n == 0) return DGNN_OK;
    DGNN_REQUIRE(vertices && tetrahedra && infinite && edge_index && centroids && scratch && n_vertices > 0, DGNN_E_INVALID, "cell_centroids_3dt: null pointer");
    int32_t* flag = scratch;
    int32_t* fin_rank = flag + n;
    int32_t* sums = fin_rank + n + 1;
    const dim3 g(dgnn_grid_cap(dgnn_cdiv(n, 256)));
    hipLaunchKernelGGL(k_finite_flags, g, dim3(256), 0, stream, infinite, n, flag);
    const int rc = dgnn_exclusive_scan_i32(flag, n, fin_rank, sums, stream);
    if (rc != DGNN_OK) return rc;
    hipLaunchKernelGGL(k_centroids_3dt, g, dim3(256), 0, stream, vertices, n_vertices, tetrahedra, n_finite, infinite, fin_rank, edge_index + stride_row,
                       stride_col, n, centroids, dgnn_async_flag_dev());
    return dgnn_check_launch("cell_centroids_3dt");
}

// scratch (int32 units): keys 2 x n u64 | vals 2 x n | hist 256 x nblk + 2 | scanned hist | scan sums | bbox partials
extern "C" int64_t dgnn_cell_order_morton_scratch_elems(int64_t n) {
    if (n < 0) return 0;
    const int64_t nblk = dgnn_cdiv(n > 0 ? n : 1, RX_TILE);
    return 4 * n + 2 * n + 2 * (256 * nblk + 2) + (dgnn_cdiv(256 * nblk, 2048) + 4) + BB_BLOCKS * 6 + 8 + 4;
}

extern "C" int dgnn_cell_order_morton(const float* centroids, int64_t n, int32_t* order, int32_t* rank, int32_t* scratch, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DGNN_REQUIRE(n >= 0 && n < INT32_MAX, DGNN_E_INVALID, "cell_order_morton: bad size");
    if (n == 0) return DGNN_OK;
    DGNN_REQUIRE(centroids && order && rank && scratch && ((uintptr_t)scratch % 8) == 0, DGNN_E_INVALID, "cell_order_morton: null / unaligned pointer");
    const int nblk = (int)dgnn_cdiv(n, RX_TILE);
    uint64_t* keys[2] = {reinterpret_cast<uint64_t*>(scratch), reinterpret_cast<uint64_t*>(scratch) + n};
    int32_t* vals[2] = {scratch + 4 * n, scratch + 5 * n};
    int32_t* hist = scratch + 6 * n;
    int32_t* scanned = hist + 256 * (int64_t)nblk + 2;
    int32_t* sums = scanned + 256 * (int64_t)nblk + 2;
    float* partial = reinterpret_cast<float*>(sums + dgnn_cdiv(256 * (int64_t)nblk, 2048) + 4);
    float* bbox = partial + BB_BLOCKS * 6;
    const int bb = (int)(dgnn_cdiv(n, 256) < BB_BLOCKS ? dgnn_cdiv(n, 256) : BB_BLOCKS);
    hipLaunchKernelGGL(k_bbox_partial, dim3(bb), dim3(256), 0, stream, centroids, n, partial);
    hipLaunchKernelGGL(k_bbox_final, dim3(1), dim3(64), 0, stream, partial, bb, bbox);
    hipLaunchKernelGGL(k_morton_keys, dim3(dgnn_grid_cap(dgnn_cdiv(n, 256))), dim3(256), 0, stream, centroids, n, bbox, keys[0], vals[0]);
    int cur = 0;
    for (int shift = 0; shift < 48; shift += 8) {
        hipLaunchKernelGGL(k_radix_hist, dim3(nblk), dim3(RX_THREADS), 0, stream, keys[cur], n, shift, nblk, hist);
        const int rc = dgnn_exclusive_scan_i32(hist, 256 * (int64_t)nblk, scanned, sums, stream);
        if (rc != DGNN_OK) return rc;
        hipLaunchKernelGGL(k_radix_scatter, dim3(nblk), dim3(RX_THREADS), 0, stream, keys[cur], vals[cur], n, shift, nblk, scanned, keys[cur ^ 1], vals[cur ^ 1]);
        cur ^= 1;
    }
    hipLaunchKernelGGL(k_order_to_rank, dim3(dgnn_grid_cap(dgnn_cdiv(n, 256))), dim3(256), 0, stream, vals[cur], n, order, rank);
    return dgnn_check_launch("cell_order_morton");
}

// scratch (int32): claim[n] | flag[4n] | pos[4n + 1] | scan sums | state[4]
extern "C" int64_t dgnn_cell_order_bfs_scratch_elems(int64_t n) { return n < 0 ? 0 : n + 4 * n + 4 * n + 1 + dgnn_cdiv(4 * n, 2048) + 4 + 4; }

extern "C" int dgnn_cell_order_bfs(const int64_t* edge_index, int64_t stride_row, int64_t stride_col, int64_t n, int32_t* order, int32_t* rank,
                                   int32_t* scratch, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DGNN_REQUIRE(n >= 0 && 4 * n + 1 < INT32_MAX, DGNN_E_INVALID, "cell_order_bfs: bad size");
    if (n == 0) return DGNN_OK;
    DGNN_REQUIRE(edge_index && order && rank && scratch, DGNN_E_INVALID, "cell_order_bfs: null pointer");
    static thread_local HostWord hw;
    if (!hw.host) {
        if (hipHostMalloc(reinterpret_cast<void**>(&hw.host), 2 * sizeof(int32_t), hipHostMallocMapped) != hipSuccess ||
            hipHostGetDevicePointer(reinterpret_cast<void**>(&hw.dev), hw.host, 0) != hipSuccess) {
            (void)hipGetLastError();
            hw.host = nullptr;
            dgnn_set_error("cell_order_bfs: pinned host word unavailable");
            return DGNN_E_LAUNCH;
        }
    }
    const int64_t* dst = edge_index + stride_row;
    int32_t* claim = scratch;
    int32_t* flag = claim + n;
    int32_t* pos = flag + 4 * n;
    int32_t* sums = pos + 4 * n + 1;
    int32_t* state = sums + dgnn_cdiv(4 * n, 2048) + 4;
    int32_t* aflag = dgnn_async_flag_dev();
    hipLaunchKernelGGL(k_bfs_init, dim3(dgnn_grid_cap(dgnn_cdiv(n, 256))), dim3(256), 0, stream, claim, n, state);
    int64_t done = 0;
    while (done < n) {
        hipLaunchKernelGGL(k_bfs_seed, dim3(1), dim3(256), 0, stream, claim, n, order, state, hw.dev);
        if (hipStreamSynchronize(stream) != hipSuccess) return dgnn_check_launch("cell_order_bfs");
        int64_t fs = hw.host[0];
        done = hw.host[1];
        while (fs > 0) {
            const dim3 g(dgnn_grid_cap(dgnn_cdiv(4 * fs, 256)));
            hipLaunchKernelGGL(k_bfs_claim, g, dim3(256), 0, stream, dst, stride_col, n, order, state, claim, aflag);
            hipLaunchKernelGGL(k_bfs_flags, g, dim3(256), 0, stream, dst, stride_col, n, order, state, claim, flag);
            const int rc = dgnn_exclusive_scan_i32(flag, 4 * fs, pos, sums, stream);
            if (rc != DGNN_OK) return rc;
            hipLaunchKernelGGL(k_bfs_append, g, dim3(256), 0, stream, dst, stride_col, flag, pos, order, state, claim);
            hipLaunchKernelGGL(k_bfs_advance, dim3(1), dim3(64), 0, stream, state, pos, hw.dev);
            if (hipStreamSynchronize(stream) != hipSuccess) return dgnn_check_launch("cell_order_bfs");
            fs = hw.host[0];
            done = hw.host[1];
        }
        if (fs == 0 && done >= n) break;
    }
    hipLaunchKernelGGL(k_order_to_rank, dim3(dgnn_grid_cap(dgnn_cdiv(n, 256))), dim3(256), 0, stream, order, n, (int32_t*)nullptr, rank);
    return dgnn_check_launch("cell_order_bfs");
}

extern "C" int dgnn_reorder_edges_ref(const int64_t* edge_index, int64_t stride_row, int64_t stride_col, int64_t n, const int32_t* order,
                                      const int32_t* rank, int64_t* pairs_out, int32_t* edge_rows_out, void* stream_) {
    DGNN_REQUIRE(n >= 0 && 4 * n < INT32_MAX, DGNN_E_INVALID, "reorder_edges_ref: bad size");
    if (n == 0) return DGNN_OK;
    DGNN_REQUIRE(edge_index && order && rank && pairs_out && edge_rows_out, DGNN_E_INVALID, "reorder_edges_ref: null pointer");
    hipLaunchKernelGGL(k_reorder_edges_ref, dim3(dgnn_grid_cap(dgnn_cdiv(4 * n, 256))), dim3(256), 0, (hipStream_t)stream_, edge_index,
                       edge_index + stride_row, stride_col, n, order, rank, pairs_out, edge_rows_out, dgnn_async_flag_dev());
    return dgnn_check_launch("reorder_edges_ref");
}
