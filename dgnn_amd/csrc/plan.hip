// Graph plan: stable destination-sorted CSR of the reference's int64 [2,E] edge_index.
//
// Reference semantics being reproduced: torch_scatter's CPU scatter_add_
// accumulates, for every destination, in ascending edge position.  The plan stores exactly that
// order, so the aggregation kernels are deterministic and can be compared with the oracle.
//
// Pipeline (all on the caller's stream, no host sync, no allocation):
//   count   deg[key[e]]++                         (int atomics; 4 MB of counters at 1M tets: L2)
//   scan    rowptr = exclusive_scan(deg)          (3 small kernels)
//   fill    slot = rowptr[d] + --deg[d]; tmp[slot] = e     (unordered inside the segment)
//   emit    sort every segment by edge position   (4-regular tet graphs: a 5-comparator network in
//           registers; up to 32: insertion sort; longer segments: block-wide rank sort) and write
//           eid[k], other[k] = edge_index[other_row][eid[k]].
#include <stdarg.h>

#include <stdlib.h>

#include "common.h"
#include <mutex>

static thread_local char g_err[512] = "ok";
void dgnn_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* dgnn_last_error_string(void) { return g_err; }

static int32_t* g_async_host = nullptr;
static int32_t* g_async_dev = nullptr;
int32_t* dgnn_async_flag_dev() {
    // first use may come from the Python thread and from the library's block-builder thread at once
    static std::once_flag once;
    std::call_once(once, [] {
        void* h = nullptr;
        if (hipHostMalloc(&h, 64, hipHostMallocPortable | hipHostMallocMapped) == hipSuccess && h) {
            *reinterpret_cast<volatile int32_t*>(h) = 0;
            void* d = nullptr;
            if (hipHostGetDevicePointer(&d, h, 0) == hipSuccess && d) {
                g_async_host = reinterpret_cast<int32_t*>(h);
                g_async_dev = reinterpret_cast<int32_t*>(d);
            } else {
                (void)hipHostFree(h);
            }
        }
        (void)hipGetLastError();
    });
    return g_async_dev;
}
extern "C" int dgnn_poll_async_error(void) {
    if (!g_async_host) return DGNN_OK;
    const int32_t bits = __atomic_exchange_n(g_async_host, 0, __ATOMIC_ACQ_REL);
    if (bits == 0) return DGNN_OK;
    dgnn_set_error("asynchronous kernel error 0x%x:%s%s (edges with such endpoints were skipped; the reference's scatter raises an "
                   "index error for them)", bits, (bits & DGNN_ASYNC_KEY_RANGE) ? " edge_index sort-key endpoint out of range;" : "",
                   (bits & DGNN_ASYNC_OTHER_RANGE) ? " edge_index other endpoint out of range;" : "");
    if (bits & DGNN_ASYNC_DUPLICATE)
        dgnn_set_error("asynchronous kernel error 0x%x: edge_chain: an edge id occurs more than once in e_id_cur / e_id_next of a block; the sparse chaining "
                       "keeps one position per edge (the reference's index / scatter autograd would sum the duplicates' gradients) -- use DGNN_CHAIN_DENSE=1 "
                       "for such blocks", bits);
    return DGNN_E_INDEX;
}
extern "C" int dgnn_version(void) { return DGNN_VERSION; }

namespace {

constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 8;  // per thread
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;

// deg[0..n) = 0 (n = 0: the persistent fallback zeroes it itself), *big_count = 0, need = (a, b, barrier counter 0): one launch
__global__ void k_plan_init(int32_t* deg, int64_t n, int32_t* big_count, int32_t* need, int32_t a, int32_t b) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) deg[i] = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) { *big_count = 0; need[0] = a; need[1] = b; need[2] = 0; }
}

__global__ void k_zero_i32(int32_t* p, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = 0;
}

// `need` (may be NULL): device flags written by the fast-path kernels -- when one of them already produced the plan the
// generic kernels return at once.  `sc` = element stride between consecutive edges of a row.
// need[0] = 1: the "already grouped by key" pass failed; need[1] = 1: the "reference layout" pass failed or was not tried.
__device__ __forceinline__ bool plan_skip(const int32_t* need) { return need != nullptr && (need[0] == 0 || need[1] == 0); }

// An edge whose key is outside [0, n_key) is left out of the plan (count and fill agree on that) and reported through the
// asynchronous error word: torch's scatter raises an index error there, silently corrupting scratch memory is not an option.
// The generic builder's passes are device functions of (block id, grid size): each is a kernel of its own when no fast path is tried, and
// a phase of the single persistent k_plan_fallback (grid-wide barriers in between) when it only stands by behind a fast path.
__device__ __forceinline__ void plan_count(int64_t bid, int64_t nblk, const int64_t* __restrict__ key, int64_t sc, int64_t E, int64_t n_key,
                                           int32_t* __restrict__ deg, int32_t* __restrict__ aflag) {
    bool bad = false;
    for (int64_t e = bid * (int64_t)blockDim.x + threadIdx.x; e < E; e += nblk * blockDim.x) {
        const int64_t k = key[e * sc];
        if (k < 0 || k >= n_key) { bad = true; continue; }
        atomicAdd(&deg[k], 1);
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) dgnn_raise_async(aflag, DGNN_ASYNC_KEY_RANGE);
}
__global__ void k_plan_count(const int64_t* __restrict__ key, int64_t sc, int64_t E, int64_t n_key, int32_t* __restrict__ deg,
                             int32_t* __restrict__ aflag) {
    plan_count(blockIdx.x, gridDim.x, key, sc, E, n_key, deg, aflag);
}

// block-local exclusive scan of SCAN_TILE items (tile `tile`); tile total -> sums[tile].  SCAN_THREADS threads.
__device__ __forceinline__ void scan_tile(int64_t tile, const int32_t* __restrict__ in, int64_t n, int32_t* __restrict__ out,
                                          int32_t* __restrict__ sums, int32_t* wsum) {
    const int64_t base = tile * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
    int32_t v[SCAN_ITEMS];
    int32_t t = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        v[i] = (base + i < n) ? in[base + i] : 0;
        t += v[i];
    }
    // inclusive scan of t across the wave
    int32_t incl = t;
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        int32_t o = __shfl_up(incl, off);
        if (lane >= off) incl += o;
    }
    const int w = threadIdx.x >> 6;
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    int32_t woff = 0;
    for (int i = 0; i < w; ++i) woff += wsum[i];
    int32_t run = woff + incl - t;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        if (base + i < n) out[base + i] = run;
        run += v[i];
    }
    if (threadIdx.x == SCAN_THREADS - 1) sums[tile] = run;
    __syncthreads();   // wsum is reused by the caller's next tile
}
__global__ void __launch_bounds__(SCAN_THREADS) k_scan_tile(const int32_t* __restrict__ in, int64_t n, int32_t* __restrict__ out,
                                                            int32_t* __restrict__ sums) {
    __shared__ int32_t wsum[SCAN_THREADS / 64];
    scan_tile(blockIdx.x, in, n, out, sums, wsum);
}

// one block (any size that is a multiple of 64, <= 1024): exclusive scan of the tile sums in place; total -> sums[nb]
__device__ __forceinline__ void scan_sums(int32_t* __restrict__ sums, int nb, int32_t* wsum, int32_t* carry_p) {
    int32_t& carry = *carry_p;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int nt = (int)blockDim.x;
    for (int base = 0; base < nb; base += nt) {
        int i = base + threadIdx.x;
        int32_t t = i < nb ? sums[i] : 0;
        int32_t incl = t;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            int32_t o = __shfl_up(incl, off);
            if (lane >= off) incl += o;
        }
        if (lane == 63) wsum[w] = incl;
        __syncthreads();
        int32_t woff = carry;
        for (int k = 0; k < w; ++k) woff += wsum[k];
        if (i < nb) sums[i] = woff + incl - t;
        __syncthreads();
        if ((int)threadIdx.x == nt - 1) carry = woff + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) sums[nb] = carry;
}
__global__ void __launch_bounds__(1024) k_scan_sums(int32_t* __restrict__ sums, int nb) {
    __shared__ int32_t wsum[16];
    __shared__ int32_t carry;
    scan_sums(sums, nb, wsum, &carry);
}

__device__ __forceinline__ void scan_add(int64_t tile, int32_t* __restrict__ out, int64_t n, const int32_t* __restrict__ sums, int nb) {
    const int32_t add = sums[tile];
    const int64_t base = tile * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i)
        if (base + i < n) out[base + i] += add;
    if (tile == 0 && threadIdx.x == 0) out[n] = sums[nb];
}
__global__ void __launch_bounds__(SCAN_THREADS) k_scan_add(int32_t* __restrict__ out, int64_t n, const int32_t* __restrict__ sums, int nb) {
    scan_add(blockIdx.x, out, n, sums, nb);
}

__device__ __forceinline__ void plan_fill(int64_t bid, int64_t nblk, const int64_t* __restrict__ key, int64_t sc, int64_t E, int64_t n_key,
                                          const int32_t* __restrict__ rowptr, int32_t* __restrict__ deg, int32_t* __restrict__ tmp) {
    for (int64_t e = bid * (int64_t)blockDim.x + threadIdx.x; e < E; e += nblk * blockDim.x) {
        const int64_t d = key[e * sc];
        if (d < 0 || d >= n_key) continue;  // reported by k_plan_count
        const int32_t slot = rowptr[d] + atomicSub(&deg[d], 1) - 1;
        tmp[slot] = (int32_t)e;
    }
}
__global__ void k_plan_fill(const int64_t* __restrict__ key, int64_t sc, int64_t E, int64_t n_key, const int32_t* __restrict__ rowptr,
                            int32_t* __restrict__ deg, int32_t* __restrict__ tmp) {
    plan_fill(blockIdx.x, gridDim.x, key, sc, E, n_key, rowptr, deg, tmp);
}

// other endpoint of sorted edge `pos`, range-checked against n_other (0 = unknown, unchecked): a bad id becomes 0 (memory-safe)
// and is reported through the asynchronous error word
__device__ __forceinline__ int32_t checked_other(const int64_t* __restrict__ other_row, int64_t pos, int64_t sc, int64_t n_other,
                                                 int32_t* aflag) {
    int64_t v = other_row[pos * sc];
    if (n_other > 0 && (v < 0 || v >= n_other)) {
        dgnn_raise_async(aflag, DGNN_ASYNC_OTHER_RANGE);
        v = 0;
    }
    return (int32_t)v;
}

__device__ __forceinline__ void cswap(int32_t& a, int32_t& b) {
    int32_t lo = min(a, b), hi = max(a, b);
    a = lo;
    b = hi;
}

// one thread per key: sort its segment of `tmp` ascending and emit eid/other.
// Segments longer than 32 are queued for k_plan_emit_big.
__device__ __forceinline__ void plan_emit(int64_t bid, int64_t nblk, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ tmp,
                                          const int64_t* __restrict__ other_row, int64_t sc, int64_t n_key, int32_t* __restrict__ eid,
                                          int32_t* __restrict__ other, int32_t* __restrict__ big_count, int32_t* __restrict__ big_list,
                                          int64_t n_other, int32_t* aflag) {
    for (int64_t d = bid * (int64_t)blockDim.x + threadIdx.x; d < n_key; d += nblk * blockDim.x) {
        const int32_t beg = rowptr[d], n = rowptr[d + 1] - beg;
        if (n <= 4) {
            int32_t v0 = n > 0 ? tmp[beg] : INT32_MAX, v1 = n > 1 ? tmp[beg + 1] : INT32_MAX;
            int32_t v2 = n > 2 ? tmp[beg + 2] : INT32_MAX, v3 = n > 3 ? tmp[beg + 3] : INT32_MAX;
            cswap(v0, v1); cswap(v2, v3); cswap(v0, v2); cswap(v1, v3); cswap(v1, v2);
            if (n > 0) { eid[beg] = v0; other[beg] = checked_other(other_row, v0, sc, n_other, aflag); }
            if (n > 1) { eid[beg + 1] = v1; other[beg + 1] = checked_other(other_row, v1, sc, n_other, aflag); }
            if (n > 2) { eid[beg + 2] = v2; other[beg + 2] = checked_other(other_row, v2, sc, n_other, aflag); }
            if (n > 3) { eid[beg + 3] = v3; other[beg + 3] = checked_other(other_row, v3, sc, n_other, aflag); }
        } else if (n <= 32) {
            int32_t v[32];
            for (int i = 0; i < n; ++i) {
                int32_t x = tmp[beg + i];
                int j = i;
                while (j > 0 && v[j - 1] > x) { v[j] = v[j - 1]; --j; }
                v[j] = x;
            }
            for (int i = 0; i < n; ++i) { eid[beg + i] = v[i]; other[beg + i] = checked_other(other_row, v[i], sc, n_other, aflag); }
        } else {
            big_list[atomicAdd(big_count, 1)] = (int32_t)d;
        }
    }
}
__global__ void k_plan_emit(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ tmp, const int64_t* __restrict__ other_row, int64_t sc,
                            int64_t n_key, int32_t* __restrict__ eid, int32_t* __restrict__ other, int32_t* __restrict__ big_count,
                            int32_t* __restrict__ big_list, int64_t n_other, int32_t* aflag) {
    plan_emit(blockIdx.x, gridDim.x, rowptr, tmp, other_row, sc, n_key, eid, other, big_count, big_list, n_other, aflag);
}

// ---- fast path 1: the edge list is already grouped by key (ascending) -------------------------------------------
// Then the stable sort is the identity: eid[k] = k, other[k] = the other endpoint, and rowptr[d] is the first position
// whose key is >= d.  One pass, no atomics.  This is what k-hop sampled blocks look like (edges ordered by destination,
// SURVEY App. B), what the by-source (transposed) plan of the reference layout is, and how dgnn_amd.partition lays out a
// rank's local edge list.  Any inversion or out-of-range key sets need[0] and the next builder takes over.
// Two kernels: the check first -- the row-pointer fill below walks the gap between consecutive keys, which is only
// bounded (n_key in total) when the keys really are ascending.
__global__ void __launch_bounds__(256) k_plan_sorted_check(const int64_t* __restrict__ key, int64_t sc, int64_t E, int64_t n_key,
                                                           int32_t* __restrict__ need) {
    bool bad = false;
    for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < E; k += (int64_t)gridDim.x * blockDim.x) {
        if (*reinterpret_cast<volatile int32_t*>(&need[0])) return;  // somebody already found an inversion
        const int64_t kk = key[k * sc], kp = k > 0 ? key[(k - 1) * sc] : 0;
        bad |= kk < kp || kk < 0 || kk >= n_key;
    }
    // one store per wave at most, and none once the flag is up (thousands of same-address atomics serialise in L2)
    if (__any(bad) && (threadIdx.x & 63) == 0 && *reinterpret_cast<volatile int32_t*>(&need[0]) == 0) atomicOr(&need[0], 1);
}

__global__ void __launch_bounds__(256) k_plan_sorted(const int64_t* __restrict__ key, const int64_t* __restrict__ oth, int64_t sc,
                                                     int64_t E, int64_t n_key, int32_t* __restrict__ rowptr,
                                                     int32_t* __restrict__ other, int32_t* __restrict__ eid,
                                                     const int32_t* __restrict__ need, int64_t n_other, int32_t* aflag) {
    // need == NULL: DGNN_PLAN_HINT_GROUPED_TRUSTED -- the caller vouches for ascending in-range keys, nothing checked them; the row-pointer
    // walk is clamped to [0, n_key] so that a wrong promise gives a wrong plan, never a write outside rowptr
    if (need && need[0] != 0) return;
    for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < E; k += (int64_t)gridDim.x * blockDim.x) {
        int64_t kk = key[k * sc], kp = k > 0 ? key[(k - 1) * sc] : -1;
        if (!need) {
            kk = kk < 0 ? 0 : (kk >= n_key ? n_key - 1 : kk);
            kp = kp < -1 ? -1 : (kp >= n_key ? n_key - 1 : kp);
        }
        eid[k] = (int32_t)k;
        other[k] = checked_other(oth, k, sc, n_other, aflag);
        for (int64_t d = kp + 1; d <= kk; ++d) rowptr[d] = (int32_t)k;
        if (k == E - 1)
            for (int64_t d = kk + 1; d <= n_key; ++d) rowptr[d] = (int32_t)E;
    }
}

// ---- fast path 2: the reference's own graph layout -----------------------------------------------------------------
// processing/data.py:434-438 hands the model `adjacencies` [4N,2]: row 4t+r = (t, r-th neighbour of cell t), every
// cell has exactly 4 rows and the relation is symmetric (cells share facets).  Then in-edges(t) = the reverse of its
// out-edges, rowptr[t] = 4t, and the stable by-destination order is: neighbours ascending, and for a neighbour s the
// slots r' of s's own rows that point back at t, ascending (edge id 4s + r').  One thread per cell builds its 4
// entries from its own 4 rows and its neighbours' rows -- no atomics, no scan, no segment sort.
// Everything is verified on the fly (other[k] == k/4, keys in range, count(s->t) == count(t->s) for every
// neighbour; the last one, checked by every cell, is exactly global symmetry).  Any violation sets need[1], and the
// generic kernels that follow in the stream then rebuild the plan from scratch.
__global__ void __launch_bounds__(256) k_plan_regular(const int64_t* __restrict__ key, const int64_t* __restrict__ oth, int64_t sc,
                                                      int64_t n_key, int32_t* __restrict__ rowptr, int32_t* __restrict__ other,
                                                      int32_t* __restrict__ eid, int32_t* __restrict__ need) {
    if (need[0] == 0) return;  // fast path 1 already produced the plan
    bool bad = false;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < n_key; t += (int64_t)gridDim.x * blockDim.x) {
        int64_t k4[4], o4[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            k4[r] = key[(4 * t + r) * sc];
            o4[r] = oth[(4 * t + r) * sc];
        }
        rowptr[t] = (int32_t)(4 * t);
        if (t == n_key - 1) rowptr[n_key] = (int32_t)(4 * n_key);
        int32_t d[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            bad |= o4[r] != t || k4[r] < 0 || k4[r] >= n_key;
            d[r] = (int32_t)(k4[r] < 0 ? 0 : (k4[r] >= n_key ? n_key - 1 : k4[r]));
        }
        cswap(d[0], d[1]); cswap(d[2], d[3]); cswap(d[0], d[2]); cswap(d[1], d[3]); cswap(d[1], d[2]);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t s_ = d[i];
            int occ = 0, mult = 0;  // which of the duplicates of s_ this is, and how many there are
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                mult += d[j] == d[i];
                occ += (j < i) && d[j] == d[i];
            }
            int cnt = 0, slot = 0;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool hit = key[(4 * s_ + r) * sc] == t;
                if (hit && cnt == occ) slot = r;
                cnt += hit;
            }
            bad |= cnt != mult;
            other[4 * t + i] = (int32_t)s_;
            eid[4 * t + i] = (int32_t)(4 * s_ + slot);
        }
    }
    if (__any(bad) && (threadIdx.x & 63) == 0 && *reinterpret_cast<volatile int32_t*>(&need[1]) == 0) atomicOr(&need[1], 1);
}

// one block per queued long segment: rank sort (edge positions are distinct)
__device__ __forceinline__ void plan_emit_big(int bid, int nblk, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ tmp,
                                              const int64_t* __restrict__ other_row, int64_t sc, int32_t* __restrict__ eid,
                                              int32_t* __restrict__ other, const int32_t* __restrict__ big_count,
                                              const int32_t* __restrict__ big_list, int64_t n_other, int32_t* aflag) {
    const int nbig = *big_count;
    for (int b = bid; b < nbig; b += nblk) {
        const int32_t d = big_list[b];
        const int32_t beg = rowptr[d], n = rowptr[d + 1] - beg;
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            const int32_t x = tmp[beg + i];
            int32_t rank = 0;
            for (int j = 0; j < n; ++j) rank += tmp[beg + j] < x;
            eid[beg + rank] = x;
            other[beg + rank] = checked_other(other_row, x, sc, n_other, aflag);
        }
    }
}
__global__ void __launch_bounds__(256) k_plan_emit_big(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ tmp,
                                                       const int64_t* __restrict__ other_row, int64_t sc, int32_t* __restrict__ eid,
                                                       int32_t* __restrict__ other, const int32_t* __restrict__ big_count,
                                                       const int32_t* __restrict__ big_list, int64_t n_other, int32_t* aflag) {
    plan_emit_big(blockIdx.x, gridDim.x, rowptr, tmp, other_row, sc, eid, other, big_count, big_list, n_other, aflag);
}

// ---- the generic builder as ONE launch: stands by behind the verified fast paths -----------------------------------------------------
// When a fast path produced the plan (need[0] == 0 or need[1] == 0, final before this kernel starts) every block returns on its first
// instruction -- one skipped launch instead of seven.  Otherwise the passes above run as phases of this persistent grid (at most
// 2 blocks per CU: all co-resident), separated by grid-wide barriers on a counter in the scratch area; agent-scope release / acquire
// around each barrier makes one phase's stores visible to the next phase's plain loads on every XCD (the per-XCD L2s are not coherent).
__device__ __forceinline__ void grid_sync(int32_t* counter, int nblk, int& phase) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int target = (phase + 1) * nblk;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(4);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // every wave's vector L1
    ++phase;
}

__global__ void __launch_bounds__(SCAN_THREADS) k_plan_fallback(const int64_t* __restrict__ key, const int64_t* __restrict__ oth, int64_t sc, int64_t E,
                                                                int64_t n_key, int64_t n_other, int32_t* __restrict__ rowptr,
                                                                int32_t* __restrict__ other, int32_t* __restrict__ eid, int32_t* __restrict__ deg,
                                                                int32_t* __restrict__ tmp, int32_t* __restrict__ sums, int nb,
                                                                int32_t* __restrict__ big_count, int32_t* __restrict__ big_list,
                                                                const int32_t* __restrict__ need, int32_t* gsync, int32_t* aflag) {
    if (plan_skip(need)) return;
    __shared__ int32_t wsum[16];
    __shared__ int32_t carry;
    const int64_t bid = blockIdx.x, nblk = gridDim.x;
    int phase = 0;
    for (int64_t i = bid * (int64_t)blockDim.x + threadIdx.x; i < n_key; i += nblk * blockDim.x) deg[i] = 0;
    grid_sync(gsync, (int)nblk, phase);
    plan_count(bid, nblk, key, sc, E, n_key, deg, aflag);
    grid_sync(gsync, (int)nblk, phase);
    for (int64_t t = bid; t < nb; t += nblk) scan_tile(t, deg, n_key, rowptr, sums, wsum);
    grid_sync(gsync, (int)nblk, phase);
    if (bid == 0) scan_sums(sums, nb, wsum, &carry);
    grid_sync(gsync, (int)nblk, phase);
    for (int64_t t = bid; t < nb; t += nblk) scan_add(t, rowptr, n_key, sums, nb);
    grid_sync(gsync, (int)nblk, phase);
    plan_fill(bid, nblk, key, sc, E, n_key, rowptr, deg, tmp);
    grid_sync(gsync, (int)nblk, phase);
    plan_emit(bid, nblk, rowptr, tmp, oth, sc, n_key, eid, other, big_count, big_list, n_other, aflag);
    grid_sync(gsync, (int)nblk, phase);
    plan_emit_big((int)bid, (int)nblk, rowptr, tmp, oth, sc, eid, other, big_count, big_list, n_other, aflag);
}

// The same pass with FOUR LANES PER CELL when the (src, dst) pairs are interleaved in memory (edge_index = the reference's transposed view of
// its [E,2] int64 array: row stride 1, column stride 2): lane q of a cell loads pair 4t+q with one 16-byte load (coalesced: a wave reads 1 KB
// contiguous) and, for each of the cell's 4 neighbours, pair q of the neighbour's 4 rows -- a quad reads one contiguous 64-byte chunk per
// neighbour, i.e. 5 sector requests per cell instead of the 24 eight-byte lane loads of the one-thread-per-cell form.  Hits are combined
// inside the quad with a ballot.  Same checks, same output.
__global__ void __launch_bounds__(256) k_plan_regular_q(const int64_t* __restrict__ pairs, int64_t n_key, int32_t* __restrict__ rowptr,
                                                        int32_t* __restrict__ other, int32_t* __restrict__ eid, int32_t* __restrict__ need) {
    if (need[0] == 0) return;  // fast path 1 already produced the plan
    typedef long long ll2 __attribute__((ext_vector_type(2)));
    const int lane = threadIdx.x & 63, q = lane & 3;
    const int64_t E = 4 * n_key;
    bool bad = false;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < E; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t t = e >> 2;
        const ll2 p = *reinterpret_cast<const ll2*>(pairs + 2 * e);
        bad |= p.x != t || p.y < 0 || p.y >= n_key;
        const int32_t d = (int32_t)(p.y < 0 ? 0 : (p.y >= n_key ? n_key - 1 : p.y));
        int32_t dj[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) dj[j] = __shfl(d, (lane & 60) + j);
        int rank = 0, occ = 0, mult = 0;   // position of this lane's neighbour in the stable ascending order; which of its duplicates it is
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            rank += (dj[j] < d) || (dj[j] == d && j < q);
            occ += dj[j] == d && j < q;
            mult += dj[j] == d;
        }
        uint32_t mine = 0;   // slots of neighbour d's rows that point back at t (4-bit mask)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const ll2 w = *reinterpret_cast<const ll2*>(pairs + 2 * (4 * (int64_t)dj[j] + q));
            const uint64_t b = __ballot(w.y == t);
            const uint32_t m = (uint32_t)(b >> (lane & 60)) & 15u;
            if (j == q) mine = m;
        }
        int cnt = 0, slot = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool hit = (mine >> r) & 1u;
            if (hit && cnt == occ) slot = r;
            cnt += hit;
        }
        bad |= cnt != mult;
        other[4 * t + rank] = d;
        eid[4 * t + rank] = 4 * d + slot;
        if (q == 0) rowptr[t] = (int32_t)(4 * t);
        if (e == E - 1) rowptr[n_key] = (int32_t)E;
    }
    if (__any(bad) && (threadIdx.x & 63) == 0 && *reinterpret_cast<volatile int32_t*>(&need[1]) == 0) atomicOr(&need[1], 1);
}

__global__ void k_gather_rows(const float* __restrict__ in, int64_t ld_in, const int32_t* __restrict__ idx, int64_t n,
                              int cols, float* __restrict__ out, int64_t ld_out) {
    // one thread per (row, col); consecutive threads walk a row -> coalesced on both sides
    const int64_t total = n * cols;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = t / cols;
        const int c = (int)(t - r * cols);
        out[r * ld_out + c] = in[(int64_t)idx[r] * ld_in + c];
    }
}

__global__ void k_scatter_rows(const float* __restrict__ in, int64_t ld_in, const int64_t* __restrict__ idx, int64_t n,
                               int cols, float* __restrict__ out, int64_t ld_out) {
    const int64_t total = n * cols;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = t / cols;
        const int c = (int)(t - r * cols);
        out[idx[r] * ld_out + c] = in[r * ld_in + c];
    }
}

template <typename T>
__global__ void k_relu(const T* __restrict__ x, int64_t n, T* __restrict__ y) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        dgnn_st(y + i, fmaxf(dgnn_ld(x + i), 0.0f));
}

template <typename T>
__global__ void k_relu_bwd(const T* __restrict__ y, const T* __restrict__ g, int64_t n, T* __restrict__ out) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        dgnn_st(out + i, dgnn_ld(y + i) > 0.f ? dgnn_ld(g + i) : 0.f);
}

}  // namespace

extern "C" int dgnn_relu_bwd(const float* y, const float* g, int64_t n, float* out, void* stream) {
    DGNN_REQUIRE(n >= 0 && (n == 0 || (y && g && out)), DGNN_E_INVALID, "relu_bwd: bad args");
    if (n == 0) return DGNN_OK;
    hipLaunchKernelGGL((k_relu_bwd<float>), dim3(dgnn_grid_cap(dgnn_cdiv(n, 256))), dim3(256), 0, (hipStream_t)stream, y, g, n, out);
    return dgnn_check_launch("relu_bwd");
}

// out[0..n] = exclusive scan of in[0..n) (out[n] = total).  sums_scratch: cdiv(n, 2048) + 2 ints.  Used by sampler.hip.
int dgnn_exclusive_scan_i32(const int32_t* in, int64_t n, int32_t* out, int32_t* sums_scratch, hipStream_t stream) {
    if (n <= 0) {
        hipLaunchKernelGGL(k_zero_i32, dim3(1), dim3(64), 0, stream, out, (int64_t)1);
        return DGNN_OK;
    }
    const int nb = (int)dgnn_cdiv(n, SCAN_TILE);
    hipLaunchKernelGGL(k_scan_tile, dim3(nb), dim3(SCAN_THREADS), 0, stream, in, n, out, sums_scratch);
    hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(1024), 0, stream, sums_scratch, nb);
    hipLaunchKernelGGL(k_scan_add, dim3(nb), dim3(SCAN_THREADS), 0, stream, out, n, sums_scratch, nb);
    return DGNN_OK;
}

// How many blocks of k_plan_fallback the CURRENT device holds at once (2 per CU at most: the phases are bandwidth-bound); cached per device.
static int plan_fallback_resident_blocks() {
    static int cached[DGNN_MAX_DEVICES];   // 0 = not asked yet
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= DGNN_MAX_DEVICES) dev = -1;
    if (dev >= 0) {
        const int c = __atomic_load_n(&cached[dev], __ATOMIC_ACQUIRE);
        if (c > 0) return c;
    }
    int per_cu = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(k_plan_fallback), SCAN_THREADS, 0) != hipSuccess || per_cu < 1) per_cu = 1;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev < 0 ? 0 : dev) != hipSuccess || cus < 1) cus = 1;
    (void)hipGetLastError();
    const int blocks = (per_cu < 2 ? per_cu : 2) * cus;
    if (dev >= 0) __atomic_store_n(&cached[dev], blocks, __ATOMIC_RELEASE);
    return blocks;
}

// scratch layout (int32): deg[n_key] | tmp[E] | sums[nb+2] | big_count[1] | big_list[E/33+2] | need[2] | grid-barrier counter[1] (+1 pad)
extern "C" int64_t dgnn_plan_scratch_elems(int64_t E, int64_t n_key) {
    if (E < 0 || n_key < 0) return 0;
    return n_key + E + (dgnn_cdiv(n_key, SCAN_TILE) + 2) + 1 + (E / 33 + 2) + 4;
}

extern "C" int dgnn_plan_build(const int64_t* edge_index, int64_t stride_row, int64_t stride_col, int64_t E, int64_t n_key,
                               int64_t n_other, int by, int hint, int32_t* rowptr, int32_t* other, int32_t* eid, int32_t* scratch,
                               void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    int32_t* const aflag = dgnn_async_flag_dev();
    DGNN_REQUIRE(E >= 0 && n_key >= 0 && (by == 0 || by == 1), DGNN_E_INVALID, "plan_build: bad sizes E=%lld n=%lld by=%d",
                 (long long)E, (long long)n_key, by);
    DGNN_REQUIRE(hint >= DGNN_PLAN_HINT_AUTO && hint <= DGNN_PLAN_HINT_GROUPED_TRUSTED, DGNN_E_INVALID, "plan_build: bad hint %d", hint);
    DGNN_REQUIRE(E < INT32_MAX && n_key < INT32_MAX, DGNN_E_UNSUPPORTED, "plan_build: E and n must fit int32");
    DGNN_REQUIRE(rowptr && scratch && (E == 0 || (edge_index && other && eid)), DGNN_E_INVALID, "plan_build: null pointer");
    DGNN_REQUIRE(E <= 1 || stride_col != 0, DGNN_E_INVALID, "plan_build: zero column stride");
    const int64_t* key = edge_index + (by ? stride_row : 0);
    const int64_t* oth = edge_index + (by ? 0 : stride_row);
    const int64_t sc = stride_col;
    int32_t* deg = scratch;
    int32_t* tmp = deg + n_key;
    const int nb = (int)dgnn_cdiv(n_key, SCAN_TILE);
    int32_t* sums = tmp + E;
    int32_t* big_count = sums + nb + 2;
    int32_t* big_list = big_count + 1;
    int32_t* need = big_list + (E / 33 + 2);

    if (n_key == 0) {
        hipLaunchKernelGGL(k_zero_i32, dim3(1), dim3(64), 0, stream, rowptr, (int64_t)1);
        return dgnn_check_launch("plan_build");
    }
    if (hint == DGNN_PLAN_HINT_GROUPED_TRUSTED && E > 0) {
        // the caller built the list grouped by key itself (a ring part, dgnn_amd/partition.py): ONE launch, no check, no builder standing by --
        // the three launches saved are 15 us of a 0.23 ms step on a 1/8 shard of the 1M-tet scene
        hipLaunchKernelGGL(k_plan_sorted, dim3(dgnn_grid_cap(dgnn_cdiv(E, 256))), dim3(256), 0, stream, key, oth, sc, E, n_key, rowptr, other, eid,
                           (const int32_t*)nullptr, n_other, aflag);
        return dgnn_check_launch("plan_build");
    }
    if (hint == DGNN_PLAN_HINT_GROUPED_TRUSTED) hint = DGNN_PLAN_HINT_GROUPED;
    // Verified fast paths first (see the kernels): "already grouped by key" and, by destination with E == 4N, the
    // reference layout.  `hint` only picks which of them is attempted (AUTO: both); whatever fails on the device is
    // caught by the flags and the generic kernels queued behind rebuild the plan -- with small grids (they are
    // grid-stride loops and return on their first instruction when a fast path succeeded).
    const bool can_regular = by == 1 && E == 4 * n_key && E > 0;
    const bool try_sorted = E > 0 && hint != DGNN_PLAN_HINT_GENERIC && !(hint == DGNN_PLAN_HINT_REFERENCE && can_regular);
    const bool try_regular = can_regular && hint != DGNN_PLAN_HINT_GROUPED && hint != DGNN_PLAN_HINT_GENERIC;
    const bool standby = try_sorted || try_regular;   // the generic builder only stands by: one persistent launch (k_plan_fallback)
    hipLaunchKernelGGL(k_plan_init, dim3(standby ? 1 : dgnn_grid_cap(dgnn_cdiv(n_key, 256))), dim3(standby ? 64 : 256), 0, stream, deg,
                       standby ? (int64_t)0 : n_key, big_count, need, try_sorted ? 0 : 1, try_regular ? 0 : 1);
    if (try_sorted) {
        const dim3 g(dgnn_grid_cap(dgnn_cdiv(E, 256)));
        hipLaunchKernelGGL(k_plan_sorted_check, g, dim3(256), 0, stream, key, sc, E, n_key, need);
        hipLaunchKernelGGL(k_plan_sorted, g, dim3(256), 0, stream, key, oth, sc, E, n_key, rowptr, other, eid, need, n_other, aflag);
    }
    if (try_regular) {
        // (src, dst) pairs interleaved in memory (the reference's transposed view of its [E,2] array): four lanes per cell, 16-byte loads
        if (stride_row == 1 && sc == 2 && ((uintptr_t)edge_index % 16) == 0)
            hipLaunchKernelGGL(k_plan_regular_q, dim3(dgnn_grid_cap(dgnn_cdiv(E, 256))), dim3(256), 0, stream, edge_index, n_key, rowptr, other, eid, need);
        else
            hipLaunchKernelGGL(k_plan_regular, dim3(dgnn_grid_cap(dgnn_cdiv(n_key, 256))), dim3(256), 0, stream, key, oth, sc, n_key, rowptr,
                               other, eid, need);
    }
    if (standby) {
        // The persistent kernel's grid barriers need every block resident at once: the grid is capped by what THIS device can hold
        // (occupancy x its real CU count -- a CPX/SPX partition or another gfx9 part has fewer than 256), so every block becomes resident as soon
        // as the kernels ahead of it drain.  DGNN_PLAN_COOPERATIVE=1 additionally launches it as a cooperative kernel (the runtime then refuses
        // a grid that cannot be co-resident instead of hanging) -- off by default: on this stack a process that has issued cooperative launches
        // slows the kernel launches of every OTHER process on the GPU to half speed for as long as it lives (measured: the training leg of
        // bench.py, a child process, 0.92 -> 1.85 ms per step).
        const int64_t want = dgnn_cdiv(E > n_key ? E : n_key, (int64_t)SCAN_THREADS * 8);
        const int cap = plan_fallback_resident_blocks();
        const int grid = (int)(want < 1 ? 1 : (want < cap ? want : cap));
        static const bool cooperative = [] { const char* e = getenv("DGNN_PLAN_COOPERATIVE"); return e && e[0] == '1'; }();
        if (!cooperative) {
            hipLaunchKernelGGL(k_plan_fallback, dim3(grid), dim3(SCAN_THREADS), 0, stream, key, oth, sc, E, n_key, n_other, rowptr, other, eid, deg, tmp,
                               sums, nb, big_count, big_list, need, need + 2, aflag);
            return dgnn_check_launch("plan_build");
        }
        int64_t sc_ = sc, E_ = E, nk_ = n_key, no_ = n_other;
        int nb_ = nb;
        int32_t* gsync = need + 2;
        const int32_t* need_c = need;
        void* args[] = {(void*)&key, (void*)&oth, &sc_, &E_, &nk_, &no_, &rowptr, &other, &eid, &deg, &tmp, &sums, &nb_, &big_count, &big_list,
                        (void*)&need_c, &gsync, (void*)&aflag};
        const hipError_t ce = hipLaunchCooperativeKernel(reinterpret_cast<const void*>(k_plan_fallback), dim3(grid), dim3(SCAN_THREADS), args, 0, stream);
        if (ce != hipSuccess) {
            (void)hipGetLastError();
            dgnn_set_error("plan_build: cooperative launch of the fallback builder: %s", hipGetErrorString(ce));
            return DGNN_E_LAUNCH;
        }
        return dgnn_check_launch("plan_build");
    }
    auto grid_for = [&](int64_t n) { return dim3((unsigned)dgnn_grid_cap(dgnn_cdiv(n, 256))); };
    if (E > 0) hipLaunchKernelGGL(k_plan_count, grid_for(E), dim3(256), 0, stream, key, sc, E, n_key, deg, aflag);
    hipLaunchKernelGGL(k_scan_tile, dim3(nb), dim3(SCAN_THREADS), 0, stream, deg, n_key, rowptr, sums);
    hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(1024), 0, stream, sums, nb);
    hipLaunchKernelGGL(k_scan_add, dim3(nb), dim3(SCAN_THREADS), 0, stream, rowptr, n_key, sums, nb);
    if (E > 0) {
        hipLaunchKernelGGL(k_plan_fill, grid_for(E), dim3(256), 0, stream, key, sc, E, n_key, rowptr, deg, tmp);
        hipLaunchKernelGGL(k_plan_emit, grid_for(n_key), dim3(256), 0, stream, rowptr, tmp, oth, sc, n_key, eid, other, big_count, big_list,
                           n_other, aflag);
        hipLaunchKernelGGL(k_plan_emit_big, dim3(256), dim3(256), 0, stream, rowptr, tmp, oth, sc, eid, other, big_count, big_list, n_other,
                           aflag);
    }
    return dgnn_check_launch("plan_build");
}

extern "C" int dgnn_gather_rows_f32(const float* in, int64_t ld_in, const int32_t* idx, int64_t n, int cols, float* out,
                                    int64_t ld_out, void* stream) {
    DGNN_REQUIRE(n >= 0 && cols >= 0 && (n == 0 || cols == 0 || (in && idx && out)), DGNN_E_INVALID, "gather_rows: bad args");
    if (n == 0 || cols == 0) return DGNN_OK;
    hipLaunchKernelGGL(k_gather_rows, dim3(dgnn_grid_cap(dgnn_cdiv(n * cols, 256))), dim3(256), 0, (hipStream_t)stream, in,
                       ld_in, idx, n, cols, out, ld_out);
    return dgnn_check_launch("gather_rows");
}

extern "C" int dgnn_scatter_rows_f32(const float* in, int64_t ld_in, const int64_t* idx, int64_t n, int cols, float* out,
                                     int64_t ld_out, void* stream) {
    DGNN_REQUIRE(n >= 0 && cols >= 0 && (n == 0 || cols == 0 || (in && idx && out)), DGNN_E_INVALID, "scatter_rows: bad args");
    if (n == 0 || cols == 0) return DGNN_OK;
    hipLaunchKernelGGL(k_scatter_rows, dim3(dgnn_grid_cap(dgnn_cdiv(n * cols, 256))), dim3(256), 0, (hipStream_t)stream, in,
                       ld_in, idx, n, cols, out, ld_out);
    return dgnn_check_launch("scatter_rows");
}

extern "C" int dgnn_relu(const float* x, int64_t n, float* y, void* stream) {
    DGNN_REQUIRE(n >= 0 && (n == 0 || (x && y)), DGNN_E_INVALID, "relu: bad args");
    if (n == 0) return DGNN_OK;
    hipLaunchKernelGGL((k_relu<float>), dim3(dgnn_grid_cap(dgnn_cdiv(n, 256))), dim3(256), 0, (hipStream_t)stream, x, n, y);
    return dgnn_check_launch("relu");
}

extern "C" int dgnn_relu_bf16(const uint16_t* x, int64_t n, uint16_t* y, void* stream) {
    DGNN_REQUIRE(n >= 0 && (n == 0 || (x && y)), DGNN_E_INVALID, "relu_bf16: bad args");
    if (n == 0) return DGNN_OK;
    hipLaunchKernelGGL((k_relu<uint16_t>), dim3(dgnn_grid_cap(dgnn_cdiv(n, 256))), dim3(256), 0, (hipStream_t)stream, x, n, y);
    return dgnn_check_launch("relu_bf16");
}

extern "C" int dgnn_relu_bwd_bf16(const uint16_t* y, const uint16_t* g, int64_t n, uint16_t* out, void* stream) {
    DGNN_REQUIRE(n >= 0 && (n == 0 || (y && g && out)), DGNN_E_INVALID, "relu_bwd_bf16: bad args");
    if (n == 0) return DGNN_OK;
    hipLaunchKernelGGL((k_relu_bwd<uint16_t>), dim3(dgnn_grid_cap(dgnn_cdiv(n, 256))), dim3(256), 0, (hipStream_t)stream, y, g, n, out);
    return dgnn_check_launch("relu_bwd_bf16");
}
