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
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <mutex>

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

namespace {
__global__ void k_take_i32(const int32_t* __restrict__ in, const int32_t* __restrict__ idx, int64_t n, int32_t* __restrict__ out) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = in[idx[i]];
}

__global__ void k_i64_to_i32_x2(const int64_t* __restrict__ a, const int64_t* __restrict__ b, int64_t n, int32_t* __restrict__ a32,
                                int32_t* __restrict__ b32) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        a32[i] = (int32_t)a[i];
        b32[i] = (int32_t)b[i];
    }
}
}  // namespace

namespace {

// One device -> host word without a stream synchronize: a one-thread kernel stores {value, ticket} into pinned host memory that
// is mapped into the GPU, the host spins on the ticket.  hipStreamSynchronize / a pageable hipMemcpy would do, but both go
// through the runtime's queue locks, which a second host thread that is launching the training step at the same time also
// needs; the spin touches no HIP API.  Falls back to hipMemcpy + hipStreamSynchronize when pinned memory is unavailable.
struct Slot {
    volatile int32_t value;
    volatile uint32_t ticket;
    int32_t pad[14];
};
constexpr int N_SLOTS = 64;
Slot* g_slots_host = nullptr;
Slot* g_slots_dev = nullptr;
unsigned g_slot_busy[N_SLOTS];
unsigned g_ticket = 0;

bool slots_init() {
    static bool tried = false, ok = false;
    static std::mutex m;
    std::lock_guard<std::mutex> lock(m);
    if (!tried) {
        tried = true;
        const char* env = getenv("DGNN_KHOP_MAILBOX");   // "0": always hipMemcpy + hipStreamSynchronize (A/B measurements)
        if (env && env[0] == '0') return false;
        void* h = nullptr;
        if (hipHostMalloc(&h, sizeof(Slot) * N_SLOTS, hipHostMallocPortable | hipHostMallocMapped) == hipSuccess && h) {
            memset(h, 0, sizeof(Slot) * N_SLOTS);
            void* d = nullptr;
            if (hipHostGetDevicePointer(&d, h, 0) == hipSuccess && d) {
                g_slots_host = reinterpret_cast<Slot*>(h);
                g_slots_dev = reinterpret_cast<Slot*>(d);
                ok = true;
            } else {
                (void)hipHostFree(h);
            }
        }
        (void)hipGetLastError();
    }
    return ok;
}

inline void cpu_relax() {   // spin-wait hint of the host architecture (the library also builds on non-x86 hosts)
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#elif defined(__aarch64__)
    asm volatile("yield" ::: "memory");
#else
    std::this_thread::yield();
#endif
}

struct Mailbox {   // one slot for the duration of a call
    int idx = -1;
    Mailbox() {
        if (!slots_init()) return;
        for (int i = 0; i < N_SLOTS; ++i)
            if (__atomic_exchange_n(&g_slot_busy[i], 1u, __ATOMIC_ACQ_REL) == 0u) {
                idx = i;
                return;
            }
    }
    ~Mailbox() {
        if (idx >= 0) __atomic_store_n(&g_slot_busy[idx], 0u, __ATOMIC_RELEASE);
    }
};

__global__ void k_publish(const int32_t* __restrict__ value, Slot* slot, uint32_t ticket) {
    slot->value = *value;
    __threadfence_system();
    slot->ticket = ticket;
}

bool read_back(const int32_t* dev_value, int32_t* out, const Mailbox& mb, hipStream_t stream) {
    if (mb.idx >= 0) {
        uint32_t ticket = __atomic_add_fetch(&g_ticket, 1u, __ATOMIC_RELAXED);
        if (ticket == 0) ticket = __atomic_add_fetch(&g_ticket, 1u, __ATOMIC_RELAXED);
        hipLaunchKernelGGL(k_publish, dim3(1), dim3(1), 0, stream, dev_value, g_slots_dev + mb.idx, ticket);
        if (hipGetLastError() == hipSuccess) {
            Slot* s = g_slots_host + mb.idx;
            const auto t0 = std::chrono::steady_clock::now();
            for (uint64_t spins = 0;; ++spins) {
                if (__atomic_load_n(&s->ticket, __ATOMIC_ACQUIRE) == ticket) {
                    *out = s->value;
                    return true;
                }
                cpu_relax();
                if ((spins & 0xFFFF) == 0xFFFF && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(20)) break;   // a hung queue
            }
            // timed out: the queued k_publish still targets this slot.  It is only handed back (Mailbox destructor) after the stream has drained,
            // which the synchronising copy below does before this function returns.
        }
    }
    return hipMemcpyAsync(out, dev_value, sizeof(int32_t), hipMemcpyDeviceToHost, stream) == hipSuccess && hipStreamSynchronize(stream) == hipSuccess;
}

}  // namespace

// All hops of one batch in ONE call, for graphs whose every node has exactly `deg` in-edges (a Delaunay scene: 4), so that a
// hop's edge count is known without reading it back and only the number of newly discovered nodes crosses to the host (one
// 4-byte read per hop, waited for INSIDE this call: a caller that runs it on a worker thread -- ctypes releases the GIL --
// overlaps those round trips with the training step it is enqueueing).  Outputs go into caller-allocated buffers with
// per-hop capacities; hop h (0 = the targets' own neighbourhood, i.e. the INNERMOST block) writes
//   ei[h]       int64 [2, cap_e[h]]  rows 0 / 1 = local source / destination ids of its counts[h] * deg edges
//   e_id[h]     int64 [cap_e[h]]     graph edge ids            src32[h] / e_id32[h] int32 [cap_e[h]] = row 0 of ei[h] (the plan's
//                                                              `src`) / e_id[h] (rows of the scene's edge_attr, read in place)
//   off[h]      int32 [cap_t[h] + 1] row offsets (the plan's rowptr)
//   n_id_out[h] int64 [cap_t[h] + cap_e[h]]  node ids: its targets, then the new sources in order of first appearance
// counts_out (HOST int64 [hops + 1]) receives the number of targets of every hop and, last, the node count of the outermost
// block.  t_rowptr != NULL: every hop's transposed plan as well (dgnn_plan_build by source: t_rowptr[h] int32 [cap_all[h] + 1],
// t_dst[h] / t_eid[h] int32 [cap_e[h]]) and t_rows[h] = e_id32[h][t_eid[h]]; plan_scratch = max_h dgnn_plan_scratch_elems(cap_e[h],
// cap_all[h]).  Returns DGNN_E_INVALID when a capacity is too small (nothing is left half-written in `pos` / `first`).
extern "C" int dgnn_khop_blocks_regular(const int32_t* rowptr, const int32_t* src, const int32_t* eid, int deg, const int64_t* batch,
                                        int64_t n_batch, int hops, int32_t* pos, int32_t* first, int64_t* const* ei, int64_t* const* e_id,
                                        int32_t* const* src32, int32_t* const* e_id32, int32_t* const* off, int64_t* const* n_id_out, const int64_t* cap_t,
                                        const int64_t* cap_e, int32_t* scratch, int32_t* n_new_dev, int32_t* const* t_rowptr, int32_t* const* t_dst,
                                        int32_t* const* t_eid, int32_t* const* t_rows, const int64_t* cap_all, int32_t* plan_scratch,
                                        int64_t* counts_out, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DGNN_REQUIRE(hops >= 1 && hops <= 16 && deg >= 1 && n_batch >= 0 && rowptr && src && eid && pos && first && ei && e_id && src32 && e_id32 && off && n_id_out &&
                     cap_t && cap_e && scratch && n_new_dev && counts_out && (n_batch == 0 || batch),
                 DGNN_E_INVALID, "khop_blocks_regular: bad args");
    const int64_t* n_id = batch;
    int64_t n_t = n_batch;
    int rc = DGNN_OK;
    int h = 0;
    Mailbox mailbox;
    for (; h < hops; ++h) {
        const int64_t n_e = n_t * deg;
        counts_out[h] = n_t;
        if (n_t > cap_t[h] || n_e > cap_e[h]) {
            dgnn_set_error("khop_blocks_regular: hop %d needs %lld targets / %lld edges, capacity %lld / %lld", h, (long long)n_t, (long long)n_e,
                           (long long)cap_t[h], (long long)cap_e[h]);
            rc = DGNN_E_INVALID;
            break;
        }
        if ((rc = dgnn_khop_count(rowptr, n_id, n_t, h == 0, pos, off[h], scratch, stream_)) != DGNN_OK) break;
        if ((rc = dgnn_khop_expand(rowptr, src, eid, n_id, n_t, off[h], n_e, pos, first, ei[h], ei[h] + cap_e[h], e_id[h], n_id_out[h], n_new_dev, scratch,
                                   stream_)) != DGNN_OK)
            break;
        int32_t n_new = 0;
        if (!read_back(n_new_dev, &n_new, mailbox, stream)) {
            dgnn_set_error("khop_blocks_regular: reading the new-node count failed: %s", hipGetErrorString(hipGetLastError()));
            rc = DGNN_E_LAUNCH;
            break;
        }
        const int64_t n_all = n_t + n_new;
        if ((rc = dgnn_khop_commit(n_id_out[h], n_t, n_all, pos, first, stream_)) != DGNN_OK) break;
        if (n_e > 0)
            hipLaunchKernelGGL(k_i64_to_i32_x2, dim3(dgnn_grid_cap(dgnn_cdiv(n_e, 256))), dim3(256), 0, stream, ei[h], e_id[h], n_e, src32[h], e_id32[h]);
        if (t_rowptr) {   // the block's source-sorted plan (what the backward pass walks) and the edge rows in its order
            if (n_all > cap_all[h]) {
                dgnn_set_error("khop_blocks_regular: hop %d reaches %lld nodes, capacity %lld", h, (long long)n_all, (long long)cap_all[h]);
                rc = DGNN_E_INVALID;
                n_id = n_id_out[h], n_t = n_all;
                ++h;
                break;
            }
            if ((rc = dgnn_plan_build(ei[h], cap_e[h], 1, n_e, n_all, n_t, 0, DGNN_PLAN_HINT_GENERIC, t_rowptr[h], t_dst[h], t_eid[h], plan_scratch,
                                      stream_)) != DGNN_OK) {
                n_id = n_id_out[h], n_t = n_all;
                ++h;
                break;
            }
            if (n_e > 0)
                hipLaunchKernelGGL(k_take_i32, dim3(dgnn_grid_cap(dgnn_cdiv(n_e, 256))), dim3(256), 0, stream, e_id32[h], t_eid[h], n_e, t_rows[h]);
        }
        n_id = n_id_out[h];
        n_t = n_all;
    }
    if (h < hops) {
        for (int q = h + 1; q <= hops; ++q) counts_out[q] = 0;
    }
    counts_out[h < hops ? h : hops] = n_t;
    // pos[...] = -1 for everything this batch touched (also on the error paths: n_id / n_t describe what has been committed)
    const int rc2 = dgnn_khop_reset(n_id, n_t, pos, stream_);
    if (rc != DGNN_OK) return rc;
    if (rc2 != DGNN_OK) return rc2;
    return dgnn_check_launch("khop_blocks_regular");
}

// ---- the same call on a library-owned host thread ------------------------------------------------------------------------------
// start() returns at once; a std::thread (no interpreter lock involved) issues the launches on `stream` and waits for the per-hop
// counts; wait() joins it.  The caller keeps enqueueing the training step on its own stream meanwhile and must not touch
// `stream`, `pos`, `first` or the output buffers between start() and wait().
#include <string>
#include <thread>
#include <vector>

namespace {
struct KhopJob {
    std::thread th;
    int rc = DGNN_OK;
    std::string err;
    std::vector<int64_t*> ei, e_id, n_id_out;
    std::vector<int32_t*> src32, e_id32, off, t_rowptr, t_dst, t_eid, t_rows;
    std::vector<int64_t> cap_t, cap_e, cap_all, counts;
};
template <typename T>
std::vector<T> copy_n(const T* p, int n) {
    return p ? std::vector<T>(p, p + n) : std::vector<T>();
}
// rows of a block gathered by the builder itself (round 4): out[i, 0:cols] = src[idx[i] * ld + 0:cols] for the block's node ids
// (which = 0: all nodes of the outermost block) or its targets (which = 1) -- the head-of-step x_all[n_id, 1:], x_all[ids], y_all[ids] of the
// training loop (learning/surfaceNetStaticEdgeFilters.py:206, learning/runModel.py:273-274), 30 us of torch indexing kernels on the step's own
// stream, now behind the block on the builder's stream a step ahead
struct KhopRows {
    int n = 0;
    const float* src[4];
    int64_t ld[4];
    int cols[4], which[4];
    float* out[4];
};
__global__ void k_gather_rows_i64(const float* __restrict__ src, int64_t ld, const int64_t* __restrict__ idx, int64_t n, int cols, float* __restrict__ out) {
    const int64_t total = n * cols;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = t / cols;
        const int c = (int)(t - r * cols);
        out[t] = src[idx[r] * ld + c];
    }
}
}  // namespace

static void* khop_start(const int32_t* rowptr, const int32_t* src, const int32_t* eid, int deg, const int64_t* batch, int64_t n_batch, int hops, int32_t* pos,
                        int32_t* first, int64_t* const* ei, int64_t* const* e_id, int32_t* const* src32, int32_t* const* e_id32, int32_t* const* off,
                        int64_t* const* n_id_out, const int64_t* cap_t, const int64_t* cap_e, int32_t* scratch, int32_t* n_new_dev, int32_t* const* t_rowptr,
                        int32_t* const* t_dst, int32_t* const* t_eid, int32_t* const* t_rows, const int64_t* cap_all, int32_t* plan_scratch, KhopRows rows,
                        void* stream) {
    if (hops < 1 || hops > 16 || !ei || !e_id || !src32 || !e_id32 || !off || !n_id_out || !cap_t || !cap_e) {
        dgnn_set_error("khop_blocks_regular_start: bad args");
        return nullptr;
    }
    int device = 0;
    (void)hipGetDevice(&device);
    KhopJob* j = new KhopJob();
    j->ei = copy_n(ei, hops), j->e_id = copy_n(e_id, hops), j->n_id_out = copy_n(n_id_out, hops);
    j->src32 = copy_n(src32, hops), j->e_id32 = copy_n(e_id32, hops), j->off = copy_n(off, hops);
    j->t_rowptr = copy_n(t_rowptr, hops), j->t_dst = copy_n(t_dst, hops), j->t_eid = copy_n(t_eid, hops), j->t_rows = copy_n(t_rows, hops);
    j->cap_t = copy_n(cap_t, hops), j->cap_e = copy_n(cap_e, hops), j->cap_all = copy_n(cap_all, hops);
    j->counts.assign(hops + 1, 0);
    const bool want_t = t_rowptr != nullptr;
    j->th = std::thread([=]() {
        if (hipSetDevice(device) != hipSuccess) {
            j->rc = DGNN_E_LAUNCH;
            j->err = "khop_blocks_regular_start: hipSetDevice failed on the builder thread";
            return;
        }
        j->rc = dgnn_khop_blocks_regular(rowptr, src, eid, deg, batch, n_batch, hops, pos, first, j->ei.data(), j->e_id.data(), j->src32.data(),
                                         j->e_id32.data(), j->off.data(), j->n_id_out.data(), j->cap_t.data(), j->cap_e.data(), scratch, n_new_dev,
                                         want_t ? j->t_rowptr.data() : nullptr, want_t ? j->t_dst.data() : nullptr, want_t ? j->t_eid.data() : nullptr,
                                         want_t ? j->t_rows.data() : nullptr, want_t ? j->cap_all.data() : nullptr, plan_scratch, j->counts.data(),
                                         stream);
        if (j->rc != DGNN_OK) {
            j->err = dgnn_last_error_string();   // the error text is per thread: carry it over
            return;
        }
        for (int i = 0; i < rows.n; ++i) {       // the block's feature / label rows, behind the block on the builder's stream
            const int64_t n = rows.which[i] ? n_batch : j->counts[hops];
            const int64_t* idx = rows.which[i] ? batch : j->n_id_out[hops - 1];
            if (n <= 0) continue;
            hipLaunchKernelGGL(k_gather_rows_i64, dim3(dgnn_grid_cap(dgnn_cdiv(n * rows.cols[i], 256))), dim3(256), 0, (hipStream_t)stream, rows.src[i], rows.ld[i],
                               idx, n, rows.cols[i], rows.out[i]);
        }
        if (rows.n) {
            j->rc = dgnn_check_launch("khop_blocks_regular_start_rows");
            if (j->rc != DGNN_OK) j->err = dgnn_last_error_string();
        }
    });
    return j;
}

extern "C" void* dgnn_khop_blocks_regular_start(const int32_t* rowptr, const int32_t* src, const int32_t* eid, int deg, const int64_t* batch,
                                                int64_t n_batch, int hops, int32_t* pos, int32_t* first, int64_t* const* ei, int64_t* const* e_id,
                                                int32_t* const* src32, int32_t* const* e_id32, int32_t* const* off, int64_t* const* n_id_out,
                                                const int64_t* cap_t, const int64_t* cap_e, int32_t* scratch, int32_t* n_new_dev,
                                                int32_t* const* t_rowptr, int32_t* const* t_dst, int32_t* const* t_eid, int32_t* const* t_rows,
                                                const int64_t* cap_all, int32_t* plan_scratch, void* stream) {
    return khop_start(rowptr, src, eid, deg, batch, n_batch, hops, pos, first, ei, e_id, src32, e_id32, off, n_id_out, cap_t, cap_e, scratch, n_new_dev, t_rowptr,
                      t_dst, t_eid, t_rows, cap_all, plan_scratch, KhopRows(), stream);
}

// ... and up to 4 row gathers behind the block: r_src[i] (fp32, already offset to its first column), row stride r_ld[i], r_cols[i] columns, r_which[i]
// 0 = all nodes of the outermost block (n_id_out[hops-1][:counts[hops]]) / 1 = the batch's targets, r_out[i] [capacity, r_cols[i]] packed rows
extern "C" void* dgnn_khop_blocks_regular_start_rows(const int32_t* rowptr, const int32_t* src, const int32_t* eid, int deg, const int64_t* batch,
                                                     int64_t n_batch, int hops, int32_t* pos, int32_t* first, int64_t* const* ei, int64_t* const* e_id,
                                                     int32_t* const* src32, int32_t* const* e_id32, int32_t* const* off, int64_t* const* n_id_out,
                                                     const int64_t* cap_t, const int64_t* cap_e, int32_t* scratch, int32_t* n_new_dev,
                                                     int32_t* const* t_rowptr, int32_t* const* t_dst, int32_t* const* t_eid, int32_t* const* t_rows,
                                                     const int64_t* cap_all, int32_t* plan_scratch, int n_rows, const float* const* r_src, const int64_t* r_ld,
                                                     const int32_t* r_cols, const int32_t* r_which, float* const* r_out, void* stream) {
    KhopRows rows;
    if (n_rows < 0 || n_rows > 4 || (n_rows && !(r_src && r_ld && r_cols && r_which && r_out))) {
        dgnn_set_error("khop_blocks_regular_start_rows: at most 4 row gathers, all arrays given");
        return nullptr;
    }
    rows.n = n_rows;
    for (int i = 0; i < n_rows; ++i) {
        if (!r_src[i] || !r_out[i] || r_cols[i] <= 0 || r_ld[i] < r_cols[i]) {
            dgnn_set_error("khop_blocks_regular_start_rows: bad row gather %d", i);
            return nullptr;
        }
        rows.src[i] = r_src[i], rows.ld[i] = r_ld[i], rows.cols[i] = r_cols[i], rows.which[i] = r_which[i], rows.out[i] = r_out[i];
    }
    return khop_start(rowptr, src, eid, deg, batch, n_batch, hops, pos, first, ei, e_id, src32, e_id32, off, n_id_out, cap_t, cap_e, scratch, n_new_dev, t_rowptr,
                      t_dst, t_eid, t_rows, cap_all, plan_scratch, rows, stream);
}

extern "C" int dgnn_khop_blocks_regular_wait(void* job, int hops, int64_t* counts_out) {
    DGNN_REQUIRE(job && counts_out, DGNN_E_INVALID, "khop_blocks_regular_wait: bad args");
    KhopJob* j = reinterpret_cast<KhopJob*>(job);
    if (j->th.joinable()) j->th.join();
    const int rc = j->rc;
    for (int h = 0; h <= hops && h < (int)j->counts.size(); ++h) counts_out[h] = j->counts[h];
    if (rc != DGNN_OK) dgnn_set_error("%s", j->err.c_str());
    delete j;
    return rc;
}
