// Halo exchange of a partitioned scene as library calls over RCCL (dgnn_halo_plan_create / dgnn_halo_exchange).
//
// Before conv layer l >= 1 every rank sends the rows its peers' boundary cells read and receives its own halo rows into the tail of its activation
// buffer [n_own + n_halo, C] (dgnn_amd/partition.py; what replaces the reference's k-hop recomputation, learning/surfaceNetStaticEdgeFilters.py:232-275
// and run.py:221-223).  Rounds 1-3 issued that exchange from Python (torch.distributed.batch_isend_irecv + one wait per op); here it is
//   start: pack kernel on the caller's stream -> event -> ON THE PLAN'S SIDE STREAM one group of ncclRecv (straight into the buffer's tail) / ncclSend
//          (every pair of GPUs of an MI355X node has its own xGMI link: single-hop neighbour exchange) -> event
//   wait : the caller's stream waits for that event
// so that the interior cells' launch, queued between the two calls, overlaps the transfer.  Nothing allocates or synchronises the host.
//
// RCCL is resolved at RUN time (dlopen, no link-time dependency: the library still loads where RCCL is absent, and a process that already carries an
// RCCL -- PyTorch ships its own -- shares that one): DGNN_RCCL_LIB, then an already loaded librccl.so / librccl.so.1, then librccl.so.1 from the
// loader's path, then /opt/rocm/lib.  The communicator is the library's own (dgnn_comm_create from a unique id the host broadcasts over whatever
// channel it has), or any ncclComm_t the caller owns.
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>

#include "common.h"

namespace {

struct Rccl {
    void* handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    const char* why = "not tried";
};

Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* env = getenv("DGNN_RCCL_LIB");
        const char* names[] = {env, "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
        for (int pass = 0; pass < 2 && !r.handle; ++pass)          // pass 0: a copy the process already holds
            for (const char* n : names) {
                if (!n || !*n) continue;
                r.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL | (pass == 0 ? RTLD_NOLOAD : 0));
                if (r.handle) break;
            }
        if (!r.handle) { r.why = "librccl.so not found (DGNN_RCCL_LIB names one)"; return; }
#define SYM(field, name) r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.handle, name))
        SYM(GetUniqueId, "ncclGetUniqueId"); SYM(CommInitRank, "ncclCommInitRank"); SYM(CommDestroy, "ncclCommDestroy");
        SYM(GroupStart, "ncclGroupStart"); SYM(GroupEnd, "ncclGroupEnd"); SYM(Send, "ncclSend"); SYM(Recv, "ncclRecv");
        SYM(GetErrorString, "ncclGetErrorString"); SYM(CommCount, "ncclCommCount");
#undef SYM
        if (!(r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.GroupStart && r.GroupEnd && r.Send && r.Recv)) {
            r.why = "librccl.so lacks a point-to-point entry point";
            r.handle = nullptr;
        }
    });
    return r;
}

int rccl_fail(const char* what, ncclResult_t rc) {
    Rccl& r = rccl();
    dgnn_set_error("%s: RCCL error %d (%s)", what, (int)rc, r.GetErrorString ? r.GetErrorString(rc) : "?");
    return DGNN_E_LAUNCH;
}

// rows of `cols` 4-byte words: out[r, :] = in[idx[r], :]
__global__ void k_halo_pack(const uint32_t* __restrict__ in, int64_t ld, const int32_t* __restrict__ idx, int64_t n, int cols, uint32_t* __restrict__ out) {
    const int64_t total = n * cols;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = t / cols;
        const int c = (int)(t - r * cols);
        out[t] = in[(int64_t)idx[r] * ld + c];
    }
}

// rows of `cols` 4-byte words: out[r * ld + c] = in[r * cols + c]  (packed staging -> the strided tail of the activation buffer)
__global__ void k_halo_unpack(const uint32_t* __restrict__ in, int64_t n, int cols, uint32_t* __restrict__ out, int64_t ld) {
    const int64_t total = n * cols;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = t / cols;
        const int c = (int)(t - r * cols);
        out[r * ld + c] = in[t];
    }
}

}  // namespace

struct dgnn_halo_plan {
    int rank, world, device;
    int64_t n_own, n_send, n_recv;
    const int32_t* send_idx;      // device, caller-owned: local ids of the owned rows to send, grouped by destination rank
    int64_t *send_off, *recv_off;   // [world + 1] row offsets per peer
    hipStream_t side;
    hipEvent_t packed, done;
    bool pending;
    // receive staging for a STRIDED buffer (ld != C): one packed message per peer lands here and k_halo_unpack spreads it, so that what goes over the
    // wire -- ONE message of rows * C * elem_bytes per peer and direction -- never depends on either side's row stride (a rank that
    // received row by row because ITS ld != C needed a peer that sent row by row, i.e. the same ld everywhere).  Owned by the plan, grown on demand.
    void* stage;
    size_t stage_bytes;
};

// Fault injection for the tests of the agreed fall-backs (tests/test_gpu_multi.py; VERDICT r5 item 8: the failure branches of communicator creation had
// only ever seen their success side): DGNN_FAULT_<POINT> = "all" or a rank as in $RANK (torch.distributed.run sets it) makes that point fail there.
//   DGNN_FAULT_RCCL_UNAVAILABLE   dgnn_rccl_available() says no
//   DGNN_FAULT_COMM_CREATE        dgnn_comm_create fails BEFORE ncclCommInitRank (use "all": a lone rank that stays out leaves its peers in the collective)
//   DGNN_FAULT_COMM_CREATE_AFTER  dgnn_comm_create joins ncclCommInitRank, destroys what it got and fails (a one-rank failure on a multi-GPU box)
//   DGNN_FAULT_HALO_PLAN          dgnn_halo_plan_create fails
// Never set outside tests.
static bool fault_here(const char* name) {
    const char* v = getenv(name);
    if (!v || !*v) return false;
    if (!strcmp(v, "all")) return true;
    const char* r = getenv("RANK");
    return r && !strcmp(v, r);
}

extern "C" int dgnn_rccl_available(void) { return rccl().handle != nullptr && !fault_here("DGNN_FAULT_RCCL_UNAVAILABLE"); }

extern "C" int dgnn_comm_unique_id(void* id128) {
    DGNN_REQUIRE(id128, DGNN_E_INVALID, "comm_unique_id: null pointer");
    Rccl& r = rccl();
    DGNN_REQUIRE(r.handle, DGNN_E_UNSUPPORTED, "comm_unique_id: RCCL unavailable: %s", r.why);
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    const ncclResult_t rc = r.GetUniqueId(reinterpret_cast<ncclUniqueId*>(id128));
    return rc == ncclSuccess ? DGNN_OK : rccl_fail("comm_unique_id", rc);
}

extern "C" int dgnn_comm_create(const void* id128, int rank, int world, void** comm_out) {
    DGNN_REQUIRE(id128 && comm_out && world >= 1 && rank >= 0 && rank < world, DGNN_E_INVALID, "comm_create: bad arguments");
    Rccl& r = rccl();
    DGNN_REQUIRE(r.handle, DGNN_E_UNSUPPORTED, "comm_create: RCCL unavailable: %s", r.why);
    if (fault_here("DGNN_FAULT_COMM_CREATE")) {
        dgnn_set_error("comm_create: injected fault (DGNN_FAULT_COMM_CREATE)");
        return DGNN_E_LAUNCH;
    }
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    ncclComm_t comm = nullptr;
    const ncclResult_t rc = r.CommInitRank(&comm, world, id, rank);     // on the calling thread's current device
    if (rc != ncclSuccess) return rccl_fail("comm_create", rc);
    if (fault_here("DGNN_FAULT_COMM_CREATE_AFTER")) {
        (void)r.CommDestroy(comm);
        dgnn_set_error("comm_create: injected fault (DGNN_FAULT_COMM_CREATE_AFTER)");
        return DGNN_E_LAUNCH;
    }
    *comm_out = comm;
    return DGNN_OK;
}

extern "C" int dgnn_comm_destroy(void* comm) {
    if (!comm) return DGNN_OK;
    Rccl& r = rccl();
    DGNN_REQUIRE(r.handle, DGNN_E_UNSUPPORTED, "comm_destroy: RCCL unavailable");
    const ncclResult_t rc = r.CommDestroy(static_cast<ncclComm_t>(comm));
    return rc == ncclSuccess ? DGNN_OK : rccl_fail("comm_destroy", rc);
}

extern "C" int dgnn_comm_count(void* comm) {
    DGNN_REQUIRE(comm, DGNN_E_INVALID, "comm_count: null communicator");
    Rccl& r = rccl();
    DGNN_REQUIRE(r.handle && r.CommCount, DGNN_E_UNSUPPORTED, "comm_count: RCCL unavailable");
    int n = 0;
    const ncclResult_t rc = r.CommCount(static_cast<ncclComm_t>(comm), &n);
    return rc == ncclSuccess ? n : rccl_fail("comm_count", rc);
}

extern "C" int dgnn_halo_plan_create(int rank, int world, int64_t n_own, const int32_t* send_idx, const int64_t* send_counts, const int64_t* recv_counts,
                                     dgnn_halo_plan** out) {
    DGNN_REQUIRE(out && world >= 1 && rank >= 0 && rank < world && n_own >= 0 && send_counts && recv_counts, DGNN_E_INVALID, "halo_plan_create: bad arguments");
    if (fault_here("DGNN_FAULT_HALO_PLAN")) {
        dgnn_set_error("halo_plan_create: injected fault (DGNN_FAULT_HALO_PLAN)");
        return DGNN_E_LAUNCH;
    }
    dgnn_halo_plan* p = new dgnn_halo_plan();
    p->rank = rank;
    p->world = world;
    p->n_own = n_own;
    p->send_idx = send_idx;
    p->send_off = new int64_t[world + 1];
    p->recv_off = new int64_t[world + 1];
    p->send_off[0] = p->recv_off[0] = 0;
    bool ok = true;
    for (int i = 0; i < world; ++i) {
        ok = ok && send_counts[i] >= 0 && recv_counts[i] >= 0;
        p->send_off[i + 1] = p->send_off[i] + send_counts[i];
        p->recv_off[i + 1] = p->recv_off[i] + recv_counts[i];
    }
    p->n_send = p->send_off[world];
    p->n_recv = p->recv_off[world];
    p->pending = false;
    p->side = nullptr;
    p->packed = p->done = nullptr;
    p->stage = nullptr;
    p->stage_bytes = 0;
    ok = ok && (p->n_send == 0 || send_idx != nullptr);
    ok = ok && hipGetDevice(&p->device) == hipSuccess && hipStreamCreateWithFlags(&p->side, hipStreamNonBlocking) == hipSuccess &&
         hipEventCreateWithFlags(&p->packed, hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&p->done, hipEventDisableTiming) == hipSuccess;
    if (!ok) {
        (void)hipGetLastError();
        dgnn_set_error("halo_plan_create: negative counts, missing send_idx or no stream / event");
        if (p->side) (void)hipStreamDestroy(p->side);
        if (p->packed) (void)hipEventDestroy(p->packed);
        if (p->done) (void)hipEventDestroy(p->done);
        delete[] p->send_off;
        delete[] p->recv_off;
        delete p;
        return DGNN_E_INVALID;
    }
    *out = p;
    return DGNN_OK;
}

extern "C" int dgnn_halo_plan_destroy(dgnn_halo_plan* p) {
    if (!p) return DGNN_OK;
    (void)hipStreamSynchronize(p->side);
    (void)hipStreamDestroy(p->side);
    (void)hipEventDestroy(p->packed);
    (void)hipEventDestroy(p->done);
    if (p->stage) (void)hipFree(p->stage);
    delete[] p->send_off;
    delete[] p->recv_off;
    delete p;
    return DGNN_OK;
}

extern "C" int64_t dgnn_halo_send_rows(const dgnn_halo_plan* p) { return p ? p->n_send : 0; }
extern "C" int64_t dgnn_halo_recv_rows(const dgnn_halo_plan* p) { return p ? p->n_recv : 0; }

// x: [n_own + n_recv, C] rows of `elem_bytes`-byte elements with row stride ld (elements); send_buf: n_send * C * elem_bytes bytes
extern "C" int dgnn_halo_exchange_start(dgnn_halo_plan* p, void* comm, void* x, int64_t ld, int C, int elem_bytes, void* send_buf, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DGNN_REQUIRE(p && C > 0 && (elem_bytes == 2 || elem_bytes == 4) && ld >= C, DGNN_E_INVALID, "halo_exchange_start: bad arguments");
    DGNN_REQUIRE(!p->pending, DGNN_E_INVALID, "halo_exchange_start: the previous exchange has not been waited for");
    if (p->n_send == 0 && p->n_recv == 0) return DGNN_OK;
    Rccl& r = rccl();
    DGNN_REQUIRE(r.handle, DGNN_E_UNSUPPORTED, "halo_exchange_start: RCCL unavailable: %s", r.why);
    DGNN_REQUIRE(comm && x && (p->n_send == 0 || send_buf), DGNN_E_INVALID, "halo_exchange_start: null pointer");
    const int64_t row_bytes = (int64_t)C * elem_bytes, ld_bytes = ld * elem_bytes;
    DGNN_REQUIRE(row_bytes % 4 == 0 && ld_bytes % 4 == 0 && ((uintptr_t)x % 4) == 0 && ((uintptr_t)send_buf % 4) == 0, DGNN_E_UNSUPPORTED,
                 "halo_exchange_start: rows must be whole 4-byte words (even width / stride for 16-bit rows)");
    if (p->n_send) {
        const int cols = (int)(row_bytes / 4);
        hipLaunchKernelGGL(k_halo_pack, dim3(dgnn_grid_cap(dgnn_cdiv(p->n_send * cols, 256))), dim3(256), 0, stream, static_cast<const uint32_t*>(x), ld_bytes / 4,
                           p->send_idx, p->n_send, cols, static_cast<uint32_t*>(send_buf));
    }
    // the side stream starts once the packed rows -- and everything before them on the caller's stream, i.e. the buffer itself -- are ready
    if (hipEventRecord(p->packed, stream) != hipSuccess || hipStreamWaitEvent(p->side, p->packed, 0) != hipSuccess) return dgnn_check_launch("halo_exchange_start");
    ncclComm_t c = static_cast<ncclComm_t>(comm);
    char* tail = static_cast<char*>(x) + p->n_own * ld_bytes;
    char* land = tail;                 // where the packed halo rows land: the tail itself when it is packed (ld == C), else the plan's staging area
    if (ld != C && p->n_recv) {
        const size_t need = (size_t)(p->n_recv * row_bytes);
        if (need > p->stage_bytes) {       // (first use / a wider layer: the one allocation this library makes, inside the opaque plan)
            (void)hipStreamSynchronize(p->side);
            if (p->stage) (void)hipFree(p->stage);
            p->stage = nullptr;
            p->stage_bytes = 0;
            if (hipMalloc(&p->stage, need) != hipSuccess) { (void)hipGetLastError(); dgnn_set_error("halo_exchange_start: no memory for %zu bytes of receive staging", need); return DGNN_E_LAUNCH; }
            p->stage_bytes = need;
        }
        land = static_cast<char*>(p->stage);
    }
    ncclResult_t rc = r.GroupStart();
    if (rc != ncclSuccess) return rccl_fail("halo_exchange_start (group start)", rc);
    for (int peer = 0; peer < p->world && rc == ncclSuccess; ++peer) {
        const int64_t nr = p->recv_off[peer + 1] - p->recv_off[peer], ns = p->send_off[peer + 1] - p->send_off[peer];
        // ONE message per peer and direction, whatever the row strides on either side
        if (nr) rc = r.Recv(land + p->recv_off[peer] * row_bytes, (size_t)(nr * row_bytes), ncclInt8, peer, c, p->side);
        if (ns && rc == ncclSuccess) rc = r.Send(static_cast<char*>(send_buf) + p->send_off[peer] * row_bytes, (size_t)(ns * row_bytes), ncclInt8, peer, c, p->side);
    }
    const ncclResult_t rc2 = r.GroupEnd();
    if (rc != ncclSuccess) return rccl_fail("halo_exchange_start (send / recv)", rc);
    if (rc2 != ncclSuccess) return rccl_fail("halo_exchange_start (group end)", rc2);
    if (land != tail) {
        const int cols = (int)(row_bytes / 4);
        hipLaunchKernelGGL(k_halo_unpack, dim3(dgnn_grid_cap(dgnn_cdiv(p->n_recv * cols, 256))), dim3(256), 0, p->side, reinterpret_cast<const uint32_t*>(land), p->n_recv, cols,
                           reinterpret_cast<uint32_t*>(tail), ld_bytes / 4);
    }
    if (hipEventRecord(p->done, p->side) != hipSuccess) return dgnn_check_launch("halo_exchange_start");
    p->pending = true;
    return DGNN_OK;
}

extern "C" int dgnn_halo_exchange_wait(dgnn_halo_plan* p, void* stream_) {
    DGNN_REQUIRE(p, DGNN_E_INVALID, "halo_exchange_wait: null plan");
    if (!p->pending) return DGNN_OK;
    p->pending = false;
    if (hipStreamWaitEvent((hipStream_t)stream_, p->done, 0) != hipSuccess) return dgnn_check_launch("halo_exchange_wait");
    return DGNN_OK;
}
