// Whole-scene inference in ONE library call (reference SurfaceNet.inference_layer, learning/surfaceNetStaticEdgeFilters.py:323-355; VERDICT r3
// item 2b): destination-sorted plan of the scene's adjacency (dgnn_plan_build), every conv layer with BatchNorm(eval) + ReLU in its launch,
// the decoder inside the last layer's launch -- the launch chain the Python mirror issued call by call (45-65 us of interpreter time per
// layer: a reconbench-size scene of 66k cells was host-bound at 0.29 ms per pass with 0.2 ms of GPU time).  Same kernels, same order, same
// arguments: results are bit-identical to the per-layer entry points (tests/test_gpu_infer.py).  Nothing allocates or synchronises.
#include "common.h"

namespace {

inline int64_t align_up(int64_t v, int64_t a) { return (v + a - 1) / a * a; }

struct Workspace {
    float* act[2];
    int64_t bytes;
};

// act0 | act1 : n x max hidden width floats each, 16-byte aligned
Workspace carve(void* base, int64_t n, int n_layers, const int32_t* widths) {
    Workspace w{};
    int maxw = 0;
    for (int l = 1; l <= n_layers; ++l) maxw = widths[l] > maxw ? widths[l] : maxw;   // outputs of layers 0 .. L-2 (and the last layer's, when the decoder runs apart)
    const int64_t each = align_up(4 * n * maxw, 16);
    char* b = static_cast<char*>(base);
    w.act[0] = reinterpret_cast<float*>(b);
    w.act[1] = reinterpret_cast<float*>(b + each);
    w.bytes = 2 * each;
    return w;
}

}  // namespace

extern "C" int64_t dgnn_static_infer_workspace_bytes(int64_t n, int n_layers, const int32_t* widths) {
    if (n < 0 || n_layers < 1 || !widths) return 0;
    return carve(nullptr, n, n_layers, widths).bytes + 16;
}

// The chain shared by the whole-scene call and the ring call: layer l runs over the destinations [0, n_dst[l]) of ONE plan (n_dst non-increasing; sources of
// those destinations lie in [0, n_dst[l-1]), in [0, n_loc) for layer 0); logits for the first n_dst[L-1] cells.
static int infer_chain(const char* who, const int64_t* edge_index, int64_t stride_row, int64_t stride_col, int64_t E, int plan_hint, int32_t* rowptr, int32_t* src,
                       int32_t* eid, int32_t* plan_scratch, bool attr_in_plan_order, int64_t n_loc, const int64_t* n_dst, const float* x, int64_t ldx,
                       const float* edge_attr, int64_t lde, int f_e, int n_layers, const int32_t* widths, const float* const* We, const float* const* be,
                       const float* const* Wj, const float* const* bj, const float* const* Wi, const float* const* scale, const float* const* shift,
                       const void* const* prepared, const float* W0, const float* b0, const float* scale1, const float* shift1, int c_hidden, const float* W3,
                       const float* b3, int n_logits, int fuse_decoder, int gemm_mode, void* workspace, float* logits, void* stream) {
    DGNN_REQUIRE(n_loc >= 0 && E >= 0 && n_layers >= 1 && n_layers <= 16 && widths && We && be && Wj && bj && Wi && scale && shift && n_dst, DGNN_E_INVALID,
                 "%s: bad sizes / null table", who);
    for (int l = 0; l < n_layers; ++l)
        DGNN_REQUIRE(n_dst[l] >= 0 && n_dst[l] <= (l ? n_dst[l - 1] : n_loc), DGNN_E_INVALID, "%s: destination counts must not grow from layer to layer", who);
    if (n_dst[n_layers - 1] == 0) return DGNN_OK;
    DGNN_REQUIRE(x && edge_attr && workspace && logits && ((uintptr_t)workspace % 16) == 0, DGNN_E_INVALID, "%s: null / unaligned pointer", who);
    const bool build = edge_index != nullptr;
    DGNN_REQUIRE(rowptr && src && (!build || (eid && plan_scratch)), DGNN_E_INVALID, "%s: plan arrays missing", who);
    const bool dec2 = W0 != nullptr;                     // Linear - BN - ReLU - Linear decoder (:180-187); W0 NULL and W3 given: a single Linear is not covered here
    DGNN_REQUIRE(!dec2 || (b0 && W3 && b3), DGNN_E_INVALID, "%s: incomplete decoder", who);
    // ---- what the chain below can run: checked before anything is launched ----------------------------------------------------------
    bool ok = f_e == 20 && lde == 20 && ((uintptr_t)edge_attr % 16) == 0 && (dec2 || W3 == nullptr);
    int maxw = 0;
    for (int l = 0; l < n_layers && ok; ++l) {
        const int ci = widths[l], co = widths[l + 1];
        ok = ci > 0 && ci <= 128 && (co == 64 || co == 128) && (ci <= 64 || (co == 128 && ci % 2 == 0)) && We[l] && be[l] && Wj[l] && Wi[l] &&
             ((scale[l] == nullptr) == (shift[l] == nullptr));
        maxw = co > maxw ? co : maxw;
    }
    ok = ok && n_loc * (int64_t)(ldx > maxw ? ldx : maxw) < ((int64_t)1 << 31);
    if (widths[0] > 64) ok = ok && ldx % 2 == 0 && ((uintptr_t)x % 8) == 0;
    if (dec2) ok = ok && widths[n_layers] == 128 && c_hidden == 64 && (n_logits == 1 || n_logits == 2);    // dgnn_decoder_fused_fwd's shapes
    if (!ok) {
        dgnn_set_error("%s: a layer shape / operand layout outside the fused kernels", who);
        return DGNN_E_UNSUPPORTED;
    }
    const Workspace ws = carve(workspace, n_dst[0], n_layers, widths);
    if (build) {
        DGNN_REQUIRE(E < INT32_MAX && n_loc < INT32_MAX, DGNN_E_UNSUPPORTED, "%s: E and n must fit int32", who);
        const int rc = dgnn_plan_build(edge_index, stride_row, stride_col, E, n_dst[0], n_loc, 1, plan_hint, rowptr, src, eid, plan_scratch, stream);
        if (rc != DGNN_OK) return rc;
    }
    const int32_t* e_ = attr_in_plan_order ? nullptr : eid;
    const float* h = x;
    int64_t ldh = ldx;
    const int64_t n_out = n_dst[n_layers - 1];
    for (int l = 0; l < n_layers; ++l) {
        const int ci = widths[l], co = widths[l + 1];
        const int64_t n = n_dst[l];
        const bool last = l == n_layers - 1;
        const void* prep = (prepared && gemm_mode == DGNN_GEMM_F16X2) ? prepared[l] : nullptr;
        if (last && dec2) {
            // the last layer's launch carries the decoder (only logits are written); shapes / layouts it does not take run layer and decoder apart
            int rc = DGNN_E_UNSUPPORTED;
            if (fuse_decoder && gemm_mode == DGNN_GEMM_F16X2 && n_logits == 2) {
                rc = prep ? dgnn_sage_layer_fused_decoder_fwd_p(rowptr, src, e_, n, h, nullptr, ldh, ci, edge_attr, lde, f_e, We[l], be[l], Wj[l], bj[l], Wi[l],
                                                                scale[l], shift[l], 1, co, W0, b0, scale1, shift1, c_hidden, W3, b3, n_logits, logits, prep, stream)
                          : dgnn_sage_layer_fused_decoder_fwd(rowptr, src, e_, n, h, nullptr, ldh, ci, edge_attr, lde, f_e, We[l], be[l], Wj[l], bj[l], Wi[l],
                                                              scale[l], shift[l], 1, co, W0, b0, scale1, shift1, c_hidden, W3, b3, n_logits, logits, stream);
            }
            if (rc != DGNN_E_UNSUPPORTED) return rc;
            if (fuse_decoder) prep = nullptr;     // (a decoder-carrying prepared block is not a plain layer's)
        }
        float* out = (last && !dec2) ? logits : ws.act[l & 1];
        int rc = DGNN_E_UNSUPPORTED;
        if (prep) rc = dgnn_sage_layer_fused_fwd_p(rowptr, src, e_, n, h, nullptr, ldh, ci, edge_attr, lde, f_e, We[l], be[l], Wj[l], bj[l], Wi[l], scale[l],
                                                   shift[l], 1, co, out, co, prep, stream);
        if (rc == DGNN_E_UNSUPPORTED)
            rc = dgnn_sage_layer_fused_fwd(rowptr, src, e_, n, h, nullptr, ldh, ci, edge_attr, lde, f_e, We[l], be[l], Wj[l], bj[l], Wi[l], scale[l], shift[l],
                                           1, co, out, co, gemm_mode, stream);
        if (rc != DGNN_OK) return rc;
        h = out;
        ldh = co;
    }
    if (dec2) return dgnn_decoder_fused_fwd(h, ldh, n_out, (int)ldh, W0, b0, scale1, shift1, c_hidden, W3, b3, n_logits, logits, n_logits, stream);
    return DGNN_OK;
}

extern "C" int dgnn_static_infer_fwd(const int64_t* edge_index, int64_t stride_row, int64_t stride_col, int64_t E, int plan_hint, int32_t* rowptr,
                                     int32_t* src, int32_t* eid, int32_t* plan_scratch, int64_t n, const float* x, int64_t ldx, const float* edge_attr, int64_t lde,
                                     int f_e, int n_layers, const int32_t* widths, const float* const* We, const float* const* be,
                                     const float* const* Wj, const float* const* bj, const float* const* Wi, const float* const* scale,
                                     const float* const* shift, const void* const* prepared, const float* W0, const float* b0, const float* scale1,
                                     const float* shift1, int c_hidden, const float* W3, const float* b3, int n_logits, int fuse_decoder, int gemm_mode,
                                     void* workspace, float* logits, void* stream) {
    DGNN_REQUIRE(n >= 0 && n_layers >= 1 && n_layers <= 16, DGNN_E_INVALID, "static_infer_fwd: bad sizes");
    int64_t n_dst[16];
    for (int l = 0; l < n_layers; ++l) n_dst[l] = n;
    return infer_chain("static_infer_fwd", edge_index, stride_row, stride_col, E, plan_hint, rowptr, src, eid, plan_scratch, /*attr_in_plan_order=*/eid == nullptr,
                       n, n_dst, x, ldx, edge_attr, lde, f_e, n_layers, widths, We, be, Wj, bj, Wi, scale, shift, prepared, W0, b0, scale1, shift1, c_hidden, W3,
                       b3, n_logits, fuse_decoder, gemm_mode, workspace, logits, stream);
}

// ---- one rank's part of a scene cut across GPUs WITHOUT a data-path exchange --------------------------------------------------------------
// The part holds, behind its owned cells, the rings of cells 1 .. L hops away (input rows resident: they are static) and layer l is computed for the owned
// cells AND the rings up to L-1-l hops out -- what the reference does per sampled batch (k-hop recomputation, learning/surfaceNetStaticEdgeFilters.py:232-275)
// applied to a whole part.  A ring of a 1/8 part of the 1M-tet scene is 3.7 % of its cells: ~5 % redundant work buys a step with no collective, no
// interior / boundary split and no transfer latency between the layers (the exchange form below spends 0.35 ms per step where this one spends 0.22).
// Cells are independent of their tile's composition in every fused kernel: the union of the ranks' logits is bit-identical to the whole scene's.
extern "C" int dgnn_static_infer_rings_fwd(const int64_t* edge_index, int64_t stride_row, int64_t stride_col, int64_t E, int plan_hint, int32_t* rowptr, int32_t* src,
                                           int32_t* eid, int32_t* plan_scratch, int attr_in_plan_order, int64_t n_loc, const int64_t* n_dst, const float* x,
                                           int64_t ldx, const float* edge_attr, int64_t lde, int f_e, int n_layers, const int32_t* widths, const float* const* We,
                                           const float* const* be, const float* const* Wj, const float* const* bj, const float* const* Wi,
                                           const float* const* scale, const float* const* shift, const void* const* prepared, const float* W0, const float* b0,
                                           const float* scale1, const float* shift1, int c_hidden, const float* W3, const float* b3, int n_logits,
                                           int fuse_decoder, int gemm_mode, void* workspace, float* logits, void* stream) {
    DGNN_REQUIRE(attr_in_plan_order || eid, DGNN_E_INVALID, "static_infer_rings_fwd: eid missing");
    return infer_chain("static_infer_rings_fwd", edge_index, stride_row, stride_col, E, plan_hint, rowptr, src, eid, plan_scratch, attr_in_plan_order != 0, n_loc,
                       n_dst, x, ldx, edge_attr, lde, f_e, n_layers, widths, We, be, Wj, bj, Wi, scale, shift, prepared, W0, b0, scale1, shift1, c_hidden, W3, b3,
                       n_logits, fuse_decoder, gemm_mode, workspace, logits, stream);
}

// ---- bf16 STORAGE (BASELINE config 3), whole scene or ring part, one call ------------------------------------------------------------------------------
// The chain SurfaceNet.inference_layer runs in bf16 storage when every layer has the fused form: layer 0 reads the caller's fp32 rows in place and
// starts the 16-bit rows (UNSIGNED when `mode` carries DGNN_BF16_ROWS_OUT_UNSIGNED: 9 significant bits behind the ReLU), the middle layers keep the
// format, the last layer's launch carries the decoder and writes fp32 logits.  Configurations outside that form: DGNN_E_UNSUPPORTED, nothing launched
// (the Python mirror then issues its per-layer chain).
extern "C" int dgnn_static_infer_rings_fwd_bf16(const int64_t* edge_index, int64_t stride_row, int64_t stride_col, int64_t E, int plan_hint, int32_t* rowptr,
                                                int32_t* src, int32_t* eid, int32_t* plan_scratch, int attr_in_plan_order, int64_t n_loc, const int64_t* n_dst,
                                                const float* x, int64_t ldx, const float* edge_attr, int64_t lde, int f_e, int n_layers, const int32_t* widths,
                                                const float* const* We, const float* const* be, const float* const* Wj, const float* const* bj,
                                                const float* const* Wi, const float* const* scale, const float* const* shift, const float* W0, const float* b0,
                                                const float* scale1, const float* shift1, int c_hidden, const float* W3, const float* b3, int n_logits, int mode,
                                                void* workspace, float* logits, void* stream) {
    const char* who = "static_infer_rings_fwd_bf16";
    DGNN_REQUIRE(n_loc >= 0 && E >= 0 && n_layers >= 2 && n_layers <= 16 && widths && We && be && Wj && bj && Wi && scale && shift && n_dst, DGNN_E_INVALID,
                 "%s: bad sizes / null table", who);
    for (int l = 0; l < n_layers; ++l)
        DGNN_REQUIRE(n_dst[l] >= 0 && n_dst[l] <= (l ? n_dst[l - 1] : n_loc), DGNN_E_INVALID, "%s: destination counts must not grow from layer to layer", who);
    if (n_dst[n_layers - 1] == 0) return DGNN_OK;
    DGNN_REQUIRE(x && edge_attr && workspace && logits && ((uintptr_t)workspace % 16) == 0, DGNN_E_INVALID, "%s: null / unaligned pointer", who);
    const bool build = edge_index != nullptr;
    DGNN_REQUIRE(rowptr && src && (!build || (eid && plan_scratch)) && (attr_in_plan_order || eid), DGNN_E_INVALID, "%s: plan arrays missing", who);
    const bool uns = (mode & DGNN_BF16_ROWS_OUT_UNSIGNED) != 0;
    const int base = mode & ~(DGNN_BF16_ROWS_IN_UNSIGNED | DGNN_BF16_ROWS_OUT_UNSIGNED);
    bool ok = W0 && b0 && W3 && b3 && base == DGNN_BF16_COMPENSATED && f_e == 20 && lde == 20 && ((uintptr_t)edge_attr % 16) == 0 && widths[0] <= 32 &&
              widths[0] % 2 == 0 && ((uintptr_t)x % 4) == 0 && widths[n_layers] == 128 && widths[n_layers - 1] > 64 && c_hidden == 64 && n_logits == 2;
    int maxw = 0;
    for (int l = 0; l < n_layers && ok; ++l) {
        const int ci = widths[l], co = widths[l + 1];
        const int nb = ci <= 32 ? 2 : (ci <= 64 ? 4 : 8);
        ok = ci > 0 && ci <= 128 && ci % nb == 0 && (co == 64 || co == 128) && We[l] && be[l] && Wj[l] && Wi[l] && ((scale[l] == nullptr) == (shift[l] == nullptr));
        maxw = co > maxw ? co : maxw;
    }
    ok = ok && n_loc * (int64_t)(ldx > maxw ? ldx : maxw) < ((int64_t)1 << 31);
    if (!ok) {
        dgnn_set_error("%s: a layer shape / operand layout / arithmetic mode outside the fully fused bf16-storage chain", who);
        return DGNN_E_UNSUPPORTED;
    }
    const Workspace ws = carve(workspace, n_dst[0], n_layers, widths);      // (sized for 4-byte rows: the 2-byte rows use half of each buffer)
    if (build) {
        DGNN_REQUIRE(E < INT32_MAX && n_loc < INT32_MAX, DGNN_E_UNSUPPORTED, "%s: E and n must fit int32", who);
        const int rc = dgnn_plan_build(edge_index, stride_row, stride_col, E, n_dst[0], n_loc, 1, plan_hint, rowptr, src, eid, plan_scratch, stream);
        if (rc != DGNN_OK) return rc;
    }
    const int32_t* e_ = attr_in_plan_order ? nullptr : eid;
    const void* h = x;
    int64_t ldh = ldx;
    for (int l = 0; l < n_layers; ++l) {
        const int ci = widths[l], co = widths[l + 1];
        const int fmt = uns ? ((l ? DGNN_BF16_ROWS_IN_UNSIGNED : 0) | DGNN_BF16_ROWS_OUT_UNSIGNED) : 0;
        if (l == n_layers - 1)
            return dgnn_sage_layer_fused_decoder_fwd_bf16(rowptr, src, e_, n_dst[l], static_cast<const uint16_t*>(h), nullptr, ldh, ci, edge_attr, lde, f_e, We[l],
                                                          be[l], Wj[l], bj[l], Wi[l], scale[l], shift[l], 1, co, W0, b0, scale1, shift1, c_hidden, W3, b3, n_logits,
                                                          logits, base | (uns ? DGNN_BF16_ROWS_IN_UNSIGNED : 0), stream);
        uint16_t* out = reinterpret_cast<uint16_t*>(ws.act[l & 1]);
        const int rc = dgnn_sage_layer_fused_fwd_bf16(rowptr, src, e_, n_dst[l], h, l == 0, nullptr, ldh, ci, edge_attr, lde, f_e, We[l], be[l], Wj[l], bj[l], Wi[l],
                                                      scale[l], shift[l], 1, co, out, co, base | fmt, stream);
        if (rc != DGNN_OK) return rc;
        h = out;
        ldh = co;
    }
    return DGNN_OK;
}

// ---- one rank's part of a scene cut across GPUs, ONE call per step (dgnn_amd/partition.py run_partitioned_layers) -------------------------
// The launch chain PartitionedScene.inference_layer issued from Python -- plan of the local bipartite graph, layer 0 over the owned cells, then per
// later layer [interior cells | wait for the halo | boundary cells] with the exchange of the layer's output rows started right behind it -- was
// host-bound at a strong-scaling shard: a 1/8 part of the 1M-tet scene (126k owned cells) has 0.2 ms of GPU work behind 0.58 ms of interpreter time
// (tools/bench_partition_rank.py).  Same kernels, ranges and order: bit-identical to the per-layer calls and therefore to the whole scene.
extern "C" int dgnn_static_infer_partitioned_fwd(const int64_t* edge_index, int64_t stride_row, int64_t stride_col, int64_t E, int plan_hint, int32_t* rowptr,
                                                 int32_t* src, int32_t* eid, int32_t* plan_scratch, int attr_in_plan_order, int64_t n_own, int64_t n_interior,
                                                 int64_t n_halo, const float* x, int64_t ldx, const float* edge_attr, int64_t lde, int f_e, int n_layers,
                                                 const int32_t* widths, const float* const* We, const float* const* be, const float* const* Wj,
                                                 const float* const* bj, const float* const* Wi, const float* const* scale, const float* const* shift,
                                                 const void* const* prepared, const float* W0, const float* b0, const float* scale1, const float* shift1,
                                                 int c_hidden, const float* W3, const float* b3, int n_logits, int fuse_decoder, int gemm_mode,
                                                 dgnn_halo_plan* halo, void* comm, void* send_buf, void* workspace, float* logits, void* stream) {
    DGNN_REQUIRE(n_own >= 0 && n_halo >= 0 && n_interior >= 0 && n_interior <= n_own && E >= 0 && n_layers >= 1 && n_layers <= 16 && widths && We && be &&
                     Wj && bj && Wi && scale && shift,
                 DGNN_E_INVALID, "static_infer_partitioned_fwd: bad sizes / null table");
    if (n_own == 0 && n_halo == 0) return DGNN_OK;
    const int64_t n_loc = n_own + n_halo;
    DGNN_REQUIRE(x && edge_attr && workspace && logits && ((uintptr_t)workspace % 16) == 0, DGNN_E_INVALID,
                 "static_infer_partitioned_fwd: null / unaligned pointer");
    const bool build = edge_index != nullptr;
    DGNN_REQUIRE(rowptr && src && (!build || (eid && plan_scratch)) && (attr_in_plan_order || eid), DGNN_E_INVALID,
                 "static_infer_partitioned_fwd: plan arrays missing");
    DGNN_REQUIRE(!halo || (dgnn_halo_recv_rows(halo) == n_halo && (dgnn_halo_send_rows(halo) == 0 || send_buf)), DGNN_E_INVALID,
                 "static_infer_partitioned_fwd: the halo plan receives %lld rows, the part has %lld halo rows (or no send buffer)",
                 (long long)(halo ? dgnn_halo_recv_rows(halo) : 0), (long long)n_halo);
    DGNN_REQUIRE(halo || n_halo == 0, DGNN_E_INVALID, "static_infer_partitioned_fwd: halo rows without a halo plan");
    const bool dec2 = W0 != nullptr;
    DGNN_REQUIRE(!dec2 || (b0 && W3 && b3), DGNN_E_INVALID, "static_infer_partitioned_fwd: incomplete decoder");
    bool ok = f_e == 20 && lde == 20 && ((uintptr_t)edge_attr % 16) == 0 && (dec2 || W3 == nullptr);
    int maxw = 0;
    for (int l = 0; l < n_layers && ok; ++l) {
        const int ci = widths[l], co = widths[l + 1];
        ok = ci > 0 && ci <= 128 && (co == 64 || co == 128) && (ci <= 64 || (co == 128 && ci % 2 == 0)) && We[l] && be[l] && Wj[l] && Wi[l] &&
             ((scale[l] == nullptr) == (shift[l] == nullptr));
        maxw = co > maxw ? co : maxw;
    }
    ok = ok && n_loc * (int64_t)(ldx > maxw ? ldx : maxw) < ((int64_t)1 << 31);
    if (widths[0] > 64) ok = ok && ldx % 2 == 0 && ((uintptr_t)x % 8) == 0;
    if (dec2) ok = ok && widths[n_layers] == 128 && c_hidden == 64 && (n_logits == 1 || n_logits == 2);
    if (!ok) {
        dgnn_set_error("static_infer_partitioned_fwd: a layer shape / operand layout outside the fused kernels");
        return DGNN_E_UNSUPPORTED;
    }
    const Workspace ws = carve(workspace, n_loc, n_layers, widths);
    if (build) {
        DGNN_REQUIRE(E < INT32_MAX && n_loc < INT32_MAX, DGNN_E_UNSUPPORTED, "static_infer_partitioned_fwd: E and n must fit int32");
        const int rc = dgnn_plan_build(edge_index, stride_row, stride_col, E, n_own, n_loc, 1, plan_hint, rowptr, src, eid, plan_scratch, stream);
        if (rc != DGNN_OK) return rc;
    }
    const int32_t* e_ = attr_in_plan_order ? nullptr : eid;
    const float* h = x;
    int64_t ldh = ldx;
    bool decoder_apart = dec2;
    for (int l = 0; l < n_layers; ++l) {
        const int ci = widths[l], co = widths[l + 1];
        const bool last = l == n_layers - 1;
        const void* prep = (prepared && gemm_mode == DGNN_GEMM_F16X2) ? prepared[l] : nullptr;
        bool with_dec = last && dec2 && fuse_decoder && gemm_mode == DGNN_GEMM_F16X2 && n_logits == 2;
        if (last && dec2 && fuse_decoder && !with_dec) prep = nullptr;      // (a decoder-carrying prepared block is not a plain layer's)
        float* out = (last && !dec2) ? logits : ws.act[l & 1];
        // layer 0 reads input rows only (the halo's are resident); later layers: interior cells, then -- the halo has landed -- boundary cells
        const int64_t cut[3] = {0, l == 0 ? n_own : n_interior, n_own};
        for (int part = 0; part < 2; ++part) {
            const int64_t b = cut[part], e = cut[part + 1];
            if (part == 1 && l > 0 && halo) {
                const int rc = dgnn_halo_exchange_wait(halo, stream);
                if (rc != DGNN_OK) return rc;
            }
            if (e <= b) continue;
            const float* xd = b ? h + b * ldh : nullptr;
            int rc = DGNN_E_UNSUPPORTED;
            if (with_dec) {
                float* lg = logits + b * n_logits;
                rc = prep ? dgnn_sage_layer_fused_decoder_fwd_p(rowptr + b, src, e_, e - b, h, xd, ldh, ci, edge_attr, lde, f_e, We[l], be[l], Wj[l], bj[l], Wi[l],
                                                                scale[l], shift[l], 1, co, W0, b0, scale1, shift1, c_hidden, W3, b3, n_logits, lg, prep, stream)
                          : dgnn_sage_layer_fused_decoder_fwd(rowptr + b, src, e_, e - b, h, xd, ldh, ci, edge_attr, lde, f_e, We[l], be[l], Wj[l], bj[l], Wi[l],
                                                              scale[l], shift[l], 1, co, W0, b0, scale1, shift1, c_hidden, W3, b3, n_logits, lg, stream);
                if (rc == DGNN_E_UNSUPPORTED) {      // decided on the layer's first launch: layer and decoder apart for both ranges
                    with_dec = false;
                    prep = nullptr;
                } else if (rc != DGNN_OK) {
                    return rc;
                } else {
                    decoder_apart = false;
                    continue;
                }
            }
            float* o = out + b * co;
            if (prep) rc = dgnn_sage_layer_fused_fwd_p(rowptr + b, src, e_, e - b, h, xd, ldh, ci, edge_attr, lde, f_e, We[l], be[l], Wj[l], bj[l], Wi[l], scale[l],
                                                       shift[l], 1, co, o, co, prep, stream);
            if (rc == DGNN_E_UNSUPPORTED)
                rc = dgnn_sage_layer_fused_fwd(rowptr + b, src, e_, e - b, h, xd, ldh, ci, edge_attr, lde, f_e, We[l], be[l], Wj[l], bj[l], Wi[l], scale[l], shift[l],
                                               1, co, o, co, gemm_mode, stream);
            if (rc != DGNN_OK) return rc;
        }
        if (!last && halo) {       // the rows the peers' boundary cells read next, the peers' rows into this buffer's tail
            const int rc = dgnn_halo_exchange_start(halo, comm, out, co, co, 4, send_buf, stream);
            if (rc != DGNN_OK) return rc;
        }
        h = out;
        ldh = co;
    }
    if (decoder_apart) return dgnn_decoder_fused_fwd(h, ldh, n_own, (int)ldh, W0, b0, scale1, shift1, c_hidden, W3, b3, n_logits, logits, n_logits, stream);
    return DGNN_OK;
}
