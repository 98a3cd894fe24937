// Fused SAGE inference layer: one persistent launch per layer computes
//
//     y_i = act( ( Wj . mean_{e->i}( x[src_e] * (We.A_e + be) ) + bj + Wi . x_i ) * scale + shift )
//
// (reference learning/surfaceNetStaticEdgeFilters.py:66-96 + BatchNorm(eval) + ReLU at :345-346) with
// the aggregate living only in LDS.  The reference materialises three [E, C_in] tensors per layer
// (phi, x_j, x_j*phi: 6 GB at 1M tets / C=128); this kernel reads x, edge_attr and the indices once
// and writes y once (1360 B/tet at 128->128).
//
// Work decomposition (512-thread workgroup = 8 wavefronts, one workgroup per CU, persistent):
//   * wavefronts 4..7  PRODUCERS (VALU + memory): each owns TILE/4 destination tets of the tile.
//     All loads of the tile are issued up front -- the 4 gathered source rows and the own row of every
//     tet (one coalesced row per wave instruction, CPL floats per lane) and the tile's edge-attribute
//     block (contiguous, edges are in plan order) -- so one HBM round trip covers the tile.  The
//     filter phi = We.A + be is 20 FMAs per channel with We held in registers and A broadcast from LDS;
//     products and the in-order 4-term sum reproduce the reference's scatter order.  The mean row and
//     the tet's own row are written side by side into the LDS A-tile [TILE][2*CIN_PAD].
//   * wavefronts 0..3  CONSUMERS (matrix cores): [a | x_i] . [Wj | Wi]^T on v_mfma_f32_32x32x2_f32.
//     Each consumer keeps its 32 output columns of the concatenated weight matrix in REGISTERS for the
//     whole launch (K/2 VGPRs), so the only per-tile LDS traffic is one ds_read_b128 of the A-tile per
//     4 MFMAs.  BatchNorm(eval) scale/shift and ReLU are applied on the accumulators and the tile is
//     stored straight to HBM (128-byte row segments per half-wave).
//   * the A-tile is double buffered; producers fill tile t+1 while consumers multiply tile t; one
//     workgroup barrier per tile.  fp32 MFMA runs at the fp32 vector rate (MI355X guide), so the matrix
//     pipe is the critical resource and VALU/memory work hides beside it.
//   * tile -> workgroup map is XCD-aware: workgroup b runs on XCD b%8 (observed placement; used for L2
//     locality only), and each XCD walks one contiguous eighth of the tets so that gathered neighbour
//     rows are mostly resident in that XCD's L2.
//
// LDS A-tile row stride is K+4 floats: the 16-lane groups of ds_read_b128 then hit 16 distinct 16-byte
// slots (conflict free) and rows stay 16-byte aligned.
#include "common.h"

namespace {

constexpr int FE = 20;

template <int CIN_PAD, int COUT>
struct FusedCfg {
    static constexpr int K = 2 * CIN_PAD;
    static constexpr int NSLICE = COUT / 32;  // column slices of 32
    static constexpr int RG = 4 / NSLICE;     // row groups per tile
    static constexpr int TILE = 32 * RG;      // destination tets per tile
    static constexpr int LDA = K + 4;
    static constexpr int TPW = TILE / 4;                  // tets per producer wave
    static constexpr int CPL = CIN_PAD > 64 ? 2 : 1;      // channels per lane in the producer
    static constexpr int EA_FLOATS = TPW * 4 * FE;        // edge-attribute floats per producer wave
    static constexpr int EA_PAD = (EA_FLOATS + 255) / 256 * 256;
    static constexpr int SMEM_FLOATS = 2 * TILE * LDA + 4 * EA_PAD;
};

template <int CPL>
__device__ __forceinline__ void ld_row(float (&v)[CPL], const float* p, bool on) {
    if (CPL == 2) {
        float2 t = on ? *reinterpret_cast<const float2*>(p) : make_float2(0.f, 0.f);
        v[0] = t.x;
        v[CPL - 1] = t.y;
    } else {
        v[0] = on ? *p : 0.f;
    }
}

template <int CIN_PAD, int COUT>
__global__ void __launch_bounds__(512, 2)
k_sage_fused(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ src, int64_t n_dst,
             const float* __restrict__ x, int64_t ldx, int c_in, const float* __restrict__ ea, int64_t lde,
             const float* __restrict__ We, const float* __restrict__ be, const float* __restrict__ Wj,
             const float* __restrict__ bj, const float* __restrict__ Wi, const float* __restrict__ scale,
             const float* __restrict__ shift, int relu, float* __restrict__ out, int64_t ldo, int64_t ntiles) {
    using C = FusedCfg<CIN_PAD, COUT>;
    constexpr int K = C::K, LDA = C::LDA, TILE = C::TILE, TPW = C::TPW, CPL = C::CPL;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const abuf = smem;                       // [2][TILE][LDA]
    float* const eabuf = smem + 2 * TILE * LDA;     // [4 producer waves][EA_PAD]

    const int lane = lane_id(), w = wave_id_uniform();
    const int h = lane >> 5, l31 = lane & 31;

    // XCD-aware persistent schedule: XCD x owns tiles [x*per, (x+1)*per); its workgroups walk them together
    const int nwg = gridDim.x, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, wg_per_xcd = (nwg + 7 - xcd) >> 3;
    const int64_t per = (ntiles + 7) / 8;
    const int64_t t_lo = xcd * per, t_hi = min(ntiles, t_lo + per);
    // number of tiles this workgroup processes (identical trip count for all 8 waves -> barriers match)
    int64_t my_n = 0;
    if (t_lo + slot < t_hi) my_n = (t_hi - t_lo - slot + wg_per_xcd - 1) / wg_per_xcd;

    if (w < 4) {
        // ------------------------------------------------------------------ consumer (MFMA)
        const int cs = w % C::NSLICE, rg = w / C::NSLICE;
        const int col = cs * 32 + l31;
        float wr[K / 2];
#pragma unroll
        for (int S = 0; S < K / 8; ++S) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = 8 * S + 4 * h + j;  // k-permutation shared with the A-tile read below
                float v = 0.f;
                if (k < CIN_PAD) {
                    if (k < c_in) v = Wj[(int64_t)col * c_in + k];
                } else {
                    if (k - CIN_PAD < c_in) v = Wi[(int64_t)col * c_in + (k - CIN_PAD)];
                }
                wr[S * 4 + j] = v;
            }
        }
        const float bb = bj ? bj[col] : 0.f;
        const float sc = scale ? scale[col] : 1.f;
        const float sh = scale ? shift[col] : 0.f;
        const bool has_scale = scale != nullptr;

        for (int64_t it = 0; it < my_n; ++it) {
            const int64_t tile = t_lo + slot + it * wg_per_xcd;
            __syncthreads();  // tile `it` is complete in abuf[it&1]
            const float* A = abuf + (it & 1) * TILE * LDA + (rg * 32 + l31) * LDA + 4 * h;
            f32x16 acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
            for (int S = 0; S < K / 8; ++S) {
                const f32x4 av = *reinterpret_cast<const f32x4*>(A + 8 * S);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], wr[S * 4 + j], acc, 0, 0, 0);
            }
            const int64_t row0 = tile * TILE + rg * 32 + 4 * h;
            float* o = out + row0 * ldo + col;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rr = (r & 3) + 8 * (r >> 2);
                if (row0 + rr < n_dst) {
                    float v = acc[r] + bb;
                    if (has_scale) v = __fmaf_rn(v, sc, sh);
                    if (relu) v = fmaxf(v, 0.f);
                    o[(int64_t)rr * ldo] = v;
                }
            }
        }
        __syncthreads();  // matches the producers' final barrier
    } else {
        // ------------------------------------------------------------------ producer (gather + filter + mean)
        const int pw = w - 4;
        const int c0 = lane * CPL;
        const bool on = c0 < c_in;  // c_in is a multiple of CPL (host-checked)
        float wl[CPL][FE], bl[CPL];
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
            bl[j] = on ? be[c0 + j] : 0.f;
#pragma unroll
            for (int f = 0; f < FE; ++f) wl[j][f] = on ? We[(int64_t)(c0 + j) * FE + f] : 0.f;
        }
        float* const myea = eabuf + pw * C::EA_PAD;

        for (int64_t it = 0; it <= my_n; ++it) {
            if (it < my_n) {
                const int64_t tile = t_lo + slot + it * wg_per_xcd;
                const int64_t i0 = tile * TILE + pw * TPW;  // first tet of this wave
                float* const Arow = abuf + (it & 1) * TILE * LDA + (pw * TPW) * LDA;
                // ---- indices (wave-uniform -> scalar loads)
                int beg[TPW + 1];
                bool regular = (i0 + TPW <= n_dst);
                if (regular) {
#pragma unroll
                    for (int r = 0; r <= TPW; ++r) beg[r] = rowptr[i0 + r];
#pragma unroll
                    for (int r = 0; r < TPW; ++r) regular = regular && (beg[r + 1] - beg[r] == 4);
                }
                if (regular) {
                    const int e0 = beg[0];
                    // ---- issue every load of this wave's TPW tets: edge-attribute block, own rows, 4 neighbours
                    constexpr int NV4 = C::EA_FLOATS / 4;  // float4 count of the attribute block (lde == FE here)
                    const float* eab = ea + (int64_t)e0 * lde;
                    f32x4 ev[(NV4 + 63) / 64];
#pragma unroll
                    for (int q = 0; q < (NV4 + 63) / 64; ++q) {
                        const int idx = q * 64 + lane;
                        ev[q] = idx < NV4 ? *reinterpret_cast<const f32x4*>(eab + 4 * idx) : f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                    int sidx[TPW * 4];
#pragma unroll
                    for (int q = 0; q < TPW * 4; ++q) sidx[q] = src[e0 + q];
                    float xd[TPW][CPL], xr[TPW * 4][CPL];
#pragma unroll
                    for (int r = 0; r < TPW; ++r) ld_row<CPL>(xd[r], x + (i0 + r) * ldx + c0, on);
#pragma unroll
                    for (int q = 0; q < TPW * 4; ++q) ld_row<CPL>(xr[q], x + (int64_t)sidx[q] * ldx + c0, on);
                    // attribute block -> this wave's LDS staging area (read back as broadcasts)
#pragma unroll
                    for (int q = 0; q < (NV4 + 63) / 64; ++q) {
                        const int idx = q * 64 + lane;
                        if (idx < NV4) *reinterpret_cast<f32x4*>(myea + 4 * idx) = ev[q];
                    }
                    // ---- filter, multiply, in-order mean; write [a | x_i] rows of the A-tile
#pragma unroll
                    for (int r = 0; r < TPW; ++r) {
                        float acc[CPL];
#pragma unroll
                        for (int j = 0; j < CPL; ++j) acc[j] = 0.f;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float* ap = myea + (r * 4 + e) * FE;
                            float A[FE];
#pragma unroll
                            for (int f = 0; f < FE; f += 4) {
                                const f32x4 t = *reinterpret_cast<const f32x4*>(ap + f);
                                A[f] = t[0]; A[f + 1] = t[1]; A[f + 2] = t[2]; A[f + 3] = t[3];
                            }
#pragma unroll
                            for (int j = 0; j < CPL; ++j) {
                                float p = bl[j];
#pragma unroll
                                for (int f = 0; f < FE; ++f) p = __fmaf_rn(wl[j][f], A[f], p);
                                acc[j] = __fadd_rn(acc[j], __fmul_rn(xr[r * 4 + e][j], p));
                            }
                        }
                        float* dst = Arow + r * LDA;
                        if (CPL == 2) {
                            if (lane < CIN_PAD / 2) {
                                *reinterpret_cast<float2*>(dst + c0) = make_float2(acc[0] * 0.25f, acc[CPL - 1] * 0.25f);
                                *reinterpret_cast<float2*>(dst + CIN_PAD + c0) = make_float2(xd[r][0], xd[r][CPL - 1]);
                            }
                        } else {
                            if (lane < CIN_PAD) {
                                dst[c0] = acc[0] * 0.25f;
                                dst[CIN_PAD + c0] = xd[r][0];
                            }
                        }
                    }
                } else {
                    // ---- generic path: any in-degree, tile tail; one edge at a time (rare)
                    for (int r = 0; r < TPW; ++r) {
                        const int64_t i = i0 + r;
                        float acc[CPL], xdv[CPL];
#pragma unroll
                        for (int j = 0; j < CPL; ++j) acc[j] = xdv[j] = 0.f;
                        if (i < n_dst) {
                            const int b = rowptr[i], e_end = rowptr[i + 1];
                            ld_row<CPL>(xdv, x + i * ldx + c0, on);
                            for (int k = b; k < e_end; ++k) {
                                const int s = src[k];
                                const float* ar = ea + (int64_t)k * lde;
                                float xv[CPL];
                                ld_row<CPL>(xv, x + (int64_t)s * ldx + c0, on);
#pragma unroll
                                for (int j = 0; j < CPL; ++j) {
                                    float p = bl[j];
#pragma unroll
                                    for (int f = 0; f < FE; ++f) p = __fmaf_rn(wl[j][f], ar[f], p);
                                    acc[j] = __fadd_rn(acc[j], __fmul_rn(xv[j], p));
                                }
                            }
                            const float cnt = (float)max(e_end - b, 1);
#pragma unroll
                            for (int j = 0; j < CPL; ++j) acc[j] = __fdiv_rn(acc[j], cnt);
                        }
                        float* dst = Arow + r * LDA;
                        if (lane * CPL < CIN_PAD) {
#pragma unroll
                            for (int j = 0; j < CPL; ++j) {
                                dst[c0 + j] = acc[j];
                                dst[CIN_PAD + c0 + j] = xdv[j];
                            }
                        }
                    }
                }
            }
            __syncthreads();  // publishes tile `it`; consumers are done with the other buffer
        }
    }
}

template <int CIN_PAD, int COUT>
int launch_fused(const int32_t* rowptr, const int32_t* src, int64_t n_dst, const float* x, int64_t ldx, int c_in,
                 const float* ea, int64_t lde, const float* We, const float* be, const float* Wj, const float* bj,
                 const float* Wi, const float* scale, const float* shift, int relu, float* out, int64_t ldo,
                 hipStream_t stream) {
    using C = FusedCfg<CIN_PAD, COUT>;
    const int64_t ntiles = dgnn_cdiv(n_dst, C::TILE);
    const size_t smem = sizeof(float) * C::SMEM_FLOATS;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_sage_fused<CIN_PAD, COUT>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        attr_set = true;
    }
    int grid = (int)(ntiles < DGNN_NUM_CU ? ntiles : DGNN_NUM_CU);
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL((k_sage_fused<CIN_PAD, COUT>), dim3(grid), dim3(512), smem, stream, rowptr, src, n_dst, x, ldx, c_in,
                       ea, lde, We, be, Wj, bj, Wi, scale, shift, relu, out, ldo, ntiles);
    return dgnn_check_launch("sage_layer_fused_fwd");
}

}  // namespace

extern "C" int dgnn_sage_layer_fused_fwd(const int32_t* rowptr, const int32_t* src, int64_t n_dst, const float* x_src,
                                         int64_t ldx, int c_in, const float* edge_attr_sorted, int64_t lde, int f_e,
                                         const float* We, const float* be, const float* Wj, const float* bj, const float* Wi,
                                         const float* scale, const float* shift, int relu, int c_out, float* out, int64_t ldo,
                                         void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DGNN_REQUIRE(n_dst >= 0 && c_in > 0 && c_out > 0, DGNN_E_INVALID, "sage_layer_fused_fwd: bad sizes");
    if (n_dst == 0) return DGNN_OK;
    DGNN_REQUIRE(rowptr && src && x_src && edge_attr_sorted && We && be && Wj && Wi && out, DGNN_E_INVALID,
                 "sage_layer_fused_fwd: null pointer");
    DGNN_REQUIRE((scale == nullptr) == (shift == nullptr), DGNN_E_INVALID, "sage_layer_fused_fwd: scale/shift must come together");
    DGNN_REQUIRE(f_e == FE && lde == FE, DGNN_E_UNSUPPORTED, "sage_layer_fused_fwd: needs f_e == 20 and packed edge rows (lde == 20)");
    DGNN_REQUIRE(((uintptr_t)edge_attr_sorted % 16) == 0, DGNN_E_UNSUPPORTED, "sage_layer_fused_fwd: edge_attr must be 16-byte aligned");
    DGNN_REQUIRE(c_in <= 128 && (c_out == 64 || c_out == 128), DGNN_E_UNSUPPORTED,
                 "sage_layer_fused_fwd: supports c_in <= 128 and c_out in {64,128} (got %d -> %d)", c_in, c_out);
#define GO(CP, CO) return launch_fused<CP, CO>(rowptr, src, n_dst, x_src, ldx, c_in, edge_attr_sorted, lde, We, be, Wj, bj, Wi, \
                                               scale, shift, relu, out, ldo, stream)
    if (c_in <= 32) { if (c_out == 64) GO(32, 64); else GO(32, 128); }
    if (c_in <= 64) { if (c_out == 64) GO(64, 64); else GO(64, 128); }
    DGNN_REQUIRE(c_in % 2 == 0 && ldx % 2 == 0 && ((uintptr_t)x_src % 8) == 0, DGNN_E_UNSUPPORTED,
                 "sage_layer_fused_fwd: c_in > 64 needs even c_in / ldx and 8-byte aligned x");
    DGNN_REQUIRE(c_out == 128, DGNN_E_UNSUPPORTED, "sage_layer_fused_fwd: c_in > 64 supports c_out == 128 only");
    GO(128, 128);
#undef GO
}
