// Fused SAGE inference layer: one persistent launch per layer computes
//
//     y_i = act( ( Wj . mean_{e->i}( x[src_e] * (We.A_e + be) ) + bj + Wi . x_i ) * scale + shift )
//
// (reference learning/surfaceNetStaticEdgeFilters.py:66-96 + BatchNorm(eval) + ReLU at :345-346) with
// the aggregate living only in LDS.  The reference materialises three [E, C_in] tensors per layer
// (phi, x_j, x_j*phi: 6 GB at 1M tets / C=128); this kernel reads x, edge_attr and the indices once
// and writes y once (1360 B/tet at 128->128).
//
// Measured facts that shape the design (tools/ubench_pipes.hip, tools/ubench_interleave.hip on MI355X):
// fp32 MFMA and fp32 VALU work on one SIMD do NOT overlap -- their times add, from different waves and
// from one wave alike -- and a wave that wants VALU issue while another wave streams dependent MFMAs
// advances about one instruction per MFMA.  So wave specialisation (gather waves beside MFMA waves)
// only starves the gather waves; what matters is (1) every SIMD always has VALU or MFMA work, (2) HBM
// latency sits behind the MFMA phase, (3) one barrier per tile.
//
// Structure (512-thread workgroup = 8 wavefronts = 2 per SIMD, one workgroup per CU, persistent).
// Every wavefront runs the same two-phase loop over the workgroup's tiles of TILE destination tets:
//   P  gather/filter phase (VALU): the wave owns TILE/8 tets.  Their 4 neighbour rows and own rows were
//      requested one tile ahead (indices two tiles ahead: one coalesced VGPR load each, turned into scalar
//      row offsets by v_readlane), the contiguous edge-attribute block arrived by LDS-DMA in a private strip.
//      phi = We.A + be is 20 FMAs per channel with We in registers and A broadcast from the strip;
//      products and the in-order 4-term sum reproduce the reference's scatter order.  The mean row and the
//      tet's own row go side by side into the LDS A-tile.  Then the wave issues all loads of its tets of
//      the NEXT tile and hits the workgroup barrier (LDS traffic only; the loads fly across it).
//   C  matrix phase (MFMA): [a | x_i] . [Wj | Wi]^T.  The 8 waves split the tile as (32-column slice) x
//      (K half: the `a` half with Wj, the `x_i` half with Wi) [x row group], so a wave keeps only its half-K
//      weights in registers for the whole launch and streams its A operand from LDS.  Two modes:
//        MODE 0  v_mfma_f32_32x32x2_f32: bit-faithful fp32 fmaf chains;
//        MODE 1  operands split exactly into 3 bf16 parts, the 6 partial products of weight >= 2^-18 on
//                v_mfma_f32_32x32x16_bf16 with fp32 accumulation (dropped terms <= 2^-25: fp32-class result
//                at 6/16 of the matrix time).
//      The two K-halves exchange half of their accumulators through LDS; each finishes 8 of the 16
//      accumulator rows (bias, BatchNorm scale/shift, ReLU) and stores them -- one tile later, right after
//      the next barrier, so that one barrier per tile orders both the A-tile hand-off and the exchange.
// HBM latency of tile t+1 is therefore covered by the whole matrix phase of tile t.
//
// tile -> workgroup map is XCD-aware: workgroup b runs on XCD b%8 (observed placement; used for L2
// locality only) and each XCD walks one contiguous eighth of the tets, so gathered neighbour rows are
// mostly resident in that XCD's L2.
//
// LDS A-tile row stride is K+4 floats: the 16-lane groups of ds_read_b128 then hit 16 distinct 16-byte
// slots (conflict free) and rows stay 16-byte aligned.
#include <type_traits>

#include "fused_common.h"

int64_t* g_dgnn_trace_buf = nullptr;
int64_t g_dgnn_trace_cap = 0;
extern "C" int dgnn_debug_trace_buffer(int64_t* dev_buf, int64_t n) {
    g_dgnn_trace_buf = dev_buf;
    g_dgnn_trace_cap = n;
    return DGNN_OK;
}

namespace {
using namespace fused;

// MODE 0: fp32 MFMA (v_mfma_f32_32x32x2_f32), bit-faithful fp32 fmaf chains.
// MODE 1: split-bf16 MFMA (v_mfma_f32_32x32x16_bf16): both operands are split exactly into 3 bf16 parts and the
//         6 partial products of relative weight >= 2^-18 are accumulated in fp32 (dropped: mid*lo, lo*mid, lo*lo
//         <= 2^-25 relative, below fp32 rounding).  Same fp32-class result at 6/16 of the matrix-pipe time.
template <int CIN_PAD, int COUT, int MODE>
struct FusedCfg {
    static constexpr int K = 2 * CIN_PAD;
    static constexpr int NSLICE = COUT / 32;           // column slices of 32
    static constexpr int RG = NWAVE / (2 * NSLICE);    // row groups per tile (K is split in 2)
    static constexpr int TILE = 32 * RG;               // destination tets per tile
    static constexpr int LDA = K + 4;
    // MODE 1 A-tile: per row K/8 octets of 48 bytes [hi 8 x bf16 | mid | lo] + 16 pad bytes (row stride = odd
    // number of 16-byte slots -> conflict-free ds_read_b128); counted in floats for the carve-up below
    static constexpr int ROWB = (K / 8) * 48 + 16;
    static constexpr int A_FLOATS = MODE == 0 ? TILE * LDA : TILE * ROWB / 4;
    static constexpr int TPW = TILE / NWAVE;           // tets gathered per wave
    static constexpr int CPL = CIN_PAD > 64 ? 2 : 1;   // channels per lane in the gather phase
    static constexpr int NQ = TPW * 4;                 // neighbour rows per wave
    static constexpr int EA_FLOATS = NQ * FE;          // edge-attribute floats per wave and tile
    static constexpr int NV4 = EA_FLOATS / 4;
    static constexpr int NEV = (NV4 + 63) / 64;        // float4 loads per lane for the attribute block
    static constexpr int EA_PAD = NEV * 256;
    static constexpr int S_STEPS = CIN_PAD / 8;        // k-steps of 8 per K half
    static constexpr int RED_FLOATS = NWAVE * 8 * 64;  // one partial-sum exchange buffer
    static constexpr int SMEM_FLOATS = 2 * A_FLOATS + NWAVE * EA_PAD + 2 * RED_FLOATS;
    static constexpr int FW_FLOATS = (CIN_PAD / 16) * 6 * 64;   // FM: the filter's matrix-core operand, [16-channel block][5 k-steps + bias][64 lanes]
};

// FM (MODE 0 only; round 6): the filter product phi = We . A + be on the fp32 matrix cores too (v_mfma_f32_16x16x4_f32) instead of 20 FMAs per channel
// and edge on the VALU.  fp32 MFMA and fp32 VALU time ADD on a SIMD (header); nominally both pipes do 32 multiply-adds per clock (SIMD-32), but measured on
// the 128 -> 128 layer the 153 M VALU instructions per 1M tets this removes cost 3.9 SIMD cycles each (dependent fmaf chains at two wavefronts per SIMD
// run at half the issue rate; SQ_ACTIVE_INST_VALU says the same) against 32 cycles for each of the 10 M matrix instructions that replace them: the
// wave's 16 edges (4 tets) x 16 channels x 4 attributes per instruction, C input = the bias, k ascending -- the same fmaf chain per (edge, channel) as the
// VALU form (be, then attributes 0..19 in order).  The C/D layout puts the 4 in-edges of tet g = lane >> 4 into the 4 accumulator registers of the lanes
// (., g): the in-order 4-term sum with the neighbour rows is in-lane, as in the fp16 kernels.  Lane (n = lane & 15, g) owns, of tet g's rows, the channels
// chan(nb, n) = 64 (nb / VW) + VW n + nb % VW, nb < CIN_PAD / 16, VW = min(4, CIN_PAD / 16): contiguous pieces of 8 / 16 bytes per lane, so that the 16
// lanes of a tet cover whole cache lines with one load instruction.
template <int CIN_PAD, int COUT, int MODE, int FM = 0>
__global__ void __launch_bounds__(512, 2)
k_sage_fused(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ src, const int32_t* __restrict__ eid, int64_t n_dst,
             const float* __restrict__ x, const float* __restrict__ xdst, int64_t ldx, int c_in, const float* __restrict__ ea, int64_t lde,
             const float* __restrict__ We, const float* __restrict__ be, const float* __restrict__ Wj,
             const float* __restrict__ bj, const float* __restrict__ Wi, const float* __restrict__ scale,
             const float* __restrict__ shift, int relu, float* __restrict__ out, int64_t ldo, int64_t ntiles,
             int64_t* __restrict__ trace, int64_t trace_cap) {
    using C = FusedCfg<CIN_PAD, COUT, MODE>;
    constexpr int LDA = C::LDA, TILE = C::TILE, TPW = C::TPW, CPL = C::CPL, NQ = C::NQ, NEV = C::NEV, NV4 = C::NV4;
    constexpr int ROWB = C::ROWB;
    static_assert(FM == 0 || (MODE == 0 && TPW % 4 == 0), "the matrix-core filter product: fp32 mode, whole blocks of 4 tets per wave");
    constexpr int EB = FM ? TPW / 4 : 1;                  // blocks of 16 edges (4 tets) per wave and tile
    constexpr int NBK = CIN_PAD / 16;                     // blocks of 16 channels
    constexpr int VW = NBK < 4 ? NBK : 4;                 // contiguous channels per lane and segment
    constexpr int NSEG = NBK / VW;                        // segments (64 channels apart)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const abuf = smem;                                  // [2][A_FLOATS]
    float* const eabuf = smem + 2 * C::A_FLOATS;               // [NWAVE][EA_PAD]
    float* const redbuf = eabuf + NWAVE * C::EA_PAD;           // [2][NWAVE][8][64]
    float* const fwbuf = redbuf + 2 * C::RED_FLOATS;           // FM only: [NBK][6][64]

    const int lane = lane_id(), w = wave_id_uniform();
    const int h = lane >> 5, l31 = lane & 31;
    const int ldx32 = (int)ldx;  // host guarantees n * ldx < 2^31: row offsets in 32-bit scalar arithmetic

    // XCD-aware persistent schedule: XCD x owns tiles [x*per, (x+1)*per); its workgroups walk them together
    const int nwg = gridDim.x, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, wg_per_xcd = (nwg + 7 - xcd) >> 3;
    const int64_t per = (ntiles + 7) / 8;
    const int64_t t_lo = xcd * per, t_hi = min(ntiles, t_lo + per);
    int64_t my_n = 0;  // tiles of this workgroup (same for all 8 waves -> barrier counts match)
    if (t_lo + slot < t_hi) my_n = (t_hi - t_lo - slot + wg_per_xcd - 1) / wg_per_xcd;

    // ---- matrix-phase role: (column slice cs, K half kh, row group rg)
    const int cs = w % C::NSLICE, kh = (w / C::NSLICE) & 1, rg = w / (2 * C::NSLICE);
    const int col = cs * 32 + l31;
    const int partner = w ^ C::NSLICE;  // same cs and rg, other K half
    constexpr int NWR = MODE == 0 ? C::S_STEPS * 4 : 1;
    constexpr int NWB = MODE == 1 ? CIN_PAD / 16 : 1;   // k-steps of 16 per K half
    float wr[NWR];
    bf16x8 wb[NWB][3];                                   // [k-step][hi, mid, lo] 8 bf16 each
    {
        const float* Wsrc = kh ? Wi : Wj;
        if (MODE == 0) {
#pragma unroll
            for (int S = 0; S < C::S_STEPS; ++S) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int k = 8 * S + 4 * h + j;  // k-permutation shared with the A-tile read
                    const float v = Wsrc[(int64_t)col * c_in + (k < c_in ? k : 0)];
                    wr[S * 4 + j] = k < c_in ? v : 0.f;
                }
            }
        } else {
#pragma unroll
            for (int S = 0; S < NWB; ++S) {
                uint32_t ph[4], pm[4], pl[4];
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const int k = 16 * S + 8 * h + 2 * d;  // lane half h holds k = 16S+8h .. +7, as the A-tile read does
                    const float v0 = Wsrc[(int64_t)col * c_in + (k < c_in ? k : 0)];
                    const float v1 = Wsrc[(int64_t)col * c_in + (k + 1 < c_in ? k + 1 : 0)];
                    split3(k < c_in ? v0 : 0.f, k + 1 < c_in ? v1 : 0.f, ph[d], pm[d], pl[d]);
                }
                wb[S][0] = __builtin_bit_cast(bf16x8, f32x4{__builtin_bit_cast(float, ph[0]), __builtin_bit_cast(float, ph[1]),
                                                            __builtin_bit_cast(float, ph[2]), __builtin_bit_cast(float, ph[3])});
                wb[S][1] = __builtin_bit_cast(bf16x8, f32x4{__builtin_bit_cast(float, pm[0]), __builtin_bit_cast(float, pm[1]),
                                                            __builtin_bit_cast(float, pm[2]), __builtin_bit_cast(float, pm[3])});
                wb[S][2] = __builtin_bit_cast(bf16x8, f32x4{__builtin_bit_cast(float, pl[0]), __builtin_bit_cast(float, pl[1]),
                                                            __builtin_bit_cast(float, pl[2]), __builtin_bit_cast(float, pl[3])});
            }
        }
    }
    const float bb = bj ? bj[col] : 0.f;
    const float sc = scale ? scale[col] : 1.f;
    const float sh = scale ? shift[col] : 0.f;
    const bool has_scale = scale != nullptr;

    // ---- gather-phase role: this wave owns tets [w*TPW, (w+1)*TPW) of every tile
    const int c0 = lane * CPL;
    const bool on = c0 < c_in;    // c_in is a multiple of CPL (host-checked)
    const int c0l = on ? c0 : 0;  // clamped channel offset: loads stay in bounds for inactive lanes
    constexpr int NWL = FM ? 1 : CPL;
    float wl[NWL][FE], bl[NWL];
    if constexpr (!FM) {
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
            const float bv = be[c0l + j];
            bl[j] = on ? bv : 0.f;
#pragma unroll
            for (int f = 0; f < FE; ++f) {
                const float wv = We[(int64_t)(c0l + j) * FE + f];
                wl[j][f] = on ? wv : 0.f;
            }
        }
    }
    float* const myea = eabuf + w * C::EA_PAD;
    const bool vec2 = c_in % 2 == 0 && ldx % 2 == 0 && (((uintptr_t)x | (uintptr_t)xdst) & 7) == 0;      // FM, VW == 2: a lane's pair of channels as one 8-byte load
    // FM: lane (n, g): B operand We[chan(nb, n)][4 ks + g] of k-step ks, C input be[chan(nb, n)]
    const int fn = lane & 15, fg = lane >> 4;
    if constexpr (FM) {
        // (in LDS, read where used: 40 + 8 registers per lane at 128 channels spilled beside the dense product's 64 resident weights)
        for (int e = threadIdx.x; e < NBK * 6 * 64; e += blockDim.x) {
            const int ln = e & 63, nk = e >> 6, nb = nk / 6, ks = nk - 6 * nb;
            const int c = 64 * (nb / VW) + VW * (ln & 15) + nb % VW;
            float v = 0.f;
            if (c < c_in) v = ks < 5 ? We[(int64_t)c * FE + 4 * ks + (ln >> 4)] : be[c];
            fwbuf[e] = v;
        }
        __syncthreads();
    }

    // state of the tile in flight
    float xd[FM ? 1 : TPW][CPL], xr[FM ? 1 : NQ][CPL];
    float fxd[FM ? EB * NBK : 1], fxr[FM ? EB * 4 * NBK : 1];      // FM: [eb * NBK + nb], [(eb * 4 + r) * NBK + nb]: lane (n, g)'s channels of tet 4 eb + g
    bool regular = false;        // loads for the current tile are in flight (4-regular fast path)
    // Index prefetch runs two tiles deep so that no index round trip sits in front of the row loads.  The
    // indices live in VGPRs, one entry per lane (a single coalesced load each), and are turned into scalar row
    // offsets with v_readlane when the row loads are issued -- no long-lived SGPR arrays, no SGPR spills.
    //   vbeg2  lane r <= TPW : rowptr[i0(it+2) + r]      (requested during P(it))
    //   vbeg1  same for tile it+1 (arrived during P(it-1));  vsrc1  lane q < NQ : src[beg1_0 + q]
    int vbeg1 = 0, vbeg2 = 0, vsrc1 = 0, veid1 = 0;
    bool ok1 = false, ok2 = false;  // tile exists, is complete and (for ok1) 4-regular

    auto tile_of = [&](int64_t it) { return t_lo + slot + it * wg_per_xcd; };
    auto load_rowptr = [&](int64_t it, int& vb) -> bool {
        if (it >= my_n) return false;
        const int64_t i0 = tile_of(it) * TILE + w * TPW;
        if (i0 + TPW > n_dst) return false;
        vb = rowptr[i0 + (lane < TPW ? lane : TPW)];
        return true;
    };
    auto load_src = [&]() {
        if (ok1) {
            const int b0 = __builtin_amdgcn_readfirstlane(vbeg1);
            ok1 = __all(lane > TPW || vbeg1 == b0 + 4 * lane) != 0;  // every in-degree of this wave's tets is 4
            if (ok1) {
                vsrc1 = src[b0 + (lane < NQ ? lane : NQ - 1)];
                if (eid) veid1 = eid[b0 + (lane < NQ ? lane : NQ - 1)];
            }
        }
    };
    // issue every load of this wave's tets of tile `it` (uses vbeg1/vsrc1)
    auto issue_loads = [&](int64_t it) {
        regular = ok1;
        if (regular) {
            const int i0 = (int)(tile_of(it) * TILE) + w * TPW;
            const float* eab = ea + (int64_t)__builtin_amdgcn_readfirstlane(vbeg1) * lde;
            if constexpr (FM) {
                // lane (n, g): VW contiguous channels per segment of tet 4 eb + g's own row and of its 4 neighbour rows (a segment past c_in: clamped, zeroed at use)
                auto ld_seg = [&](float* dst, const float* rowp) {
#pragma unroll
                    for (int sg = 0; sg < NSEG; ++sg) {
                        const int cseg = 64 * sg + VW * fn;
                        const float* pp = rowp + (cseg < c_in ? cseg : 0);
                        if constexpr (VW == 4) {
                            const f32x4 t = *reinterpret_cast<const f32x4*>(pp);
                            dst[4 * sg] = t[0], dst[4 * sg + 1] = t[1], dst[4 * sg + 2] = t[2], dst[4 * sg + 3] = t[3];
                        } else if (vec2) {
                            const float2 t = *reinterpret_cast<const float2*>(pp);
                            dst[2 * sg] = t.x, dst[2 * sg + 1] = t.y;
                        } else {      // rows that are not 8-byte aligned (the scene's feature rows behind a column slice) or an odd c_in: two dwords
                            dst[2 * sg] = pp[0];
                            dst[2 * sg + 1] = rowp[cseg + 1 < c_in ? cseg + 1 : 0];
                        }
                    }
                };
#pragma unroll
                for (int eb = 0; eb < EB; ++eb) {
                    ld_seg(fxd + eb * NBK, xdst + (uint32_t)((i0 + 4 * eb + fg) * ldx32));
#pragma unroll
                    for (int r = 0; r < 4; ++r) ld_seg(fxr + (eb * 4 + r) * NBK, x + (uint32_t)(__shfl(vsrc1, 16 * eb + 4 * fg + r) * ldx32));
                }
            } else {
#pragma unroll
                for (int r = 0; r < TPW; ++r) ld_row<CPL>(xd[r], xdst + (uint32_t)((i0 + r) * ldx32) + c0l);
#pragma unroll
                for (int q = 0; q < NQ; ++q)
                    ld_row<CPL>(xr[q], x + (uint32_t)(__builtin_amdgcn_readlane(vsrc1, q) * ldx32) + c0l);
            }
            // edge-attribute block of this wave's tets: async DMA straight into its private LDS strip (issued last:
            // hipcc answers any later wait on an ordinary load with vmcnt(0) while an LDS-DMA is in flight)
            // eid == nullptr: rows already in plan order (one contiguous block); otherwise row k comes from edge_attr[eid[k]]
#pragma unroll
            for (int q = 0; q < NEV; ++q) {
                const int idx = q * 64 + lane, ch = idx < NV4 ? idx : 0;  // 16-byte chunk of the strip: edge ch/5, part ch%5
                if (eid) {
                    const int e = (ch * 0x3334) >> 16;
                    glds16(ea + (int64_t)__shfl(veid1, e) * FE + (ch - 5 * e) * 4, myea + q * 256);
                } else {
                    glds16(eab + 4 * ch, myea + q * 256);
                }
            }
        }
    };
    // shift the index pipeline by one tile
    auto advance_idx = [&](int64_t it_next) {
        ok1 = ok2;
        vbeg1 = vbeg2;
        load_src();
        ok2 = load_rowptr(it_next + 1, vbeg2);
    };

    ok1 = load_rowptr(0, vbeg1);
    load_src();
    ok2 = load_rowptr(1, vbeg2);
    // write one finished tet into the A-tile: columns [0,CIN_PAD) = mean row, [CIN_PAD,2*CIN_PAD) = own row
    auto put_row = [&](int buf, int row, const float (&av)[CPL], const float (&xv)[CPL]) {
        if (MODE == 0) {
            float* dst = abuf + buf * C::A_FLOATS + row * LDA;
            if (CPL == 2) {
                if (lane < CIN_PAD / 2) {
                    *reinterpret_cast<float2*>(dst + c0) = make_float2(av[0], av[CPL - 1]);
                    *reinterpret_cast<float2*>(dst + CIN_PAD + c0) = make_float2(xv[0], xv[CPL - 1]);
                }
            } else if (lane < CIN_PAD) {
                dst[c0] = av[0];
                dst[CIN_PAD + c0] = xv[0];
            }
        } else {
            char* dst = reinterpret_cast<char*>(abuf + buf * C::A_FLOATS) + row * ROWB;
            if (CPL == 2) {
                // channels (c0, c0+1) = one packed bf16 pair per part: octet c0/8, dword (c0%8)/2
                if (lane < CIN_PAD / 2) {
                    uint32_t hh, mm, ll;
                    char* pa = dst + (c0 >> 3) * 48 + (c0 & 7) * 2;
                    split3(av[0], av[CPL - 1], hh, mm, ll);
                    *reinterpret_cast<uint32_t*>(pa) = hh;
                    *reinterpret_cast<uint32_t*>(pa + 16) = mm;
                    *reinterpret_cast<uint32_t*>(pa + 32) = ll;
                    char* px = pa + (CIN_PAD / 8) * 48;
                    split3(xv[0], xv[CPL - 1], hh, mm, ll);
                    *reinterpret_cast<uint32_t*>(px) = hh;
                    *reinterpret_cast<uint32_t*>(px + 16) = mm;
                    *reinterpret_cast<uint32_t*>(px + 32) = ll;
                }
            } else if (lane < CIN_PAD) {
                uint32_t hh, mm, ll;
                char* pa = dst + (c0 >> 3) * 48 + (c0 & 7) * 2;
                split3(av[0], xv[0], hh, mm, ll);  // low halves: mean value, high halves: own-row value
                *reinterpret_cast<uint16_t*>(pa) = (uint16_t)hh;
                *reinterpret_cast<uint16_t*>(pa + 16) = (uint16_t)mm;
                *reinterpret_cast<uint16_t*>(pa + 32) = (uint16_t)ll;
                char* px = pa + (CIN_PAD / 8) * 48;
                *reinterpret_cast<uint16_t*>(px) = (uint16_t)(hh >> 16);
                *reinterpret_cast<uint16_t*>(px + 16) = (uint16_t)(mm >> 16);
                *reinterpret_cast<uint16_t*>(px + 32) = (uint16_t)(ll >> 16);
            }
        }
    };

    issue_loads(0);
    float mine[8];  // own half of the accumulators of the previous tile, finished after the next barrier
#pragma unroll
    for (int r = 0; r < 8; ++r) mine[r] = 0.f;

    for (int64_t it = 0; it <= my_n; ++it) {
        if (it < my_n) {
            // ================================================================ P: gather / filter / mean
            const int64_t i0 = tile_of(it) * TILE + w * TPW;
            stamp(trace, trace_cap, it, w, 0);
            const bool was_regular = regular;
            // rows and the LDS-DMA'd attribute strip of this tile: the wave's own counted wait orders the reads below
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            advance_idx(it + 1);  // index scalar loads fly under the VALU work below
            stamp(trace, trace_cap, it, w, 1);
            if constexpr (FM) {
              // FULL: c_in == CIN_PAD (wave-uniform): no channel of a lane's piece is padding, no masks
              auto fm_tile = [&](auto full_c) {
                constexpr bool FULL = decltype(full_c)::value;
#pragma unroll
                for (int eb = 0; eb < EB; ++eb) {
                    // A operand: lane (m = edge 16 eb + n of the wave, g) holds attribute 4 ks + g of its edge
                    float fa[5];
#pragma unroll
                    for (int ks = 0; ks < 5; ++ks) fa[ks] = myea[(16 * eb + fn) * FE + 4 * ks + fg];
                    float* dst = abuf + (int)(it & 1) * C::A_FLOATS + (w * TPW + 4 * eb + fg) * LDA;
#pragma unroll
                    for (int sg = 0; sg < NSEG; ++sg) {
                        // the VW channel blocks of a segment advance together: independent chains of 5 dependent matrix instructions each
                        f32x4 d[VW];
                        float fwv[VW][5];
#pragma unroll
                        for (int u = 0; u < VW; ++u) {
                            const float* fwp = fwbuf + (sg * VW + u) * 6 * 64 + lane;
                            const float fb_ = fwp[5 * 64];
                            d[u] = f32x4{fb_, fb_, fb_, fb_};
#pragma unroll
                            for (int ks = 0; ks < 5; ++ks) fwv[u][ks] = fwp[ks * 64];
                        }
#pragma unroll
                        for (int ks = 0; ks < 5; ++ks)
#pragma unroll
                            for (int u = 0; u < VW; ++u) d[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[ks], fwv[u][ks], d[u], 0, 0, 0);
                        float av_[VW], xv_[VW];
#pragma unroll
                        for (int u = 0; u < VW; ++u) {
                            const int nb = sg * VW + u;
                            const bool con = FULL || 64 * sg + VW * fn + u < c_in;
                            float a_ = 0.f;
#pragma unroll
                            for (int r = 0; r < 4; ++r) a_ = __fadd_rn(a_, __fmul_rn(con ? fxr[(eb * 4 + r) * NBK + nb] : 0.f, d[u][r]));
                            av_[u] = a_ * 0.25f;
                            xv_[u] = con ? fxd[eb * NBK + nb] : 0.f;
                        }
                        if constexpr (VW == 4) {
                            *reinterpret_cast<f32x4*>(dst + 64 * sg + 4 * fn) = f32x4{av_[0], av_[1], av_[2], av_[3]};
                            *reinterpret_cast<f32x4*>(dst + CIN_PAD + 64 * sg + 4 * fn) = f32x4{xv_[0], xv_[1], xv_[2], xv_[3]};
                        } else {
                            *reinterpret_cast<float2*>(dst + 2 * fn) = make_float2(av_[0], av_[VW - 1]);
                            *reinterpret_cast<float2*>(dst + CIN_PAD + 2 * fn) = make_float2(xv_[0], xv_[VW - 1]);
                        }
                    }
                }
              };
              if (was_regular) {
                  if (c_in == CIN_PAD) fm_tile(std::true_type{});
                  else fm_tile(std::false_type{});
              }
            }
            if (!FM && was_regular) {
#pragma unroll
                for (int r = 0; r < TPW; ++r) {
                    float acc[CPL];
                    // ILP form: the 4 edges of a tet advance together -> 4*CPL independent FMA chains (a single chain
                    // issues at half rate because of the 4-cycle dependent latency).  The register-tight configuration
                    // (bf16x3 at C=128) keeps the edge-serial form.  Either way each chain runs f-ascending and the
                    // 4 products are summed in edge order, so the two forms are bit-identical.
                    constexpr bool ILP = !(MODE == 1 && CIN_PAD == 128);
                    if (ILP) {
                        float p[4][CPL];
#pragma unroll
                        for (int e = 0; e < 4; ++e)
#pragma unroll
                            for (int j = 0; j < CPL; ++j) p[e][j] = bl[j];
#pragma unroll
                        for (int f = 0; f < FE; f += 4) {
                            f32x4 Aq[4];
#pragma unroll
                            for (int e = 0; e < 4; ++e) Aq[e] = *reinterpret_cast<const f32x4*>(myea + (r * 4 + e) * FE + f);
#pragma unroll
                            for (int t = 0; t < 4; ++t)
#pragma unroll
                                for (int e = 0; e < 4; ++e)
#pragma unroll
                                    for (int j = 0; j < CPL; ++j) p[e][j] = __fmaf_rn(wl[j][f + t], Aq[e][t], p[e][j]);
                        }
#pragma unroll
                        for (int j = 0; j < CPL; ++j) {
                            acc[j] = 0.f;
#pragma unroll
                            for (int e = 0; e < 4; ++e) acc[j] = __fadd_rn(acc[j], __fmul_rn(on ? xr[r * 4 + e][j] : 0.f, p[e][j]));
                        }
                    } else {
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
                                acc[j] = __fadd_rn(acc[j], __fmul_rn(on ? xr[r * 4 + e][j] : 0.f, p));
                            }
                        }
                    }
                    float xv[CPL];
#pragma unroll
                    for (int j = 0; j < CPL; ++j) {
                        acc[j] *= 0.25f;
                        xv[j] = on ? xd[r][j] : 0.f;
                    }
                    put_row((int)(it & 1), w * TPW + r, acc, xv);
                }
            }
            if (!was_regular) {
                // generic path: any in-degree, tile tail; one edge at a time (rare).  FM: the filter's weights come from memory here (they are not in this
                // wave's registers in channel-per-lane order) -- the same chain, be then attributes 0..19
                for (int r = 0; r < TPW; ++r) {
                    const int64_t i = i0 + r;
                    float acc[CPL], xdv[CPL];
#pragma unroll
                    for (int j = 0; j < CPL; ++j) acc[j] = xdv[j] = 0.f;
                    if (i < n_dst) {
                        const int b = rowptr[i], e_end = rowptr[i + 1];
                        ld_row<CPL>(xdv, xdst + i * ldx + c0l);
                        for (int k = b; k < e_end; ++k) {
                            const int s = src[k];
                            const float* ar = ea + (int64_t)(eid ? eid[k] : k) * lde;
                            float xv[CPL];
                            ld_row<CPL>(xv, x + (int64_t)s * ldx + c0l);
#pragma unroll
                            for (int j = 0; j < CPL; ++j) {
                                float p;
                                if constexpr (FM) {
                                    p = on ? be[c0l + j] : 0.f;
                                    for (int f = 0; f < FE; ++f) p = __fmaf_rn(on ? We[(int64_t)(c0l + j) * FE + f] : 0.f, ar[f], p);
                                } else {
                                    p = bl[j];
#pragma unroll
                                    for (int f = 0; f < FE; ++f) p = __fmaf_rn(wl[j][f], ar[f], p);
                                }
                                acc[j] = __fadd_rn(acc[j], __fmul_rn(on ? xv[j] : 0.f, p));
                            }
                        }
                        const float cnt = (float)max(e_end - b, 1);
#pragma unroll
                        for (int j = 0; j < CPL; ++j) acc[j] = __fdiv_rn(acc[j], cnt);
                    }
#pragma unroll
                    for (int j = 0; j < CPL; ++j) xdv[j] = on ? xdv[j] : 0.f;
                    put_row((int)(it & 1), w * TPW + r, acc, xdv);
                }
            }
            stamp(trace, trace_cap, it, w, 2);
            // every read of the attribute strip must have returned before the next tile's LDS-DMA is issued (the DMA
            // is not ordered against this wave's LDS queue and the compiler models it as touching 16 bytes only)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            // the index loads issued at the start of this phase are consumed HERE (they landed long ago); otherwise the
            // compiler waits for them later with vmcnt(0), draining the row loads that must fly across the barrier
            asm volatile("" : "+v"(vbeg2), "+v"(vsrc1), "+v"(vbeg1), "+v"(veid1));
            issue_loads(it + 1);  // in flight during the barrier wait and the whole matrix phase
            stamp(trace, trace_cap, it, w, 3);
        }
        tile_barrier();  // A-tile `it` complete; partial sums of tile `it-1` complete
        stamp(trace, trace_cap, it, w, 4);

        if (it > 0) {
            // ============================================================ delayed epilogue of tile it-1
            const int64_t tile = tile_of(it - 1);
            const float* red = redbuf + ((it - 1) & 1) * C::RED_FLOATS + partner * 512 + lane;
            const int64_t row0 = tile * TILE + rg * 32 + 4 * h + 16 * kh;  // kh=0: acc regs 0..7, kh=1: regs 8..15
            float* o = out + row0 * ldo + col;
            float v[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                v[r] = (mine[r] + red[r * 64]) + bb;
                if (has_scale) v[r] = __fmaf_rn(v[r], sc, sh);
                if (relu) v[r] = fmaxf(v[r], 0.f);
            }
            if ((tile + 1) * TILE <= n_dst) {  // full tile (wave-uniform): unconditional row stores
#pragma unroll
                for (int r = 0; r < 8; ++r) o[(int64_t)((r & 3) + 8 * (r >> 2)) * ldo] = v[r];
            } else {
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const int rr = (r & 3) + 8 * (r >> 2);
                    if (row0 + rr < n_dst) o[(int64_t)rr * ldo] = v[r];
                }
            }
        }
        if (it < my_n) {
            // ================================================================ C: matrix phase
            f32x16 acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
            if (MODE == 0) {
                const float* A = abuf + (it & 1) * C::A_FLOATS + (rg * 32 + l31) * LDA + kh * CIN_PAD + 4 * h;
#pragma unroll
                for (int S = 0; S < C::S_STEPS; ++S) {
                    const f32x4 av = *reinterpret_cast<const f32x4*>(A + 8 * S);
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], wr[S * 4 + j], acc, 0, 0, 0);
                }
            } else {
                const char* A = reinterpret_cast<const char*>(abuf + (it & 1) * C::A_FLOATS) + (rg * 32 + l31) * ROWB +
                                (kh * (CIN_PAD / 8) + h) * 48;
#pragma unroll
                for (int S = 0; S < NWB; ++S) {
                    const bf16x8 ah = *reinterpret_cast<const bf16x8*>(A + S * 96);
                    const bf16x8 am = *reinterpret_cast<const bf16x8*>(A + S * 96 + 16);
                    const bf16x8 al = *reinterpret_cast<const bf16x8*>(A + S * 96 + 32);
                    // smallest partial products first
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, wb[S][0], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, wb[S][2], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, wb[S][1], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, wb[S][0], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, wb[S][1], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, wb[S][0], acc, 0, 0, 0);
                }
            }
            // keep the half this wave finishes, hand the other half to the partner
            float* red = redbuf + (it & 1) * C::RED_FLOATS + w * 512 + lane;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                mine[r] = kh ? acc[8 + r] : acc[r];
                red[r * 64] = kh ? acc[r] : acc[8 + r];
            }
            stamp(trace, trace_cap, it, w, 5);
        }
    }
}

template <int CIN_PAD, int COUT, int MODE, int FM = 0>
int launch_fused(const int32_t* rowptr, const int32_t* src, const int32_t* eid, int64_t n_dst, const float* x, const float* xdst,
                 int64_t ldx, int c_in,
                 const float* ea, int64_t lde, const float* We, const float* be, const float* Wj, const float* bj,
                 const float* Wi, const float* scale, const float* shift, int relu, float* out, int64_t ldo,
                 hipStream_t stream) {
    using C = FusedCfg<CIN_PAD, COUT, MODE>;
    const int64_t ntiles = dgnn_cdiv(n_dst, C::TILE);
    const size_t smem = sizeof(float) * (C::SMEM_FLOATS + (FM ? C::FW_FLOATS : 0));
    static bool attr_set[DGNN_MAX_DEVICES] = {};
    dgnn_allow_dynamic_lds(reinterpret_cast<const void*>(&k_sage_fused<CIN_PAD, COUT, MODE, FM>), smem, attr_set);
    int grid = (int)(ntiles < DGNN_NUM_CU ? ntiles : DGNN_NUM_CU);
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL((k_sage_fused<CIN_PAD, COUT, MODE, FM>), dim3(grid), dim3(512), smem, stream, rowptr, src, eid, n_dst, x, xdst, ldx, c_in,
                       ea, lde, We, be, Wj, bj, Wi, scale, shift, relu, out, ldo, ntiles, g_dgnn_trace_buf, g_dgnn_trace_cap);
    return dgnn_check_launch("sage_layer_fused_fwd");
}

}  // namespace

int dgnn_sage_layer_fused_mfma_try(const int32_t* rowptr, const int32_t* src, const int32_t* eid, int64_t n_dst, const float* x_src,
                                   const float* x_dst, int64_t ldx,
                                   int c_in, const float* edge_attr, int64_t lde, const float* We, const float* be,
                                   const float* Wj, const float* bj, const float* Wi, const float* scale, const float* shift,
                                   int relu, int c_out, float* out, int64_t ldo, int f16_parts, hipStream_t stream, void* prep = nullptr,
                                   int prep_mode = 0);  // fused_mfma.hip (prep: prepared parameters, see dgnn_sage_layer_prepare)

extern "C" int dgnn_sage_layer_fused_fwd(const int32_t* rowptr, const int32_t* src, const int32_t* eid, int64_t n_dst, const float* x_src,
                                         const float* x_dst, int64_t ldx, int c_in, const float* edge_attr, int64_t lde, int f_e,
                                         const float* We, const float* be, const float* Wj, const float* bj, const float* Wi,
                                         const float* scale, const float* shift, int relu, int c_out, float* out, int64_t ldo,
                                         int gemm_mode, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DGNN_REQUIRE(n_dst >= 0 && c_in > 0 && c_out > 0, DGNN_E_INVALID, "sage_layer_fused_fwd: bad sizes");
    DGNN_REQUIRE(gemm_mode >= DGNN_GEMM_F32 && gemm_mode <= DGNN_GEMM_F16X2, DGNN_E_INVALID,
                 "sage_layer_fused_fwd: bad gemm_mode %d", gemm_mode);
    if (n_dst == 0) return DGNN_OK;
    DGNN_REQUIRE(rowptr && src && x_src && edge_attr && We && be && Wj && Wi && out, DGNN_E_INVALID,
                 "sage_layer_fused_fwd: null pointer");
    DGNN_REQUIRE((scale == nullptr) == (shift == nullptr), DGNN_E_INVALID, "sage_layer_fused_fwd: scale/shift must come together");
    if (x_dst == nullptr) x_dst = x_src;  // the reference's (x, x[:n_dst]) pair
    DGNN_REQUIRE(f_e == FE && lde == FE, DGNN_E_UNSUPPORTED, "sage_layer_fused_fwd: needs f_e == 20 and packed edge rows (lde == 20)");
    DGNN_REQUIRE(((uintptr_t)edge_attr % 16) == 0, DGNN_E_UNSUPPORTED, "sage_layer_fused_fwd: edge_attr must be 16-byte aligned");
    DGNN_REQUIRE(n_dst * ldx < ((int64_t)1 << 31), DGNN_E_UNSUPPORTED, "sage_layer_fused_fwd: activations beyond 2^31 elements");
    DGNN_REQUIRE(c_in <= 128 && (c_out == 64 || c_out == 128), DGNN_E_UNSUPPORTED,
                 "sage_layer_fused_fwd: supports c_in <= 128 and c_out in {64,128} (got %d -> %d)", c_in, c_out);
    if (gemm_mode >= DGNN_GEMM_BF16X3_FILTER) {
        const int rc = dgnn_sage_layer_fused_mfma_try(rowptr, src, eid, n_dst, x_src, x_dst, ldx, c_in, edge_attr, lde, We, be, Wj, bj, Wi,
                                                      scale, shift, relu, c_out, out, ldo,
                                                      gemm_mode == DGNN_GEMM_F16X2 ? 2 : (gemm_mode == DGNN_GEMM_F16X2_DENSE ? 1 : 0), stream);
        if (rc != DGNN_E_UNSUPPORTED) return rc;
        gemm_mode = DGNN_GEMM_BF16X3;  // shape not covered by the all-MFMA variant
    }
    // the filter product on the fp32 matrix cores (FM, see the kernel): rows in pieces of VW = min(4, CP / 16) channels per lane.  DGNN_FILTER_MFMA=0: the VALU form
    static const bool fm_on = !(getenv("DGNN_FILTER_MFMA") && getenv("DGNN_FILTER_MFMA")[0] == '0');
    auto fm_ok = [&](int cp) {
        const int vw = cp / 16 < 4 ? cp / 16 : 4;      // (pairs: the kernel falls back to dword loads by itself; quads: only aligned rows)
        return fm_on && gemm_mode == DGNN_GEMM_F32 && (vw < 4 || (c_in % 4 == 0 && ldx % 4 == 0 && (((uintptr_t)x_src | (uintptr_t)x_dst) % 16) == 0));
    };
#define GO(CP, CO)                                                                                                              \
    do {                                                                                                                        \
        if (fm_ok(CP))                                                                                                          \
            return launch_fused<CP, CO, 0, 1>(rowptr, src, eid, n_dst, x_src, x_dst, ldx, c_in, edge_attr, lde, We, be, Wj, bj, Wi, scale, \
                                              shift, relu, out, ldo, stream);                                                   \
        if (gemm_mode == DGNN_GEMM_F32)                                                                                         \
            return launch_fused<CP, CO, 0>(rowptr, src, eid, n_dst, x_src, x_dst, ldx, c_in, edge_attr, lde, We, be, Wj, bj, Wi, scale, \
                                           shift, relu, out, ldo, stream);                                                      \
        return launch_fused<CP, CO, 1>(rowptr, src, eid, n_dst, x_src, x_dst, ldx, c_in, edge_attr, lde, We, be, Wj, bj, Wi, scale,     \
                                       shift, relu, out, ldo, stream);                                                          \
    } while (0)
    if (c_in <= 32) { if (c_out == 64) GO(32, 64); else GO(32, 128); }
    if (c_in <= 64) { if (c_out == 64) GO(64, 64); else GO(64, 128); }
    DGNN_REQUIRE(c_in % 2 == 0 && ldx % 2 == 0 && (((uintptr_t)x_src | (uintptr_t)x_dst) % 8) == 0, DGNN_E_UNSUPPORTED,
                 "sage_layer_fused_fwd: c_in > 64 needs even c_in / ldx and 8-byte aligned x");
    DGNN_REQUIRE(c_out == 128, DGNN_E_UNSUPPORTED, "sage_layer_fused_fwd: c_in > 64 supports c_out == 128 only");
    GO(128, 128);
#undef GO
}
