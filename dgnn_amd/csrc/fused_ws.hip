// Wave-specialised fused conv layer, C_in in {64, 128} -> 128 (round 5; VERDICT r4 item 3, DESIGN 10 "next" of round 4).
//
// Reference: SAGEConv.forward + BatchNorm(eval) + ReLU, learning/surfaceNetStaticEdgeFilters.py:66-96, :345-346 -- the shipped model's layers 1, 2 and 3
// (the description below is for C_in = 128; 64: half the channels per producer lane, 512-byte ring rows, 4 k-steps).
// k_sage_fused_mfma<128,128> runs eight wavefronts that ALL walk the same two phases (filter / mean, barrier, dense product): 44-48 % of its wave cycles
// are parked, the matrix pipe is 26 % busy, and at 229-250 VGPRs two wavefronts per SIMD is all it admits (docs/history_r1-r4.md 5a, 5b).  The probe
// tools/probe_producer.py says what the split buys: the producer side ALONE (gathers, filter product, mean, row split -- k_agg_sr<8, false> with its HBM
// stores off) runs 1M cells in 0.31-0.34 ms at 4 resp. 2 wavefronts per SIMD: it is bound by its reads, not by latency, once nothing else shares its
// wavefronts.  So here a 1024-thread workgroup (16 wavefronts, 4 per SIMD, <= 128 VGPRs each) is
//   8 PRODUCERS  one group of 4 cells each per 32-cell tile: index chain two tiles ahead, 4 neighbour rows + own row as 16-byte loads, filter product on
//                v_mfma_f32_16x16x32_f16 (per-edge power-of-two scales, 3 products), in-lane 4-term mean, the [a | x_i] row scaled by one power of two,
//                split (hi, lo) and parked in an LDS ring slot (1 KB per cell, XOR-swizzled 16-byte pieces, conflict-free for both sides);
//   8 CONSUMERS  16 output channels each, their [Wj | Wi] rows resident as fp16 (hi, lo) fragments (64 VGPRs): D[channel][cell] on v_mfma_f32_16x16x32_f16
//                with the weights as the A operand, 48 products per tile; epilogue per lane = one cell x 4 consecutive channels: inverse scales,
//                bias / BatchNorm / ReLU, one 16-byte non-temporal store;
//   The ring slots are handed over by LDS counters, no barrier in the loop: `ready[slot]` +1 per producer that has parked its rows of a tile, `done[slot]` +1
//   per consumer that has read them; every wavefront runs at its own pace, a producer only waits when the slot it is about to write (two tiles back) has
//   not been drained.  EVERY adder waits for the slot first, the producer groups past the end of the graph included -- a count that lands early can stand
//   in for a late wavefront's (that race, and the same one in the decoder's counters, are pinned by tests/test_gpu_infer.py::
//   test_ignatius_layers_repeat_bit_for_bit_with_cold_caches and tests/test_gpu_parity.py::test_last_layer_and_decoder_in_one_launch).  No K split, no
//   partial-sum exchange, no resident weights in the gathering wavefronts.
// Where the time goes (a build with s_memtime around the polls, profiles/r05_ws_phases.md): the producers bound the plain launch (they wait 8-12 % of their
// time, the consumers 35-45 %); a consumer's 48 products take ~2 us of a 3.3 us tile because the two consumers of a SIMD start together and share its
// matrix pipe with two producers.
// Arithmetic: the fp16 two-part form (fused_common.h): scaling groups per cell row [a | x_i], per edge, per [We | be], per consumer (its 16 weight rows).
// A cell's result depends on its own inputs only: destination sub-ranges, ring parts and differently tiled runs agree bit for bit.  Groups of any
// in-degree other than 4 take a per-lane fp32 path (never on Delaunay scenes).
#include <stdlib.h>

#include "common.h"
#include "fused_common.h"

namespace {
using namespace fused;

typedef float f32x4_t __attribute__((ext_vector_type(4)));
#define H8(v) __builtin_bit_cast(f16x8, v)

constexpr int WS_C = 128;                    // channels out (and in: CIN = 128, the shipped model's layers 2 and 3; CIN = 64: its layer 1)
constexpr int WS_TILE = 32;                  // cells per tile
constexpr int WS_NP = 8;                     // producer wavefronts (the other 8 of the 16 are consumers)
// per input width CIN: ring bytes per cell = [hi of a | hi of x | lo of a | lo of x], CIN fp16 each (1 KB / 512 B); a ring slot = 32 cells (32 / 16 KB);
// the filter operand [cb < CIN / 16][hi | lo][48] x 16 B
template <int CIN> struct WsC {
    static constexpr int NCH = CIN / 16;        // channels of a cell's rows per producer lane (8 / 4)
    static constexpr int NV = NCH / 4;          // ... as 16-byte loads
    static constexpr int KH = CIN / 32;         // k-steps of the dense product per half (a | x_i)
    static constexpr int PP = CIN / 8;          // 16-byte pieces per quarter of a ring row
    static constexpr int ROWB = 8 * CIN;
    static constexpr int SLOT = WS_TILE * ROWB;
    static constexpr int BP = NCH * 2 * 48 * 16;
};
// RING slots of the hand-off.  FLAGS = false (RING 2 only): one s_barrier per tile instead of the counters (producers on tile t, consumers on t - 1, everybody
// meets once per tile: ties all 16 wavefronts to the slowest gather of every tile; DGNN_WS_RING=2, kept as the simple form to test against).
// FLAGS = true: the counters; 2, 3 or 4 slots measure the same (DGNN_WS_RING=22 / 3 / 4), the decoder-carrying launch has LDS for 2.
template <int CIN, int RING> struct WsL {
    static constexpr int OFF_BP = RING * WsC<CIN>::SLOT;
    static constexpr int OFF_ROWF = OFF_BP + WsC<CIN>::BP;          // [RING][32] inverse row scales
    static constexpr int OFF_CST = OFF_ROWF + RING * WS_TILE * 4;   // bias | scale | shift [128]
    static constexpr int OFF_SC = OFF_CST + 3 * WS_C * 4;           // max |We|, |be|; then ready[RING], done[RING], ycnt[2], yfree[2], lcnt[2]
    static constexpr int OFF_DC = OFF_SC + 16 + 2 * RING * 4 + 16 + 16;   // decoder constants: A1 | B1 [64], W3 [2][64], b3 [2] (+2 pad)
    static constexpr int OFF_PLOG = OFF_DC + (64 + 64 + 128 + 4) * 4;     // partial logits [2 tiles][4 hidden blocks][32 cells][2]
    static constexpr int OFF_YS = OFF_PLOG + 2 * 4 * 32 * 8;              // inverse scales of the parked rows [2 tiles][8 slices][32 cells]
    static constexpr int OFF_W0 = (OFF_YS + 2 * 8 * 32 * 4 + 255) & ~255; // W0 as A-operand fragments [4 hidden blocks][8 slices][64 lanes] x 16 B (hi x 4 | lo x 4)
    static constexpr int OFF_YT = OFF_W0 + 4 * 8 * 64 * 16;               // the layer's finished tile as B-operand fragments [2 tiles][8 slices][2 blocks][64 lanes] x 16 B
    static constexpr int SMEM = OFF_SC + 16 + 2 * RING * 4 + 16 + 16;
    static constexpr int SMEM_DEC = OFF_YT + 2 * 8 * 2 * 64 * 16;
};

// the decoder behind the last conv layer (reference learning/surfaceNetStaticEdgeFilters.py:180-187, applied at :350-351): Linear(128 -> 64) - BatchNorm(eval) -
// ReLU - Linear(64 -> 2).  In the DEC instantiation the consumers do not store the layer's rows.  Three stages, each one tile behind the one before it, so
// that no wavefront ever waits for data another one has only just produced (the first form of this kernel exchanged fp32 partial hidden rows of the SAME
// tile -- three lock-step meetings of the eight consumers per tile, 1.1 us of a tile's 4.4):
//   A(t)    consumer s: its 16 finished channels of the 32 cells, one power-of-two scale per cell and slice, split (hi, lo) -- a lane's 4 channels of one
//           cell ARE a B-operand fragment of v_mfma_f32_16x16x16_f16 -- parked in LDS (16 bytes per lane and row block, linear);
//   B(t-1)  job (hb, b) of 8: hidden units 16 hb .. + 15 of cells 16 b .. + 15 = 8 slices x 3 products against W0's fragments (LDS, laid out once per
//           workgroup), every slice scaled back and added in slice order; BatchNorm / ReLU / W3 over the lane's 4 units, the 4 lanes of a cell added in
//           one order: partial logits per cell and hidden block.  A tile's 8 jobs go to the 8 wavefronts of ONE role: the consumers on three tiles of
//           four, the producers -- between issuing a tile's gathers and using them -- on the fourth (measured: producers always 0.54 ms, half 0.507,
//           three of four 0.493, consumers always 0.525 on one box; the producers are the busier role but have the gather latency to fill);
//   C(t-2)  consumer 0 adds the four partial logits of a cell in one order and stores 8 bytes per cell.
// Every sum has one order and touches one cell's values only (sub-ranges and ring parts give the same bits).
struct WsDec {
    const float* W0;      // [64, 128]
    const float* b0;      // [64]
    const float* scale1;  // [64] folded BatchNorm of the decoder, or NULL
    const float* shift1;
    const float* W3;      // [2, 64]
    const float* b3;      // [2]
    float* logits;        // [n_dst, 2]
};
typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));


__device__ __forceinline__ void st_nt16(float* p, f32x4_t v) { __builtin_nontemporal_store(v, reinterpret_cast<f32x4_t*>(p)); }
// IO16: the rows in HBM are the UNSIGNED 16-bit rows of the bf16-storage chain (fused_bf16.hip "UB": value = bits << 15 -- 8 exponent and 8 explicit
// mantissa bits of a value that is never negative, written behind a ReLU; round to nearest even on bit 15 of the fp32 pattern)
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float ub16_lo(uint32_t u) { return __builtin_fabsf(__builtin_bit_cast(float, u << 15)); }
__device__ __forceinline__ float ub16_hi(uint32_t u) { return __builtin_bit_cast(float, (u >> 16) << 15); }
__device__ __forceinline__ uint32_t ub16_enc(float v) {
    const uint32_t b = __builtin_bit_cast(uint32_t, v) & 0x7FFFFFFFu;
    return (b + 0x3FFFu + ((b >> 15) & 1u)) >> 15;
}
__device__ __forceinline__ uint32_t bits(float f) { return __builtin_bit_cast(uint32_t, f); }

#ifndef DGNN_WS_LINE_OWN
#define DGNN_WS_LINE_OWN 1      // 0: round 5's ownership (8 contiguous channels per lane) -- for A/B builds only
#endif
template <int CIN, int RING, bool FLAGS, bool DEC, bool IO16 = false>
__global__ void __launch_bounds__(1024) k_sage_fused_ws(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ src, const int32_t* __restrict__ eid,
                                                        int64_t n_dst, const float* __restrict__ x, const float* __restrict__ xdst, int64_t ldx,
                                                        const float* __restrict__ ea, int64_t lde, const float* __restrict__ We, const float* __restrict__ be,
                                                        const float* __restrict__ Wj, const float* __restrict__ bj, const float* __restrict__ Wi,
                                                        const float* __restrict__ scale, const float* __restrict__ shift, int relu, float* __restrict__ out,
                                                        int64_t ldo, int64_t ntiles, int knobs, WsDec dec) {
    static_assert(!DEC || FLAGS, "the decoder stage hands over by counters");
    static_assert(CIN == 128 || (CIN == 64 && !DEC), "input widths of the shipped model's 128-wide layers");
    static_assert(!IO16 || !DEC, "the decoder-carrying launch of the bf16-storage chain stays with fused_bf16.hip");
    // IO16: x / xdst / out are uint16_t rows (ldx / ldo in elements of that type)
    const uint16_t* const x16 = reinterpret_cast<const uint16_t*>(x);
    const uint16_t* const xdst16 = reinterpret_cast<const uint16_t*>(xdst);
    uint16_t* const out16 = reinterpret_cast<uint16_t*>(out);
    typedef WsC<CIN> G;
    constexpr int NCH = G::NCH, NV = G::NV, KH = G::KH, PP = G::PP;
    extern __shared__ __attribute__((aligned(16))) char ws_smem[];
    const bool nt_store = (knobs & 1) != 0;
    typedef WsL<CIN, RING> L;
    char* const ring = ws_smem;
    char* const bpbuf = ws_smem + L::OFF_BP;
    float* const rowf = reinterpret_cast<float*>(ws_smem + L::OFF_ROWF);
    float* const cst = reinterpret_cast<float*>(ws_smem + L::OFF_CST);
    uint32_t* const scbuf = reinterpret_cast<uint32_t*>(ws_smem + L::OFF_SC);
    volatile uint32_t* const ready = reinterpret_cast<volatile uint32_t*>(ws_smem + L::OFF_SC + 16);
    volatile uint32_t* const done = ready + RING;
    // DEC: one counter per buffer (tile parity), like ready / done per ring slot.  A single running counter is NOT enough: a fast consumer's count for the
    // next tile can stand in for a slow consumer's count of this one wherever nothing else holds the fast one back (the first tiles, the drain)
    volatile uint32_t* const ycnt = done + RING;          // [2] +1 per consumer whose 16 channels of a finished tile are parked (stage A)
    volatile uint32_t* const yfree = ycnt + 2;            // [2] +1 per consumer that has read a parked tile (stage B)
    volatile uint32_t* const lcnt = ycnt + 4;             // [2] +1 per consumer whose partial logits of a tile are parked (stage B)
    float* const dcst = reinterpret_cast<float*>(ws_smem + L::OFF_DC);
    float* const plog = reinterpret_cast<float*>(ws_smem + L::OFF_PLOG);
    float* const ysc = reinterpret_cast<float*>(ws_smem + L::OFF_YS);
    char* const w0buf = ws_smem + L::OFF_W0;
    char* const ytile = ws_smem + L::OFF_YT;
    // wait until *ctr >= target (LDS word, wave-uniform): a short sleep between polls keeps the LDS and the issue slots for the working wavefronts
    // The compiler barriers matter: the poll is a volatile load, which orders nothing against the ORDINARY LDS loads behind it -- without the second one
    // the compiler hoisted the consumer's read of a tile's row scales above the poll (a stale scale for one producer's four cells, once in thousands of
    // tiles: caught by tests/test_gpu_parity.py::test_fused_layers_row_level_on_larger_graph) -- nor the ordinary stores in front of it.
    typedef __attribute__((address_space(3))) volatile uint32_t lds_vu32;
    auto wait_for = [&](volatile uint32_t* ctr, uint32_t target) {
        asm volatile("" ::: "memory");
        // (the poll through an LDS-space pointer = ds_read_b32: the generic pointer compiled to a flat load, whose s_waitcnt vmcnt(0) also drained the
        // wavefront's gathers in flight)
        lds_vu32* c3 = (lds_vu32*)ctr;
        while ((int32_t)(__builtin_amdgcn_readfirstlane((int)*c3) - (int)target) < 0) __builtin_amdgcn_s_sleep(1);
        asm volatile("" ::: "memory");
    };
    const int lane = lane_id(), w = wave_id_uniform();
    const int jcol = lane & 15, tq = lane >> 4;

    // tiles of this workgroup: XCD b % 8 walks one contiguous eighth of the cells (gathered neighbour rows stay in that XCD's L2)
    const int nwg = gridDim.x, xcd = blockIdx.x & 7, slot_ = blockIdx.x >> 3, wg_per_xcd = (nwg + 7 - xcd) >> 3;
    const int64_t per = (ntiles + 7) / 8, t_lo = xcd * per, t_hi = min(ntiles, t_lo + per);
    int64_t my_n = 0;
    if (t_lo + slot_ < t_hi) my_n = (t_hi - t_lo - slot_ + wg_per_xcd - 1) / wg_per_xcd;
    auto tile_of = [&](int64_t it) { return t_lo + slot_ + it * wg_per_xcd; };

    // ---- prologue: power-of-two scale of [We | be], the split filter operand in LDS, the epilogue's per-channel constants
    if (threadIdx.x == 0) scbuf[0] = 0u;
    if (threadIdx.x < 2 * RING + 8) ready[threadIdx.x] = 0u;
    if constexpr (DEC) {
        for (int n_ = threadIdx.x; n_ < 64; n_ += blockDim.x) {
            const float s1 = dec.scale1 ? dec.scale1[n_] : 1.f, h1 = dec.scale1 ? dec.shift1[n_] : 0.f;
            dcst[n_] = s1;                                   // h = relu(sum * A1 + B1)
            dcst[64 + n_] = __fmaf_rn(dec.b0[n_], s1, h1);
            dcst[128 + n_] = dec.W3[n_];
            dcst[192 + n_] = dec.W3[64 + n_];
        }
        if (threadIdx.x < 2) dcst[256 + threadIdx.x] = dec.b3[threadIdx.x];
    }
    __syncthreads();
    {
        uint32_t me = 0u;
        for (int e = threadIdx.x; e < CIN * FE; e += blockDim.x) me = umax(me, absbits(We[e]));
        for (int e = threadIdx.x; e < CIN; e += blockDim.x) me = umax(me, absbits(be[e]));
        me = wave_umax(me);
        if (lane == 0) atomicMax(&scbuf[0], me);
    }
    for (int c = threadIdx.x; c < WS_C; c += blockDim.x) {
        cst[c] = bj ? bj[c] : 0.f;
        cst[WS_C + c] = scale ? scale[c] : 1.f;
        cst[2 * WS_C + c] = scale ? shift[c] : 0.f;
    }
    __syncthreads();
    float sWe, inv_sWe;
    pow2_scales(scbuf[0], sWe, inv_sWe);
    // Which channels lane j of a cell's 16 lanes owns.  fp32 rows of 128 channels (two 16-byte loads per lane and row): channels 4 j .. 4 j + 3 and
    // 64 + 4 j .. + 3, so that ONE load instruction covers whole 128-byte lines (16 lanes x 16 bytes = lines 0-1, then lines 2-3).  With 8 contiguous channels
    // per lane (round 5) both instructions touched all four lines of a row, half of each: twice the tag look-ups in the vector L1, the second one a hit on
    // a line still in flight.  Same bits (a channel's arithmetic does not depend on which lane owns it); measured on one box (profiles/r06_l1_probe.md):
    // 128 -> 128 0.408 -> 0.406 ms, last layer + decoder 0.500 -> 0.490 ms.
    constexpr bool LINE_OWN = !IO16 && NV == 2 && DGNN_WS_LINE_OWN;
    auto chan = [](int j, int cb) { return LINE_OWN ? ((cb >> 2) * 64 + 4 * j + (cb & 3)) : NCH * j + cb; };
    // entry (cb, part, g, j): channel c = chan(j, cb), k = 8 g .. 8 g + 7 (g < 3: attributes 0..19, the bias at k = 20, zeros)
    for (int e = threadIdx.x; e < NCH * 48; e += blockDim.x) {
        const int cb = e / 48, gj = e - cb * 48, g = gj >> 4, j = gj & 15;
        const int c = chan(j, cb);
        uint32_t ph[4], pl[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            float v[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int k = 8 * g + 2 * d + u;
                v[u] = k < FE ? We[(int64_t)c * FE + k] : (k == FE ? be[c] : 0.f);
            }
            split2h(v[0] * sWe, v[1] * sWe, ph[d], pl[d]);
        }
        uint4* dst = reinterpret_cast<uint4*>(bpbuf + ((cb * 2) * 48 + gj) * 16);
        dst[0] = make_uint4(ph[0], ph[1], ph[2], ph[3]);
        dst[48] = make_uint4(pl[0], pl[1], pl[2], pl[3]);
    }

    // DEC, stage B of tile t, job (hb, b) (either role runs it, see above): hidden units 16 hb .. + 15 of cells 16 b .. + 15 -- lane (cell n = jcol, tq) ends up with
    // units 16 hb + 4 tq .. + 3
    auto stage_b = [&](uint32_t t, int hb, int b) {
        wait_for(ycnt + (t & 1), 8u * (t / 2 + 1));
        int ln = lane;
        asm volatile("" : "+v"(ln));                            // (addresses recomputed per tile, see `key` below)
        const char* yt = ytile + ((t & 1) << 14) + ((b * 64 + ln) << 4);
        const char* wp = w0buf + ((hb * 512 + ln) << 4);
        const float* sp = ysc + (t & 1) * 8 * WS_TILE + 16 * b + (ln & 15);
        f32x4_t hs = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const uint4 wa = *reinterpret_cast<const uint4*>(wp + (s << 10)), ya = *reinterpret_cast<const uint4*>(yt + (s << 11));
            const float inv = sp[s * WS_TILE];
            const f16x4_t w0h = __builtin_bit_cast(f16x4_t, make_uint2(wa.x, wa.y)), w0l = __builtin_bit_cast(f16x4_t, make_uint2(wa.z, wa.w));
            const f16x4_t yh = __builtin_bit_cast(f16x4_t, make_uint2(ya.x, ya.y)), yl = __builtin_bit_cast(f16x4_t, make_uint2(ya.z, ya.w));
            f32x4_t d = {0.f, 0.f, 0.f, 0.f};
            d = __builtin_amdgcn_mfma_f32_16x16x16f16(w0l, yh, d, 0, 0, 0);
            d = __builtin_amdgcn_mfma_f32_16x16x16f16(w0h, yl, d, 0, 0, 0);
            d = __builtin_amdgcn_mfma_f32_16x16x16f16(w0h, yh, d, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) hs[i] += d[i] * inv;     // slices 0 .. 7 in order (the scale is a power of two: exact)
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) atomicAdd(const_cast<uint32_t*>(yfree + (t & 1)), 1u);
        const int u0 = 16 * hb + 4 * (ln >> 4);
        float l0 = 0.f, l1 = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float hv = fmaxf(__fmaf_rn(hs[i], dcst[u0 + i], dcst[64 + u0 + i]), 0.f);
            l0 = __fmaf_rn(hv, dcst[128 + u0 + i], l0);
            l1 = __fmaf_rn(hv, dcst[192 + u0 + i], l1);
        }
        // the cell's 4 lanes (16 apart): (q0 + q1) + (q2 + q3) in every one of them
        l0 += __shfl_xor(l0, 16);
        l1 += __shfl_xor(l1, 16);
        l0 += __shfl_xor(l0, 32);
        l1 += __shfl_xor(l1, 32);
        if (ln < 16) *reinterpret_cast<float2*>(plog + (((t & 1) * 4 + hb) * WS_TILE + 16 * b + ln) * 2) = make_float2(l0, l1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) atomicAdd(const_cast<uint32_t*>(lcnt + (t & 1)), 1u);
    };

    // who runs stage B of tile t: every wavefront of one role takes one of the tile's 8 jobs.  (knobs bits 4-5: 0 = producers always, 1 = consumers on even
    // tiles, 2 = consumers on three tiles of four, 3 = consumers always)
    const int bsel = (knobs >> 4) & 3;
    auto b_on_consumers = [&](uint32_t t) { return bsel == 0 ? false : bsel == 1 ? (t & 1) == 0 : bsel == 2 ? (t & 3) != 3 : true; };

    if (w >= WS_NP) {
        // =================================================================== CONSUMER: channels [16 cw, 16 cw + 16)
        const int cw = w - WS_NP;
        // resident weights: lane (m = jcol: channel 16 cw + m, g = tq) holds K index 32 s + 8 g + i of k-step s: s < KH -> Wj (the mean half), else Wi
        const int col = 16 * cw + jcol;
        uint32_t mw = 0u;
#pragma unroll
        for (int s = 0; s < 2 * KH; ++s) {
            const float* wr = (s < KH ? Wj : Wi) + (int64_t)col * CIN + 32 * (s % KH) + 8 * tq;
            const f32x4_t r0 = *reinterpret_cast<const f32x4_t*>(wr), r1 = *reinterpret_cast<const f32x4_t*>(wr + 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) mw = umax(mw, umax(absbits(r0[i]), absbits(r1[i])));
        }
        float sW, inv_sW;
        pow2_scales(wave_umax(mw), sW, inv_sW);     // one scale per consumer (its 16 weight rows)
        f16x8 wh[2 * KH], wl[2 * KH];
#pragma unroll
        for (int s = 0; s < 2 * KH; ++s) {
            const float* wr = (s < KH ? Wj : Wi) + (int64_t)col * CIN + 32 * (s % KH) + 8 * tq;
            const f32x4_t r0 = *reinterpret_cast<const f32x4_t*>(wr), r1 = *reinterpret_cast<const f32x4_t*>(wr + 4);
            uint32_t ph[4], pl[4];
            split2h(r0[0] * sW, r0[1] * sW, ph[0], pl[0]);
            split2h(r0[2] * sW, r0[3] * sW, ph[1], pl[1]);
            split2h(r1[0] * sW, r1[1] * sW, ph[2], pl[2]);
            split2h(r1[2] * sW, r1[3] * sW, ph[3], pl[3]);
            wh[s] = pack8h(ph);
            wl[s] = pack8h(pl);
        }
        // DEC: W0[:, 16 cw .. 16 cw + 15] (slice cw) as A-operand fragments of v_mfma_f32_16x16x16_f16, for everybody: lane (m = jcol, kg = tq) of hidden block hb
        // holds hidden unit 16 hb + m, channels 16 cw + 4 kg .. + 3; one power-of-two scale per slice
        float inv_sW0 = 1.f;
        if constexpr (DEC) {
            uint32_t m0 = 0u;
            f32x4_t raw[4];
#pragma unroll
            for (int hb = 0; hb < 4; ++hb) {
                raw[hb] = *reinterpret_cast<const f32x4_t*>(dec.W0 + (int64_t)(16 * hb + jcol) * WS_C + 16 * cw + 4 * tq);
#pragma unroll
                for (int i = 0; i < 4; ++i) m0 = umax(m0, absbits(raw[hb][i]));
            }
            float sW0;
            pow2_scales(wave_umax(m0), sW0, inv_sW0);
#pragma unroll
            for (int hb = 0; hb < 4; ++hb) {
                uint32_t ph[2], pl[2];
                split2h(raw[hb][0] * sW0, raw[hb][1] * sW0, ph[0], pl[0]);
                split2h(raw[hb][2] * sW0, raw[hb][3] * sW0, ph[1], pl[1]);
                *reinterpret_cast<uint4*>(w0buf + (((hb * 8 + cw) * 64 + lane) << 4)) = make_uint4(ph[0], ph[1], pl[0], pl[1]);
            }
        }
        __syncthreads();   // (the producers' prologue barrier)
        const int c0 = 16 * cw + 4 * tq;          // this lane's 4 consecutive output channels
        const int64_t n_it = my_n + (DEC ? 2 : 0);       // DEC: stage B runs one tile behind the product, stage C two
        for (int64_t it = FLAGS ? 1 : 0; it <= n_it; ++it) {
            if constexpr (DEC) {
                // C(it - 3): consumer 0 adds the four partial logits of a cell in one order and stores them.  (First in the iteration: the others' B(it - 1)
                // -- next iteration -- reuses this buffer, and they get there only behind this wavefront's A(it - 1) below.)
                if (cw == 0 && it >= 3) {
                    const uint32_t t = (uint32_t)(it - 3);
                    wait_for(lcnt + (t & 1), 8u * (t / 2 + 1));
                    if (lane < WS_TILE) {
                        float s0 = dcst[256], s1 = dcst[257];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float* pq = plog + (((t & 1) * 4 + q) * WS_TILE + lane) * 2;
                            s0 += pq[0];
                            s1 += pq[1];
                        }
                        const int64_t cell = tile_of(t) * WS_TILE + lane;
                        if (cell < n_dst) *reinterpret_cast<float2*>(dec.logits + cell * 2) = make_float2(s0, s1);
                    }
                }
            }
            if (it >= 1 && it <= my_n) {
                const int sl = (int)((it - 1) % RING);
                if constexpr (FLAGS) wait_for(ready + sl, (uint32_t)(WS_NP * ((it - 1) / RING + 1)));
                const char* tb = ring + sl * G::SLOT;
                f32x4_t acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
                // (the swizzle key passes through an empty asm every tile: left alone, the compiler keeps all 32 loop-invariant read addresses in registers
                // -- 19 spilled in the decoder-carrying instantiation -- instead of two integer instructions per read)
                int key = jcol;
                asm volatile("" : "+v"(key));
                // ring reads one whole k-step ahead of the products (two register sets).  The scheduling barrier pins that order: left alone, the compiler
                // sinks every read to just in front of its first use and waits for it on the spot -- sixteen exposed LDS round trips per tile.  (Two steps
                // ahead = three sets: over the 128-register budget, spills inside the loop, twice the time.)
                f16x8 xh[2][2], xl[2][2];
                auto ld_step = [&](int s, int set) {
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        const char* rp = tb + (16 * b + jcol) * G::ROWB;
                        xh[set][b] = H8(*reinterpret_cast<const uint4*>(rp + (((4 * s + tq) ^ key) << 4)));
                        xl[set][b] = H8(*reinterpret_cast<const uint4*>(rp + (((2 * PP + 4 * s + tq) ^ key) << 4)));
                    }
                };
                ld_step(0, 0);
#pragma unroll
                for (int s = 0; s < 2 * KH; ++s) {
                    const int set = s & 1;
                    if (s + 1 < 2 * KH) ld_step(s + 1, set ^ 1);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int b = 0; b < 2; ++b) acc[b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[s], xh[set][b], acc[b], 0, 0, 0);     // small terms first
#pragma unroll
                    for (int b = 0; b < 2; ++b) acc[b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[s], xl[set][b], acc[b], 0, 0, 0);
#pragma unroll
                    for (int b = 0; b < 2; ++b) acc[b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[s], xh[set][b], acc[b], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
                float invr[2];
#pragma unroll
                for (int b = 0; b < 2; ++b) invr[b] = rowf[sl * WS_TILE + 16 * b + jcol] * inv_sW;
                if constexpr (FLAGS) {
                    // every read of the slot has returned: hand it back
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    if (lane == 0) atomicAdd(const_cast<uint32_t*>(done + sl), 1u);
                }
                // (the per-channel constants are re-read from LDS every tile: 12 registers the decoder stage needs)
                const f32x4_t bb = *reinterpret_cast<const f32x4_t*>(cst + c0), sc = *reinterpret_cast<const f32x4_t*>(cst + WS_C + c0),
                              sh = *reinterpret_cast<const f32x4_t*>(cst + 2 * WS_C + c0);
                f32x4_t yv[2];
#pragma unroll
                for (int b = 0; b < 2; ++b) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float t_ = __fmaf_rn(acc[b][i], invr[b], bb[i]);
                        t_ = __fmaf_rn(t_, sc[i], sh[i]);
                        yv[b][i] = relu ? fmaxf(t_, 0.f) : t_;
                    }
                }
                if constexpr (!DEC) {
                    const int64_t cell0 = tile_of(it - 1) * WS_TILE;
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        const int64_t cell = cell0 + 16 * b + jcol;
                        if (cell < n_dst) {
                            if constexpr (IO16) {
                                const u32x2_t pk = {ub16_enc(yv[b][0]) | (ub16_enc(yv[b][1]) << 16), ub16_enc(yv[b][2]) | (ub16_enc(yv[b][3]) << 16)};
                                if (nt_store) __builtin_nontemporal_store(pk, reinterpret_cast<u32x2_t*>(out16 + cell * ldo + c0));
                                else *reinterpret_cast<u32x2_t*>(out16 + cell * ldo + c0) = pk;
                            } else {
                                if (nt_store) st_nt16(out + cell * ldo + c0, yv[b]);
                                else *reinterpret_cast<f32x4_t*>(out + cell * ldo + c0) = yv[b];
                            }
                        }
                    }
                } else {
                    // A(it - 1): this consumer's 16 channels of the tile, one scale per cell and slice (the cell's 4 lanes are 16 apart), parked as B-operand
                    // fragments; the buffer's previous tile (two back) has been read by everybody
                    const uint32_t t = (uint32_t)(it - 1);
                    wait_for(yfree + (t & 1), 8u * (t / 2));
                    char* yt = ytile + ((t & 1) << 14) + ((cw * 128 + lane) << 4);
                    float* ys = ysc + ((t & 1) * 8 + cw) * WS_TILE + jcol;
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        const float my = fmaxf(fmaxf(fabsf(yv[b][0]), fabsf(yv[b][1])), fmaxf(fabsf(yv[b][2]), fabsf(yv[b][3])));
                        float sY, inv_sY;
                        pow2_scales(cross_row_umax(bits(my)), sY, inv_sY);
                        uint32_t ph[2], pl[2];
                        split2h(yv[b][0] * sY, yv[b][1] * sY, ph[0], pl[0]);
                        split2h(yv[b][2] * sY, yv[b][3] * sY, ph[1], pl[1]);
                        *reinterpret_cast<uint4*>(yt + (b << 10)) = make_uint4(ph[0], ph[1], pl[0], pl[1]);
                        if (tq == 0) ys[16 * b] = inv_sY * inv_sW0;
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    if (lane == 0) atomicAdd(const_cast<uint32_t*>(ycnt + (t & 1)), 1u);
                }
            }
            if constexpr (DEC) {
                if (it >= 2 && it <= my_n + 1 && b_on_consumers((uint32_t)(it - 2))) stage_b((uint32_t)(it - 2), cw & 3, cw >> 2);
            }
            if constexpr (!FLAGS) tile_barrier();
        }
        return;
    }

    // ======================================================================= PRODUCER: group p of 4 cells of every tile
    __syncthreads();   // filter operand and constants in place
    const int p = w;
    const int P0 = NCH * jcol;                     // the lane's NCH channels (contiguous ownership)
    const int Q0 = 4 * jcol;                       // LINE_OWN: its two quads start at Q0 and 64 + Q0
    auto load_rp = [&](int64_t it, int& vb) {
        if (it < my_n) {
            const int64_t i0 = tile_of(it) * WS_TILE + 4 * p;
            const int nv = (int)max((int64_t)0, min((int64_t)4, n_dst - i0));
            vb = nv > 0 ? rowptr[i0 + (lane < nv ? lane : nv)] : 0;
        }
    };
    auto load_idx = [&](int64_t it, int vb, bool& reg, int& vsrc, int& veid) {
        reg = false;
        if (it < my_n) {
            const int64_t i0 = tile_of(it) * WS_TILE + 4 * p;
            const int nv = (int)max((int64_t)0, min((int64_t)4, n_dst - i0));
            if (nv > 0) {
                const int b0 = __builtin_amdgcn_readfirstlane(vb);
                reg = __all(vb == b0 + 4 * (lane < nv ? lane : nv)) != 0;
                if (reg) {
                    const int k_me = b0 + (lane < 4 * nv ? lane : 4 * nv - 1);
                    vsrc = src[k_me];
                    veid = eid ? eid[k_me] : k_me;
                }
            }
        }
    };
    int vb1 = 0, vb2 = 0, vsrc1 = 0, veid1 = 0;
    bool reg1 = false;
    load_rp(0, vb1);
    load_rp(1, vb2);
    load_idx(0, vb1, reg1, vsrc1, veid1);

    // DEC: on the tiles the rule gives to the producers (b_on_consumers) this wavefront also runs stage B of tile it - 2 (hidden block p & 3 of row block
    // p >> 2) -- BETWEEN issuing its gathers for tile `it` and using them: the job fills the gather latency, and its data (the consumers' stage A of tile
    // it - 2) is complete about when the slot the producer is going to write is handed back anyway.  Two more iterations drain the last two tiles.
    for (int64_t it = 0; it < my_n + (FLAGS ? (DEC ? 2 : 0) : 1); ++it) {
        int nv = 0, sl = 0, tl = 0, vsrc = 0, veid = 0;
        bool regular = false;
        f32x4_t xo[NV], q0, q1, rr[4][NV];              // the gathered rows (DEC: in flight across stage B); read only on the path that loaded them
        uint32_t xo16[NCH / 2], rr16[4][NCH / 2];       // IO16: the same rows as packed pairs of 16-bit values, decoded where they are used
        int64_t cell = 0;
        auto own_row = [&]() {
            if constexpr (IO16) {
                const uint16_t* rp0 = xdst16 + cell * ldx + P0;
                if constexpr (NCH == 8) {
                    const uint4 u = *reinterpret_cast<const uint4*>(rp0);
                    xo16[0] = u.x, xo16[1] = u.y, xo16[2] = u.z, xo16[3] = u.w;
                } else {
                    const uint2 u = *reinterpret_cast<const uint2*>(rp0);
                    xo16[0] = u.x, xo16[1] = u.y;
                }
            } else {
                const float* rp0 = xdst + cell * ldx + (LINE_OWN ? Q0 : P0);
#pragma unroll
                for (int v = 0; v < NV; ++v) xo[v] = *reinterpret_cast<const f32x4_t*>(rp0 + (LINE_OWN ? 64 : 4) * v);
            }
        };
        auto neighbour_rows = [&]() {
            const float* er = ea + (int64_t)__shfl(veid, jcol) * lde;
            q0 = *reinterpret_cast<const f32x4_t*>(er + 8 * (tq < 2 ? tq : 2));
            q1 = *reinterpret_cast<const f32x4_t*>(er + 8 * (tq < 1 ? tq : 1) + 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if constexpr (IO16) {
                    const uint16_t* rp = x16 + (int64_t)__shfl(vsrc, tl * 4 + r) * ldx + P0;
                    if constexpr (NCH == 8) {
                        const uint4 u = *reinterpret_cast<const uint4*>(rp);
                        rr16[r][0] = u.x, rr16[r][1] = u.y, rr16[r][2] = u.z, rr16[r][3] = u.w;
                    } else {
                        const uint2 u = *reinterpret_cast<const uint2*>(rp);
                        rr16[r][0] = u.x, rr16[r][1] = u.y;
                    }
                } else {
                    const float* rp = x + (int64_t)__shfl(vsrc, tl * 4 + r) * ldx + (LINE_OWN ? Q0 : P0);
#pragma unroll
                    for (int v = 0; v < NV; ++v) rr[r][v] = *reinterpret_cast<const f32x4_t*>(rp + (LINE_OWN ? 64 : 4) * v);
                }
            }
        };
        if (it < my_n) {
            const int64_t i0 = tile_of(it) * WS_TILE + 4 * p;
            sl = (int)(it % RING);
            nv = (int)max((int64_t)0, min((int64_t)4, n_dst - i0));
            regular = reg1;
            vsrc = vsrc1;
            veid = veid1;
            vb1 = vb2;
            load_idx(it + 1, vb1, reg1, vsrc1, veid1);
            load_rp(it + 2, vb2);
            tl = tq < nv ? tq : nv - 1;      // a short group at the end of the graph: clamped (duplicated) cells
            cell = i0 + tl;
        }
        if constexpr (DEC) {
            // (the gathers leave ahead of stage B only in this instantiation: in the plain one the loads and their use stay in ONE basic block -- split, the
            // plain launch lost 9 %: 0.412 -> 0.450 ms)
            if (it < my_n && nv > 0) {
                own_row();
                if (regular) neighbour_rows();
            }
            if (it >= 2 && !b_on_consumers((uint32_t)(it - 2))) stage_b((uint32_t)(it - 2), p & 3, p >> 2);
        }
        if (it < my_n) {
            if (nv > 0) {
                if constexpr (!DEC) own_row();
                float aout[NCH], xv[NCH];
#pragma unroll
                for (int v = 0; v < NV; ++v) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if constexpr (IO16) xv[4 * v + i] = (i & 1) ? ub16_hi(xo16[2 * v + (i >> 1)]) : ub16_lo(xo16[2 * v + (i >> 1)]);
                        else xv[4 * v + i] = xo[v][i];
                    }
                }
                if (regular) {
                    if constexpr (!DEC) neighbour_rows();
                    float av[8];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        av[i] = tq < 3 ? q0[i] : 0.f;
                        av[4 + i] = tq < 2 ? q1[i] : 0.f;
                    }
                    if (tq == 2) av[4] = 1.0f;
                    float mf = 0.f;
#pragma unroll
                    for (int i = 0; i < 8; i += 2) mf = fmaxf(fmaxf(mf, fabsf(av[i])), fabsf(av[i + 1]));
                    float sA, inv_sA;
                    pow2_scales(cross_row_umax(bits(mf)), sA, inv_sA);      // one scale per EDGE (a row of the operand)
                    uint32_t ph[4], pl[4];
#pragma unroll
                    for (int d = 0; d < 4; ++d) split2h(av[2 * d] * sA, av[2 * d + 1] * sA, ph[d], pl[d]);
                    float inv_e[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) inv_e[r] = __shfl(inv_sA, 4 * tq + r);
                    const f16x8 ah = pack8h(ph), al = pack8h(pl);
#pragma unroll
                    for (int c4 = 0; c4 < NCH; c4 += 2) {
                        f16x8 bh[2], bl[2];
                        f32x4_t d[2];
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            const char* bp = bpbuf + (((c4 + u) * 2) * 48 + (tq < 3 ? tq : 0) * 16 + jcol) * 16;   // k-group 3 re-reads group 0: its A operand is zero
                            bh[u] = H8(*reinterpret_cast<const uint4*>(bp));
                            bl[u] = H8(*reinterpret_cast<const uint4*>(bp + 768));
                            d[u] = f32x4_t{0.f, 0.f, 0.f, 0.f};
                        }
#pragma unroll
                        for (int u = 0; u < 2; ++u) d[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[u], d[u], 0, 0, 0);
#pragma unroll
                        for (int u = 0; u < 2; ++u) d[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[u], d[u], 0, 0, 0);
#pragma unroll
                        for (int u = 0; u < 2; ++u) d[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[u], d[u], 0, 0, 0);
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            const int c8 = c4 + u;
#pragma unroll
                            for (int r = 0; r < 4; ++r) d[u][r] *= inv_e[r];     // exact: powers of two
                            auto nb_ = [&](int r) {
                                if constexpr (IO16) return (c8 & 1) ? ub16_hi(rr16[r][c8 >> 1]) : ub16_lo(rr16[r][c8 >> 1]);
                                else return rr[r][c8 >> 2][c8 & 3];
                            };
                            float a = __fmul_rn(nb_(0), d[u][0]);
#pragma unroll
                            for (int r = 1; r < 4; ++r) a = __fmaf_rn(nb_(r), d[u][r], a);
                            aout[c8] = a * (0.25f * inv_sWe);
                        }
                    }
                } else {
                    // generic path (a group with any in-degree other than 4): plain fp32 per lane, one edge at a time (never on Delaunay scenes)
#pragma unroll
                    for (int cb = 0; cb < NCH; ++cb) aout[cb] = 0.f;
                    if (tq < nv) {
                        const int b = rowptr[cell], e_end = rowptr[cell + 1];
                        for (int k = b; k < e_end; ++k) {
                            const int s_ = src[k];
                            const float* ar = ea + (int64_t)(eid ? eid[k] : k) * lde;
#pragma unroll 1
                            for (int cb = 0; cb < NCH; ++cb) {
                                const int c = chan(jcol, cb);
                                float pf = be[c];
                                for (int f = 0; f < FE; ++f) pf = __fmaf_rn(We[(int64_t)c * FE + f], ar[f], pf);
                                const float xs_ = IO16 ? __builtin_bit_cast(float, (uint32_t)x16[(int64_t)s_ * ldx + c] << 15) : x[(int64_t)s_ * ldx + c];
                                aout[cb] = __fadd_rn(aout[cb], __fmul_rn(xs_, pf));
                            }
                        }
                        const float cnt = (float)max(e_end - b, 1);
#pragma unroll
                        for (int cb = 0; cb < NCH; ++cb) aout[cb] = __fdiv_rn(aout[cb], cnt);
                    }
                }
                // the [mean | own] row: one power-of-two scale (its 16 lanes), split, parked in the ring slot of this tile
                float ma = 0.f;
#pragma unroll
                for (int i = 0; i < NCH; i += 2) {
                    ma = fmaxf(fmaxf(ma, fabsf(aout[i])), fabsf(aout[i + 1]));
                    ma = fmaxf(fmaxf(ma, fabsf(xv[i])), fabsf(xv[i + 1]));
                }
                float s_, inv_;
                pow2_scales(row16_umax(bits(ma)), s_, inv_);
                const int T = 4 * p + tq;
                if constexpr (FLAGS) wait_for(done + sl, (uint32_t)(8 * (it / RING)));      // the slot's previous tile has been read by all 8 consumers
                if (jcol == 0) rowf[sl * WS_TILE + T] = inv_;
                uint32_t ah_[NCH / 2], al_[NCH / 2], xh_[NCH / 2], xl_[NCH / 2];
#pragma unroll
                for (int d = 0; d < NCH / 2; ++d) {
                    split2h(aout[2 * d] * s_, aout[2 * d + 1] * s_, ah_[d], al_[d]);
                    split2h(xv[2 * d] * s_, xv[2 * d + 1] * s_, xh_[d], xl_[d]);
                }
                char* rowp = ring + sl * G::SLOT + T * G::ROWB;
                const int key = T & 15;
                if constexpr (LINE_OWN) {
                    // the lane's two quads of channels = half a piece each: quad q sits in piece 8 q + jcol / 2 of each quarter, half jcol & 1
                    const int half = (jcol & 1) << 3;
#pragma unroll
                    for (int qd = 0; qd < 2; ++qd) {
                        const int pc = 8 * qd + (jcol >> 1);
                        *reinterpret_cast<uint2*>(rowp + ((pc ^ key) << 4) + half) = make_uint2(ah_[2 * qd], ah_[2 * qd + 1]);
                        *reinterpret_cast<uint2*>(rowp + (((PP + pc) ^ key) << 4) + half) = make_uint2(xh_[2 * qd], xh_[2 * qd + 1]);
                        *reinterpret_cast<uint2*>(rowp + (((2 * PP + pc) ^ key) << 4) + half) = make_uint2(al_[2 * qd], al_[2 * qd + 1]);
                        *reinterpret_cast<uint2*>(rowp + (((3 * PP + pc) ^ key) << 4) + half) = make_uint2(xl_[2 * qd], xl_[2 * qd + 1]);
                    }
                } else if constexpr (NCH == 8) {
                    // the lane's 8 channels = one 16-byte piece of each quarter
                    *reinterpret_cast<uint4*>(rowp + ((jcol ^ key) << 4)) = make_uint4(ah_[0], ah_[1], ah_[2], ah_[3]);
                    *reinterpret_cast<uint4*>(rowp + (((PP + jcol) ^ key) << 4)) = make_uint4(xh_[0], xh_[1], xh_[2], xh_[3]);
                    *reinterpret_cast<uint4*>(rowp + (((2 * PP + jcol) ^ key) << 4)) = make_uint4(al_[0], al_[1], al_[2], al_[3]);
                    *reinterpret_cast<uint4*>(rowp + (((3 * PP + jcol) ^ key) << 4)) = make_uint4(xl_[0], xl_[1], xl_[2], xl_[3]);
                } else {
                    // the lane's 4 channels = half a piece: lanes 2 i, 2 i + 1 fill piece i of each quarter
                    const int pc = jcol >> 1, half = (jcol & 1) << 3;
                    *reinterpret_cast<uint2*>(rowp + ((pc ^ key) << 4) + half) = make_uint2(ah_[0], ah_[1]);
                    *reinterpret_cast<uint2*>(rowp + (((PP + pc) ^ key) << 4) + half) = make_uint2(xh_[0], xh_[1]);
                    *reinterpret_cast<uint2*>(rowp + (((2 * PP + pc) ^ key) << 4) + half) = make_uint2(al_[0], al_[1]);
                    *reinterpret_cast<uint2*>(rowp + (((3 * PP + pc) ^ key) << 4) + half) = make_uint2(xl_[0], xl_[1]);
                }
            } else if constexpr (FLAGS) {
                // a group past the end of the graph parks nothing, but its count must not land in `ready` while the slot's PREVIOUS tile is still being
                // produced: the consumers would take seven real producers plus this one for eight and read a late producer's rows before they are written
                // (seen on the workgroup that owns the last, partial tile of a scene: tools/det_ws_real.py)
                wait_for(done + sl, (uint32_t)(8 * (it / RING)));
            }
            if constexpr (FLAGS) {
                // this producer's rows of the tile are parked (LDS operations of a wavefront complete in order)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (lane == 0) atomicAdd(const_cast<uint32_t*>(ready + sl), 1u);
            }
        }
        if constexpr (!FLAGS) tile_barrier();
    }
}

}  // namespace

// 1 = the wave-specialised kernel takes the 64 -> 128 / 128 -> 128 layers in the default arithmetic (DGNN_WS=0: k_sage_fused_mfma<.., 128> as in rounds 2-4)
int dgnn_ws_enabled() {
    static const int v = getenv("DGNN_WS") ? atoi(getenv("DGNN_WS")) : 1;
    return v;
}

// C ABI (include/dgnn_hip.h): which kernel dgnn_sage_layer_fused_fwd / _decoder_fwd launch for a 64 -> 128 / 128 -> 128 layer in DGNN_GEMM_F16X2
extern "C" int dgnn_wave_specialised_enabled(void) { return dgnn_ws_enabled() != 0 ? 1 : 0; }

// same contract as dgnn_sage_layer_fused_mfma_try for c_in in {64, 128}, c_out == 128; with W0 != NULL (c_in == 128) the launch carries the decoder
// 128 -> 64 -> 2 and writes logits [n_dst, 2] instead of rows (`out` unused).  DGNN_E_UNSUPPORTED: the caller keeps the two-phase kernel
int dgnn_sage_layer_fused_ws_try(const int32_t* rowptr, const int32_t* src, const int32_t* eid, int64_t n_dst, const float* x_src, const float* x_dst, int64_t ldx,
                                 int c_in, const float* edge_attr, int64_t lde, const float* We, const float* be, const float* Wj, const float* bj, const float* Wi,
                                 const float* scale, const float* shift, int relu, int c_out, float* out, int64_t ldo, hipStream_t stream, const float* W0,
                                 const float* b0, const float* scale1, const float* shift1, const float* W3, const float* b3, float* logits) {
    const bool dec = W0 != nullptr;
    if ((c_in != WS_C && !(c_in == 64 && !dec)) || c_out != WS_C || lde != FE || ldx % 4 != 0 || (!dec && ldo % 4 != 0) ||
        ((((uintptr_t)x_src | (uintptr_t)x_dst | (uintptr_t)edge_attr | (uintptr_t)(dec ? nullptr : out) | (uintptr_t)Wj | (uintptr_t)Wi | (uintptr_t)W0) % 16) != 0) ||
        (dec && ((uintptr_t)logits % 8) != 0))
        return DGNN_E_UNSUPPORTED;
    static const int ws64 = getenv("DGNN_WS_64") ? atoi(getenv("DGNN_WS_64")) : 1;     // 0: the 64 -> 128 layer stays on the two-phase kernel
    if (c_in == 64 && !ws64) return DGNN_E_UNSUPPORTED;
    const int64_t ntiles = dgnn_cdiv(n_dst, WS_TILE);
    int grid = (int)(ntiles < DGNN_NUM_CU ? ntiles : DGNN_NUM_CU);
    if (grid < 1) grid = 1;
    static const int ring = getenv("DGNN_WS_RING") ? atoi(getenv("DGNN_WS_RING")) : 22;          // 2 = one barrier per tile; 22 / 3 / 4 = counters, 2 / 3 / 4 slots (measured: equal)
    static const int knobs = getenv("DGNN_WS_NT") ? atoi(getenv("DGNN_WS_NT")) : 33;   // 1: non-temporal row stores; bits 4-5: who runs the decoder's stage B (2 = consumers on 3 tiles of 4)
    const WsDec d{W0, b0, scale1, shift1, W3, b3, logits};
#define DGNN_WS_GO(C_, R_, F_, D_)                                                                                                                          \
    do {                                                                                                                                                    \
        static bool attr_[DGNN_MAX_DEVICES] = {};                                                                                                           \
        const size_t sm_ = D_ ? WsL<C_, R_>::SMEM_DEC : WsL<C_, R_>::SMEM;                                                                                  \
        dgnn_allow_dynamic_lds(reinterpret_cast<const void*>(&k_sage_fused_ws<C_, R_, F_, D_>), sm_, attr_);                                                \
        hipLaunchKernelGGL((k_sage_fused_ws<C_, R_, F_, D_>), dim3(grid), dim3(1024), sm_, stream, rowptr, src, eid, n_dst, x_src, x_dst, ldx, edge_attr, lde, We, \
                           be, Wj, bj, Wi, scale, shift, relu, out, ldo, ntiles, knobs, d);                                                                 \
    } while (0)
    if (c_in == 64) {
        if (ring == 2) DGNN_WS_GO(64, 2, false, false);
        else if (ring == 22) DGNN_WS_GO(64, 2, true, false);
        else DGNN_WS_GO(64, 4, true, false);          // (DGNN_WS_RING=3 / 4: four 16 KB slots)
    } else if (dec) DGNN_WS_GO(128, 2, true, true);
    else if (ring == 2) DGNN_WS_GO(128, 2, false, false);
    else if (ring == 3) DGNN_WS_GO(128, 3, true, false);
    else if (ring == 4) DGNN_WS_GO(128, 4, true, false);
    else DGNN_WS_GO(128, 2, true, false);
#undef DGNN_WS_GO
    return dgnn_check_launch(dec ? "sage_layer_fused_decoder_fwd(wave-specialised)" : "sage_layer_fused_fwd(wave-specialised)");
}

// The same kernel on the UNSIGNED 16-bit rows of the bf16-storage chain (DGNN_BF16_COMPENSATED | _ROWS_IN_UNSIGNED | _ROWS_OUT_UNSIGNED, fused_bf16.hip): rows
// decoded to fp32 where the producers use them (exact), the fp16 two-part arithmetic of the fp32-I/O kernel in between, rows encoded (round to nearest even
// on bit 15) in the consumers' epilogue.  c_in in {64, 128}, c_out == 128, relu (the format has no sign).  DGNN_E_UNSUPPORTED: the caller keeps its own kernel.
int dgnn_sage_layer_fused_ws16_try(const int32_t* rowptr, const int32_t* src, const int32_t* eid, int64_t n_dst, const uint16_t* x_src, const uint16_t* x_dst,
                                   int64_t ldx, int c_in, const float* edge_attr, int64_t lde, const float* We, const float* be, const float* Wj, const float* bj,
                                   const float* Wi, const float* scale, const float* shift, int relu, int c_out, uint16_t* out, int64_t ldo, hipStream_t stream) {
    static const int on = getenv("DGNN_WS_16") ? atoi(getenv("DGNN_WS_16")) : 1;
    const int nch = c_in / 16;
    if (!on || !dgnn_ws_enabled() || (c_in != 128 && c_in != 64) || c_out != WS_C || !relu || lde != FE || ldx % nch != 0 || ldo % 4 != 0 ||
        ((((uintptr_t)x_src | (uintptr_t)x_dst) % (2 * nch)) != 0) || ((uintptr_t)out % 8) != 0 ||
        ((((uintptr_t)edge_attr | (uintptr_t)Wj | (uintptr_t)Wi) % 16) != 0))
        return DGNN_E_UNSUPPORTED;
    const int64_t ntiles = dgnn_cdiv(n_dst, WS_TILE);
    int grid = (int)(ntiles < DGNN_NUM_CU ? ntiles : DGNN_NUM_CU);
    if (grid < 1) grid = 1;
    static const int knobs = getenv("DGNN_WS_NT") ? atoi(getenv("DGNN_WS_NT")) : 33;
    const WsDec d{};
    const float* xs = reinterpret_cast<const float*>(x_src);
    const float* xd = reinterpret_cast<const float*>(x_dst);
    float* o = reinterpret_cast<float*>(out);
    if (c_in == 64) {
        static bool attr_[DGNN_MAX_DEVICES] = {};
        const size_t sm_ = WsL<64, 4>::SMEM;
        dgnn_allow_dynamic_lds(reinterpret_cast<const void*>(&k_sage_fused_ws<64, 4, true, false, true>), sm_, attr_);
        hipLaunchKernelGGL((k_sage_fused_ws<64, 4, true, false, true>), dim3(grid), dim3(1024), sm_, stream, rowptr, src, eid, n_dst, xs, xd, ldx, edge_attr, lde, We, be,
                           Wj, bj, Wi, scale, shift, relu, o, ldo, ntiles, knobs, d);
    } else {
        static bool attr_[DGNN_MAX_DEVICES] = {};
        const size_t sm_ = WsL<128, 2>::SMEM;
        dgnn_allow_dynamic_lds(reinterpret_cast<const void*>(&k_sage_fused_ws<128, 2, true, false, true>), sm_, attr_);
        hipLaunchKernelGGL((k_sage_fused_ws<128, 2, true, false, true>), dim3(grid), dim3(1024), sm_, stream, rowptr, src, eid, n_dst, xs, xd, ldx, edge_attr, lde, We, be,
                           Wj, bj, Wi, scale, shift, relu, o, ldo, ntiles, knobs, d);
    }
    return dgnn_check_launch("sage_layer_fused_fwd_bf16(wave-specialised)");
}
