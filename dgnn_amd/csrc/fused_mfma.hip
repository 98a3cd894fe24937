// Fused SAGE inference layer, "all matrix core" variant (gemm_mode DGNN_GEMM_BF16X3_FILTER):
// the filter MLP phi = We.A + be runs on v_mfma_f32_16x16x32_bf16 as well, not on the VALU.
//
// Why: MFMA and fp32 VALU work largely serialise on a SIMD (fused.hip header; profiles/r01_ubench_valu_kinds.txt has the
// per-instruction-kind picture), and in the split-bf16 variant the 20-tap filter is ~70 % of the remaining VALU instructions.  As a matrix product it is tiny and cheap:
//     PHI[16 edges x 16 channels] = A[16 edges x 32] . B[32 x 16 channels]
// with A = the wave's 16 edge-attribute rows (k < 20 attributes, k = 20 the constant 1 that carries the bias,
// rest 0) and B = We^T | be.  Both are split exactly into 3 bf16 parts and the 6 partial products of weight
// >= 2^-18 are accumulated in fp32 -- the same fp32-class scheme as the dense part (fused.hip MODE 1).
//
// The MFMA accumulator layout does the segmented reduction for free: in v_mfma_f32_16x16x32_bf16 lane l holds
// D[row = 4*(l>>4) + r][col = l&15] for r = 0..3 -- rows 4t..4t+3 are exactly the 4 in-edges of the wave's
// t-th tet (edges are in plan order), so one lane owns all 4 messages of (tet t = l>>4, channel) and the
// mean is an in-lane, in-order 4-term sum (the reference's scatter order), no shuffles.
// Column block cb of lane column j is channel NB*j + cb (NB = C_in/16 blocks), so a lane's channels over
// all blocks are NB contiguous floats: its x rows are fetched as one 16/32-byte piece per row, and its
// outputs form whole 8-k octets of the split-bf16 A-tile (3 x ds_write_b128 per row segment).
//
// Everything else is fused.hip's uniform two-phase loop: loads of tile t+1 (neighbour rows, own rows,
// LDS-DMA of the attribute block) are issued before the single per-tile barrier and land under the matrix
// phase; the dense part is the 32x32x16 split-bf16 product -- K split between wave pairs with a delayed epilogue at
// C_in = 128 (8 waves), full K per wave and an immediate epilogue for C_in <= 64 (4 waves, two workgroups per CU; Cfg2).
// Edge attributes are read in the caller's order through the plan's eid (per-lane DMA addresses) when eid is given.
//
// Template parameters DSP / FSP choose the arithmetic of the dense / filter product: 3 = the exact 3-part bf16 split described above
// (6 products), 2 = the fp16 two-part form of fused_common.h (power-of-two scale per A-tile row, per edge and per weight matrix, 3 products;
// gemm modes DGNN_GEMM_F16X2_DENSE / DGNN_GEMM_F16X2, the Python host's default).  The compile-time switches below record what was measured
// and not kept; DGNN_WHATIF builds remove one ingredient at a time.
#include "fused_common.h"

#ifndef DGNN_SMALL_NW
#define DGNN_SMALL_NW 4  // wavefronts per workgroup for C_in <= 64 (4: two independent workgroups per CU; 8: one)
#endif
#ifndef DGNN_EARLY_ISSUE
#define DGNN_EARLY_ISSUE 0  // measured: the tile period does not move (5.03 -> 5.19 us): the issue phase is address arithmetic and index shuffles, not memory stalls, and it costs ~100 register moves
#endif
#ifndef DGNN_DENSE_PREFETCH
#define DGNN_DENSE_PREFETCH 0  // measured: no gain (0.527 vs 0.521 ms): the second wavefront of the SIMD already covers the LDS round trips
#endif
#ifndef DGNN_FILTER_PIPE
#define DGNN_FILTER_PIPE 0  // measured: no gain (0.531 vs 0.527 ms at 128 -> 128)
#endif
#ifndef DGNN_SKEW
#define DGNN_SKEW 0  // measured: 0.52 -> 0.84 ms at 128 -> 128.  A lone filter-phase wavefront per SIMD takes as long as two interleaved ones (1.14 us: the phase is a chain of LDS -> matrix core -> vector ALU dependencies, latency-bound per wavefront), so a slot lasts a whole P and the tile two of them
#endif
#ifndef DGNN_FILTER_BREG
#define DGNN_FILTER_BREG 0  // measured: slower on both small layers (0.24 -> 0.25, 0.36 -> 0.38 ms; 16 / 32 more live registers, spills at 64 -> 128)
#endif
#ifndef DGNN_SMALL_OCC
#define DGNN_SMALL_OCC 2
#endif
#ifndef DGNN_TR
#define DGNN_TR 0  // measured (tools/variants.py, same box): 16-byte stores of 32-byte row pieces cost the small layers 7 % (0.229 -> 0.247, 0.360 -> 0.380 ms: each store
                   // instruction touches 32 cache lines instead of 2) and leave 128 -> 128 where it was
#endif
// What-if builds (tools/build_variant.sh <name> -DDGNN_WHATIF=<bits>): one ingredient removed, results garbage, only the time matters.
//   1 no dense-phase products, 2 no filter products, 4 no output stores (K-split kernels), 8 neighbour rows = own row (no gathers),
//   16 no attribute DMA.  Reading them needs care: removing the DMA made every layer 17-27 % faster, yet neither requesting it a whole
//   phase earlier (double-buffered strips) nor replacing it by coalesced reads of a pre-split operand cache written by the first layer
//   moved the time at all -- with the DMA gone the filter operand no longer changes from tile to tile, and this part clocks visibly
//   higher on quieter operands.
#ifndef DGNN_WHATIF
#define DGNN_WHATIF 0
#endif
#ifndef DGNN_PLANAR
#define DGNN_PLANAR 1
#endif
#ifndef DGNN_ROW_SHIFT
#define DGNN_ROW_SHIFT 1
#endif
#ifndef DGNN_NT_STORES
#define DGNN_NT_STORES 1
#endif
#ifndef DGNN_NT_ATTR
#define DGNN_NT_ATTR 0  // measured: streaming the attribute DMA past the caches costs 1.7 % (1.764 -> 1.795 ms per step)
#endif
#ifndef DGNN_PHASE_PRIO
#define DGNN_PHASE_PRIO 0  // measured: balances the barrier waits (0.6/0.4 us instead of 1.4/0.2) but the tile period does not move
#endif
#ifndef DGNN_YOUNG_PRIO
#define DGNN_YOUNG_PRIO 0  // measured: it only swaps which wave of a SIMD waits at the barrier (zero-sum)
#endif

namespace {
using namespace fused;

typedef float f32x4_t __attribute__((ext_vector_type(4)));
#define H8(v) __builtin_bit_cast(f16x8, v)
template <int V> struct IC { static constexpr int value = V; };
// output rows are written once and read by the NEXT launch: streamed past the caches (DGNN_NT_STORES) they do not evict the feature rows
// the gathers of this launch hit in L2 / Infinity Cache
// the edge attributes are read once per launch: non-temporal DMA (aux bit 1 = nt) keeps them from displacing feature rows
__device__ __forceinline__ void glds16_s(const float* g, float* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0,
                                     DGNN_NT_ATTR ? 2 : 0);
}
__device__ __forceinline__ void glds4_s(const float* g, float* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)lds_wave_base, 4, 0,
                                     DGNN_NT_ATTR ? 2 : 0);
}
__device__ __forceinline__ void st_out(float* p, float v) {
#if DGNN_NT_STORES
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}

// NW wavefronts per workgroup, KS = how many ways the dense phase splits K between wavefronts.
//   (8, 2): 8 waves = (32-column slice) x (K half) [x row group]; the only arrangement whose resident weights fit at C_in = 128.
//   (4, 1): 4 waves = (32-column slice) [x row group], full K per wave, no partial-sum exchange; 70-76 KB of LDS, so TWO
//           workgroups share a CU with independent barriers -- one fills the other's barrier / latency bubbles.  Used for
//           C_in <= 64, where a tile is too little work to hide its own fixed latencies.
//   DSP / FSP = parts per operand of the dense product / the filter product: 3 = bf16 x 3 (6 products), 2 = fp16 x 2 with
//   power-of-two group scales (3 products; fused_common.h).  Scaling groups: one per A-tile row (a tet's [mean | own] row, so a tet's
//   result does not depend on its neighbours in the tile), one per weight matrix pair [Wj | Wi], one per [We | be], one per 16-edge
//   attribute block of the filter product (FSP == 2).
//   DEC (128 -> 128, fp16 forms only): the launch also carries the decoder Linear(128 -> 64) - BN - ReLU - Linear(64 -> 2) and writes logits
//   instead of the layer's rows (see the DEC block in the kernel).
template <int CIN_PAD, int COUT, int NW = 8, int KS = 2, int DSP = 3, int FSP = 3, bool DEC = false>
struct Cfg2 {
    static constexpr int K = 2 * CIN_PAD;
    static constexpr int NSLICE = COUT / 32;
    static constexpr int RG = NW / (KS * NSLICE);
    static constexpr int TILE = 32 * RG;
    static constexpr int OCT = 16 * DSP;                  // one k-octet of an A-tile row: [hi|mid|lo] or [hi|lo], 16 B each
    static constexpr int ROWB = (K / 8) * OCT + 16;       // A-tile row: K/8 octets + pad
    static constexpr int A_BYTES = TILE * ROWB;
    static constexpr int TPW = TILE / NW;                 // tets per wave (4, 8 or 16)
    static constexpr int RB = TPW / 4;                    // 16-edge row blocks per wave
    static constexpr int NQ = TPW * 4;                    // edges per wave
    static constexpr int NB = CIN_PAD / 16;               // column blocks = contiguous channels per lane (8, 4, 2)
    static constexpr int EA_BYTES = NQ * FE * 4;          // 1280, 2560 or 5120: whole KiB by 16-B DMA, the rest by 4-B DMA
    static constexpr int EA_FULL = EA_BYTES / 1024, EA_TAIL = (EA_BYTES % 1024) / 256;
    static constexpr int BP_BYTES = NB * FSP * 768;       // [cb][part][g<3][j<16] x 16 B filter operand parts
    static constexpr int RED_BYTES = KS == 2 ? NW * 8 * 64 * 4 : 0;
    static constexpr int ROWF_BYTES = DSP == 2 ? 4 * TILE * 4 : 0;  // per-row inverse scales, 4 tiles deep (written in P(it), read up to the
                                                                    // delayed epilogue after barrier it+1 while P(it+2) may already write)
    static constexpr int SC_BYTES = 16;                   // launch-wide weight maxima (prologue)
    static constexpr int COLP_BYTES = (DSP == 2 && (DGNN_TR || DEC)) ? 3 * COUT * 4 : 0;  // [bias | scale | shift][COUT]: the transposed epilogue's lanes own 8 / 16 columns
    // decoder stage: W0 as matrix-core fragments [2 output blocks][8 column slabs][hi | lo][64 lanes] x 16 B, then A1 | B1 [64] (folded bias / BatchNorm),
    // W3 [2][64], b3 [2] (+2 pad), per-(slab, row) inverse scales [8][32], the second output block's partial logits [32][2], a hand-off flag
    static constexpr int DEC_W0 = 2 * 8 * 2 * 1024, DEC_CONST = (64 + 64 + 128 + 4) * 4, DEC_SCALE = 8 * 32 * 4, DEC_PBUF = 32 * 2 * 4 + 16;
    static constexpr int DEC_BYTES = DEC ? DEC_W0 + DEC_CONST + DEC_SCALE + DEC_PBUF : 0;
    static constexpr int SMEM_BYTES = 2 * A_BYTES + NW * EA_BYTES + 2 * RED_BYTES + BP_BYTES + ROWF_BYTES + SC_BYTES + COLP_BYTES + DEC_BYTES;
    // prepared-parameter buffer: 16 B header (sW, 1/sW, sWe, 1/sWe) | filter operand parts | wb fragments [NW][K/16/KS][DSP][64 lanes] x 16 B | DEC: W0 fragments, constants
    static constexpr int PREP_WB = NW * (K / 16 / KS) * DSP * 1024;
    static constexpr int PREP_BYTES = 16 + BP_BYTES + PREP_WB + (DEC ? DEC_W0 + DEC_CONST : 0);
    static_assert(!DEC || (CIN_PAD == 128 && COUT == 128 && NW == 8 && KS == 2 && DSP == 2), "decoder stage: 128 -> 128, eight waves, fp16 dense form");
    static_assert(SMEM_BYTES <= 160 * 1024, "LDS");
    static constexpr int NWB = K / 16 / KS;               // dense part: k-steps of 16 per wave
    static_assert(EA_BYTES % 256 == 0, "attribute block must be DMA-able");
    static_assert(RG >= 1 && NQ <= 64, "wave roles");
};

// part: -1 = the lane's NB channels; 0 / 1 = their first / second half (NB == 8, where the row piece is two 16-byte loads)
template <int NB, int part = -1>
__device__ __forceinline__ void ld_vec(float (&v)[NB], const float* p, bool vec) {
    if (NB == 8 && (vec || part >= 0)) {
        if (part != 1) {
            const f32x4_t a = *reinterpret_cast<const f32x4_t*>(p);
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = a[i];
        }
        if (part != 0) {
            const f32x4_t b = *reinterpret_cast<const f32x4_t*>(p + 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) v[(4 + i) % NB] = b[i];
        }
    } else if (part == 1) {
    } else if (NB == 4 && vec) {
        const f32x4_t a = *reinterpret_cast<const f32x4_t*>(p);
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i % NB] = a[i];
    } else {
#pragma unroll
        for (int i = 0; i < NB; ++i) v[i] = p[i];
    }
}

// wavefronts per SIMD the register allocation aims at: 2 (eight-wave workgroup, or two four-wave ones per CU); DGNN_SMALL_OCC = 3 asks for three
// four-wave workgroups per CU where their LDS allows it (<= 53 KB each)
template <int CIN_PAD, int COUT, int NW, int KS, int DSP, int FSP>
constexpr int occ_of() { return (NW == 4 && Cfg2<CIN_PAD, COUT, NW, KS, DSP, FSP>::SMEM_BYTES <= 53 * 1024) ? DGNN_SMALL_OCC : 2; }

// parameters of the decoder stage (DEC): reference learning/surfaceNetStaticEdgeFilters.py:180-187, applied at :350-351
struct DecArgs {
    const float* W0;      // [64, 128]
    const float* b0;      // [64]
    const float* scale1;  // [64] folded BatchNorm(eval) of the decoder, or NULL
    const float* shift1;  // [64]
    const float* W3;      // [2, 64]
    const float* b3;      // [2]
    float* logits;        // [n_dst, 2] (row stride 2)
    // prepared parameters (fp16 forms): what the prologue computes from the weights -- power-of-two scales, the filter operand parts, every
    // wavefront's resident dense-weight fragments, the decoder's W0 fragments and constants -- kept in a buffer across launches.
    // prep_mode 0: compute them (the launch's first ~15 us); 1: compute them, WRITE them to `prep` and return (dgnn_sage_layer_prepare);
    // 2: READ them from `prep` (coalesced 16-byte loads).  Same values either way: bit-identical results.
    void* prep;
    int prep_mode;
};

template <int CIN_PAD, int COUT, int NW, int KS, int DSP, int FSP, bool DEC = false>
__global__ void __launch_bounds__(64 * NW, (occ_of<CIN_PAD, COUT, NW, KS, DSP, FSP>()))
k_sage_fused_mfma(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ src, const int32_t* __restrict__ eid, int64_t n_dst,
                  const float* __restrict__ x, const float* __restrict__ xdst, int64_t ldx, int c_in, const float* __restrict__ ea, int64_t lde,
                  const float* __restrict__ We, const float* __restrict__ be, const float* __restrict__ Wj,
                  const float* __restrict__ bj, const float* __restrict__ Wi, const float* __restrict__ scale,
                  const float* __restrict__ shift, int relu, float* __restrict__ out, int64_t ldo, int64_t ntiles,
                  int xvec, int64_t* __restrict__ trace, int64_t trace_cap, DecArgs dec) {
    using C = Cfg2<CIN_PAD, COUT, NW, KS, DSP, FSP, DEC>;
    constexpr int ROWB = C::ROWB, TILE = C::TILE, TPW = C::TPW, RB = C::RB, NB = C::NB, NWB = C::NWB, OCT = C::OCT;
    // the next tile's gathers are issued between the pieces of the filter phase (fp16 forms: the bf16 x 3 form at 128 -> 128 has no
    // registers left for the overlap of old and new rows) or as one burst at its end
    constexpr bool EARLY = DSP == 2 && DGNN_EARLY_ISSUE;
    extern __shared__ __attribute__((aligned(16))) char smem2[];
    char* const abuf = smem2;                                        // [2][A_BYTES]
    char* const eabuf = smem2 + 2 * C::A_BYTES;                      // [NW][EA_BYTES] fp32 attribute strips
    float* const redbuf = reinterpret_cast<float*>(eabuf + NW * C::EA_BYTES);     // [2][NW][8][64] (KS == 2 only)
    char* const bpbuf = reinterpret_cast<char*>(redbuf) + 2 * C::RED_BYTES;       // filter operand parts
    float* const rowf = reinterpret_cast<float*>(bpbuf + C::BP_BYTES);            // [4][TILE] (DSP == 2 only)
    uint32_t* const scbuf = reinterpret_cast<uint32_t*>(bpbuf + C::BP_BYTES + C::ROWF_BYTES);
    float* const colp = reinterpret_cast<float*>(bpbuf + C::BP_BYTES + C::ROWF_BYTES + C::SC_BYTES);
    // DSP == 2: the dense product is taken transposed (weights as the A operand, tet rows as the B operand -- the per-lane fragments are the
    // same either way), so a lane ends up with ONE tet row and 4-column runs of it: the result leaves as 16-byte stores (2 or 4 per lane and
    // tile instead of 8 or 16 dword stores whose addresses each cost 64-bit arithmetic), the row's inverse scale is one LDS word per lane.
    constexpr bool TR = DSP == 2 && (DGNN_TR || DEC);
    // decoder stage buffers (DEC)
    char* const w0buf = reinterpret_cast<char*>(colp) + C::COLP_BYTES;                 // [2][8][2][64] x 16 B
    float* const dconst = reinterpret_cast<float*>(w0buf + C::DEC_W0);                 // A1[64] | B1[64] | W3[2][64] | b3[2]
    float* const dscale = dconst + C::DEC_CONST / 4;                                   // [8 slabs][32 rows]
    float* const pbuf = dscale + C::DEC_SCALE / 4;                                     // [32 rows][2]
    volatile int32_t* const pflag = reinterpret_cast<volatile int32_t*>(pbuf + 64);
    constexpr bool PLANAR = DSP == 2 && DGNN_PLANAR;
    // Skewed schedule (128 -> 128, fp16 dense form): the eight waves are two groups, A = the K-half-0 waves 0..3 and B = the K-half-1 waves
    // 4..7, one of each per SIMD, and at any time one group is in its filter phase P (vector ALU, LDS, gathers) while the other is in its dense
    // phase C (matrix cores, stores).  In the plain schedule both wavefronts of a SIMD are always in the SAME phase and queue for the same
    // unit -- the memory front end in front of the barrier, the matrix cores behind it -- and the phases' latencies have nobody to hide them.
    // Costs a second barrier per tile.  The tile's rows 0..15 come from group A, rows 16..31 from group B; C(t) of either group runs only
    // after both have delivered (two slots after P_A(t), one after P_B(t)).
    constexpr bool SKEW = DGNN_SKEW && DSP == 2 && KS == 2 && C::NSLICE == 4 && NW == 8;

    const int lane = lane_id(), w = wave_id_uniform();
#if DGNN_YOUNG_PRIO
    // the second wave of every SIMD loses VALU/MFMA arbitration to the older one on every phase (priority, then age);
    // one static priority bump evens the two out so neither idles long at the tile barrier
    if (w >= 4) __builtin_amdgcn_s_setprio(1);
#endif
    const int h = lane >> 5, l31 = lane & 31;
    const int jcol = lane & 15, tq = lane >> 4;  // filter phase: channel group / tet within a row block; also MFMA (col, k-group)
    const int ldx32 = (int)ldx;
    // row offset = row * ldx: a shift when the stride is a power of two (64 / 128 floats in layers 1..3; v_mul_lo_u32 runs at a quarter of the rate)
    const int ldx_sh = (DGNN_ROW_SHIFT && (ldx32 & (ldx32 - 1)) == 0) ? __builtin_ctz((unsigned)ldx32) : -1;
    constexpr bool vec = NB >= 4;  // the host side only takes rows that can be read as 16-byte pieces when NB >= 4 (xvec); NB == 2 reads 8 bytes
    (void)xvec;

    const int nwg = gridDim.x, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, wg_per_xcd = (nwg + 7 - xcd) >> 3;
    const int64_t per = (ntiles + 7) / 8;
    const int64_t t_lo = xcd * per, t_hi = min(ntiles, t_lo + per);
    int64_t my_n = 0;
    if (t_lo + slot < t_hi) my_n = (t_hi - t_lo - slot + wg_per_xcd - 1) / wg_per_xcd;
    auto tile_of = [&](int64_t it) { return t_lo + slot + it * wg_per_xcd; };

    // ---- fp16 forms: power-of-two scales of the two weight groups from their largest magnitudes (every workgroup reads both
    // matrices once, coalesced; they are L2-resident)
    float sW = 1.f, inv_sW = 1.f, sWe = 1.f, inv_sWe = 1.f;
    if constexpr (TR) {
        for (int c = threadIdx.x; c < COUT; c += blockDim.x) {
            colp[c] = bj ? bj[c] : 0.f;
            colp[COUT + c] = scale ? scale[c] : 1.f;
            colp[2 * COUT + c] = scale ? shift[c] : 0.f;
        }
    }
    const int prep_mode = (DSP == 2 && FSP == 2) ? dec.prep_mode : 0;
    char* const prepb = reinterpret_cast<char*>(dec.prep);
    if (prep_mode == 2) {
        const f32x4_t hd = *reinterpret_cast<const f32x4_t*>(prepb);
        sW = hd[0]; inv_sW = hd[1]; sWe = hd[2]; inv_sWe = hd[3];
    } else if constexpr (DSP == 2 || FSP == 2) {
        if (threadIdx.x < 2) scbuf[threadIdx.x] = 0u;
        __syncthreads();
        uint32_t mw = 0u, me = 0u;
        if constexpr (DSP == 2)
            for (int e = threadIdx.x; e < COUT * c_in; e += blockDim.x) mw = umax(mw, umax(absbits(Wj[e]), absbits(Wi[e])));
        if constexpr (FSP == 2) {
            for (int e = threadIdx.x; e < c_in * FE; e += blockDim.x) me = umax(me, absbits(We[e]));
            for (int e = threadIdx.x; e < c_in; e += blockDim.x) me = umax(me, absbits(be[e]));
        }
        mw = wave_umax(mw);
        me = wave_umax(me);
        if (lane == 0) {
            atomicMax(&scbuf[0], mw);
            atomicMax(&scbuf[1], me);
        }
        __syncthreads();
        if constexpr (DSP == 2) pow2_scales(scbuf[0], sW, inv_sW);
        if constexpr (FSP == 2) pow2_scales(scbuf[1], sWe, inv_sWe);
    }

    // ---- filter operand B = [We^T ; be ; 0] split in FSP parts -> LDS, once per launch.
    // entry (cb, g, j): channel c = NB*j + cb, k = 8g .. 8g+7 (g < 3; the k-group 3 of the MFMA is all zero)
    if (prep_mode == 2) {
        for (int i = threadIdx.x; i < C::BP_BYTES / 16; i += blockDim.x) reinterpret_cast<uint4*>(bpbuf)[i] = reinterpret_cast<const uint4*>(prepb + 16)[i];
    } else
    for (int e = threadIdx.x; e < NB * 48; e += blockDim.x) {
        const int cb = e / 48, gj = e - cb * 48, g = gj >> 4, j = gj & 15;
        const int c = NB * j + cb;
        uint32_t ph[4], pm[4], pl[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            float v[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int k = 8 * g + 2 * d + u;
                v[u] = 0.f;
                if (c < c_in) v[u] = k < FE ? We[(int64_t)c * FE + k] : (k == FE ? be[c] : 0.f);
            }
            if constexpr (FSP == 2) split2h(v[0] * sWe, v[1] * sWe, ph[d], pl[d]);
            else split3(v[0], v[1], ph[d], pm[d], pl[d]);
        }
        uint4* dst = reinterpret_cast<uint4*>(bpbuf + ((cb * FSP) * 48 + gj) * 16);
        dst[0] = make_uint4(ph[0], ph[1], ph[2], ph[3]);
        if constexpr (FSP == 3) dst[48] = make_uint4(pm[0], pm[1], pm[2], pm[3]);
        dst[48 * (FSP - 1)] = make_uint4(pl[0], pl[1], pl[2], pl[3]);
    }

    // ---- decoder stage (DEC): W0 scaled by one power of two and split into (hi, lo) fp16 fragments, once per launch.
    // Fragment (ob, sl, part, lane): output ob*32 + (lane & 31), k-group h = lane >> 5, element e = 0..7 <-> column
    // 32*(sl >> 1) + 16*(sl & 1) + 4*h + 8*(e >> 2) + (e & 3) -- the order in which a lane of the transposed epilogue holds its 8 finished
    // values of slab sl, so that a wave's finished rows ARE the other operand of v_mfma_f32_32x32x16_f16 (no transposition through LDS).
    if constexpr (DEC) {
        if (threadIdx.x == 0) scbuf[2] = 0u;
        if (threadIdx.x == 1) *pflag = 0;
        if (prep_mode == 2) {
            const uint4* src_ = reinterpret_cast<const uint4*>(prepb + 16 + C::BP_BYTES + C::PREP_WB);
            for (int i = threadIdx.x; i < (C::DEC_W0 + C::DEC_CONST) / 16; i += blockDim.x) reinterpret_cast<uint4*>(w0buf)[i] = src_[i];   // w0buf | dconst are adjacent
        } else {
        __syncthreads();
        uint32_t m0 = 0u;
        for (int e = threadIdx.x; e < 64 * 128; e += blockDim.x) m0 = umax(m0, absbits(dec.W0[e]));
        m0 = wave_umax(m0);
        if (lane_id() == 0) atomicMax(&scbuf[2], m0);
        __syncthreads();
        float sW0, inv_sW0;
        pow2_scales(scbuf[2], sW0, inv_sW0);
        for (int e = threadIdx.x; e < 2 * 8 * 64; e += blockDim.x) {
            const int ln = e & 63, sl = (e >> 6) & 7, ob = e >> 9;
            const int n_ = ob * 32 + (ln & 31), hh = ln >> 5, cb_ = 32 * (sl >> 1) + 16 * (sl & 1) + 4 * hh;
            uint32_t ph[4], pl[4];
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const int e0 = 2 * d, e1 = 2 * d + 1;
                const float v0 = dec.W0[n_ * 128 + cb_ + 8 * (e0 >> 2) + (e0 & 3)], v1 = dec.W0[n_ * 128 + cb_ + 8 * (e1 >> 2) + (e1 & 3)];
                split2h(v0 * sW0, v1 * sW0, ph[d], pl[d]);
            }
            uint4* dst = reinterpret_cast<uint4*>(w0buf + ((ob * 8 + sl) * 2) * 1024 + ln * 16);
            dst[0] = make_uint4(ph[0], ph[1], ph[2], ph[3]);
            dst[64] = make_uint4(pl[0], pl[1], pl[2], pl[3]);
        }
        for (int n_ = threadIdx.x; n_ < 64; n_ += blockDim.x) {
            const float s1 = dec.scale1 ? dec.scale1[n_] : 1.f, h1 = dec.scale1 ? dec.shift1[n_] : 0.f;
            dconst[n_] = inv_sW0 * s1;                              // h1 = relu(acc * A1 + B1), acc in units of 1 / sW0
            dconst[64 + n_] = __fmaf_rn(dec.b0[n_], s1, h1);
            dconst[128 + n_] = dec.W3[n_];
            dconst[192 + n_] = dec.W3[64 + n_];
        }
        if (threadIdx.x < 2) dconst[256 + threadIdx.x] = dec.b3[threadIdx.x];
        }
    }

    // ---- dense-phase role: (column slice cs, K part kh, row group rg); this wave's share of K resident as 3 bf16 parts
    const int cs = w % C::NSLICE, kh = KS == 2 ? (w / C::NSLICE) & 1 : 0, rg = w / (KS * C::NSLICE);
    const int col = cs * 32 + l31;
    const int partner = w ^ C::NSLICE;
    bf16x8 wb[NWB][DSP];  // DSP == 2: fp16 bit patterns of (hi, lo) of W * sW
    if (prep_mode == 2) {
        const uint4* src_ = reinterpret_cast<const uint4*>(prepb + 16 + C::BP_BYTES) + (int64_t)w * NWB * DSP * 64 + lane;
#pragma unroll
        for (int S = 0; S < NWB; ++S)
#pragma unroll
            for (int pp = 0; pp < DSP; ++pp) wb[S][pp] = __builtin_bit_cast(bf16x8, src_[(S * DSP + pp) * 64]);
    } else {
#pragma unroll
        for (int S = 0; S < NWB; ++S) {
            // k-steps 0 .. CIN_PAD/16-1 of the A row are the mean half (Wj), the rest the own-row half (Wi)
            const bool second = KS == 2 ? kh != 0 : S >= CIN_PAD / 16;
            const float* Wsrc = second ? Wi : Wj;
            const int S_ = (KS == 1 && S >= CIN_PAD / 16) ? S - CIN_PAD / 16 : S;
            uint32_t ph[4], pm[4], pl[4];
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const int k = 16 * S_ + 8 * h + 2 * d;
                const float v0 = Wsrc[(int64_t)col * c_in + (k < c_in ? k : 0)];
                const float v1 = Wsrc[(int64_t)col * c_in + (k + 1 < c_in ? k + 1 : 0)];
                if constexpr (DSP == 2) split2h(k < c_in ? v0 * sW : 0.f, k + 1 < c_in ? v1 * sW : 0.f, ph[d], pl[d]);
                else split3(k < c_in ? v0 : 0.f, k + 1 < c_in ? v1 : 0.f, ph[d], pm[d], pl[d]);
            }
            wb[S][0] = pack8(ph);
            if constexpr (DSP == 3) wb[S][1] = pack8(pm);
            wb[S][DSP - 1] = pack8(pl);
        }
    }
    const float bb = bj ? bj[col] : 0.f;
    const float sc = scale ? scale[col] : 1.f;
    const float sh = scale ? shift[col] : 0.f;
    const bool has_scale = scale != nullptr;
    __syncthreads();  // filter operand parts visible
    if (prep_mode == 1) {   // dgnn_sage_layer_prepare: one workgroup, no tiles -- park what the prologue made and leave
        if (threadIdx.x == 0) *reinterpret_cast<f32x4_t*>(prepb) = f32x4_t{sW, inv_sW, sWe, inv_sWe};
        for (int i = threadIdx.x; i < C::BP_BYTES / 16; i += blockDim.x) reinterpret_cast<uint4*>(prepb + 16)[i] = reinterpret_cast<const uint4*>(bpbuf)[i];
        uint4* dstw = reinterpret_cast<uint4*>(prepb + 16 + C::BP_BYTES) + (int64_t)w * NWB * DSP * 64 + lane;
#pragma unroll
        for (int S = 0; S < NWB; ++S)
#pragma unroll
            for (int pp = 0; pp < DSP; ++pp) dstw[(S * DSP + pp) * 64] = __builtin_bit_cast(uint4, wb[S][pp]);
        if constexpr (DEC) {
            uint4* dstd = reinterpret_cast<uint4*>(prepb + 16 + C::BP_BYTES + C::PREP_WB);
            for (int i = threadIdx.x; i < (C::DEC_W0 + C::DEC_CONST) / 16; i += blockDim.x) dstd[i] = reinterpret_cast<const uint4*>(w0buf)[i];
        }
        return;
    }

    // ---- filter operand parts of this lane kept in registers where they fit (NB <= 4: 8 registers per channel block): no LDS reads
    // inside the channel-block loop
    constexpr bool BREG = DGNN_FILTER_BREG && FSP == 2 && NB <= 4;
    uint4 breg[BREG ? NB : 1][2];
    if constexpr (BREG) {
#pragma unroll
        for (int cb = 0; cb < NB; ++cb) {
            const char* bp = bpbuf + ((cb * FSP) * 48 + (tq < 3 ? tq : 0) * 16 + jcol) * 16;
            breg[cb][0] = *reinterpret_cast<const uint4*>(bp);
            breg[cb][1] = *reinterpret_cast<const uint4*>(bp + 768);
        }
    }

    // ---- filter-phase role
    const int c0 = NB * jcol;            // this lane's NB contiguous channels
    const bool on = c0 < c_in;           // c_in is a multiple of NB (host-checked)
    const int c0l = on ? c0 : 0;
    float* const myea = reinterpret_cast<float*>(eabuf + w * C::EA_BYTES);

    float xd[RB][NB], xr[RB][4][NB];
    bool regular = false;
    // Index pipeline, three tiles deep.  When P(it) starts (after its vmcnt(0)):  S1 = tile it+1: row starts, sources and edge ids have
    // landed, so its gathers can be issued at any point of P(it);  S2 = tile it+2: row starts landed, sources / edge ids are requested
    // now;  S3 = tile it+3: row starts requested now.
    int vbeg1 = 0, vbeg2 = 0, vbeg3 = 0, vsrc1 = 0, vsrc2 = 0, veid1 = 0, veid2 = 0;
    bool ok1 = false, ok2 = false, ok3 = false;
    // nv = valid tets of this wave's group (TPW except in the last tile).  A short group still takes the matrix-core
    // path with clamped (duplicated) rows so that a tet's result does not depend on where the tile grid cuts the
    // graph: whole-graph and partitioned runs stay bit-identical on 4-regular scenes.
    int nv1 = 0, nv2 = 0, nv3 = 0;

    auto load_rowptr = [&](int64_t it, int& vb, int& nv) -> bool {
        if (it >= my_n) return false;
        const int64_t i0 = tile_of(it) * TILE + w * TPW;
        if (i0 >= n_dst) return false;
        nv = (int)(n_dst - i0 < TPW ? n_dst - i0 : TPW);
        vb = rowptr[i0 + (lane < nv ? lane : nv)];
        return true;
    };
    auto load_src = [&](bool& ok, int nv, int vbeg, int& vsrc, int& veid) {
        if (ok) {
            const int b0 = __builtin_amdgcn_readfirstlane(vbeg);
            ok = __all(vbeg == b0 + 4 * (lane < nv ? lane : nv)) != 0;
            if (ok) {
                vsrc = src[b0 + (lane < 4 * nv ? lane : 4 * nv - 1)];
                if (eid) veid = eid[b0 + (lane < 4 * nv ? lane : 4 * nv - 1)];
            }
        }
    };
    // Gathers of tile `it` (S1), in pieces so that they can be issued between the pieces of P(it-1)'s arithmetic instead of as one
    // burst in front of the barrier (the CU's vector-memory front end takes ~0.6 us per tile to accept them: 92 KB at 64 B/clk).
    //   issue_x(it, rb, part): own row + the 4 neighbour rows of row block rb (part: see ld_vec)
    //   issue_ea(it): LDS-DMA of the attribute block into the private strip -- the strip's reads of the current tile must have returned
    int sidx[RB][4];
    auto issue_x = [&](int64_t it, int rb, auto part_c) {
        constexpr int part = decltype(part_c)::value;
        if (!regular) return;
        const int i0 = (int)(tile_of(it) * TILE) + w * TPW;
        const int tl = rb * 4 + tq;  // this lane's tet within the wave
        const int own = i0 + (tl < nv1 ? tl : nv1 - 1);
        // all index shuffles first, then every row offset under ONE wave-uniform choice of shift / multiply, then the loads back to back
        // (a per-row choice compiles to a branch and an LDS wait in front of every pair of loads)
        constexpr bool BATCH = !(DSP == 3 && CIN_PAD == 128);  // the bf16 x 3 form at 128 -> 128 has no registers for five offsets at once (2 spills)
        auto off_of = [&](int row) -> uint32_t { return ldx_sh >= 0 ? (uint32_t)row << ldx_sh : (uint32_t)(row * ldx32); };
        if constexpr (BATCH) {
            if (part != 1) {
#pragma unroll
                for (int r = 0; r < 4; ++r) sidx[rb][r] = (DGNN_WHATIF & 8) ? own : __shfl(vsrc1, tl * 4 + r);
            }
            uint32_t off[5];
            if (ldx_sh >= 0) {
                off[4] = (uint32_t)own << ldx_sh;
#pragma unroll
                for (int r = 0; r < 4; ++r) off[r] = (uint32_t)sidx[rb][r] << ldx_sh;
            } else {
                off[4] = (uint32_t)(own * ldx32);
#pragma unroll
                for (int r = 0; r < 4; ++r) off[r] = (uint32_t)(sidx[rb][r] * ldx32);
            }
            ld_vec<NB, part>(xd[rb], xdst + off[4] + c0l, vec);
#pragma unroll
            for (int r = 0; r < 4; ++r) ld_vec<NB, part>(xr[rb][r], x + off[r] + c0l, vec);
        } else {
            ld_vec<NB, part>(xd[rb], xdst + off_of(own) + c0l, vec);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (part != 1) sidx[rb][r] = (DGNN_WHATIF & 8) ? own : __shfl(vsrc1, tl * 4 + r);
                ld_vec<NB, part>(xr[rb][r], x + off_of(sidx[rb][r]) + c0l, vec);
            }
        }
    };
    auto issue_ea = [&](int64_t it) {
        if (!regular || (DGNN_WHATIF & 16)) return;
        // eid == nullptr: the rows are in plan order, one contiguous block.  Otherwise every 80-byte row is
        // fetched from its place in the caller's edge_attr (row eid[k]) -- no staging copy of the edge features.
        if (eid) {
            // float index within the strip -> (edge of the wave, offset in its row); the edge ids are shuffled in one batch
            auto row_ptr = [&](int fi) -> const float* {
                const int e = (fi * 0xCCD) >> 16;  // fi / 20 for fi < 8192
                return ea + (int64_t)__shfl(veid1, e) * FE + (fi - e * FE);
            };
            if constexpr (DSP == 3 && CIN_PAD == 128) {  // no registers to hold all addresses at once
#pragma unroll
                for (int q = 0; q < C::EA_FULL; ++q) glds16_s(row_ptr(q * 256 + lane * 4), myea + q * 256);
#pragma unroll
                for (int q = 0; q < C::EA_TAIL; ++q) glds4_s(row_ptr(C::EA_FULL * 256 + q * 64 + lane), myea + C::EA_FULL * 256 + q * 64);
            } else {
                const float* gp[C::EA_FULL + C::EA_TAIL];
#pragma unroll
                for (int q = 0; q < C::EA_FULL + C::EA_TAIL; ++q)
                    gp[q] = row_ptr(q < C::EA_FULL ? q * 256 + lane * 4 : C::EA_FULL * 256 + (q - C::EA_FULL) * 64 + lane);
#pragma unroll
                for (int q = 0; q < C::EA_FULL; ++q) glds16_s(gp[q], myea + q * 256);
#pragma unroll
                for (int q = 0; q < C::EA_TAIL; ++q) glds4_s(gp[C::EA_FULL + q], myea + C::EA_FULL * 256 + q * 64);
            }
        } else {
            const float* eab = ea + (int64_t)__builtin_amdgcn_readfirstlane(vbeg1) * lde;
            const int ea_last = nv1 * 4 * FE - 4;  // last 16-byte chunk of the group's attribute block
#pragma unroll
            for (int q = 0; q < C::EA_FULL; ++q) glds16_s(eab + min(q * 256 + lane * 4, ea_last), myea + q * 256);
#pragma unroll
            for (int q = 0; q < C::EA_TAIL; ++q)
                glds4_s(eab + min(C::EA_FULL * 256 + q * 64 + lane, ea_last + 3), myea + C::EA_FULL * 256 + q * 64);
        }
    };
    auto issue_loads = [&](int64_t it) {  // everything at once (prologue, and after a group that took the generic path)
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) issue_x(it, rb, IC<-1>{});
        issue_ea(it);
    };
    // start of P(it), everything requested so far has landed: the stages move up, then S2's sources and S3's row starts are requested
    auto advance_idx = [&](int64_t it) {
        ok1 = ok2; nv1 = nv2; vbeg1 = vbeg2; vsrc1 = vsrc2; veid1 = veid2;
        ok2 = ok3; nv2 = nv3; vbeg2 = vbeg3;
        load_src(ok2, nv2, vbeg2, vsrc2, veid2);
        ok3 = load_rowptr(it + 3, vbeg3, nv3);
    };
    // one finished (tet row, NB channels) segment -> A-tile: columns [c0, c0+NB) of the mean half and of the own-row half
    // av = araw * fpre, fpre a power of two (the 1/4 of the mean times the inverse scales of the filter product; 1 on the generic path)
    auto put_seg = [&](int64_t it, int row, const float (&araw)[NB], const float (&xv)[NB], float fpre) {
        const int buf = (int)(it & 1);
        // PLANAR (fp16 form): a row is [hi of all K | lo of all K], so the 16 lanes of a tet write 16 x NB*2 contiguous bytes per part
        // (octet-interleaved [hi|lo] pieces put lanes j and j+8 on the same banks: 2-way conflicts on every A-tile write)
        char* dst = abuf + buf * C::A_BYTES + row * ROWB + (PLANAR ? c0 * 2 : (c0 >> 3) * OCT + (c0 & 7) * 2);
        char* dsx = dst + (PLANAR ? CIN_PAD * 2 : (CIN_PAD / 8) * OCT);
        uint32_t ph[NB / 2], pm[NB / 2], pl[NB / 2], qh[NB / 2], qm[NB / 2], ql[NB / 2];
        if constexpr (DSP == 2) {
            // row scale: the 16 lanes of this tet hold the whole [mean | own] row between them
            float ma = 0.f, mx = 0.f;  // v_max3_f32 with |.| source modifiers: one instruction per two values
#pragma unroll
            for (int i = 0; i < NB; i += 2) {
                ma = fmaxf(fmaxf(ma, fabsf(araw[i])), fabsf(araw[i + 1]));
                mx = fmaxf(fmaxf(mx, fabsf(xv[i])), fabsf(xv[i + 1]));
            }
            const uint32_t m = row16_umax(__builtin_bit_cast(uint32_t, fmaxf(ma * fpre, mx)));
            float s_, inv_;
            pow2_scales(m, s_, inv_);
            if (jcol == 0) rowf[(int)(it & 3) * TILE + row] = inv_ * inv_sW;
            const float sa = fpre * s_;  // powers of two: the scaling of the mean and of the row in one exact multiplication
#pragma unroll
            for (int d = 0; d < NB / 2; ++d) {
                split2h(araw[2 * d] * sa, araw[2 * d + 1] * sa, ph[d], pl[d]);
                split2h(xv[2 * d] * s_, xv[2 * d + 1] * s_, qh[d], ql[d]);
            }
        } else {
            float av[NB];
#pragma unroll
            for (int i = 0; i < NB; ++i) av[i] = araw[i] * fpre;
#pragma unroll
            for (int d = 0; d < NB / 2; ++d) {
                split3(av[2 * d], av[2 * d + 1], ph[d], pm[d], pl[d]);
                split3(xv[2 * d], xv[2 * d + 1], qh[d], qm[d], ql[d]);
            }
        }
        constexpr int LO = PLANAR ? C::K * 2 : 16 * (DSP - 1);
        if (NB == 8) {
            *reinterpret_cast<uint4*>(dst) = make_uint4(ph[0], ph[1], ph[2 % (NB / 2)], ph[3 % (NB / 2)]);
            if constexpr (DSP == 3) *reinterpret_cast<uint4*>(dst + 16) = make_uint4(pm[0], pm[1], pm[2 % (NB / 2)], pm[3 % (NB / 2)]);
            *reinterpret_cast<uint4*>(dst + LO) = make_uint4(pl[0], pl[1], pl[2 % (NB / 2)], pl[3 % (NB / 2)]);
            *reinterpret_cast<uint4*>(dsx) = make_uint4(qh[0], qh[1], qh[2 % (NB / 2)], qh[3 % (NB / 2)]);
            if constexpr (DSP == 3) *reinterpret_cast<uint4*>(dsx + 16) = make_uint4(qm[0], qm[1], qm[2 % (NB / 2)], qm[3 % (NB / 2)]);
            *reinterpret_cast<uint4*>(dsx + LO) = make_uint4(ql[0], ql[1], ql[2 % (NB / 2)], ql[3 % (NB / 2)]);
        } else if (NB == 4) {
            *reinterpret_cast<uint2*>(dst) = make_uint2(ph[0], ph[1 % (NB / 2)]);
            if constexpr (DSP == 3) *reinterpret_cast<uint2*>(dst + 16) = make_uint2(pm[0], pm[1 % (NB / 2)]);
            *reinterpret_cast<uint2*>(dst + LO) = make_uint2(pl[0], pl[1 % (NB / 2)]);
            *reinterpret_cast<uint2*>(dsx) = make_uint2(qh[0], qh[1 % (NB / 2)]);
            if constexpr (DSP == 3) *reinterpret_cast<uint2*>(dsx + 16) = make_uint2(qm[0], qm[1 % (NB / 2)]);
            *reinterpret_cast<uint2*>(dsx + LO) = make_uint2(ql[0], ql[1 % (NB / 2)]);
        } else {
            *reinterpret_cast<uint32_t*>(dst) = ph[0];
            if constexpr (DSP == 3) *reinterpret_cast<uint32_t*>(dst + 16) = pm[0];
            *reinterpret_cast<uint32_t*>(dst + LO) = pl[0];
            *reinterpret_cast<uint32_t*>(dsx) = qh[0];
            if constexpr (DSP == 3) *reinterpret_cast<uint32_t*>(dsx + 16) = qm[0];
            *reinterpret_cast<uint32_t*>(dsx + LO) = ql[0];
        }
    };

    ok1 = load_rowptr(0, vbeg1, nv1);
    ok2 = load_rowptr(1, vbeg2, nv2);
    ok3 = load_rowptr(2, vbeg3, nv3);
    load_src(ok1, nv1, vbeg1, vsrc1, veid1);
    regular = ok1;
    issue_loads(0);
    load_src(ok2, nv2, vbeg2, vsrc2, veid2);  // state as at the end of a P phase: S1 = tile 0 (gathers issued), S2 = tile 1, S3 = tile 2
    float mine[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) mine[r] = 0.f;

    // ---- the three pieces of a tile
    // P(it): filter product on the matrix cores, mean, the wave's rows of A-tile `it`; requests the gathers of tile it+1
    auto phaseP = [&](int64_t it) {
        // ================================================================ P: filter on the matrix cores + mean
        const int64_t i0 = tile_of(it) * TILE + w * TPW;
        stamp(trace, trace_cap, it, w, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // rows + LDS-DMA'd strip of this tile, index loads of the tiles behind it
        // the index loads are consumed HERE (the compiler would otherwise place its own, conservative, waits at their later uses)
        asm volatile("" : "+v"(vbeg2), "+v"(vbeg3), "+v"(vsrc2), "+v"(veid2));
        const bool was_regular = regular;
        advance_idx(it);
        regular = ok1;  // tile it+1: its gathers are issued during this P
        stamp(trace, trace_cap, it, w, 1);
        if (was_regular) {
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                // A operand: lane (edge i = lane&15, k-group g = lane>>4) holds attributes 8g..8g+7 of its edge;
                // k = 20 is the constant 1 multiplying the bias row, everything beyond is 0
                const float* er = myea + (rb * 16 + jcol) * FE;
                const f32x4_t q0 = *reinterpret_cast<const f32x4_t*>(er + 8 * (tq < 2 ? tq : 2));
                const f32x4_t q1 = *reinterpret_cast<const f32x4_t*>(er + 8 * (tq < 1 ? tq : 1) + 4);
                float av[8];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    av[i] = tq < 3 ? q0[i] : 0.f;
                    av[4 + i] = tq < 2 ? q1[i] : 0.f;
                }
                if (tq == 2) av[4] = 1.0f;
                if (EARLY && rb == RB - 1) {
                    // the strip has been read for the last time: the next tile's attribute block may land in it
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    issue_ea(it + 1);
                }
                uint32_t ph[4], pm[4], pl[4];
                float fmean = 0.25f;  // 1/4 (regular group) times the inverse scale of the filter weights
                float inv_e[4];       // FSP == 2: inverse scales of the 4 in-edges of this lane's tet
                if constexpr (FSP == 2) {
                    // one scale per EDGE (a row of the A operand; the constant 1 of the bias column is part of it): an edge's phi is as accurate
                    // as its own attributes allow whatever the other 15 edges of the block look like.  The row's 4 k-group lanes are 16 apart.
                    float mf = 0.f;
#pragma unroll
                    for (int i = 0; i < 8; i += 2) mf = fmaxf(fmaxf(mf, fabsf(av[i])), fabsf(av[i + 1]));
                    float sA, inv_sA;
                    pow2_scales(cross_row_umax(__builtin_bit_cast(uint32_t, mf)), sA, inv_sA);
                    fmean = 0.25f * inv_sWe;
#pragma unroll
                    for (int d = 0; d < 4; ++d) split2h(av[2 * d] * sA, av[2 * d + 1] * sA, ph[d], pl[d]);
                    // lane (tet tq, channel group): edge 4 tq + r is row 4 tq + r of the operand, whose scale lanes 4 tq + r (+16, +32, +48) hold
#pragma unroll
                    for (int r = 0; r < 4; ++r) inv_e[r] = __shfl(inv_sA, 4 * tq + r);
                } else {
#pragma unroll
                    for (int d = 0; d < 4; ++d) split3(av[2 * d], av[2 * d + 1], ph[d], pm[d], pl[d]);
                }
                const bf16x8 ah = pack8(ph), am = pack8(FSP == 3 ? pm : ph), al = pack8(pl);

                float aout[NB], xv[NB];
#pragma unroll
                for (int cb = 0; cb < NB; ++cb) {
                    const char* bp = bpbuf + ((cb * FSP) * 48 + (tq < 3 ? tq : 0) * 16 + jcol) * 16;
                    // k-group 3 re-reads group 0: its A operand is all zero, so any finite B will do (saves 12 selects)
                    uint4 u0, u1, u2;
                    if constexpr (BREG) {
                        u0 = breg[cb][0];
                        u1 = u2 = breg[cb][1];
                    } else {
                        u0 = *reinterpret_cast<const uint4*>(bp), u1 = *reinterpret_cast<const uint4*>(bp + 768),
                        u2 = *reinterpret_cast<const uint4*>(bp + 768 * (FSP - 1));
                    }
                    const bf16x8 bh = __builtin_bit_cast(bf16x8, u0), bm = __builtin_bit_cast(bf16x8, u1),
                                 bl = __builtin_bit_cast(bf16x8, u2);
                    f32x4_t d = {0.f, 0.f, 0.f, 0.f};
                    if (DGNN_WHATIF & 2) {
                        d[0] = __builtin_bit_cast(float, u0.x) + __builtin_bit_cast(float, ph[0]);
                        d[1] = __builtin_bit_cast(float, u2.y);
                        d[2] = __builtin_bit_cast(float, u0.z);
                        d[3] = __builtin_bit_cast(float, pl[1]);
                    } else if constexpr (FSP == 2) {
                        d = __builtin_amdgcn_mfma_f32_16x16x32_f16(H8(al), H8(bh), d, 0, 0, 0);
                        d = __builtin_amdgcn_mfma_f32_16x16x32_f16(H8(ah), H8(bl), d, 0, 0, 0);
                        d = __builtin_amdgcn_mfma_f32_16x16x32_f16(H8(ah), H8(bh), d, 0, 0, 0);
                    } else {
                        d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, d, 0, 0, 0);
                        d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, d, 0, 0, 0);
                        d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm, d, 0, 0, 0);
                        d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, d, 0, 0, 0);
                        d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, d, 0, 0, 0);
                        d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, d, 0, 0, 0);
                    }
                    // d[r] = phi of edge 4*tq + r (the r-th in-edge of this lane's tet), channel c0 + cb
                    // in-order sum over the tet's 4 in-edges.  Lanes past c_in need no masking: their filter
                    // operand and their rows of Wj|Wi are zero, and what they loaded (channel 0..) is finite.
                    if constexpr (FSP == 2) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) d[r] *= inv_e[r];  // exact: powers of two
                    }
                    float a = __fmul_rn(xr[rb][0][cb], d[0]);
#pragma unroll
                    for (int r = 1; r < 4; ++r) a = __fmaf_rn(xr[rb][r][cb], d[r], a);
                    aout[cb] = a;
                    xv[cb] = xd[rb][cb];
                    // the registers of this row block are free again: request its rows of the next tile right here, between
                    // the arithmetic (NB == 8: in two halves, the first as soon as channels 0..3 are through)
                    if (EARLY && RB == 1 && NB == 8 && cb == 3) issue_x(it + 1, rb, IC<0>{});
                }
                if (EARLY) issue_x(it + 1, rb, IC<(RB == 1 && NB == 8) ? 1 : -1>{});
                if constexpr (FSP == 2 && DGNN_FILTER_PIPE) {
                    // instruction order of the channel-block loop: operand reads two blocks ahead of their products, the x.phi sums of a
                    // block behind the next block's products (left alone, every read is waited for right where it is issued)
                    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
                    for (int cb = 0; cb < NB; ++cb) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
                        if (cb + 2 < NB) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                        if (cb > 0) __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                }
                put_seg(it, w * TPW + rb * 4 + tq, aout, xv, fmean);
            }
        } else {
            // generic path (a group with any in-degree other than 4, or past the end): plain fp32 per lane, one edge at a time (rare)
#pragma unroll 1
            for (int rb = 0; rb < RB; ++rb) {
                const int64_t i = i0 + rb * 4 + tq;
                float aout[NB], xv[NB];
#pragma unroll
                for (int cb = 0; cb < NB; ++cb) aout[cb] = xv[cb] = 0.f;
                if (i < n_dst && on) {
                    const int b = rowptr[i], e_end = rowptr[i + 1];
#pragma unroll
                    for (int cb = 0; cb < NB; ++cb) xv[cb] = xdst[i * ldx + c0 + cb];
                    for (int k = b; k < e_end; ++k) {
                        const int s_ = src[k];
                        const float* ar = ea + (int64_t)(eid ? eid[k] : k) * lde;
#pragma unroll 1
                        for (int cb = 0; cb < NB; ++cb) {
                            float p = be[c0 + cb];
                            for (int f = 0; f < FE; ++f) p = __fmaf_rn(We[(int64_t)(c0 + cb) * FE + f], ar[f], p);
                            aout[cb] = __fadd_rn(aout[cb], __fmul_rn(x[(int64_t)s_ * ldx + c0 + cb], p));
                        }
                    }
                    const float cnt = (float)max(e_end - b, 1);
#pragma unroll
                    for (int cb = 0; cb < NB; ++cb) aout[cb] = __fdiv_rn(aout[cb], cnt);
                }
                put_seg(it, w * TPW + rb * 4 + tq, aout, xv, 1.f);
            }
        }
        stamp(trace, trace_cap, it, w, 2);
        if (!was_regular || !EARLY) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // strip reads returned before the next DMA may land
            issue_loads(it + 1);
        }
        stamp(trace, trace_cap, it, w, 3);
    };
    // C(it): this wave's block of the dense product over its share of K
    auto phaseC = [&](int64_t it, f32x16& acc) {
        // ================================================================ C: dense part, split-bf16, K half
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        const char* A = abuf + (it & 1) * C::A_BYTES + (rg * 32 + l31) * ROWB + (PLANAR ? (kh * CIN_PAD + 8 * h) * 2 : (kh * (CIN_PAD / 8) + h) * OCT);
        constexpr int SSTEP = PLANAR ? 32 : 2 * OCT, LOFF = PLANAR ? C::K * 2 : 16 * (DSP - 1);
#pragma unroll
        for (int S = 0; S < ((DGNN_WHATIF & 1) ? 1 : NWB); ++S) {
            const bf16x8 ah = *reinterpret_cast<const bf16x8*>(A + S * SSTEP);
            const bf16x8 al = *reinterpret_cast<const bf16x8*>(A + S * SSTEP + LOFF);
            if constexpr (DSP == 2) {
                // first product: C = inline constant 0 (no 16 register moves to clear the accumulator)
                if constexpr (TR) {  // weights as the A operand: the accumulator holds the block transposed
                    if (S == 0) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(H8(wb[S][0]), H8(al), f32x16{}, 0, 0, 0);
                    else acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(H8(wb[S][0]), H8(al), acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(H8(wb[S][1]), H8(ah), acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(H8(wb[S][0]), H8(ah), acc, 0, 0, 0);
                } else {
                    if (S == 0) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(H8(al), H8(wb[S][0]), f32x16{}, 0, 0, 0);
                    else acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(H8(al), H8(wb[S][0]), acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(H8(ah), H8(wb[S][1]), acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(H8(ah), H8(wb[S][0]), acc, 0, 0, 0);
                }
            } else {
                const bf16x8 am = *reinterpret_cast<const bf16x8*>(A + S * 2 * OCT + 16);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, wb[S][0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, wb[S][DSP - 1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, wb[S][1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, wb[S][0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, wb[S][1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, wb[S][0], acc, 0, 0, 0);
            }
        }
        if constexpr (DSP == 2 && DGNN_DENSE_PREFETCH > 0) {
            // instruction order of the block above: the A fragments of DGNN_DENSE_PREFETCH k-steps are requested ahead of the products
            // that use them (left alone, the scheduler requests each pair right in front of its use and every k-step pays an LDS round trip)
            __builtin_amdgcn_sched_group_barrier(0x100, 2 * DGNN_DENSE_PREFETCH, 0);
#pragma unroll
            for (int S = 0; S < NWB; ++S) {
                __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
                if (S + DGNN_DENSE_PREFETCH < NWB) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            }
        }
    };
    // K split between wave pairs: the sums of tile `ie` are finished from this wave's half (`mine`) and the partner's (LDS)
    auto epilogueK2 = [&](int64_t ie) {
        // ============================================================ delayed epilogue of tile it-1
        const int64_t tile = tile_of(ie);
        const float* red = redbuf + (ie & 1) * (C::RED_BYTES / 4) + partner * 512 + lane;
        if constexpr (TR) {
            // this wave finishes columns cs*32 + 16*kh + 4h + 8g + c (g < 2, c < 4) of tet row rg*32 + l31
            const int row = rg * 32 + l31, cbase = cs * 32 + 16 * kh + 4 * h;
            const int64_t grow = tile * TILE + row;
            const float rf = rowf[(int)(ie & 3) * TILE + row];
            float* o = out + grow * ldo + cbase;
            float v8[8];
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const f32x4_t b4 = *reinterpret_cast<const f32x4_t*>(colp + cbase + 8 * g);
                f32x4_t v;
#pragma unroll
                for (int c = 0; c < 4; ++c) v[c] = __fmaf_rn(mine[4 * g + c] + red[(4 * g + c) * 64], rf, b4[c]);
                if (has_scale) {
                    const f32x4_t s4 = *reinterpret_cast<const f32x4_t*>(colp + COUT + cbase + 8 * g);
                    const f32x4_t h4 = *reinterpret_cast<const f32x4_t*>(colp + 2 * COUT + cbase + 8 * g);
#pragma unroll
                    for (int c = 0; c < 4; ++c) v[c] = __fmaf_rn(v[c], s4[c], h4[c]);
                }
                if (relu) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) v[c] = fmaxf(v[c], 0.f);
                }
                if constexpr (DEC) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) v8[4 * g + c] = v[c];
                } else {
                    if (grow < n_dst) *reinterpret_cast<f32x4_t*>(o + 8 * g) = v;
                }
            }
            if constexpr (DEC) {
                // The finished rows do not leave the CU: this lane's 8 values of (tet row l31, slab 2*cs + kh) become one fragment of the
                // decoder's first product -- scaled by a power of two of their own (the slab's 16 values of this row live in lanes l31 and
                // l31 + 32), split into (hi, lo) and parked in the partner's region of the partial-sum buffer, which this wave alone has
                // just read.  The slab's inverse scale goes to dscale.
                float mx = 0.f;
#pragma unroll
                for (int i = 0; i < 8; i += 2) mx = fmaxf(fmaxf(mx, fabsf(v8[i])), fabsf(v8[i + 1]));
                uint32_t mb = __builtin_bit_cast(uint32_t, mx);
                const auto sw = __builtin_amdgcn_permlane32_swap(mb, mb, false, false);
                mb = umax(sw[0], sw[1]);
                float s_, inv_;
                pow2_scales(mb, s_, inv_);
                if (h == 0) dscale[(2 * cs + kh) * 32 + l31] = inv_;
                uint32_t ph[4], pl[4];
#pragma unroll
                for (int d = 0; d < 4; ++d) split2h(v8[2 * d] * s_, v8[2 * d + 1] * s_, ph[d], pl[d]);
                uint4* dst = reinterpret_cast<uint4*>(redbuf + (ie & 1) * (C::RED_BYTES / 4) + partner * 512) + lane;
                dst[0] = make_uint4(ph[0], ph[1], ph[2], ph[3]);
                dst[64] = make_uint4(pl[0], pl[1], pl[2], pl[3]);
            }
        } else {
        const int64_t row0 = tile * TILE + rg * 32 + 4 * h + 16 * kh;
        float* o = out + row0 * ldo + col;
        float v[8], rf[8];
        if constexpr (DSP == 2) {
            const float* rfp = rowf + (int)(ie & 3) * TILE + rg * 32 + 4 * h + 16 * kh;
            const f32x4_t r0 = *reinterpret_cast<const f32x4_t*>(rfp), r1 = *reinterpret_cast<const f32x4_t*>(rfp + 8);
#pragma unroll
            for (int r = 0; r < 4; ++r) { rf[r] = r0[r]; rf[4 + r] = r1[r]; }
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if constexpr (DSP == 2) v[r] = __fmaf_rn(mine[r] + red[r * 64], rf[r], bb);
            else v[r] = (mine[r] + red[r * 64]) + bb;
            if (has_scale) v[r] = __fmaf_rn(v[r], sc, sh);
            if (relu) v[r] = fmaxf(v[r], 0.f);
        }
        if (DGNN_WHATIF & 4) {
            float t = 0.f;
#pragma unroll
            for (int r = 0; r < 8; ++r) t += v[r];
            if (t == 123.456f) o[0] = t;
        } else if ((tile + 1) * TILE <= n_dst) {
#pragma unroll
            for (int r = 0; r < 8; ++r) st_out(&o[(int64_t)((r & 3) + 8 * (r >> 2)) * ldo], v[r]);
        } else {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int rr = (r & 3) + 8 * (r >> 2);
                if (row0 + rr < n_dst) st_out(&o[(int64_t)rr * ldo], v[r]);
            }
        }
        }
    };
    // D(ie) (DEC; waves 0 and 1, one per output block of 32): h1 = relu(BN(W0 . y + b0)) for the tile's 32 rows, then this block's share of
    // W3 . h1.  Per slab: 3 products on the matrix cores from zero, times the slab's inverse scale into the fp32 sum (slabs of one row carry
    // different powers of two).  Wave 1 hands its partial logits to wave 0 through LDS; wave 0 adds, adds b3 and stores 8 bytes per tet.
    auto decoderD = [&](int64_t ie) {
        if constexpr (DEC) {
            const int ob = w;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            const char* fb = reinterpret_cast<const char*>(redbuf + (ie & 1) * (C::RED_BYTES / 4));
#pragma unroll
            for (int sl = 0; sl < 8; ++sl) {
                const int wp = (sl >> 1) + C::NSLICE * (sl & 1);          // the wave that finished slab sl; its fragments sit in its partner's region
                const char* fr = fb + (wp ^ C::NSLICE) * 2048 + lane * 16;
                const bf16x8 bh = *reinterpret_cast<const bf16x8*>(fr), bl = *reinterpret_cast<const bf16x8*>(fr + 1024);
                const char* wr = w0buf + ((ob * 8 + sl) * 2) * 1024 + lane * 16;
                const bf16x8 ah = *reinterpret_cast<const bf16x8*>(wr), al = *reinterpret_cast<const bf16x8*>(wr + 1024);
                f32x16 pr = __builtin_amdgcn_mfma_f32_32x32x16_f16(H8(al), H8(bh), f32x16{}, 0, 0, 0);
                pr = __builtin_amdgcn_mfma_f32_32x32x16_f16(H8(ah), H8(bl), pr, 0, 0, 0);
                pr = __builtin_amdgcn_mfma_f32_32x32x16_f16(H8(ah), H8(bh), pr, 0, 0, 0);
                const float inv = dscale[sl * 32 + l31];
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = __fmaf_rn(pr[r], inv, acc[r]);
            }
            // acc[r] = pre-activation of decoder output ob*32 + (r&3) + 8(r>>2) + 4h for tet row l31
            float p0 = 0.f, p1 = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n0 = ob * 32 + 8 * q + 4 * h;
                const f32x4_t a1 = *reinterpret_cast<const f32x4_t*>(dconst + n0), b1 = *reinterpret_cast<const f32x4_t*>(dconst + 64 + n0);
                const f32x4_t u0 = *reinterpret_cast<const f32x4_t*>(dconst + 128 + n0), u1 = *reinterpret_cast<const f32x4_t*>(dconst + 192 + n0);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float hv = fmaxf(__fmaf_rn(acc[4 * q + c], a1[c], b1[c]), 0.f);
                    p0 = __fmaf_rn(hv, u0[c], p0);
                    p1 = __fmaf_rn(hv, u1[c], p1);
                }
            }
            // the two k-group lanes of a row (l31, l31 + 32) hold disjoint halves of the block's outputs
            const float q0 = __shfl_xor(p0, 32), q1 = __shfl_xor(p1, 32);
            p0 = h == 0 ? p0 + q0 : q0 + p0;
            p1 = h == 0 ? p1 + q1 : q1 + p1;
            const int32_t ticket = (int32_t)(ie + 1);
            if (ob == 1) {
                if (h == 0) *reinterpret_cast<float2*>(pbuf + 2 * l31) = make_float2(p0, p1);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (lane == 0) *pflag = ticket;
            } else {
                while (*pflag != ticket) __builtin_amdgcn_s_sleep(1);
                asm volatile("" ::: "memory");
                const volatile float* pv = pbuf + 2 * l31;
                const float o10 = pv[0], o11 = pv[1];
                const int64_t grow = tile_of(ie) * TILE + l31;
                if (h == 0 && grow < n_dst)
                    *reinterpret_cast<float2*>(dec.logits + grow * 2) = make_float2((p0 + o10) + dconst[256], (p1 + o11) + dconst[257]);
            }
        }
    };
    // what follows the products of tile `it`: hand the partner its half (K split) or finish the block right away (full K)
    auto finishC = [&](int64_t it, f32x16& acc) {
        if constexpr (KS == 2) {
            float* red = redbuf + (it & 1) * (C::RED_BYTES / 4) + w * 512 + lane;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                mine[r] = kh ? acc[8 + r] : acc[r];
                red[r * 64] = kh ? acc[r] : acc[8 + r];
            }
        } else {
            // full K in this wave: finish the 32 x 32 block right away (row (r&3) + 8(r>>2) + 4h, column `col`)
            const int64_t tile = tile_of(it);
            if constexpr (TR) {
                // transposed block: columns cs*32 + 4h + 8g + c (g < 4, c < 4) of tet row rg*32 + l31
                const int row = rg * 32 + l31, cbase = cs * 32 + 4 * h;
                const int64_t grow = tile * TILE + row;
                const float rf = rowf[(int)(it & 3) * TILE + row];
                float* o = out + grow * ldo + cbase;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4_t b4 = *reinterpret_cast<const f32x4_t*>(colp + cbase + 8 * g);
                    f32x4_t v;
#pragma unroll
                    for (int c = 0; c < 4; ++c) v[c] = __fmaf_rn(acc[4 * g + c], rf, b4[c]);
                    if (has_scale) {
                        const f32x4_t s4 = *reinterpret_cast<const f32x4_t*>(colp + COUT + cbase + 8 * g);
                        const f32x4_t h4 = *reinterpret_cast<const f32x4_t*>(colp + 2 * COUT + cbase + 8 * g);
#pragma unroll
                        for (int c = 0; c < 4; ++c) v[c] = __fmaf_rn(v[c], s4[c], h4[c]);
                    }
                    if (relu) {
#pragma unroll
                        for (int c = 0; c < 4; ++c) v[c] = fmaxf(v[c], 0.f);
                    }
                    if (grow < n_dst) *reinterpret_cast<f32x4_t*>(o + 8 * g) = v;
                }
            } else {
            const int64_t row0 = tile * TILE + rg * 32 + 4 * h;
            float* o = out + row0 * ldo + col;
            const bool full = (tile + 1) * TILE <= n_dst;
            float rf[16];
            if constexpr (DSP == 2) {
                const float* rfp = rowf + (int)(it & 3) * TILE + rg * 32 + 4 * h;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4_t t = *reinterpret_cast<const f32x4_t*>(rfp + 8 * q);
#pragma unroll
                    for (int r = 0; r < 4; ++r) rf[4 * q + r] = t[r];
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = DSP == 2 ? __fmaf_rn(acc[r], rf[r], bb) : acc[r] + bb;
                if (has_scale) v = __fmaf_rn(v, sc, sh);
                if (relu) v = fmaxf(v, 0.f);
                const int rr = (r & 3) + 8 * (r >> 2);
                if (full || row0 + rr < n_dst) st_out(&o[(int64_t)rr * ldo], v);
            }
            }
        }
    };

    if constexpr (SKEW) {
        // Skewed schedule (see SKEW above).  Slot 2t+1: group A runs P(t+1), group B runs C(t) and finishes tile t-1;
        // slot 2t+2: group A runs C(t) and finishes tile t, group B runs P(t+1).  One barrier per slot.
        const bool grpA = kh == 0;
        if (my_n > 0) phaseP(0);
        tile_barrier();
        for (int64_t slot = 1; slot <= 2 * my_n + 1; ++slot) {
            const int64_t t = (slot - 1) >> 1;
            const bool doP = ((slot & 1) != 0) == grpA;
            if (doP) {
                if (t + 1 < my_n) phaseP(t + 1);
            } else {
                const bool hasC = t < my_n;
                f32x16 acc;
                stamp(trace, trace_cap, t, w, 4);
                if (hasC) {
                    phaseC(t, acc);
                    stamp(trace, trace_cap, t, w, 6);
                    float* red = redbuf + (t & 1) * (C::RED_BYTES / 4) + w * 512 + lane;
#pragma unroll
                    for (int r = 0; r < 8; ++r) red[r * 64] = kh ? acc[r] : acc[8 + r];
                }
                if (grpA) {
                    // the partner's half of tile t was written in the previous slot
                    if (hasC) {
#pragma unroll
                        for (int r = 0; r < 8; ++r) mine[r] = acc[r];
                        epilogueK2(t);
                    }
                } else {
                    if (t >= 1) epilogueK2(t - 1);
                    if (hasC) {
#pragma unroll
                        for (int r = 0; r < 8; ++r) mine[r] = acc[8 + r];
                    }
                }
                stamp(trace, trace_cap, t, w, 5);
            }
            tile_barrier();
        }
    } else {
    for (int64_t it = 0; it <= my_n; ++it) {
        if (it < my_n) phaseP(it);
        tile_barrier();  // A-tile `it` complete; partial sums of tile `it-1` complete
        stamp(trace, trace_cap, it, w, 4);
        if (KS == 2 && it > 0) {
            epilogueK2(it - 1);  // delayed epilogue of tile it-1
            if constexpr (DEC) {
                tile_barrier();  // every wave's fragments of tile it-1 are in LDS
                if (w < 2) decoderD(it - 1);
            }
        }
        stamp(trace, trace_cap, it, w, 6);
        if (it < my_n) {
            f32x16 acc;
            phaseC(it, acc);
            finishC(it, acc);
            stamp(trace, trace_cap, it, w, 5);
        }
    }
    }
}

template <int CIN_PAD, int COUT, int DSP, int FSP, bool DEC = false>
int launch2(const int32_t* rowptr, const int32_t* src, const int32_t* eid, int64_t n_dst, const float* x, const float* xdst, int64_t ldx, int c_in, const float* ea,
            int64_t lde, const float* We, const float* be, const float* Wj, const float* bj, const float* Wi,
            const float* scale, const float* shift, int relu, float* out, int64_t ldo, int xvec, hipStream_t stream, DecArgs dec = DecArgs{}) {
    // C_in <= 64: four-wave workgroups, two per CU (see Cfg2); C_in = 128: eight waves with the K split
    // (64 -> 64 would need 16 tets per wave with 4 channels per lane: 110 spilled registers -- it keeps the 8-wave form)
    constexpr int NW = (CIN_PAD <= 64 && !(CIN_PAD == 64 && COUT == 64)) ? DGNN_SMALL_NW : 8, KS = NW == 8 ? 2 : 1;
    using C = Cfg2<CIN_PAD, COUT, NW, KS, DSP, FSP, DEC>;
    const int64_t ntiles = dgnn_cdiv(n_dst, C::TILE);
    const size_t smem = C::SMEM_BYTES;
    static bool attr_set[DGNN_MAX_DEVICES] = {};
    dgnn_allow_dynamic_lds(reinterpret_cast<const void*>(&k_sage_fused_mfma<CIN_PAD, COUT, NW, KS, DSP, FSP, DEC>), smem, attr_set);
    const int wg_max = DGNN_NUM_CU * (NW == 8 ? 1 : occ_of<CIN_PAD, COUT, NW, KS, DSP, FSP>());
    int grid = (int)(ntiles < wg_max ? ntiles : wg_max);
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL((k_sage_fused_mfma<CIN_PAD, COUT, NW, KS, DSP, FSP, DEC>), dim3(grid), dim3(64 * NW), smem, stream, rowptr, src, eid, n_dst, x, xdst,
                       ldx, c_in, ea, lde, We, be, Wj, bj, Wi, scale, shift, relu, out, ldo, ntiles, xvec, g_dgnn_trace_buf,
                       g_dgnn_trace_cap, dec);
    return dgnn_check_launch(DEC ? "sage_layer_fused_decoder_fwd" : "sage_layer_fused_fwd(mfma filter)");
}

}  // namespace

int dgnn_ws_enabled();      // fused_ws.hip
int dgnn_sage_layer_fused_ws_try(const int32_t* rowptr, const int32_t* src, const int32_t* eid, int64_t n_dst, const float* x_src, const float* x_dst, int64_t ldx,
                                 int c_in, const float* edge_attr, int64_t lde, const float* We, const float* be, const float* Wj, const float* bj, const float* Wi,
                                 const float* scale, const float* shift, int relu, int c_out, float* out, int64_t ldo, hipStream_t stream, const float* W0 = nullptr,
                                 const float* b0 = nullptr, const float* scale1 = nullptr, const float* shift1 = nullptr, const float* W3 = nullptr,
                                 const float* b3 = nullptr, float* logits = nullptr);

// Returns DGNN_E_UNSUPPORTED when the shape does not fit this variant (the caller then uses fused.hip MODE 1).
int dgnn_sage_layer_fused_mfma_try(const int32_t* rowptr, const int32_t* src, const int32_t* eid, int64_t n_dst, const float* x_src,
                                   const float* x_dst, int64_t ldx, int c_in, const float* edge_attr, int64_t lde, const float* We, const float* be,
                                   const float* Wj, const float* bj, const float* Wi, const float* scale, const float* shift,
                                   int relu, int c_out, float* out, int64_t ldo, int f16_parts, hipStream_t stream, void* prep, int prep_mode) {
    const int cin_pad = c_in <= 32 ? 32 : (c_in <= 64 ? 64 : 128);
    const int nb = cin_pad / 16;
    if (c_in % nb != 0 || (c_out != 64 && c_out != 128) || (cin_pad == 128 && c_out != 128)) return DGNN_E_UNSUPPORTED;
    if ((c_in == 128 || c_in == 64) && c_out == 128 && f16_parts == 2 && prep_mode != 1 && dgnn_ws_enabled()) {
        // round 5: the plain 128 -> 128 and 64 -> 128 layers in the default arithmetic run wave-specialised (fused_ws.hip); it derives what it needs from the weights
        // itself (a prepared buffer is not read)
        const int rc = dgnn_sage_layer_fused_ws_try(rowptr, src, eid, n_dst, x_src, x_dst, ldx, c_in, edge_attr, lde, We, be, Wj, bj, Wi, scale, shift, relu, c_out, out,
                                                    ldo, stream);
        if (rc != DGNN_E_UNSUPPORTED) return rc;
    }
    const int xvec = ((((uintptr_t)x_src | (uintptr_t)x_dst) % 16) == 0 && ldx % 4 == 0) ? 1 : 0;
    if (nb >= 4 && !xvec && prep_mode != 1) return DGNN_E_UNSUPPORTED;
    if (f16_parts && prep_mode != 1 && (ldo % 4 != 0 || ((uintptr_t)out % 16) != 0)) f16_parts = 0;  // the fp16 forms store 16-byte pieces of the output rows
    if (prep_mode != 0 && f16_parts != 2) return DGNN_E_UNSUPPORTED;   // prepared parameters exist for the default arithmetic only
    DecArgs dargs{};
    dargs.prep = prep;
    dargs.prep_mode = prep_mode;
    // f16_parts: 0 = bf16 x 3 everywhere, 1 = dense product on fp16 x 2 (filter product bf16 x 3), 2 = both on fp16 x 2
#define GO3(CP, CO, D, F) return launch2<CP, CO, D, F>(rowptr, src, eid, n_dst, x_src, x_dst, ldx, c_in, edge_attr, lde, We, be, Wj, bj, Wi, scale, \
                                                       shift, relu, out, ldo, xvec, stream, dargs)
#define GO2(CP, CO)                            \
    do {                                       \
        if (f16_parts == 2) GO3(CP, CO, 2, 2); \
        if (f16_parts == 1) GO3(CP, CO, 2, 3); \
        GO3(CP, CO, 3, 3);                     \
    } while (0)
    if (cin_pad == 32) { if (c_out == 64) GO2(32, 64); else GO2(32, 128); }
    if (cin_pad == 64) { if (c_out == 64) GO2(64, 64); else GO2(64, 128); }
    GO2(128, 128);
#undef GO2
#undef GO3
}

// The last conv layer of the shipped model together with the decoder (reference :180-187 applied at :350-351) in ONE launch: the finished
// 32-tet tile goes through Linear(128 -> 64) - BatchNorm(eval, folded) - ReLU - Linear(64 -> 2) inside the workgroup and only the logits are
// written (8 B per tet instead of 512 B out and 512 B back in).  Shapes: c_in <= 128 padded to 128 with c_in % 8 == 0, c_out = 128, f_e = 20,
// decoder 128 -> 64 -> 2; fp16 two-part arithmetic (gemm mode DGNN_GEMM_F16X2) throughout.  Anything else: DGNN_E_UNSUPPORTED (the caller
// then runs the layer and dgnn_decoder_fused_fwd as two launches).
namespace {
int fused_decoder_impl(const int32_t* rowptr, const int32_t* src, const int32_t* eid, int64_t n_dst, const float* x_src, const float* x_dst, int64_t ldx, int c_in,
                       const float* edge_attr, int64_t lde, int f_e, const float* We, const float* be, const float* Wj, const float* bj, const float* Wi,
                       const float* scale, const float* shift, int relu, int c_out, const float* W0, const float* b0, const float* scale1,
                       const float* shift1, int c_hidden, const float* W3, const float* b3, int n_logits, float* logits, void* prep, int prep_mode,
                       hipStream_t stream) {
    DGNN_REQUIRE(n_dst >= 0, DGNN_E_INVALID, "sage_layer_fused_decoder_fwd: bad sizes");
    if (n_dst == 0 && prep_mode != 1) return DGNN_OK;
    if (prep_mode != 1) {
        DGNN_REQUIRE(rowptr && src && x_src && edge_attr && logits, DGNN_E_INVALID, "sage_layer_fused_decoder_fwd: null pointer");
        if (x_dst == nullptr) x_dst = x_src;
    }
    DGNN_REQUIRE(We && be && Wj && Wi && W0 && b0 && W3 && b3, DGNN_E_INVALID, "sage_layer_fused_decoder_fwd: null parameter pointer");
    DGNN_REQUIRE((scale == nullptr) == (shift == nullptr) && (scale1 == nullptr) == (shift1 == nullptr), DGNN_E_INVALID,
                 "sage_layer_fused_decoder_fwd: scale/shift must come together");
    bool ok = c_out == 128 && c_in > 64 && c_in <= 128 && c_in % 8 == 0 && f_e == fused::FE && c_hidden == 64 && n_logits == 2;
    if (prep_mode != 1)
        ok = ok && lde == fused::FE && ((((uintptr_t)x_src | (uintptr_t)x_dst | (uintptr_t)edge_attr) % 16) == 0) && ldx % 4 == 0 &&
             ((uintptr_t)logits % 8) == 0 && n_dst * ldx < ((int64_t)1 << 31);
    if (prep_mode != 0) ok = ok && prep != nullptr && ((uintptr_t)prep % 16) == 0;
    if (!ok) return DGNN_E_UNSUPPORTED;
    if (prep_mode != 1 && c_in == 128 && dgnn_ws_enabled()) {      // round 5: the wave-specialised kernel carries the decoder too (fused_ws.hip)
        const int rc = dgnn_sage_layer_fused_ws_try(rowptr, src, eid, n_dst, x_src, x_dst, ldx, c_in, edge_attr, lde, We, be, Wj, bj, Wi, scale, shift, relu, c_out, nullptr,
                                                    0, stream, W0, b0, scale1, shift1, W3, b3, logits);
        if (rc != DGNN_E_UNSUPPORTED) return rc;
    }
    DecArgs dec{W0, b0, scale1, shift1, W3, b3, logits, prep, prep_mode};
    return launch2<128, 128, 2, 2, true>(rowptr, src, eid, n_dst, x_src, x_dst, ldx, c_in, edge_attr, lde, We, be, Wj, bj, Wi, scale, shift, relu, nullptr, 0, 1,
                                         stream, dec);
}
}  // namespace

extern "C" int dgnn_sage_layer_fused_decoder_fwd(const int32_t* rowptr, const int32_t* src, const int32_t* eid, int64_t n_dst, const float* x_src,
                                                 const float* x_dst, int64_t ldx, int c_in, const float* edge_attr, int64_t lde, int f_e,
                                                 const float* We, const float* be, const float* Wj, const float* bj, const float* Wi,
                                                 const float* scale, const float* shift, int relu, int c_out, const float* W0, const float* b0,
                                                 const float* scale1, const float* shift1, int c_hidden, const float* W3, const float* b3, int n_logits,
                                                 float* logits, void* stream_) {
    return fused_decoder_impl(rowptr, src, eid, n_dst, x_src, x_dst, ldx, c_in, edge_attr, lde, f_e, We, be, Wj, bj, Wi, scale, shift, relu, c_out, W0, b0, scale1,
                              shift1, c_hidden, W3, b3, n_logits, logits, nullptr, 0, (hipStream_t)stream_);
}

// ---- prepared parameters (fp16 two-part arithmetic, DGNN_GEMM_F16X2) --------------------------------------------------------------------
// What a fused launch derives from the layer's parameters before its first tile -- two power-of-two scales, the split filter operand, every
// wavefront's resident fragments of [Wj | Wi] (128 KB at 128 -> 128), the decoder's W0 fragments and folded constants -- costs 15-20 us per launch
// (each of the 256 workgroups reads the weights column-wise and splits them).  dgnn_sage_layer_prepare runs that prologue ONCE (one workgroup)
// and parks the result; the *_p entry points read it back with coalesced 16-byte loads.  The values are the same: results are bit-identical.
// The buffer belongs to (We, be, Wj, Wi[, W0, b0, scale1, shift1, W3, b3]) and to the layer shape; the caller prepares again when they change.
extern "C" int64_t dgnn_sage_layer_prepared_bytes(int c_in, int c_out, int with_decoder) {
    if (c_in <= 0 || c_in > 128 || (c_out != 64 && c_out != 128)) return 0;
    const int cp = c_in <= 32 ? 32 : (c_in <= 64 ? 64 : 128);
    if (cp == 128 && c_out != 128) return 0;
    if (with_decoder) return (cp == 128 && c_out == 128) ? Cfg2<128, 128, 8, 2, 2, 2, true>::PREP_BYTES : 0;
    if (cp == 32) return c_out == 64 ? Cfg2<32, 64, DGNN_SMALL_NW, DGNN_SMALL_NW == 8 ? 2 : 1, 2, 2>::PREP_BYTES : Cfg2<32, 128, DGNN_SMALL_NW, DGNN_SMALL_NW == 8 ? 2 : 1, 2, 2>::PREP_BYTES;
    if (cp == 64) return c_out == 64 ? Cfg2<64, 64, 8, 2, 2, 2>::PREP_BYTES : Cfg2<64, 128, DGNN_SMALL_NW, DGNN_SMALL_NW == 8 ? 2 : 1, 2, 2>::PREP_BYTES;
    return Cfg2<128, 128, 8, 2, 2, 2>::PREP_BYTES;
}

extern "C" int dgnn_sage_layer_prepare(int c_in, int c_out, const float* We, const float* be, const float* Wj, const float* Wi, const float* W0,
                                       const float* b0, const float* scale1, const float* shift1, const float* W3, const float* b3, void* prepared,
                                       void* stream_) {
    DGNN_REQUIRE(We && be && Wj && Wi && prepared && ((uintptr_t)prepared % 16) == 0, DGNN_E_INVALID, "sage_layer_prepare: null / unaligned pointer");
    if (W0)
        return fused_decoder_impl(nullptr, nullptr, nullptr, 0, nullptr, nullptr, 0, c_in, nullptr, 0, fused::FE, We, be, Wj, nullptr, Wi, nullptr, nullptr, 1, c_out, W0,
                                  b0, scale1, shift1, 64, W3, b3, 2, nullptr, prepared, 1, (hipStream_t)stream_);
    const int cin_pad = c_in <= 32 ? 32 : (c_in <= 64 ? 64 : 128);
    if (c_in % (cin_pad / 16) != 0) return DGNN_E_UNSUPPORTED;
    return dgnn_sage_layer_fused_mfma_try(nullptr, nullptr, nullptr, 0, nullptr, nullptr, 0, c_in, nullptr, 0, We, be, Wj, nullptr, Wi, nullptr, nullptr, 1, c_out, nullptr,
                                          0, 2, (hipStream_t)stream_, prepared, 1);
}

// dgnn_sage_layer_fused_fwd (gemm mode DGNN_GEMM_F16X2) / dgnn_sage_layer_fused_decoder_fwd with the parameters prepared by dgnn_sage_layer_prepare
extern "C" int dgnn_sage_layer_fused_fwd_p(const int32_t* rowptr, const int32_t* src, const int32_t* eid, int64_t n_dst, const float* x_src,
                                           const float* x_dst, int64_t ldx, int c_in, const float* edge_attr, int64_t lde, int f_e, const float* We,
                                           const float* be, const float* Wj, const float* bj, const float* Wi, const float* scale, const float* shift,
                                           int relu, int c_out, float* out, int64_t ldo, const void* prepared, void* stream_) {
    DGNN_REQUIRE(n_dst >= 0 && c_in > 0 && c_out > 0, DGNN_E_INVALID, "sage_layer_fused_fwd_p: bad sizes");
    if (n_dst == 0) return DGNN_OK;
    DGNN_REQUIRE(rowptr && src && x_src && edge_attr && We && be && Wj && Wi && out && prepared, DGNN_E_INVALID, "sage_layer_fused_fwd_p: null pointer");
    DGNN_REQUIRE((scale == nullptr) == (shift == nullptr), DGNN_E_INVALID, "sage_layer_fused_fwd_p: scale/shift must come together");
    if (x_dst == nullptr) x_dst = x_src;
    if (f_e != fused::FE || lde != fused::FE || ((uintptr_t)edge_attr % 16) != 0 || ((uintptr_t)prepared % 16) != 0 || n_dst * ldx >= ((int64_t)1 << 31) ||
        ldo % 4 != 0 || ((uintptr_t)out % 16) != 0)
        return DGNN_E_UNSUPPORTED;
    return dgnn_sage_layer_fused_mfma_try(rowptr, src, eid, n_dst, x_src, x_dst, ldx, c_in, edge_attr, lde, We, be, Wj, bj, Wi, scale, shift, relu, c_out, out, ldo, 2,
                                          (hipStream_t)stream_, const_cast<void*>(prepared), 2);
}

extern "C" int dgnn_sage_layer_fused_decoder_fwd_p(const int32_t* rowptr, const int32_t* src, const int32_t* eid, int64_t n_dst, const float* x_src,
                                                   const float* x_dst, int64_t ldx, int c_in, const float* edge_attr, int64_t lde, int f_e,
                                                   const float* We, const float* be, const float* Wj, const float* bj, const float* Wi,
                                                   const float* scale, const float* shift, int relu, int c_out, const float* W0, const float* b0,
                                                   const float* scale1, const float* shift1, int c_hidden, const float* W3, const float* b3,
                                                   int n_logits, float* logits, const void* prepared, void* stream_) {
    DGNN_REQUIRE(prepared != nullptr, DGNN_E_INVALID, "sage_layer_fused_decoder_fwd_p: null prepared buffer");
    return fused_decoder_impl(rowptr, src, eid, n_dst, x_src, x_dst, ldx, c_in, edge_attr, lde, f_e, We, be, Wj, bj, Wi, scale, shift, relu, c_out, W0, b0, scale1,
                              shift1, c_hidden, W3, b3, n_logits, logits, const_cast<void*>(prepared), 2, (hipStream_t)stream_);
}
