// Fused SAGE inference layer with bf16 STORAGE (BASELINE config 3 / SURVEY 7 step 7): activations live in HBM as bf16,
// every product runs ONCE on the bf16 matrix cores (v_mfma_f32_16x16x32_bf16 for the filter MLP, v_mfma_f32_32x32x16_bf16
// for the dense part) with fp32 accumulation; parameters stay fp32 in HBM (master weights) and are rounded to bf16 when a
// workgroup stages them.  Stated tolerance of this path: |dlogit| <= 5e-2 * max(1, |logit| / 8), arg-max agreement >= 99.9 %
// against the fp32 reference; it is NOT the fp32-class path of fused_mfma.hip.
//
// Same two-phase, one-barrier-per-tile loop as fused_mfma.hip, but with a third of the matrix work and no operand splits the
// layer is HBM-bound, so the shape is chosen for bytes in flight instead of issue slots:
//   * four-wave workgroups with the full K per wave (no partial-sum exchange).  Resident weights are K/16 x 4 VGPRs
//     (64 at C_in = 128), LDS 50 KB at 128 -> 128: three workgroups share a CU, each with its own barrier;
//   * a tile's rows (4 neighbour rows + own row per tet, 2*C_in bytes each) are issued one tile ahead as 16-byte loads and
//     land under the dense phase; edge-attribute rows (fp32, the caller's tensor, read in place through the plan's eid)
//     arrive by LDS-DMA exactly as in the fp32 kernel;
//   * the filter product's C/D layout again leaves the 4 in-edges of a tet in the 4 accumulator registers of one lane:
//     in-order fp32 sum over the gathered rows (unpacked bf16), x 0.25, rounded to bf16 into the LDS A-tile next to the tet's
//     own row (copied as is);
//   * epilogue: bias / BatchNorm(eval) / ReLU in fp32, neighbouring columns exchanged by DPP so that every lane stores one
//     4-byte pair (2 bf16) -- 64-byte row segments per half-wave instead of 2-byte scatters.
// Algorithmic bytes per tet: 2*C_in (own row) + 320 (4 attribute rows) + 16 (4 source ids) + 2*C_out.
#include "fused_common.h"

#ifndef DGNN_BF16_DENSE_GROUP
#define DGNN_BF16_DENSE_GROUP 0
#endif
#ifndef DGNN_BF16_CB_GROUP
#define DGNN_BF16_CB_GROUP 2
#endif

namespace {
using namespace fused;

typedef float f32x4_t __attribute__((ext_vector_type(4)));

// PR (compensated mode, the default): what is STORED is bf16, but nothing else is rounded to bf16 on the way -- the fp32 mean
// `a`, the fp32 attributes and every fp32 parameter enter the matrix cores as a (hi, lo) pair of bf16 values (16 significant
// bits; the hi x lo cross terms are kept, lo x lo is dropped):  filter = 3 products, a . Wj = 3 products, x_i . Wi = 2 products
// (x_i is exactly its stored bf16 value).  The only roundings left per layer are the ones storage implies: the gathered /
// own rows as they were stored and the layer's output.  PR = 0 is the plain single-product form (every operand rounded to
// bf16 once); it is ~2x further from the fp32 reference (measured: rms 1.0e-2 vs the compensated mode's on the 1M-tet graph).
// XF: the layer's INPUT rows are fp32 (the first layer reads the caller's fp32 feature matrix in place -- no cast pass, and
// real standardised features reach 174 sigma, where a bf16 rounding of the input alone costs 0.3 absolute): gathered rows enter
// the fp32 products as they are, the own row goes to the matrix cores as a (hi, lo) pair (PR) or rounded once (PR = 0).
// Eight-wave form (D16): 16-byte slot of channel block jb (8 channels) of part p (0 = mean hi, 1 = mean lo, 2 = own row) inside an A-tile row.
// The dense phase reads the tile with ds_read_b128, which the LDS serves in four NON-contiguous 16-lane groups ({0-3, 12-15, 20-27}, ...:
// MI355X_MICROARCH.md, LDS): a group mixes rows jcol of k-group tq with other rows of k-group tq + 1, so with the k-groups of a k-step in
// neighbouring slots -- the round-2/3 layout [part][k-step][k-group] -- two lanes of every group met on one bank: 25 conflict cycles per tet,
// 29 % of the launch's LDS cycles (profiles/r03g_bf16.md).  Here the two k-groups a lane group mixes (tq, tq ^ 1) sit 16 slots (= 256 B, one
// bank row) apart, so the 16 lanes of a group fall on 16 different slots; the writers (8-lane groups of ds_write_b128, banks modulo 128 B) still
// store contiguous 128-byte runs.  k-group kg of k-step S holds channel block jb = 8 * (kg & 1) + 4 * (kg >> 1) + S (weights are loaded to match).
__host__ __device__ constexpr int d16_slot(int p, int jb) { return (p & 1) * 8 + (p >> 1) * 32 + 16 * (jb >> 3) + (jb & 7); }   // p = 3: the own row's lo part (UB rows)
__host__ __device__ constexpr int d16_block(int S, int kg) { return 8 * (kg & 1) + 4 * (kg >> 1) + S; }

struct DecB {      // the decoder behind the last conv layer (DEC instantiation); all NULL otherwise
    const float *W0, *b0, *scale1, *shift1, *W3, *b3;
    float* logits;
};

// UB (round 4): rows of a layer that ends in a ReLU are never negative, so their sign bit carries nothing: UNSIGNED rows keep the fp32 bits
// [30:15] -- bf16's 8 exponent bits and EIGHT explicit mantissa bits (9 significant bits) in the same two bytes; the rounding that storage costs
// halves (2^-10 of the value instead of 2^-9).  Measured on the 1M-tet metric graph with all three stored layers unsigned: max |dlogit| 5.9e-2 ->
// 2.1e-2, rms 4.1e-3 -> 2.1e-3 -- SURVEY 8c's flat 5e-2 then holds for every logit.  UB bit 0: the INPUT rows are unsigned (gathered rows
// decode with one more shift; the own row enters the matrix cores as a (hi, lo) bf16 pair like an fp32 row), bit 1: the OUTPUT rows are.
template <int CIN_PAD, int COUT, int NW = 4, int PR = 0, int XF = 0, bool DEC = false, int UB = 0>
struct CfgB {
    static constexpr bool UBI = (UB & 1) != 0, UBO = (UB & 2) != 0;
#ifndef DGNN_UB_OWN_PAIR
#define DGNN_UB_OWN_PAIR 0
#endif
    // own-row parts in the A-tile.  An unsigned own row (UBI) goes to the matrix cores rounded to bf16 (one part): its ninth bit would cost a (hi, lo)
    // split per tet and a sixth product per k-step for 2^-10 of ONE of the layer's two terms -- measured on the 1M-tet graph (CPU model of the
    // storage roundings): max |dlogit| 2.1e-2 -> 2.8e-2 without it, rms 2.1e-3 -> 2.6e-3; the gathered rows keep all nine bits
    // (DGNN_UB_OWN_PAIR=1 builds the pair form)
    static constexpr int XPARTS = (PR && (XF || (UBI && DGNN_UB_OWN_PAIR))) ? 2 : 1;
    static constexpr int APARTS = PR ? 2 : 1;             // mean parts
    static constexpr int K = (APARTS + XPARTS) * CIN_PAD; // A-tile row: [a_hi | a_lo | x_hi | x_lo] ... [a | x_i]
    // NW == 4: four waves, each a 32-column slice of v_mfma_f32_32x32x16_bf16 blocks.  NW == 8: eight waves, each a 16-column slice
    // of v_mfma_f32_16x16x32_bf16 blocks -- half the resident weights per wave (the compensated 128 -> 128 layer keeps 4 weight
    // parts resident: 128 VGPRs in the 4-wave form = spills, 64 here) and 4 tets per wave in the filter phase.
    static constexpr bool D16 = NW == 8;
    static constexpr int NSLICE = COUT / (D16 ? 16 : 32);
    static constexpr int RG = NW / NSLICE;
    static constexpr int TILE = 32 * RG;
    static constexpr int ROWB = D16 ? (XPARTS == 2 ? 65 : 57) * 16 : K * 2 + 16;   // A-tile row: K bf16 + 16 B pad (odd number of 16-B slots); D16: see d16_slot
    static constexpr int A_BYTES = TILE * ROWB;
    static constexpr int TPW = TILE / NW;                 // tets per wave (8 or 16)
    static constexpr int RB = TPW / 4;                    // 16-edge row blocks per wave
    static constexpr int NQ = TPW * 4;                    // edges per wave
    static constexpr int NB = CIN_PAD / 16;               // contiguous channels per lane (8, 4, 2)
    static constexpr int EA_BYTES = NQ * FE * 4;
    static constexpr int EA_FULL = EA_BYTES / 1024, EA_TAIL = (EA_BYTES % 1024) / 256;
    static constexpr int BP_BYTES = (PR ? 2 : 1) * NB * 768;   // [part][cb][g<3][j<16] x 16 B filter operand
    // DEC: W0 fragments (hi, lo: 2 x 16 KB), the finished tile as decoder operand fragments (2 buffers x (hi, lo) x 8 KB), constants
    // (A1 | B1 | W3[2] | b3 | bj | scale | shift), partial logits (2 buffers x 4 hidden blocks x 32 tets x 2)
    static constexpr int W0_BYTES = DEC ? 2 * 16384 : 0, Y_BYTES = DEC ? 2 * 2 * 8192 : 0, C_FLOATS = DEC ? (64 + 64 + 128 + 4 + 3 * 128) : 0,
                         L_FLOATS = DEC ? 2 * 4 * 32 * 2 : 0;
    static constexpr int SMEM_BYTES = 2 * A_BYTES + NW * EA_BYTES + BP_BYTES + W0_BYTES + Y_BYTES + 4 * (C_FLOATS + L_FLOATS);
    static constexpr int NS = CIN_PAD / (D16 ? 32 : 16);  // k-steps (of 16, D16: of 32) per operand part (full K per wave)
    static_assert(RG >= 1 && NQ <= 64 && EA_BYTES % 256 == 0 && !(XF && NW == 8), "wave roles");
    static_assert(!D16 || (PR == 1 && XF == 0 && CIN_PAD == 128 && COUT == 128), "the eight-wave form is the compensated 128 -> 128 layer");
    static_assert(!DEC || D16, "the decoder rides in the eight-wave form only");
    static_assert(UB == 0 || PR == 1, "unsigned rows belong to the compensated arithmetic");
    static_assert(!(XF && UBI) && !(DEC && UBO), "fp32 input rows are not unsigned rows; the decoder-carrying launch stores logits");
};

__device__ __forceinline__ uint32_t pack_bf16(float a, float b) {
    const bf16x2_t h = __builtin_convertvector(f32x2_t{a, b}, bf16x2_t);
    return __builtin_bit_cast(uint32_t, h);
}
__device__ __forceinline__ float bf_lo(uint32_t u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float bf_hi(uint32_t u) { return __builtin_bit_cast(float, u & 0xFFFF0000u); }
__device__ __forceinline__ float bf16_round(float v) { return bf_lo(pack_bf16(v, 0.f)); }
// unsigned rows (UB): value = bits << 15; two values per dword, round to nearest even on bit 15 of the fp32 pattern (the sign is cleared: -0 -> +0)
// (one instruction each, like the bf16 decodes: the low half's shift leaves the neighbour's last bit in the sign position, which the |.| source
// modifier of the consuming instruction drops for free; the high half is a sub-dword (SDWA WORD_1) shift)
#ifdef DGNN_UB_MASK_DECODE
__device__ __forceinline__ float ub_lo(uint32_t u) { return __builtin_bit_cast(float, (u & 0xFFFFu) << 15); }
__device__ __forceinline__ float ub_hi(uint32_t u) { return __builtin_bit_cast(float, (u >> 1) & 0x7FFF8000u); }
#else
__device__ __forceinline__ float ub_lo(uint32_t u) { return __builtin_fabsf(__builtin_bit_cast(float, u << 15)); }
__device__ __forceinline__ float ub_hi(uint32_t u) { return __builtin_bit_cast(float, (u >> 16) << 15); }
#endif
__device__ __forceinline__ uint32_t ub_enc(float v) {
    const uint32_t b = __builtin_bit_cast(uint32_t, v) & 0x7FFFFFFFu;
    return (b + 0x3FFFu + ((b >> 15) & 1u)) >> 15;
}
__device__ __forceinline__ uint32_t pack_ub(float a, float b) { return ub_enc(a) | (ub_enc(b) << 16); }
// (x0, x1) -> packed hi pair and packed lo pair: x = hi + lo + O(2^-17 |x|), both parts bf16
__device__ __forceinline__ void split2(float x0, float x1, uint32_t& hi, uint32_t& lo) {
    hi = pack_bf16(x0, x1);
    lo = pack_bf16(x0 - bf_lo(hi), x1 - bf_hi(hi));
}

// NB bf16 = NB/2 dwords per lane
template <int NB>
__device__ __forceinline__ void ld_bf(uint32_t (&v)[NB / 2], const uint16_t* p) {
    if (NB == 8) {
        const uint4 a = *reinterpret_cast<const uint4*>(p);
        v[0] = a.x; v[1 % (NB / 2)] = a.y; v[2 % (NB / 2)] = a.z; v[3 % (NB / 2)] = a.w;
    } else if (NB == 4) {
        const uint2 a = *reinterpret_cast<const uint2*>(p);
        v[0] = a.x; v[1 % (NB / 2)] = a.y;
    } else {
        v[0] = *reinterpret_cast<const uint32_t*>(p);
    }
}

template <int CIN_PAD, int COUT, int NW, int OCC, int PR, int XF, bool DEC = false, int UB = 0>
__global__ void __launch_bounds__(64 * NW, OCC)
k_sage_fused_bf16(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ src, const int32_t* __restrict__ eid, int64_t n_dst,
                  const void* __restrict__ x_, const void* __restrict__ xdst_, int64_t ldx, int c_in, const float* __restrict__ ea,
                  int64_t lde, const float* __restrict__ We, const float* __restrict__ be, const float* __restrict__ Wj,
                  const float* __restrict__ bj, const float* __restrict__ Wi, const float* __restrict__ scale,
                  const float* __restrict__ shift, int relu, uint16_t* __restrict__ out, int64_t ldo, int64_t ntiles, DecB dec) {
    using C = CfgB<CIN_PAD, COUT, NW, PR, XF, DEC, UB>;
    constexpr bool UBI = C::UBI, UBO = C::UBO;
    constexpr int ROWB = C::ROWB, TILE = C::TILE, TPW = C::TPW, RB = C::RB, NB = C::NB, NS = C::NS, NH = NB / 2;
    constexpr int NX = XF ? NB : NH;                                 // dwords per lane and row: NB floats or NB/2 bf16 pairs
    constexpr int XOFF = C::APARTS * CIN_PAD * 2;                    // byte offset of the own-row part(s) in an A-tile row
    const uint16_t* const x = reinterpret_cast<const uint16_t*>(x_);
    const uint16_t* const xdst = reinterpret_cast<const uint16_t*>(xdst_);
    const float* const xf = reinterpret_cast<const float*>(x_);
    const float* const xdstf = reinterpret_cast<const float*>(xdst_);
    // row fragment of this lane: NB channels from c0l
    auto ld_row = [&](uint32_t (&v)[NX], const void* base, int64_t elem_off) {
        if constexpr (XF) {
            const float* p = reinterpret_cast<const float*>(base) + elem_off;
#pragma unroll
            for (int i = 0; i < NB; ++i) v[i] = __builtin_bit_cast(uint32_t, p[i]);
        } else {
            uint32_t t[NH];
            ld_bf<NB>(t, reinterpret_cast<const uint16_t*>(base) + elem_off);
#pragma unroll
            for (int i = 0; i < NH; ++i) v[i] = t[i];
        }
    };
    auto chan = [&](const uint32_t (&v)[NX], int cb) -> float {    // channel cb of a fragment as fp32
        if constexpr (XF) return __builtin_bit_cast(float, v[cb]);
        else if constexpr (UBI) return (cb & 1) ? ub_hi(v[cb >> 1]) : ub_lo(v[cb >> 1]);
        else return (cb & 1) ? bf_hi(v[cb >> 1]) : bf_lo(v[cb >> 1]);
    };
    (void)x; (void)xdst; (void)xf; (void)xdstf;
    extern __shared__ __attribute__((aligned(16))) char smemb[];
    char* const abuf = smemb;                                        // [2][A_BYTES]
    char* const eabuf = smemb + 2 * C::A_BYTES;                      // [NW][EA_BYTES] fp32 attribute strips
    char* const bpbuf = eabuf + NW * C::EA_BYTES;                    // filter operand (bf16)
    char* const w0buf = bpbuf + C::BP_BYTES;                         // DEC: W0 fragments [part][hidden block][k-step][k-group][16 rows] x 16 B
    char* const ybuf = w0buf + C::W0_BYTES;                          // DEC: [2][part][k-step][k-group][32 tets] x 16 B
    float* const cbuf = reinterpret_cast<float*>(ybuf + C::Y_BYTES); // DEC: A1[64] | B1[64] | W3[2][64] | b3[2] pad 2 | bj[128] | scale[128] | shift[128]
    float* const lbuf = cbuf + C::C_FLOATS;                          // DEC: [2][4 hidden blocks][32 tets][2]
    (void)w0buf; (void)ybuf; (void)cbuf; (void)lbuf;

    const int lane = lane_id(), w = wave_id_uniform();
    const int h = lane >> 5, l31 = lane & 31;
    const int jcol = lane & 15, tq = lane >> 4;
    const int ldx32 = (int)ldx;

    const int nwg = gridDim.x, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, wg_per_xcd = (nwg + 7 - xcd) >> 3;
    const int64_t per = (ntiles + 7) / 8;
    const int64_t t_lo = xcd * per, t_hi = min(ntiles, t_lo + per);
    int64_t my_n = 0;
    if (t_lo + slot < t_hi) my_n = (t_hi - t_lo - slot + wg_per_xcd - 1) / wg_per_xcd;
    auto tile_of = [&](int64_t it) { return t_lo + slot + it * wg_per_xcd; };

    // ---- filter operand B = [We^T ; be ; 0] as bf16 -> LDS; entry (cb, g, j): channel c = NB*j + cb, k = 8g .. 8g+7
    for (int e = threadIdx.x; e < NB * 48; e += blockDim.x) {
        const int cb = e / 48, gj = e - cb * 48, g = gj >> 4, j = gj & 15;
        const int c = NB * j + cb;
        uint32_t p[4], q[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            float v[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int k = 8 * g + 2 * d + u;
                v[u] = 0.f;
                if (c < c_in) v[u] = k < FE ? We[(int64_t)c * FE + k] : (k == FE ? be[c] : 0.f);
            }
            split2(v[0], v[1], p[d], q[d]);
        }
        *reinterpret_cast<uint4*>(bpbuf + (cb * 48 + gj) * 16) = make_uint4(p[0], p[1], p[2], p[3]);
        if (PR) *reinterpret_cast<uint4*>(bpbuf + NB * 768 + (cb * 48 + gj) * 16) = make_uint4(q[0], q[1], q[2], q[3]);
    }

    // ---- dense-phase role: (column slice cs, row group rg); the whole K of this slice resident as bf16
    constexpr bool D16 = C::D16;
    const int cs = w % C::NSLICE, rg = w / C::NSLICE;
    const int col = D16 ? cs * 16 + jcol : cs * 32 + l31;
    const int kg = D16 ? tq : h;                                     // this lane's k-group inside a k-step (8 consecutive k)
    bf16x8 wjh[NS], wih[NS], wjl[PR ? NS : 1], wil[PR ? NS : 1];   // Wj / Wi rows of this slice: hi parts (+ lo parts when PR)
#pragma unroll
    for (int S = 0; S < NS; ++S) {
        uint32_t ph[4], pl[4], qh[4], ql[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const int k = (D16 ? 8 * d16_block(S, kg) : 16 * S + 8 * kg) + 2 * d;
            const int64_t o0 = (int64_t)col * c_in + (k < c_in ? k : 0), o1 = (int64_t)col * c_in + (k + 1 < c_in ? k + 1 : 0);
            split2(k < c_in ? Wj[o0] : 0.f, k + 1 < c_in ? Wj[o1] : 0.f, ph[d], pl[d]);
            split2(k < c_in ? Wi[o0] : 0.f, k + 1 < c_in ? Wi[o1] : 0.f, qh[d], ql[d]);
        }
        wjh[S] = pack8(ph);
        wih[S] = pack8(qh);
        if (PR) {
            wjl[S] = pack8(pl);
            wil[S] = pack8(ql);
        }
    }
    const float bb = bj ? bj[col] : 0.f;
    const float sc = scale ? scale[col] : 1.f;
    const float sh = scale ? shift[col] : 0.f;
    const bool has_scale = scale != nullptr;
    if constexpr (DEC) {
        // decoder operand A = W0 (64 x 128) as (hi, lo) bf16 fragments: entry (hidden row i, 8-channel block kb) -> k-step kb / 4, k-group kb % 4
        for (int e = threadIdx.x; e < 64 * 16; e += blockDim.x) {
            const int i = e >> 4, kb = e & 15;
            const float* wr = dec.W0 + (int64_t)i * 128 + 8 * kb;
            uint32_t ph[4], pl[4];
#pragma unroll
            for (int d = 0; d < 4; ++d) split2(wr[2 * d], wr[2 * d + 1], ph[d], pl[d]);
            char* dst = w0buf + ((((i >> 4) * 4 + (kb >> 2)) * 4 + (kb & 3)) * 16 + (i & 15)) * 16;
            *reinterpret_cast<uint4*>(dst) = make_uint4(ph[0], ph[1], ph[2], ph[3]);
            *reinterpret_cast<uint4*>(dst + 16384) = make_uint4(pl[0], pl[1], pl[2], pl[3]);
        }
        for (int e = threadIdx.x; e < 64; e += blockDim.x) {
            const float a1 = dec.scale1 ? dec.scale1[e] : 1.f;
            cbuf[e] = a1;
            cbuf[64 + e] = __fmaf_rn(dec.b0[e], a1, dec.scale1 ? dec.shift1[e] : 0.f);     // relu((z + b0) * scale1 + shift1) = relu(z * A1 + B1)
            cbuf[128 + e] = dec.W3[e];
            cbuf[192 + e] = dec.W3[64 + e];
        }
        if (threadIdx.x < 2) cbuf[256 + threadIdx.x] = dec.b3[threadIdx.x];
        for (int e = threadIdx.x; e < 128; e += blockDim.x) {
            cbuf[260 + e] = bj ? bj[e] : 0.f;
            cbuf[388 + e] = scale ? scale[e] : 1.f;
            cbuf[516 + e] = scale ? shift[e] : 0.f;
        }
    }
    __syncthreads();

    // ---- filter-phase role
    const int c0 = NB * jcol;
    const bool on = c0 < c_in;
    const int c0l = on ? c0 : 0;
    float* const myea = reinterpret_cast<float*>(eabuf + w * C::EA_BYTES);

    uint32_t xd[RB][NX], xr[RB][4][NX];
    bool regular = false;
    int vbeg1 = 0, vbeg2 = 0, vsrc1 = 0, veid1 = 0;
    bool ok1 = false, ok2 = false;
    int nv1 = 0, nv2 = 0;

    auto load_rowptr = [&](int64_t it, int& vb, int& nv) -> bool {
        if (it >= my_n) return false;
        const int64_t i0 = tile_of(it) * TILE + w * TPW;
        if (i0 >= n_dst) return false;
        nv = (int)(n_dst - i0 < TPW ? n_dst - i0 : TPW);
        vb = rowptr[i0 + (lane < nv ? lane : nv)];
        return true;
    };
    auto load_src = [&]() {
        if (ok1) {
            const int b0 = __builtin_amdgcn_readfirstlane(vbeg1);
            ok1 = __all(vbeg1 == b0 + 4 * (lane < nv1 ? lane : nv1)) != 0;
            if (ok1) {
                vsrc1 = src[b0 + (lane < 4 * nv1 ? lane : 4 * nv1 - 1)];
                if (eid) veid1 = eid[b0 + (lane < 4 * nv1 ? lane : 4 * nv1 - 1)];
            }
        }
    };
    auto issue_loads = [&](int64_t it) {
        regular = ok1;
        if (regular) {
            const int i0 = (int)(tile_of(it) * TILE) + w * TPW;
            const float* eab = ea + (int64_t)__builtin_amdgcn_readfirstlane(vbeg1) * lde;
            const int ea_last = nv1 * 4 * FE - 4;
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const int tl = rb * 4 + tq;
                ld_row(xd[rb], xdst_, (int64_t)(uint32_t)((i0 + (tl < nv1 ? tl : nv1 - 1)) * ldx32) + c0l);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int s_ = __shfl(vsrc1, tl * 4 + r);
                    ld_row(xr[rb][r], x_, (int64_t)(uint32_t)(s_ * ldx32) + c0l);
                }
            }
            if (eid) {
                auto row_ptr = [&](int fi) -> const float* {
                    const int e = (fi * 0xCCD) >> 16;             // fi / 20 for fi < 8192
                    return ea + (int64_t)__shfl(veid1, e) * FE + (fi - e * FE);
                };
#pragma unroll
                for (int q = 0; q < C::EA_FULL; ++q) glds16(row_ptr(q * 256 + lane * 4), myea + q * 256);
#pragma unroll
                for (int q = 0; q < C::EA_TAIL; ++q) glds4(row_ptr(C::EA_FULL * 256 + q * 64 + lane), myea + C::EA_FULL * 256 + q * 64);
            } else {
#pragma unroll
                for (int q = 0; q < C::EA_FULL; ++q) glds16(eab + min(q * 256 + lane * 4, ea_last), myea + q * 256);
#pragma unroll
                for (int q = 0; q < C::EA_TAIL; ++q)
                    glds4(eab + min(C::EA_FULL * 256 + q * 64 + lane, ea_last + 3), myea + C::EA_FULL * 256 + q * 64);
            }
        }
    };
    auto advance_idx = [&](int64_t it_next) {
        ok1 = ok2;
        vbeg1 = vbeg2;
        nv1 = nv2;
        load_src();
        ok2 = load_rowptr(it_next + 1, vbeg2, nv2);
    };
    // one finished (tet row, NB channels) segment -> A-tile: bf16 columns [c0, c0+NB) of the mean part(s) and of the own-row part
    auto put16 = [&](char* d, const uint32_t (&v)[NH]) {
        if (NB == 8) *reinterpret_cast<uint4*>(d) = make_uint4(v[0], v[1 % NH], v[2 % NH], v[3 % NH]);
        else if (NB == 4) *reinterpret_cast<uint2*>(d) = make_uint2(v[0], v[1 % NH]);
        else *reinterpret_cast<uint32_t*>(d) = v[0];
    };
    auto put_seg = [&](int buf, int row, const uint32_t (&av)[NH], const uint32_t (&al)[NH], const uint32_t (&xv)[NX]) {
        // own row as the matrix cores take it: bf16 rows are copied as they are; fp32 rows (XF) and unsigned rows (UB) become a (hi, lo) bf16 pair
        uint32_t xh[NH], xl[NH];
        if constexpr (XF) {
#pragma unroll
            for (int q = 0; q < NH; ++q) split2(__builtin_bit_cast(float, xv[2 * q]), __builtin_bit_cast(float, xv[2 * q + 1]), xh[q], xl[q]);
        } else if constexpr (UBI) {
#pragma unroll
            for (int q = 0; q < NH; ++q) {
                if constexpr (C::XPARTS == 2) split2(ub_lo(xv[q]), ub_hi(xv[q]), xh[q], xl[q]);
                else { xh[q] = pack_bf16(ub_lo(xv[q]), ub_hi(xv[q])); xl[q] = 0u; }
            }
        } else {
#pragma unroll
            for (int q = 0; q < NH; ++q) { xh[q] = xv[q]; xl[q] = 0u; }
        }
        if constexpr (C::D16) {     // conflict-free slots for the dense phase's ds_read_b128 lane groups (d16_slot)
            char* rowp = abuf + buf * C::A_BYTES + row * ROWB + (16 * (jcol >> 3) + (jcol & 7)) * 16;
            put16(rowp, av);
            put16(rowp + 8 * 16, al);
            put16(rowp + 32 * 16, xh);
            if constexpr (C::XPARTS == 2) put16(rowp + 40 * 16, xl);
            return;
        }
        char* dst = abuf + buf * C::A_BYTES + row * ROWB + c0 * 2;
        put16(dst, av);
        if (PR) put16(dst + CIN_PAD * 2, al);
        put16(dst + XOFF, xh);
        if constexpr (C::XPARTS == 2) put16(dst + XOFF + CIN_PAD * 2, xl);
    };

    // ---- DEC: the decoder on a finished tile (reference :180-187 applied at :350-351), one tile behind the layer: tile t's finished rows are parked
    // in LDS as (hi, lo) bf16 operand fragments by the dense phase of iteration t, multiplied with W0 after the tile barrier of iteration t + 1
    // (wave w: hidden block w & 3, tet block w >> 2; Linear - BatchNorm - ReLU - the 64 -> 2 Linear's partial sums over the wave's 16 hidden
    // units), and the four partial sums of a tet are added in a fixed order and stored after the barrier of iteration t + 2.  No barrier of its own.
    auto decoder_stage = [&](int64_t itd) {
        if constexpr (DEC) {
            const int hb = w & 3, tb = w >> 2;
            const char* W = w0buf + ((hb * 4 * 4 + tq) * 16 + jcol) * 16;
            const char* Y = ybuf + (itd & 1) * 16384 + ((tq * 32) + tb * 16 + jcol) * 16;
            f32x4_t d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int S = 0; S < 4; ++S) {
                const bf16x8 wh = *reinterpret_cast<const bf16x8*>(W + S * 1024), wl = *reinterpret_cast<const bf16x8*>(W + S * 1024 + 16384);
                const bf16x8 yh = *reinterpret_cast<const bf16x8*>(Y + S * 2048), yl = *reinterpret_cast<const bf16x8*>(Y + S * 2048 + 8192);
                d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, yh, d, 0, 0, 0);
                d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, yl, d, 0, 0, 0);
                d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, yh, d, 0, 0, 0);
            }
            // d[r] = (W0 . y) of hidden unit hb * 16 + 4 * tq + r, tet tb * 16 + jcol
            const int i0h = hb * 16 + 4 * tq;
            const f32x4_t a1 = *reinterpret_cast<const f32x4_t*>(cbuf + i0h), b1 = *reinterpret_cast<const f32x4_t*>(cbuf + 64 + i0h);
            const f32x4_t w30 = *reinterpret_cast<const f32x4_t*>(cbuf + 128 + i0h), w31 = *reinterpret_cast<const f32x4_t*>(cbuf + 192 + i0h);
            float p0 = 0.f, p1 = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float hv = fmaxf(__fmaf_rn(d[r], a1[r], b1[r]), 0.f);
                p0 = __fmaf_rn(hv, w30[r], p0);
                p1 = __fmaf_rn(hv, w31[r], p1);
            }
            // the four k-group lanes of a tet (l, l ^ 16, l ^ 32, l ^ 48): pairwise sums, the same value in all four
            p0 = cross_row_sum(p0);
            p1 = cross_row_sum(p1);
            if (tq == 0) *reinterpret_cast<float2*>(lbuf + (((itd & 1) * 4 + hb) * 32 + tb * 16 + jcol) * 2) = make_float2(p0, p1);
        }
    };
    auto logits_stage = [&](int64_t itf) {       // wave 0: lane -> (tet lane >> 1, logit lane & 1)
        if constexpr (DEC) {
            const float* L = lbuf + (itf & 1) * 256 + lane;
            const float v = ((L[0] + L[64]) + L[128]) + L[192] + cbuf[256 + (lane & 1)];
            const int64_t row = tile_of(itf) * TILE + (lane >> 1);
            if (row < n_dst) dec.logits[row * 2 + (lane & 1)] = v;
        }
    };

    ok1 = load_rowptr(0, vbeg1, nv1);
    load_src();
    ok2 = load_rowptr(1, vbeg2, nv2);
    issue_loads(0);

    for (int64_t it = 0; it < my_n; ++it) {
        // ================================================================ P: filter on the matrix cores + mean
        const int64_t i0 = tile_of(it) * TILE + w * TPW;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // rows + LDS-DMA'd strip of this tile
        const bool was_regular = regular;
        advance_idx(it + 1);
        if (was_regular) {
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                // A operand: lane (edge i = lane&15, k-group g = lane>>4) holds attributes 8g..8g+7 of its edge as bf16;
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
                uint32_t pa[4], pl[4];
#pragma unroll
                for (int d = 0; d < 4; ++d) split2(av[2 * d], av[2 * d + 1], pa[d], pl[d]);
                const bf16x8 aop = pack8(pa), aol = pack8(pl);

                uint32_t aout[NH], alo[NH];
                float prev = 0.f;
#pragma unroll
                for (int cb = 0; cb < NB; ++cb) {
                    const char* bp = bpbuf + (cb * 48 + (tq < 3 ? tq : 0) * 16 + jcol) * 16;
                    const uint4 u0 = *reinterpret_cast<const uint4*>(bp);
                    f32x4_t d = {0.f, 0.f, 0.f, 0.f};
                    if (PR) {
                        const uint4 u1 = *reinterpret_cast<const uint4*>(bp + NB * 768);
                        d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aol, __builtin_bit_cast(bf16x8, u0), d, 0, 0, 0);
                        d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aop, __builtin_bit_cast(bf16x8, u1), d, 0, 0, 0);
                    }
                    d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aop, __builtin_bit_cast(bf16x8, u0), d, 0, 0, 0);
                    // d[r] = phi of the r-th in-edge of this lane's tet, channel c0 + cb; in-order sum over the 4 in-edges
                    float a = __fmul_rn(chan(xr[rb][0], cb), d[0]);
                    a = __fmaf_rn(chan(xr[rb][1], cb), d[1], a);
                    a = __fmaf_rn(chan(xr[rb][2], cb), d[2], a);
                    a = __fmaf_rn(chan(xr[rb][3], cb), d[3], a);
                    a *= 0.25f;
                    if (cb & 1) split2(prev, a, aout[cb >> 1], alo[cb >> 1]);
                    else prev = a;
                    // keep at most DGNN_BF16_CB_GROUP channel blocks in flight: fully interleaved, their operands and results
                    // push the 128 -> 128 compensated kernel over the register file (73 spilled VGPRs)
                    if (PR && NB == 8 && (cb % DGNN_BF16_CB_GROUP) == DGNN_BF16_CB_GROUP - 1) __builtin_amdgcn_sched_barrier(0);
                }
                put_seg((int)(it & 1), w * TPW + rb * 4 + tq, aout, alo, xd[rb]);
            }
        } else {
            // generic path (a group with any in-degree other than 4, or past the end): per lane, one edge at a time (rare).
            // Operands are rounded to bf16 like the matrix-core path (PR: left in fp32, as the compensated products are fp32-class),
            // products and sums are fp32.
#pragma unroll 1
            for (int rb = 0; rb < RB; ++rb) {
                const int64_t i = i0 + rb * 4 + tq;
                float af[NB];
                uint32_t aout[NH], alo[NH], xv[NX];
#pragma unroll
                for (int cb = 0; cb < NB; ++cb) af[cb] = 0.f;
#pragma unroll
                for (int q = 0; q < NX; ++q) xv[q] = 0u;
                if (i < n_dst && on) {
                    const int b = rowptr[i], e_end = rowptr[i + 1];
                    ld_row(xv, xdst_, i * ldx + c0);
                    for (int k = b; k < e_end; ++k) {
                        const int s_ = src[k];
                        const float* ar = ea + (int64_t)(eid ? eid[k] : k) * lde;
                        uint32_t xs[NX];
                        ld_row(xs, x_, (int64_t)s_ * ldx + c0);
#pragma unroll 1
                        for (int cb = 0; cb < NB; ++cb) {
                            float p = 0.f;
                            if (c0 + cb < c_in) {
                                p = PR ? be[c0 + cb] : bf16_round(be[c0 + cb]);
                                for (int f = 0; f < FE; ++f)
                                    p = PR ? __fmaf_rn(We[(int64_t)(c0 + cb) * FE + f], ar[f], p)
                                           : __fmaf_rn(bf16_round(We[(int64_t)(c0 + cb) * FE + f]), bf16_round(ar[f]), p);
                            }
                            float xc = 0.f;   // (runtime cb: select, no dynamic register indexing)
#pragma unroll
                            for (int q = 0; q < NB; ++q) xc = q == cb ? chan(xs, q) : xc;
                            af[cb] = __fadd_rn(af[cb], __fmul_rn(xc, p));
                        }
                    }
                    const float cnt = (float)max(e_end - b, 1);
#pragma unroll
                    for (int cb = 0; cb < NB; ++cb) af[cb] = __fdiv_rn(af[cb], cnt);
                }
#pragma unroll
                for (int q = 0; q < NH; ++q) split2(af[2 * q], af[2 * q + 1], aout[q], alo[q]);
                put_seg((int)(it & 1), w * TPW + rb * 4 + tq, aout, alo, xv);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // strip reads returned before the next DMA may land
        asm volatile("" : "+v"(vbeg2), "+v"(vsrc1), "+v"(vbeg1), "+v"(veid1));
        issue_loads(it + 1);
        tile_barrier();  // A-tile `it` complete
        if constexpr (DEC) {
            if (it >= 1) decoder_stage(it - 1);
            if (it >= 2 && w == 0) logits_stage(it - 2);
        }

        // ================================================================ C: dense part on bf16 operands, full K per wave
        const int64_t tile = tile_of(it);
        const bool full = (tile + 1) * TILE <= n_dst;
        const int odd = lane & 1;
        if constexpr (!D16) {
            f32x16 acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
            const char* A = abuf + (it & 1) * C::A_BYTES + (rg * 32 + l31) * ROWB + h * 16;
#pragma unroll
            for (int S = 0; S < NS; ++S) {
                const bf16x8 ah = *reinterpret_cast<const bf16x8*>(A + S * 32);
                const bf16x8 xi = *reinterpret_cast<const bf16x8*>(A + XOFF + S * 32);
                if (PR) {   // small terms first
                    const bf16x8 al = *reinterpret_cast<const bf16x8*>(A + CIN_PAD * 2 + S * 32);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, wjh[S], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, wjl[S], acc, 0, 0, 0);
                    if constexpr (C::XPARTS == 2) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(A + XOFF + CIN_PAD * 2 + S * 32), wih[S], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xi, wil[S], acc, 0, 0, 0);
                }
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, wjh[S], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xi, wih[S], acc, 0, 0, 0);
            }
            // epilogue: row (r&3) + 8(r>>2) + 4h, column `col`; columns (col, col^1) of one row pair up into a 4-byte store:
            // even lanes store row r of the pair, odd lanes row r+1
            const int64_t row0 = tile * TILE + rg * 32 + 4 * h;
            uint16_t* o = out + row0 * ldo + (col & ~1);
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                float v0 = acc[r] + bb, v1 = acc[r + 1] + bb;
                if (has_scale) { v0 = __fmaf_rn(v0, sc, sh); v1 = __fmaf_rn(v1, sc, sh); }
                if (relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); }
                const float n0 = __shfl_xor(v0, 1), n1 = __shfl_xor(v1, 1);
                const uint32_t pk = UBO ? (odd ? pack_ub(n1, v1) : pack_ub(v0, n0)) : (odd ? pack_bf16(n1, v1) : pack_bf16(v0, n0));
                const int rr = (r & 3) + 8 * (r >> 2) + odd;
                if (full || row0 + rr < n_dst) *reinterpret_cast<uint32_t*>(o + (int64_t)rr * ldo) = pk;
            }
        } else {
            // sixteen columns per wave: two 16-row blocks of v_mfma_f32_16x16x32_bf16, lane (row jcol, k-group tq); k-step S of the lane's
            // k-group = channel block d16_block(S, tq), parked at slot base + 16 * (tq & 1) + 4 * (tq >> 1) + S of the row (d16_slot)
            f32x4_t acc2[2];
#pragma unroll
            for (int m = 0; m < 2; ++m) acc2[m] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            const char* A = abuf + (it & 1) * C::A_BYTES + (rg * 32 + jcol) * ROWB + (16 * (tq & 1) + 4 * (tq >> 1)) * 16;
#pragma unroll
            for (int S = 0; S < NS; ++S) {
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const char* Am = A + m * 16 * ROWB + S * 16;
                    const bf16x8 ah = *reinterpret_cast<const bf16x8*>(Am);
                    const bf16x8 al = *reinterpret_cast<const bf16x8*>(Am + 8 * 16);
                    const bf16x8 xi = *reinterpret_cast<const bf16x8*>(Am + 32 * 16);
                    f32x4_t c = acc2[m];
                    if constexpr (C::XPARTS == 2) {      // unsigned input rows: the own row's lo part (its ninth significant bit)
                        const bf16x8 xl = *reinterpret_cast<const bf16x8*>(Am + 40 * 16);
                        if constexpr (DEC) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wih[S], xl, c, 0, 0, 0);
                        else c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xl, wih[S], c, 0, 0, 0);
                    }
                    if constexpr (DEC) {
                        // transposed product (weights as the A operand): the lane ends with ONE tet (jcol + 16 m) and 4 consecutive output
                        // columns cs * 16 + 4 tq + r -- the shape the decoder's operand fragments are cut from; same products, same order
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wjh[S], al, c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wjl[S], ah, c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wil[S], xi, c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wjh[S], ah, c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wih[S], xi, c, 0, 0, 0);
                    } else {
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, wjh[S], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, wjl[S], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xi, wil[S], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, wjh[S], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xi, wih[S], c, 0, 0, 0);
                    }
                    acc2[m] = c;
                }
            }
            if constexpr (DEC) {
                // bias / BatchNorm(eval) / ReLU of columns cs * 16 + 4 tq + r, then the 8 consecutive columns a lane pair (tq, tq ^ 1) holds become
                // one decoder operand fragment: both lanes collect all 8 values (v_permlane16_swap), the even lane writes their bf16 hi parts, the odd lane the lo parts (y = hi + lo to 16 bits; y itself is never rounded
                // to bf16 -- the rounding next to the logits that the two-launch form pays)
                const int cb4 = cs * 16 + 4 * tq;
                const f32x4_t bbv = *reinterpret_cast<const f32x4_t*>(cbuf + 260 + cb4), scv = *reinterpret_cast<const f32x4_t*>(cbuf + 388 + cb4),
                              shv = *reinterpret_cast<const f32x4_t*>(cbuf + 516 + cb4);
                char* Yw = ybuf + (it & 1) * 16384 + (tq & 1) * 8192 + ((((cs >> 1) * 4 + (cs & 1) * 2 + (tq >> 1)) * 32) + jcol) * 16;
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    float v8[8];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float v = acc2[m][r] + bbv[r];
                        if (has_scale) v = __fmaf_rn(v, scv[r], shv[r]);
                        if (relu) v = fmaxf(v, 0.f);
                        const float pv = partner16(v);      // lane l ^ 16's value, no LDS
                        v8[r] = (tq & 1) ? pv : v;
                        v8[4 + r] = (tq & 1) ? v : pv;
                    }
                    uint32_t ph[4], pl[4];
#pragma unroll
                    for (int d = 0; d < 4; ++d) split2(v8[2 * d], v8[2 * d + 1], ph[d], pl[d]);
                    const bool lo_part = (tq & 1) != 0;
                    *reinterpret_cast<uint4*>(Yw + m * 256) = make_uint4(lo_part ? pl[0] : ph[0], lo_part ? pl[1] : ph[1], lo_part ? pl[2] : ph[2], lo_part ? pl[3] : ph[3]);
                }
            } else {
            // C/D layout: column jcol, rows 4*tq + r of the block; (col, col^1) pair up as above
            uint16_t* o = out + (tile * TILE + rg * 32) * ldo + (col & ~1);
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int r = 0; r < 4; r += 2) {
                    float v0 = acc2[m][r] + bb, v1 = acc2[m][r + 1] + bb;
                    if (has_scale) { v0 = __fmaf_rn(v0, sc, sh); v1 = __fmaf_rn(v1, sc, sh); }
                    if (relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); }
                    const float n0 = __shfl_xor(v0, 1), n1 = __shfl_xor(v1, 1);
                    const uint32_t pk = UBO ? (odd ? pack_ub(n1, v1) : pack_ub(v0, n0)) : (odd ? pack_bf16(n1, v1) : pack_bf16(v0, n0));
                    const int rr = m * 16 + 4 * tq + r + odd;
                    if (full || tile * TILE + rg * 32 + rr < n_dst) *reinterpret_cast<uint32_t*>(o + (int64_t)rr * ldo) = pk;
                }
            }
        }
    }
    if constexpr (DEC) {
        // drain: the last tile's decoder, the last two tiles' logits
        if (my_n >= 1) {
            tile_barrier();
            decoder_stage(my_n - 1);
            if (my_n >= 2 && w == 0) logits_stage(my_n - 2);
            tile_barrier();
            if (w == 0) logits_stage(my_n - 1);
        }
    }
}

template <int CIN_PAD, int COUT, int OCC, int PR, int NW = 4, int XF = 0, bool DEC = false, int UB = 0>
int launch_b(const int32_t* rowptr, const int32_t* src, const int32_t* eid, int64_t n_dst, const void* x, const void* xdst, int64_t ldx,
             int c_in, const float* ea, int64_t lde, const float* We, const float* be, const float* Wj, const float* bj, const float* Wi,
             const float* scale, const float* shift, int relu, uint16_t* out, int64_t ldo, hipStream_t stream, DecB dec = DecB{}) {
    using C = CfgB<CIN_PAD, COUT, NW, PR, XF, DEC, UB>;
    const int64_t ntiles = dgnn_cdiv(n_dst, C::TILE);
    const size_t smem = C::SMEM_BYTES;
    static bool attr_set[DGNN_MAX_DEVICES] = {};
    dgnn_allow_dynamic_lds(reinterpret_cast<const void*>(&k_sage_fused_bf16<CIN_PAD, COUT, NW, OCC, PR, XF, DEC, UB>), smem, attr_set);
    const int per_cu = (int)(160 * 1024 / smem) < OCC ? (int)(160 * 1024 / smem) : OCC;
    const int wg_max = DGNN_NUM_CU * (per_cu < 1 ? 1 : per_cu);
    int grid = (int)(ntiles < wg_max ? ntiles : wg_max);
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL((k_sage_fused_bf16<CIN_PAD, COUT, NW, OCC, PR, XF, DEC, UB>), dim3(grid), dim3(64 * NW), smem, stream, rowptr, src, eid, n_dst, x, xdst,
                       ldx, c_in, ea, lde, We, be, Wj, bj, Wi, scale, shift, relu, out, ldo, ntiles, dec);
    return dgnn_check_launch(DEC ? "sage_layer_fused_decoder_fwd_bf16" : "sage_layer_fused_fwd_bf16");
}

// out[r, c] = bf16(in[r, c]) for c < cols, 0 for cols <= c < cols_pad (row stride ld_out >= cols_pad)
__global__ void k_cast_f32_bf16(const float* __restrict__ in, int64_t ld_in, int64_t n, int cols, int cols_pad,
                                uint16_t* __restrict__ out, int64_t ld_out) {
    const int64_t total = n * (cols_pad / 2);
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = t / (cols_pad / 2);
        const int c = (int)(t - r * (cols_pad / 2)) * 2;
        const float a = c < cols ? in[r * ld_in + c] : 0.f, b = c + 1 < cols ? in[r * ld_in + c + 1] : 0.f;
        *reinterpret_cast<uint32_t*>(out + r * ld_out + c) = pack_bf16(a, b);
    }
}

// unsigned rows (value = bits << 15) -> plain bf16 (round to nearest even): for consumers outside the fused layers
__global__ void k_rows_unsigned_bf16(const uint16_t* __restrict__ in, int64_t ld_in, int64_t n, int cols, uint16_t* __restrict__ out, int64_t ld_out) {
    const int64_t total = n * (cols / 2);
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = t / (cols / 2);
        const int c = (int)(t - r * (cols / 2)) * 2;
        const uint32_t u = *reinterpret_cast<const uint32_t*>(in + r * ld_in + c);
        *reinterpret_cast<uint32_t*>(out + r * ld_out + c) = pack_bf16(ub_lo(u), ub_hi(u));
    }
}

__global__ void k_cast_bf16_f32(const uint16_t* __restrict__ in, int64_t ld_in, int64_t n, int cols, float* __restrict__ out,
                                int64_t ld_out) {
    const int64_t total = n * cols;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = t / cols;
        const int c = (int)(t - r * cols);
        out[r * ld_out + c] = __builtin_bit_cast(float, (uint32_t)in[r * ld_in + c] << 16);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Fused decoder on bf16 activations (reference :180-187 applied at :350-351): logits = W3 . relu((W0 . y + b0)*scale + shift) + b3.
// 8 independent wavefronts per workgroup, each streams its own 32-row tiles: a y row segment of 8 bf16 IS the MFMA
// A operand of its lane (row l31, k-group g), so rows go global -> registers -> matrix core with no staging; W0 (bf16) is
// parked in LDS in B-operand order once per workgroup; the 32 x 64 hidden tile goes through a wave-private LDS strip and
// lane (row, o) finishes logit o with 64 fp32 FMAs.  Reads 2*K bytes per row, writes 4*n_out (logits stay fp32).
// ---------------------------------------------------------------------------------------------------------------
constexpr int HIDB = 64;

template <int K, int PR>
__global__ void __launch_bounds__(512) k_decoder_rows_bf16(const uint16_t* __restrict__ y, int64_t ldy, int64_t M,
                                                           const float* __restrict__ W0, const float* __restrict__ b0,
                                                           const float* __restrict__ scale, const float* __restrict__ shift,
                                                           const float* __restrict__ W3, const float* __restrict__ b3, int n_out,
                                                           float* __restrict__ out, int64_t ldo) {
    constexpr int NS = K / 16, LDH = HIDB + 4;
    extern __shared__ __attribute__((aligned(16))) char dsmb[];
    uint4* const Bs = reinterpret_cast<uint4*>(dsmb);                         // [parts][2 col blocks][NS][64 lanes] x 16 B
    float* const W3s = reinterpret_cast<float*>(dsmb + (PR ? 2 : 1) * 2 * NS * 64 * 16);     // [2][HIDB]
    float* const Hall = W3s + 2 * HIDB;                                       // [8 waves][32][LDH]
    const int lane = lane_id(), w = wave_id_uniform();
    const int g = lane >> 5, l31 = lane & 31;
    for (int e = threadIdx.x; e < 2 * NS * 64; e += blockDim.x) {
        const int ln = e & 63, S = (e >> 6) % NS, cblk = e / (64 * NS);
        const float* wr = W0 + (int64_t)(cblk * 32 + (ln & 31)) * K + 16 * S + 8 * (ln >> 5);
        uint32_t ph[4], pl[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) split2(wr[2 * d], wr[2 * d + 1], ph[d], pl[d]);
        Bs[(cblk * NS + S) * 64 + ln] = make_uint4(ph[0], ph[1], ph[2], ph[3]);
        if (PR) Bs[2 * NS * 64 + (cblk * NS + S) * 64 + ln] = make_uint4(pl[0], pl[1], pl[2], pl[3]);   // W0 = hi + lo: y . W0 to 16 bits
    }
    for (int e = threadIdx.x; e < 2 * HIDB; e += blockDim.x) W3s[e] = (e / HIDB) < n_out ? W3[e] : 0.f;
    float bb[2], sc[2], sh[2];
#pragma unroll
    for (int cblk = 0; cblk < 2; ++cblk) {
        const int col = cblk * 32 + l31;
        bb[cblk] = b0 ? b0[col] : 0.f;
        sc[cblk] = scale ? scale[col] : 1.f;
        sh[cblk] = scale ? shift[col] : 0.f;
    }
    const bool has_scale = scale != nullptr;
    const int po = g;
    const float b3v = (b3 && po < n_out) ? b3[po] : 0.f;
    float* const Hs = Hall + w * 32 * LDH;
    __syncthreads();

    const int64_t ntiles = (M + 31) / 32, stride = (int64_t)gridDim.x * 8;
    uint4 a[NS], an[NS];
    auto row_ptr = [&](int64_t tile) {
        const int64_t r = tile * 32 + l31;
        return y + (r < M ? r : M - 1) * ldy + 8 * g;
    };
    int64_t tile = (int64_t)blockIdx.x * 8 + w;
    if (tile < ntiles) {
        const uint16_t* p = row_ptr(tile);
#pragma unroll
        for (int S = 0; S < NS; ++S) a[S] = *reinterpret_cast<const uint4*>(p + 16 * S);
    }
    for (; tile < ntiles; tile += stride) {
        const bool more = tile + stride < ntiles;
        const uint16_t* pn = row_ptr(more ? tile + stride : tile);
        if (more) {
#pragma unroll
            for (int S = 0; S < NS; ++S) an[S] = *reinterpret_cast<const uint4*>(pn + 16 * S);   // flies under this tile's work
        }
        f32x16 acc[2];
#pragma unroll
        for (int cblk = 0; cblk < 2; ++cblk)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[cblk][i] = 0.f;
#pragma unroll
        for (int S = 0; S < NS; ++S)
#pragma unroll
            for (int cblk = 0; cblk < 2; ++cblk) {
                if (PR)
                    acc[cblk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[S]),
                                                                        __builtin_bit_cast(bf16x8, Bs[2 * NS * 64 + (cblk * NS + S) * 64 + lane]),
                                                                        acc[cblk], 0, 0, 0);
                acc[cblk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[S]),
                                                                    __builtin_bit_cast(bf16x8, Bs[(cblk * NS + S) * 64 + lane]), acc[cblk], 0, 0, 0);
            }
#pragma unroll
        for (int cblk = 0; cblk < 2; ++cblk)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[cblk][r] + bb[cblk];
                if (has_scale) v = __fmaf_rn(v, sc[cblk], sh[cblk]);
                Hs[((r & 3) + 8 * (r >> 2) + 4 * g) * LDH + cblk * 32 + l31] = fmaxf(v, 0.f);
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        float sacc = b3v;
        const float* hr = Hs + l31 * LDH;
        const float* w3 = W3s + po * HIDB;
#pragma unroll
        for (int c = 0; c < HIDB; c += 4) {
            const f32x4 hv = *reinterpret_cast<const f32x4*>(hr + c);
            const f32x4 wv = *reinterpret_cast<const f32x4*>(w3 + c);
#pragma unroll
            for (int j = 0; j < 4; ++j) sacc = __fmaf_rn(hv[j], wv[j], sacc);
        }
        const int64_t row = tile * 32 + l31;
        if (row < M && po < n_out) out[row * ldo + po] = sacc;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (more) {
#pragma unroll
            for (int S = 0; S < NS; ++S) a[S] = an[S];
        }
    }
}

}  // namespace

extern "C" int dgnn_cast_f32_to_bf16(const float* in, int64_t ld_in, int64_t n, int cols, int cols_pad, uint16_t* out, int64_t ld_out,
                                     void* stream) {
    DGNN_REQUIRE(n >= 0 && cols >= 0 && cols_pad >= cols && cols_pad % 2 == 0 && ld_out >= cols_pad && ld_out % 2 == 0, DGNN_E_INVALID,
                 "cast_f32_to_bf16: bad sizes");
    if (n == 0 || cols_pad == 0) return DGNN_OK;
    DGNN_REQUIRE(in && out && ((uintptr_t)out % 4) == 0, DGNN_E_INVALID, "cast_f32_to_bf16: null / unaligned pointer");
    hipLaunchKernelGGL(k_cast_f32_bf16, dim3(dgnn_grid_cap(dgnn_cdiv(n * (cols_pad / 2), 256))), dim3(256), 0, (hipStream_t)stream, in, ld_in, n,
                       cols, cols_pad, out, ld_out);
    return dgnn_check_launch("cast_f32_to_bf16");
}

extern "C" int dgnn_rows_unsigned_to_bf16(const uint16_t* in, int64_t ld_in, int64_t n, int cols, uint16_t* out, int64_t ld_out, void* stream) {
    DGNN_REQUIRE(n >= 0 && cols >= 0 && cols % 2 == 0 && ld_in % 2 == 0 && ld_out % 2 == 0, DGNN_E_INVALID, "rows_unsigned_to_bf16: bad sizes (even widths / strides)");
    if (n == 0 || cols == 0) return DGNN_OK;
    DGNN_REQUIRE(in && out && (((uintptr_t)in | (uintptr_t)out) % 4) == 0, DGNN_E_INVALID, "rows_unsigned_to_bf16: null / unaligned pointer");
    hipLaunchKernelGGL(k_rows_unsigned_bf16, dim3(dgnn_grid_cap(dgnn_cdiv(n * (cols / 2), 256))), dim3(256), 0, (hipStream_t)stream, in, ld_in, n, cols, out, ld_out);
    return dgnn_check_launch("rows_unsigned_to_bf16");
}

extern "C" int dgnn_cast_bf16_to_f32(const uint16_t* in, int64_t ld_in, int64_t n, int cols, float* out, int64_t ld_out, void* stream) {
    DGNN_REQUIRE(n >= 0 && cols >= 0, DGNN_E_INVALID, "cast_bf16_to_f32: bad sizes");
    if (n == 0 || cols == 0) return DGNN_OK;
    DGNN_REQUIRE(in && out, DGNN_E_INVALID, "cast_bf16_to_f32: null pointer");
    hipLaunchKernelGGL(k_cast_bf16_f32, dim3(dgnn_grid_cap(dgnn_cdiv(n * cols, 256))), dim3(256), 0, (hipStream_t)stream, in, ld_in, n, cols, out,
                       ld_out);
    return dgnn_check_launch("cast_bf16_to_f32");
}

int dgnn_sage_layer_fused_ws16_try(const int32_t* rowptr, const int32_t* src, const int32_t* eid, int64_t n_dst, const uint16_t* x_src, const uint16_t* x_dst,
                                   int64_t ldx, int c_in, const float* edge_attr, int64_t lde, const float* We, const float* be, const float* Wj, const float* bj,
                                   const float* Wi, const float* scale, const float* shift, int relu, int c_out, uint16_t* out, int64_t ldo, hipStream_t stream);   // fused_ws.hip

extern "C" int dgnn_sage_layer_fused_fwd_bf16(const int32_t* rowptr, const int32_t* src, const int32_t* eid, int64_t n_dst,
                                              const void* x_src, int x_f32, const void* x_dst, int64_t ldx, int c_in, const float* edge_attr,
                                              int64_t lde, int f_e, const float* We, const float* be, const float* Wj, const float* bj,
                                              const float* Wi, const float* scale, const float* shift, int relu, int c_out, uint16_t* out,
                                              int64_t ldo, int mode, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DGNN_REQUIRE(n_dst >= 0 && c_in > 0 && c_out > 0, DGNN_E_INVALID, "sage_layer_fused_fwd_bf16: bad sizes");
    if (n_dst == 0) return DGNN_OK;
    DGNN_REQUIRE(rowptr && src && x_src && edge_attr && We && be && Wj && Wi && out, DGNN_E_INVALID, "sage_layer_fused_fwd_bf16: null pointer");
    DGNN_REQUIRE((scale == nullptr) == (shift == nullptr), DGNN_E_INVALID, "sage_layer_fused_fwd_bf16: scale/shift must come together");
    if (x_dst == nullptr) x_dst = x_src;
    DGNN_REQUIRE(f_e == FE && lde == FE && ((uintptr_t)edge_attr % 16) == 0, DGNN_E_UNSUPPORTED,
                 "sage_layer_fused_fwd_bf16: needs f_e == 20, packed 16-byte aligned edge rows");
    DGNN_REQUIRE(n_dst * ldx < ((int64_t)1 << 31), DGNN_E_UNSUPPORTED, "sage_layer_fused_fwd_bf16: activations beyond 2^31 elements");
    const int cin_pad = c_in <= 32 ? 32 : (c_in <= 64 ? 64 : 128);
    const int nb = cin_pad / 16;
    DGNN_REQUIRE(c_in <= 128 && (c_out == 64 || c_out == 128), DGNN_E_UNSUPPORTED,
                 "sage_layer_fused_fwd_bf16: supports c_in <= 128 and c_out in {64,128} (got %d -> %d)", c_in, c_out);
    // a lane reads nb contiguous bf16 of a row: rows must be aligned to that, and every lane's piece must lie inside the row
    DGNN_REQUIRE(c_in % nb == 0, DGNN_E_UNSUPPORTED, "sage_layer_fused_fwd_bf16: c_in must be a multiple of %d", nb);
    DGNN_REQUIRE(x_f32 || (ldx % nb == 0 && (((uintptr_t)x_src | (uintptr_t)x_dst) % (2 * nb)) == 0), DGNN_E_UNSUPPORTED,
                 "sage_layer_fused_fwd_bf16: the row stride must be a multiple of %d, rows %d-byte aligned", nb, 2 * nb);
    DGNN_REQUIRE(!x_f32 || (cin_pad == 32 && (((uintptr_t)x_src | (uintptr_t)x_dst) % 4) == 0), DGNN_E_UNSUPPORTED,
                 "sage_layer_fused_fwd_bf16: fp32 input rows are supported for c_in <= 32 (the first layer)");
    DGNN_REQUIRE(ldo % 2 == 0 && ((uintptr_t)out % 4) == 0, DGNN_E_UNSUPPORTED, "sage_layer_fused_fwd_bf16: out rows must be 4-byte aligned");
    const bool ub_in = (mode & DGNN_BF16_ROWS_IN_UNSIGNED) != 0, ub_out = (mode & DGNN_BF16_ROWS_OUT_UNSIGNED) != 0;
    mode &= ~(DGNN_BF16_ROWS_IN_UNSIGNED | DGNN_BF16_ROWS_OUT_UNSIGNED);
    DGNN_REQUIRE(mode == DGNN_BF16_SINGLE || mode == DGNN_BF16_COMPENSATED, DGNN_E_INVALID, "sage_layer_fused_fwd_bf16: bad mode %d", mode);
    // unsigned rows: compensated arithmetic; the output format needs the ReLU (values >= 0); a layer either keeps the format (in and out) or, reading
    // fp32 rows, starts it
    DGNN_REQUIRE(!(ub_in || ub_out) || mode == DGNN_BF16_COMPENSATED, DGNN_E_UNSUPPORTED, "sage_layer_fused_fwd_bf16: unsigned rows need the compensated mode");
    DGNN_REQUIRE(!ub_out || relu, DGNN_E_INVALID, "sage_layer_fused_fwd_bf16: unsigned output rows need relu (values >= 0)");
    DGNN_REQUIRE(x_f32 ? !ub_in : ub_in == ub_out, DGNN_E_UNSUPPORTED, "sage_layer_fused_fwd_bf16: a layer on bf16 rows keeps the row format (in == out)");
    const bool ub = ub_out;
    if (ub && ub_in && !x_f32 && c_out == 128 && (c_in == 128 || c_in == 64)) {
        // round 5: the wave-specialised kernel (fused_ws.hip) reads and writes unsigned 16-bit rows too (DGNN_WS=0 / DGNN_WS_16=0: the kernels below)
        const int rc = dgnn_sage_layer_fused_ws16_try(rowptr, src, eid, n_dst, static_cast<const uint16_t*>(x_src), static_cast<const uint16_t*>(x_dst), ldx, c_in,
                                                      edge_attr, lde, We, be, Wj, bj, Wi, scale, shift, relu, c_out, out, ldo, stream);
        if (rc != DGNN_E_UNSUPPORTED) return rc;
    }
#define GOB(CP, CO, OCC)                                                                                                                  \
    do {                                                                                                                                  \
        if (ub)                                                                                                                           \
            return launch_b<CP, CO, OCC, 1, 4, 0, false, 3>(rowptr, src, eid, n_dst, x_src, x_dst, ldx, c_in, edge_attr, lde, We, be, Wj, bj, Wi, \
                                                            scale, shift, relu, out, ldo, stream);                                        \
        if (mode == DGNN_BF16_COMPENSATED)                                                                                                \
            return launch_b<CP, CO, OCC, 1>(rowptr, src, eid, n_dst, x_src, x_dst, ldx, c_in, edge_attr, lde, We, be, Wj, bj, Wi, scale,  \
                                            shift, relu, out, ldo, stream);                                                               \
        return launch_b<CP, CO, OCC, 0>(rowptr, src, eid, n_dst, x_src, x_dst, ldx, c_in, edge_attr, lde, We, be, Wj, bj, Wi, scale,      \
                                        shift, relu, out, ldo, stream);                                                                   \
    } while (0)
    if (x_f32) {
        const bool pr = mode == DGNN_BF16_COMPENSATED;
#define GOX(CO, PRV, UBV) return launch_b<32, CO, 2, PRV, 4, 1, false, UBV>(rowptr, src, eid, n_dst, x_src, x_dst, ldx, c_in, edge_attr, lde, We, be, Wj, bj, Wi, \
                                                                          scale, shift, relu, out, ldo, stream)
        if (c_out == 64) { if (ub) GOX(64, 1, 2); if (pr) GOX(64, 1, 0); else GOX(64, 0, 0); }
        if (ub) GOX(128, 1, 2);
        if (pr) GOX(128, 1, 0); else GOX(128, 0, 0);
#undef GOX
    }
    if (cin_pad == 32) { if (c_out == 64) GOB(32, 64, 2); else GOB(32, 128, 2); }
    if (cin_pad == 64) { if (c_out == 64) GOB(64, 64, 2); else GOB(64, 128, 2); }
    if (c_out == 64) GOB(128, 64, 2);
    // 128 -> 128: the compensated form keeps four weight parts resident -> eight-wave workgroups with 16-column slices
    if (ub)
        return launch_b<128, 128, 1, 1, 8, 0, false, 3>(rowptr, src, eid, n_dst, x_src, x_dst, ldx, c_in, edge_attr, lde, We, be, Wj, bj, Wi, scale, shift, relu,
                                                        out, ldo, stream);
    if (mode == DGNN_BF16_COMPENSATED)
        return launch_b<128, 128, 1, 1, 8>(rowptr, src, eid, n_dst, x_src, x_dst, ldx, c_in, edge_attr, lde, We, be, Wj, bj, Wi, scale, shift, relu,
                                           out, ldo, stream);
    GOB(128, 128, 2);
#undef GOB
}

// The LAST conv layer in bf16 storage with the decoder inside its launch (compensated arithmetic): the finished tile never leaves the compute unit
// and is never rounded to bf16 -- it reaches the decoder's matrix products as (hi, lo) pairs -- and only fp32 logits are written.
extern "C" int dgnn_sage_layer_fused_decoder_fwd_bf16(const int32_t* rowptr, const int32_t* src, const int32_t* eid, int64_t n_dst, const uint16_t* x_src,
                                                      const uint16_t* x_dst, int64_t ldx, int c_in, const float* edge_attr, int64_t lde, int f_e,
                                                      const float* We, const float* be, const float* Wj, const float* bj, const float* Wi,
                                                      const float* scale, const float* shift, int relu, int c_out, const float* W0, const float* b0,
                                                      const float* scale1, const float* shift1, int c_hidden, const float* W3, const float* b3,
                                                      int n_logits, float* logits, int mode, void* stream_) {
    DGNN_REQUIRE(n_dst >= 0 && c_in > 0 && c_out > 0, DGNN_E_INVALID, "sage_layer_fused_decoder_fwd_bf16: bad sizes");
    if (n_dst == 0) return DGNN_OK;
    DGNN_REQUIRE(rowptr && src && x_src && edge_attr && We && be && Wj && Wi && W0 && b0 && W3 && b3 && logits, DGNN_E_INVALID,
                 "sage_layer_fused_decoder_fwd_bf16: null pointer");
    DGNN_REQUIRE((scale == nullptr) == (shift == nullptr) && (scale1 == nullptr) == (shift1 == nullptr), DGNN_E_INVALID,
                 "sage_layer_fused_decoder_fwd_bf16: scale/shift must come together");
    if (x_dst == nullptr) x_dst = x_src;
    const bool ub_in = (mode & DGNN_BF16_ROWS_IN_UNSIGNED) != 0;
    mode &= ~DGNN_BF16_ROWS_IN_UNSIGNED;
    const bool ok = mode == DGNN_BF16_COMPENSATED && c_out == 128 && c_in > 64 && c_in <= 128 && c_in % 8 == 0 && f_e == FE && lde == FE && c_hidden == 64 &&
                    n_logits == 2 && ((uintptr_t)edge_attr % 16) == 0 && ldx % 8 == 0 && (((uintptr_t)x_src | (uintptr_t)x_dst) % 16) == 0 &&
                    ((uintptr_t)logits % 8) == 0 && n_dst * ldx < ((int64_t)1 << 31);
    if (!ok) return DGNN_E_UNSUPPORTED;     // the caller runs dgnn_sage_layer_fused_fwd_bf16 and dgnn_decoder_fused_fwd_bf16
    if (ub_in)
        return launch_b<128, 128, 1, 1, 8, 0, true, 1>(rowptr, src, eid, n_dst, x_src, x_dst, ldx, c_in, edge_attr, lde, We, be, Wj, bj, Wi, scale, shift, relu, nullptr,
                                                       0, (hipStream_t)stream_, DecB{W0, b0, scale1, shift1, W3, b3, logits});
    return launch_b<128, 128, 1, 1, 8, 0, true>(rowptr, src, eid, n_dst, x_src, x_dst, ldx, c_in, edge_attr, lde, We, be, Wj, bj, Wi, scale, shift, relu, nullptr, 0,
                                                (hipStream_t)stream_, DecB{W0, b0, scale1, shift1, W3, b3, logits});
}

extern "C" int dgnn_decoder_fused_fwd_bf16(const uint16_t* y, int64_t ldy, int64_t M, int k, const float* W0, const float* b0, const float* scale,
                                           const float* shift, int hidden, const float* W3, const float* b3, int n_out, float* out, int64_t ldo,
                                           int mode, void* stream) {
    DGNN_REQUIRE(M >= 0 && k > 0 && hidden > 0 && n_out > 0, DGNN_E_INVALID, "decoder_fused_fwd_bf16: bad sizes");
    if (M == 0) return DGNN_OK;
    DGNN_REQUIRE(y && W0 && W3 && out, DGNN_E_INVALID, "decoder_fused_fwd_bf16: null pointer");
    DGNN_REQUIRE((scale == nullptr) == (shift == nullptr), DGNN_E_INVALID, "decoder_fused_fwd_bf16: scale/shift must come together");
    DGNN_REQUIRE(k == 128 && hidden == HIDB && n_out <= 2 && ((uintptr_t)y % 16) == 0 && ldy % 8 == 0, DGNN_E_UNSUPPORTED,
                 "decoder_fused_fwd_bf16: supports 128 -> 64 -> {1,2} on 16-byte aligned rows (got %d -> %d -> %d)", k, hidden, n_out);
    constexpr int K = 128, LDH = HIDB + 4;
    DGNN_REQUIRE(mode == DGNN_BF16_SINGLE || mode == DGNN_BF16_COMPENSATED, DGNN_E_INVALID, "decoder_fused_fwd_bf16: bad mode %d", mode);
    const int parts = mode == DGNN_BF16_COMPENSATED ? 2 : 1;
    const size_t smem = parts * 2 * (K / 16) * 64 * 16 + 2 * HIDB * 4 + 8 * 32 * LDH * 4;
    const int64_t nt = dgnn_cdiv(M, 32);
    const int grid = (int)(dgnn_cdiv(nt, 8) < DGNN_NUM_CU ? dgnn_cdiv(nt, 8) : DGNN_NUM_CU);
    if (parts == 2) {
        static bool attr_set[DGNN_MAX_DEVICES] = {};
        dgnn_allow_dynamic_lds(reinterpret_cast<const void*>(&k_decoder_rows_bf16<128, 1>), smem, attr_set);
        hipLaunchKernelGGL((k_decoder_rows_bf16<128, 1>), dim3(grid), dim3(512), smem, (hipStream_t)stream, y, ldy, M, W0, b0, scale, shift, W3,
                           b3, n_out, out, ldo);
    } else {
        static bool attr_set[DGNN_MAX_DEVICES] = {};
        dgnn_allow_dynamic_lds(reinterpret_cast<const void*>(&k_decoder_rows_bf16<128, 0>), smem, attr_set);
        hipLaunchKernelGGL((k_decoder_rows_bf16<128, 0>), dim3(grid), dim3(512), smem, (hipStream_t)stream, y, ldy, M, W0, b0, scale, shift, W3,
                           b3, n_out, out, ldo);
    }
    return dgnn_check_launch("decoder_fused_fwd_bf16");
}
