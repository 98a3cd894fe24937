// Wide conv layers (C_in in {128, 256, 512}, C_out in {256, 512, 1024}) on SPLIT ROWS  (round 5).
//
// Reference: SAGEConv.forward, learning/surfaceNetStaticEdgeFilters.py:66-96, at the widths the reference's real configs use
// (configs/eth.yaml:56, aerial.yaml:57, terrestrial.yaml:56: [64,128,256,512]; configs/modelnet.yaml:56, shapenet.yaml:56: [128,256,512,1024]).
// At these widths a layer is its dense product [a | x_i] . [Wj | Wi]^T, and rounds 2-4 found the fp32-class GEMM (fp16 two-part form, 3 matrix
// products per fp32 product) bound by MOVING and SPLITTING fp32 operands: a pass over A for the row scales, 128-byte pieces of fp32 rows, ~320 split /
// address instructions per 48 matrix instructions (profiles/r04j_wide_gemm_experiments.md).  Here every wide activation lives in HBM already in the
// form the matrix cores eat:
//
//   SPLIT ROW ("SR"): a row of C channels = C/32 chunks of 128 bytes; chunk q = [hi x 32 | lo x 32] fp16 of x * s, s a power of two PER GROUP of
//   256 channels (8 chunks; one fp32 per row and group, next to the rows), hi = RN16(x s), lo = RN16(x s - hi): 22 significand bits, 4 bytes per
//   element -- what fp32 weighs.  Position p of a chunk holds channel 32 q + PI(p), PI(p) = (p & 3) | (p >> 4) << 2 | ((p >> 2) & 3) << 3: the order
//   in which a lane of the TRANSPOSED 32x32x16 product (weights as the A operand) holds its 16 finished values, so the producing epilogue stores
//   32 contiguous bytes per lane and part; the K order of every consumer's weights carries the same permutation (prepared once per model), which
//   makes it free.  A group whose largest magnitude is below 2^-112 is stored as zeros with the scale marker 2^127.
//
//   k_agg_sr    a = mean_j x_j * phi_j (reference :75-80, :89-96) for 4 cells per wavefront step: phi = [A | 1] . [We ; be] on v_mfma_f32_16x16x32_f16
//               (per-edge power-of-two scales, 3 products: the fused layers' filter product), neighbour rows gathered as 16-byte pieces of split rows
//               (or fp32 rows behind a fused layer), the in-order 4-term sum in registers, the finished row scaled per 256 channels, split and
//               written as a split row.  Reads 4 rows + 320 B of attributes, writes one row: no fp32 `a` in HBM, no conversion pass.
//   k_gemm_sr   out = act(([a | x_i] . [Wj | Wi]^T + b) * scale + shift): 256 cells x 256 output channels per 512-thread workgroup, both operands
//               global -> LDS by DMA (16 bytes per lane, XOR-swizzled 128-byte rows, two buffers, one barrier per 32-wide K chunk), no VALU in the
//               loop but the products; the accumulator of a cell changes units where the K walk crosses into a group with another scale (one exact
//               multiplication by a power of two per accumulator register, the unit only ever follows the LARGEST group seen: no overflow; a group
//               more than 2^40 below it is dropped -- its whole contribution is below the fp32 rounding of the sum).  Epilogue, transposed: a lane
//               owns ONE cell and 64 of the tile's channels: bias / BatchNorm / ReLU, row maximum in registers (+ one lane swap, + one LDS word from
//               the other channel half), scale, split, 32-byte stores -- the next layer's operand, written by its producer.
//   The decoder's Linear(h3 -> h3/2) + BN + ReLU is the same GEMM on one operand with fp32 output.
//
// Arithmetic: the fp16 two-part form of fused_common.h throughout (22 significand bits per operand, lo.lo dropped, fp32 accumulation); scaling
// groups: per cell row and 256 channels, per edge (filter operand), per weight row / weight matrix [We | be].  A cell's result depends on its own
// inputs only (whole-scene, partitioned and differently tiled runs stay bit-identical).
#include <stdlib.h>

#include <mutex>

#include "common.h"
#include "fused_common.h"

namespace {
using namespace fused;

typedef float f32x4_t __attribute__((ext_vector_type(4)));
#define H8(v) __builtin_bit_cast(f16x8, v)
template <int V> struct IC { static constexpr int value = V; };

constexpr int SRB = 128;   // bytes of one chunk of a split row: [hi x 32 | lo x 32] fp16
constexpr int TM = 256;    // cells per GEMM tile
constexpr int TN = 256;    // output channels per GEMM tile
constexpr int GT = 512;    // threads of the GEMM workgroup

__device__ __forceinline__ int sr_chan(int pos) { return (pos & 3) | ((pos >> 4) << 2) | (((pos >> 2) & 3) << 3); }
__device__ __forceinline__ float f_of(uint32_t b) { return __builtin_bit_cast(float, b); }
__device__ __forceinline__ uint32_t b_of(float f) { return __builtin_bit_cast(uint32_t, f); }
constexpr uint32_t SR_ZERO_BITS = 254u << 23;   // 2^127: the scale of an all-zero (flushed) group
// scale of a group from the bit pattern of its largest magnitude: s_store goes next to the row, s_mul multiplies the values (0 for a flushed group)
__device__ __forceinline__ void sr_scale(uint32_t maxbits, float& s_store, float& s_mul) {
    uint32_t E = maxbits >> 23;
    if (E <= 14u) {
        s_store = f_of(SR_ZERO_BITS);
        s_mul = 0.f;
        return;
    }
    E = E > 254u ? 254u : E;
    s_store = s_mul = f_of((268u - E) << 23);
}
// a 16-byte store of a split row: the row is written once and read by the NEXT launch -- streamed past the caches (nt) it does not evict the rows this
// launch gathers / re-reads from L2 (the fused layers gained 1-7 % from the same hint)
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st16(char* p, uint32_t a, uint32_t b, uint32_t c, uint32_t d, bool nt) {
    const u32x4_t v = {a, b, c, d};
    if (nt) __builtin_nontemporal_store(v, reinterpret_cast<u32x4_t*>(p));
    else *reinterpret_cast<u32x4_t*>(p) = v;
}
__device__ __forceinline__ float pow2_inv(float s) { return f_of((254u << 23) - b_of(s)); }   // 1 / s for a power of two in [2^-126, 2^126]

// ---- fp32 rows -> split rows (weights at prepare time; activations at a boundary; tests) ----------------------------------------------------
// one wavefront per row of [A1 | A2] (k1, k2 multiples of 32): `gch` chunks per scale group (0: the whole row is one group)
__global__ void __launch_bounds__(256) k_sr_pack(const float* __restrict__ A1, int64_t ld1, int k1, const float* __restrict__ A2, int64_t ld2, int k2,
                                                 int64_t rows, int gch, char* __restrict__ dst, int64_t dst_row_bytes, float* __restrict__ scales, int ng) {
    const int lane = lane_id();
    const int nch1 = k1 / 32, nch = nch1 + k2 / 32;
    if (gch <= 0) gch = nch;
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    for (int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += nwaves) {
        const float* p1 = A1 + row * ld1;
        const float* p2 = A2 ? A2 + row * ld2 : nullptr;
        char* d = dst + row * dst_row_bytes;
        for (int g = 0; g * gch < nch; ++g) {
            const int q0 = g * gch, q1 = min(nch, q0 + gch);
            uint32_t m = 0u;
            for (int e = q0 * 32 + lane; e < q1 * 32; e += 64) m = umax(m, absbits(e < k1 ? p1[e] : p2[e - k1]));
            m = wave_umax(m);
            float s_store, s_mul;
            sr_scale(m, s_store, s_mul);
            if (lane == 0) scales[row * ng + g] = s_store;
            const int pr = lane & 15, sub = lane >> 4;
            for (int q = q0 + sub; q < q1; q += 4) {
                const float* src_ = q < nch1 ? p1 + q * 32 : p2 + (q - nch1) * 32;
                const int c0 = sr_chan(2 * pr);     // positions 2 pr, 2 pr + 1 hold channels c0, c0 + 1
                uint32_t hi, lo;
                split2h(src_[c0] * s_mul, src_[c0 + 1] * s_mul, hi, lo);
                uint32_t* o = reinterpret_cast<uint32_t*>(d + (int64_t)q * SRB) + pr;
                o[0] = hi;
                o[16] = lo;
            }
        }
    }
}

// split rows -> fp32 rows (tests; a consumer outside the wide kernels)
__global__ void __launch_bounds__(256) k_sr_unpack(const char* __restrict__ src, int64_t row_bytes, const float* __restrict__ scales, int ng, int gch,
                                                   int C, int64_t rows, float* __restrict__ out, int64_t ldo) {
    const int64_t total = rows * (C / 2);
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = t / (C / 2);
        const int e = (int)(t - row * (C / 2));      // pair index within the row: chunk q = e / 16, pair pr = e % 16
        const int q = e >> 4, pr = e & 15;
        const uint32_t* p = reinterpret_cast<const uint32_t*>(src + row * row_bytes + (int64_t)q * SRB) + pr;
        const uint32_t hi = p[0], lo = p[16];
        const float s = scales[row * ng + (gch > 0 ? q / gch : 0)];
        const float inv = b_of(s) == SR_ZERO_BITS ? 0.f : pow2_inv(s);
        const f16x2_t h = __builtin_bit_cast(f16x2_t, hi), l = __builtin_bit_cast(f16x2_t, lo);
        const int c0 = sr_chan(2 * pr);
        float* o = out + row * ldo + q * 32 + c0;
        o[0] = ((float)h[0] + (float)l[0]) * inv;
        o[1] = ((float)h[1] + (float)l[1]) * inv;
    }
}

// ---- the filter operand [We^T ; be ; 0] of k_agg_sr, prepared once per set of weights ----------------------------------------------------------
// k_agg_sr walks a row in PASSES of 16 lanes x NB positions: C = 128: one pass, NB = 8; C = 256: one pass, NB = 16; C = 512: two passes of 256
// channels, NB = 16 (a launch per pass: the 4 neighbour rows of a pass are 1 KB each, so the rows the cells in flight on an XCD share stay in its L2).
// buffer: 16-byte header (sWe, 1 / sWe, 0, 0), then per pass the entries (cb, part, g, j) x 16 bytes: cb < NB (the lane's cb-th position), part
// hi / lo, k-group g < 3 (k = 8 g .. 8 g + 7: attributes 0..19, the bias at k = 20, zeros), j < 16 (lane): position P = 256 pass + NB j + cb of the
// row, i.e. channel 32 (P / 32) + PI(P % 32)
__device__ __host__ inline int agg_nb(int C) { return C == 128 ? 8 : 16; }
__device__ __host__ inline int agg_passes(int C) { return C / (16 * agg_nb(C)); }

__global__ void __launch_bounds__(256) k_sr_prepare_filter(const float* __restrict__ We, const float* __restrict__ be, int C, char* __restrict__ buf) {
    __shared__ uint32_t mx;
    if (threadIdx.x == 0) mx = 0u;
    __syncthreads();
    uint32_t m = 0u;
    for (int e = threadIdx.x; e < C * FE; e += blockDim.x) m = umax(m, absbits(We[e]));
    for (int e = threadIdx.x; e < C; e += blockDim.x) m = umax(m, absbits(be[e]));
    m = wave_umax(m);
    if (lane_id() == 0) atomicMax(&mx, m);
    __syncthreads();
    float sWe, inv_sWe;
    pow2_scales(mx, sWe, inv_sWe);
    if (threadIdx.x == 0) *reinterpret_cast<f32x4_t*>(buf) = f32x4_t{sWe, inv_sWe, 0.f, 0.f};
    const int NB = agg_nb(C), np = agg_passes(C);
    for (int e = threadIdx.x; e < np * NB * 48; e += blockDim.x) {
        const int pass = e / (NB * 48), e1 = e - pass * NB * 48;
        const int cb = e1 / 48, gj = e1 - cb * 48, g = gj >> 4, j = gj & 15;
        const int P = pass * 256 + NB * j + cb, c = (P & ~31) + sr_chan(P & 31);
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
        uint4* dst = reinterpret_cast<uint4*>(buf + 16 + ((int64_t)pass * NB * 2 * 48 + (cb * 2) * 48 + gj) * 16);
        dst[0] = make_uint4(ph[0], ph[1], ph[2], ph[3]);
        dst[48] = make_uint4(pl[0], pl[1], pl[2], pl[3]);
    }
}

// ---- k_agg_sr ----------------------------------------------------------------------------------------------------------------------------------
// Lane (j = lane & 15, t = lane >> 4) of a wavefront step owns NB contiguous POSITIONS [P0, P0 + NB), P0 = 256 pass + NB j, of cell t's row (of each
// of its 4 neighbour rows, and of the finished row); the filter product's C/D layout puts the 4 in-edges of cell t into the 4 accumulator registers
// of the lanes (., t), so sum_j x_j * phi_j is an in-lane, in-order sum (plan order = the reference's CPU scatter order).  The 16 lanes of a cell
// hold one scale group (256 channels, or the 128 of the narrowest layer).
// XSR: the source rows are split rows (row stride xrb bytes, scales xs [n_src][ngx]); else fp32 rows (ldx floats) -- the layer behind a fused layer --
// and, when xo != NULL, the cell's OWN row is also written as a split row (the x_i operand of this layer's GEMM).
// The index chain of a step (row starts -> sources / edge ids) is requested one and two steps ahead; all row pieces of a step are requested at once.
template <int NB, bool XSR, int ILV>
__global__ void __launch_bounds__(512, 2) k_agg_sr(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ src, const int32_t* __restrict__ eid,
                                                   int64_t n_dst, const void* __restrict__ x_, int64_t ldx, int64_t xrb, const float* __restrict__ xs, int ngx,
                                                   const float* __restrict__ ea, int64_t lde, const float* __restrict__ We, const float* __restrict__ be,
                                                   const char* __restrict__ prep, int pass, char* __restrict__ ao, int64_t arb, float* __restrict__ as, int nga,
                                                   char* __restrict__ xo, float* __restrict__ xos, int nt_, int* __restrict__ tickets) {
    const bool nt = (nt_ & 1) != 0;
    const bool no_store = (nt_ & 2) != 0;       // DGNN_SR_NT=2 / 3: timing probe of the producer side alone (tools/probe_producer.py); never set by the product
    constexpr int NSB = NB / 8;                                        // sub-blocks of 8 positions per lane
    extern __shared__ __attribute__((aligned(16))) char agg_smem[];
    char* const bpbuf = agg_smem;                                      // [cb][part][48] x 16 B
    const int lane = lane_id(), w = wave_id_uniform();
    const int jcol = lane & 15, tq = lane >> 4;
    const f32x4_t hd = *reinterpret_cast<const f32x4_t*>(prep);
    const float inv_sWe = hd[1];
    for (int i = threadIdx.x; i < NB * 2 * 48; i += blockDim.x) reinterpret_cast<uint4*>(bpbuf)[i] = reinterpret_cast<const uint4*>(prep + 16 + (int64_t)pass * NB * 2 * 48 * 16)[i];
    __syncthreads();

    const int64_t nq = (n_dst + 3) / 4;
    const int nwg = gridDim.x, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, wg_per_xcd = (nwg + 7 - xcd) >> 3;
    const int64_t per = (nq + 7) / 8, q_lo = xcd * per, q_hi = min(nq, q_lo + per);
    const int wpx = wg_per_xcd * 8;                                    // wavefronts walking this XCD's eighth of the cells
    const int P0 = pass * 256 + NB * jcol;                             // the lane's first position
    const float* const xf = static_cast<const float*>(x_);
    const char* const xb = static_cast<const char*>(x_);

    // index pipeline: vbN = row starts of the step after next, (vb1, vsrc1, veid1) = the next step's, loaded while the current step computes
    auto load_rp = [&](int64_t q, int& vb) {
        if (q < q_hi) {
            const int64_t i0 = q * 4;
            const int nv = (int)(n_dst - i0 < 4 ? n_dst - i0 : 4);
            vb = rowptr[i0 + (lane < nv ? lane : nv)];
        }
    };
    auto load_idx = [&](int64_t q, int vb, bool& reg, int& vsrc, int& veid) {
        reg = false;
        if (q < q_hi) {
            const int64_t i0 = q * 4;
            const int nv = (int)(n_dst - i0 < 4 ? n_dst - i0 : 4);
            const int b0 = __builtin_amdgcn_readfirstlane(vb);
            reg = __all(vb == b0 + 4 * (lane < nv ? lane : nv)) != 0;
            if (reg) {
                const int k_me = b0 + (lane < 4 * nv ? lane : 4 * nv - 1);
                vsrc = src[k_me];
                veid = eid ? eid[k_me] : k_me;
            }
        }
    };
    // Which groups of 4 cells a wavefront takes.  Static (tickets == NULL): q_first, q_first + wpx, ... -- the XCD's wavefronts sweep its eighth side by
    // side.  Tickets (round 6): every wavefront draws its next group from the XCD's counter (one returning atomic per step, requested a whole step before
    // it is needed), so the groups in flight are always the most recently started ones however the wavefronts drift apart: with the static walk a
    // wavefront that runs 10 % ahead after 60 steps works 6 x 2048 cells away from the slowest, and the rows the XCD's L2 has to hold are several
    // times the 2 MB of one sweep line (tools/locality_model.py: an LRU cache of 4096 rows misses 30 % of the requests of an in-order walk; the
    // launch measured 48 %).
    // A ticket is TK_G consecutive groups (16 cells); the counters of the 8 XCDs sit 256 bytes apart (one cache line each: eight counters in one line
    // were served as one word, 88 draws / us for the whole chip, and a draw per group made the launch 3.7 x slower); a ticket is drawn when its
    // predecessor is opened and read two groups before it is needed, so its latency is never waited for.
    constexpr int TK_STRIDE = 64;
    const int TK_G = nt_ >> 8;          // groups per ticket (launch parameter, bits 8.. of nt_)
    int64_t it_base = 0;
    int it_idx = 0, it_raw = 0;
    auto draw = [&]() {                  // lane 0 draws; the value stays in its register until the ticket is opened
        if (lane == 0) it_raw = atomicAdd(tickets + TK_STRIDE * xcd, 1);
    };
    auto next_q = [&]() -> int64_t {     // the next group this wavefront takes (static walk: + wpx)
        if (!tickets) {
            const int64_t q_ = it_base;
            it_base += wpx;
            return q_;
        }
        if (it_idx == TK_G) {
            it_base = q_lo + (int64_t)__builtin_amdgcn_readfirstlane(it_raw) * TK_G;
            it_idx = 0;
            draw();
        }
        return it_base + it_idx++;
    };
    if (tickets) {
        draw();
        it_idx = TK_G;                   // the first next_q() opens the ticket just drawn and draws the one after it
    } else {
        it_base = q_lo + slot * 8 + w;
    }
    int64_t qa = next_q(), qb = next_q(), qc = next_q();
    int vb1 = 0, vb2 = 0, vsrc1 = 0, veid1 = 0;
    bool reg1 = false;
    load_rp(qa, vb1);
    load_rp(qb, vb2);
    load_idx(qa, vb1, reg1, vsrc1, veid1);

    for (int64_t q = qa; q < q_hi;) {
        const int64_t i0 = q * 4;
        const int nv = (int)(n_dst - i0 < 4 ? n_dst - i0 : 4);
        const bool regular = reg1;
        const int vsrc = vsrc1, veid = veid1;
        // the next step's sources / edge ids (its row starts arrived a step ago) and the row starts of the step behind it
        vb1 = vb2;
        load_idx(qb, vb1, reg1, vsrc1, veid1);
        load_rp(qc, vb2);
        q = qb, qb = qc;
        qc = next_q();
        const int tl = tq < nv ? tq : nv - 1;                          // a short group at the end of the graph: clamped (duplicated) cells
        const int64_t cell = i0 + tl;
        float aout[NB];
        if (regular) {
            // A operand: lane (edge jcol, k-group tq) holds attributes 8 tq .. 8 tq + 7 of its edge; k = 20 is the constant 1 of the bias row
            const float* er = ea + (int64_t)__shfl(veid, jcol) * lde;
            const f32x4_t q0 = *reinterpret_cast<const f32x4_t*>(er + 8 * (tq < 2 ? tq : 2));
            const f32x4_t q1 = *reinterpret_cast<const f32x4_t*>(er + 8 * (tq < 1 ? tq : 1) + 4);
            int sidx[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) sidx[r] = __shfl(vsrc, tl * 4 + r);
            // every row piece of the step at once: NSB x 4 neighbours x (hi, lo | two fp32 quads)
            uint4 ra[NSB][4], rb_[NSB][4];
            float isx[4];
#pragma unroll
            for (int sb = 0; sb < NSB; ++sb) {
                const int P = P0 + 8 * sb, ch = P >> 5, p0 = P & 31;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if constexpr (XSR) {
                        const char* rp = xb + (int64_t)sidx[r] * xrb + ch * SRB + p0 * 2;
                        ra[sb][r] = *reinterpret_cast<const uint4*>(rp);
                        rb_[sb][r] = *reinterpret_cast<const uint4*>(rp + 64);
                    } else {
                        // positions p0 .. p0 + 7 of chunk ch = channels 32 ch + PI(p0) .. + 3 and 32 ch + PI(p0 + 4) .. + 3
                        const float* rp = xf + (int64_t)sidx[r] * ldx + ch * 32;
                        ra[sb][r] = *reinterpret_cast<const uint4*>(rp + sr_chan(p0));
                        rb_[sb][r] = *reinterpret_cast<const uint4*>(rp + sr_chan(p0 + 4));
                    }
                }
            }
            if constexpr (XSR) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float s_ = xs[(int64_t)sidx[r] * ngx + (P0 >> 8)];
                    isx[r] = b_of(s_) == SR_ZERO_BITS ? 0.f : pow2_inv(s_);
                }
            }
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
            pow2_scales(cross_row_umax(b_of(mf)), sA, inv_sA);      // one scale per EDGE (a row of the operand)
            uint32_t ph[4], pl[4];
#pragma unroll
            for (int d = 0; d < 4; ++d) split2h(av[2 * d] * sA, av[2 * d + 1] * sA, ph[d], pl[d]);
            float inv_e[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) inv_e[r] = __shfl(inv_sA, 4 * tq + r);
            const f16x8 ah = pack8h(ph), al = pack8h(pl);
#pragma unroll
            for (int sb = 0; sb < NSB; ++sb) {
                float xr[4][8];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const uint32_t aw[4] = {ra[sb][r].x, ra[sb][r].y, ra[sb][r].z, ra[sb][r].w}, bw[4] = {rb_[sb][r].x, rb_[sb][r].y, rb_[sb][r].z, rb_[sb][r].w};
                    if constexpr (XSR) {
#pragma unroll
                        for (int d = 0; d < 4; ++d) {
                            const f16x2_t h2 = __builtin_bit_cast(f16x2_t, aw[d]), l2 = __builtin_bit_cast(f16x2_t, bw[d]);
                            xr[r][2 * d] = ((float)h2[0] + (float)l2[0]) * isx[r];      // hi + lo is exact in fp32 (two disjoint 11-bit pieces)
                            xr[r][2 * d + 1] = ((float)h2[1] + (float)l2[1]) * isx[r];
                        }
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            xr[r][i] = f_of(aw[i]);
                            xr[r][4 + i] = f_of(bw[i]);
                        }
                    }
                }
#pragma unroll
                for (int c4 = 0; c4 < 8; c4 += ILV) {
                    // ILV channel blocks at a time: their three-product chains are independent, issued interleaved (a lone chain waits for its own
                    // results: SQ_WAIT_INST_ANY was 64 % of the wave cycles)
                    f16x8 bh[ILV], bl[ILV];
                    f32x4_t d[ILV];
#pragma unroll
                    for (int u = 0; u < ILV; ++u) {
                        const char* bp = bpbuf + (((8 * sb + c4 + u) * 2) * 48 + (tq < 3 ? tq : 0) * 16 + jcol) * 16;   // k-group 3 re-reads group 0: its A operand is zero
                        bh[u] = H8(*reinterpret_cast<const uint4*>(bp));
                        bl[u] = H8(*reinterpret_cast<const uint4*>(bp + 768));
                        d[u] = f32x4_t{0.f, 0.f, 0.f, 0.f};
                    }
#pragma unroll
                    for (int u = 0; u < ILV; ++u) d[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[u], d[u], 0, 0, 0);
#pragma unroll
                    for (int u = 0; u < ILV; ++u) d[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[u], d[u], 0, 0, 0);
#pragma unroll
                    for (int u = 0; u < ILV; ++u) d[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[u], d[u], 0, 0, 0);
#pragma unroll
                    for (int u = 0; u < ILV; ++u) {
                        const int c8 = c4 + u;
#pragma unroll
                        for (int r = 0; r < 4; ++r) d[u][r] *= inv_e[r];     // exact: powers of two
                        float a = __fmul_rn(xr[0][c8], d[u][0]);
#pragma unroll
                        for (int r = 1; r < 4; ++r) a = __fmaf_rn(xr[r][c8], d[u][r], a);
                        aout[8 * sb + c8] = a * (0.25f * inv_sWe);
                    }
                }
            }
        } else {
            // generic path (a group with any in-degree other than 4): plain fp32 per lane, one edge at a time (never on Delaunay scenes)
#pragma unroll
            for (int cb = 0; cb < NB; ++cb) aout[cb] = 0.f;
            if (tq < nv) {
                const int b = rowptr[cell], e_end = rowptr[cell + 1];
                for (int k = b; k < e_end; ++k) {
                    const int s_ = src[k];
                    const float* ar = ea + (int64_t)(eid ? eid[k] : k) * lde;
#pragma unroll 1
                    for (int cb = 0; cb < NB; ++cb) {
                        const int P = P0 + cb, c = (P & ~31) + sr_chan(P & 31);
                        float p = be[c];
                        for (int f = 0; f < FE; ++f) p = __fmaf_rn(We[(int64_t)c * FE + f], ar[f], p);
                        float xv;
                        if constexpr (XSR) {
                            const uint16_t* hp = reinterpret_cast<const uint16_t*>(xb + (int64_t)s_ * xrb + (P >> 5) * SRB) + (P & 31);
                            const float sx = xs[(int64_t)s_ * ngx + (P >> 8)];
                            const float isx_ = b_of(sx) == SR_ZERO_BITS ? 0.f : pow2_inv(sx);
                            xv = ((float)__builtin_bit_cast(_Float16, hp[0]) + (float)__builtin_bit_cast(_Float16, hp[32])) * isx_;
                        } else {
                            xv = xf[(int64_t)s_ * ldx + c];
                        }
                        aout[cb] = __fadd_rn(aout[cb], __fmul_rn(xv, p));
                    }
                }
                const float cnt = (float)max(e_end - b, 1);
#pragma unroll
                for (int cb = 0; cb < NB; ++cb) aout[cb] = __fdiv_rn(aout[cb], cnt);
            }
        }
        // the finished positions: the pass's scale group (the cell's 16 lanes), split, stored as the lane's NB positions of the split row
        auto put_row = [&](const float (&v)[NB], char* orow, float* oscale) {
            float mx = 0.f;
#pragma unroll
            for (int i = 0; i < NB; i += 2) mx = fmaxf(fmaxf(mx, fabsf(v[i])), fabsf(v[i + 1]));
            const uint32_t m = row16_umax(b_of(mx));
            float s_store, s_mul;
            sr_scale(m, s_store, s_mul);
            if (tq < nv && !(no_store && s_mul != 12345.f)) {
                if (jcol == 0) oscale[pass] = s_store;
#pragma unroll
                for (int sb = 0; sb < NSB; ++sb) {
                    const int P = P0 + 8 * sb;
                    uint32_t hi[4], lo[4];
#pragma unroll
                    for (int d = 0; d < 4; ++d) split2h(v[8 * sb + 2 * d] * s_mul, v[8 * sb + 2 * d + 1] * s_mul, hi[d], lo[d]);
                    char* o = orow + (P >> 5) * SRB + (P & 31) * 2;
                    st16(o, hi[0], hi[1], hi[2], hi[3], nt);
                    st16(o + 64, lo[0], lo[1], lo[2], lo[3], nt);
                }
            }
        };
        put_row(aout, ao + cell * arb, as + cell * nga);
        if constexpr (!XSR) {
            if (xo) {   // the cell's own fp32 row as a split row (this layer's x_i operand)
                float xv[NB];
#pragma unroll
                for (int sb = 0; sb < NSB; ++sb) {
                    const int P = P0 + 8 * sb, ch = P >> 5, p0 = P & 31;
                    const float* rp = xf + cell * ldx + ch * 32;
                    const f32x4_t a4 = *reinterpret_cast<const f32x4_t*>(rp + sr_chan(p0)), b4 = *reinterpret_cast<const f32x4_t*>(rp + sr_chan(p0 + 4));
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        xv[8 * sb + i] = a4[i];
                        xv[8 * sb + 4 + i] = b4[i];
                    }
                }
                put_row(xv, xo + cell * arb, xos + cell * nga);
            }
        }
    }
}

// ---- k_gemm_sr ---------------------------------------------------------------------------------------------------------------------------------
struct SrPart {
    const char* base;       // split rows [M][nch x 128 B]
    int64_t row_bytes;
    const float* scales;    // [M][ng]
    int nch, ng, gch;       // chunks, scale groups per row, chunks per group
};
struct SrOut {
    char* sr;               // mode 0: split rows out [M][n_out / 32 x 128 B] + scales [M][n_out / 256]
    int64_t row_bytes;
    float* scales;
    int ng;
    float* f32;             // mode 1: fp32 rows [M][n_out] in channel order, row stride ldo
    int64_t ldo;
    int mode;
};

__device__ __forceinline__ void dma16(const char* g, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// LDS: [2 buffers][W 256 rows | X 256 rows] x 128 B, then bias | bn scale | bn shift | 1 / sw [256] each, then row maxima [2][256]
constexpr int G_BUF = (TN + TM) * SRB;
constexpr int G_CST = 2 * G_BUF;
constexpr int G_RMAX = G_CST + 4 * TN * 4;
constexpr int G_SMEM = G_RMAX + 2 * TM * 4;

__global__ void __launch_bounds__(GT, 1) k_gemm_sr(SrPart p1, SrPart p2, const char* __restrict__ Wp, int64_t w_row_bytes, const float* __restrict__ sw,
                                                   const float* __restrict__ bias, const float* __restrict__ scale, const float* __restrict__ shift, int relu,
                                                   int64_t M, int n_out, SrOut out) {
    extern __shared__ __attribute__((aligned(16))) char g_smem[];
    float* const cst = reinterpret_cast<float*>(g_smem + G_CST);
    uint32_t* const rmax = reinterpret_cast<uint32_t*>(g_smem + G_RMAX);
    const int lane = lane_id(), w = wave_id_uniform();
    const int wm = w >> 2, wn = w & 3, h = lane >> 5, l31 = lane & 31;
    const int ncb = n_out / TN;   // XCD-aware tile map: the column tiles of a row panel run on ONE XCD back to back (the panel crosses the fabric once)
    const int64_t rb = (int64_t)((blockIdx.x >> 3) / ncb) * 8 + (blockIdx.x & 7);
    if (rb * TM >= M) return;
    const int64_t row0 = rb * TM;
    const int ct = (int)((blockIdx.x >> 3) % ncb), col0 = ct * TN;
    for (int c = threadIdx.x; c < TN; c += GT) {
        cst[c] = bias ? bias[col0 + c] : 0.f;
        cst[TN + c] = scale ? scale[col0 + c] : 1.f;
        cst[2 * TN + c] = scale ? shift[col0 + c] : 0.f;
        cst[3 * TN + c] = pow2_inv(sw[col0 + c]);
    }
    const int nch1 = p1.nch, nch = nch1 + p2.nch, ng1 = p1.ng, ngt = ng1 + p2.ng;

    // DMA roles: one instruction moves 8 rows x 128 B (1 KB of LDS, contiguous); wave w takes instructions w, w + 8, .. of the 32 (W) + 32 (X).
    // lane -> (row of the group rr = lane >> 3, LDS slot q = lane & 7); slot q of row r holds the row's 16-byte piece q ^ (r & 7)
    const int rr = lane >> 3, qs = lane & 7;
    // wave-uniform bases (SGPRs) + 32-bit per-lane offsets: the tile's weight rows span <= 1024 x 8 KB, its cell rows 256 x 4 KB
    const char* const wbase = Wp + (int64_t)col0 * w_row_bytes;
    const char* const xbase1 = p1.base + row0 * p1.row_bytes;
    const char* const xbase2 = p2.nch ? p2.base + row0 * p2.row_bytes - (int64_t)nch1 * SRB : xbase1;
    const int64_t rows_left = M - row0;
    uint32_t woff[4], xoff1[4], xoff2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r_t = (w + 8 * j) * 8 + rr;
        const int piece = qs ^ ((r_t >> 1) & 7);
        woff[j] = (uint32_t)(r_t * (int)w_row_bytes + piece * 16);
        const int gr = r_t < rows_left ? r_t : (int)rows_left - 1;
        xoff1[j] = (uint32_t)(gr * (int)p1.row_bytes + piece * 16);
        xoff2[j] = (uint32_t)(gr * (int)(p2.nch ? p2.row_bytes : p1.row_bytes) + piece * 16);
    }
    auto dma_chunk = [&](int ch) {
        char* buf = g_smem + (ch & 1) * G_BUF;
        const char* wb_ = wbase + (int64_t)ch * SRB;
        const bool first = ch < nch1;
        const char* xb_ = (first ? xbase1 : xbase2) + (int64_t)ch * SRB;
#pragma unroll
        for (int j = 0; j < 4; ++j) dma16(wb_ + woff[j], buf + (w + 8 * j) * 1024);
#pragma unroll
        for (int j = 0; j < 4; ++j) dma16(xb_ + (first ? xoff1[j] : xoff2[j]), buf + TN * SRB + (w + 8 * j) * 1024);
    };
    // the scale of group gi of this lane's two cells (b = 0, 1)
    int64_t cellb[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int64_t c_ = row0 + wn * 64 + b * 32 + l31;
        cellb[b] = c_ < M ? c_ : M - 1;
    }
    auto scale_of = [&](int gi, int b) -> float { return gi < ng1 ? p1.scales[cellb[b] * ng1 + gi] : p2.scales[cellb[b] * p2.ng + (gi - ng1)]; };
    f32x16 acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
    float s_cur[2] = {0.f, 0.f}, s_min[2] = {0.f, 0.f}, s_nx[2];
    uint32_t mk[2] = {~0u, ~0u};
    int gi = 0, next_b = 0;     // next group to enter, chunk at which it starts
    s_nx[0] = scale_of(0, 0);
    s_nx[1] = scale_of(0, 1);
    dma_chunk(0);
    const int sw_ = (l31 >> 1) & 7;
    for (int ch = 0; ch < nch; ++ch) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of chunk ch has landed (and the prefetched scales)
        __syncthreads();                                     // everybody's has; everybody is done with the other buffer
        if (ch + 1 < nch) dma_chunk(ch + 1);                 // in flight under the products below
        if (ch == next_b) {
            // the K walk enters group gi: the accumulators of a cell stay in units of s * sw for the LARGEST group seen so far
            bool any_mul = false;
            float mul[2];
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const float s_ = s_nx[b];
                mul[b] = 1.f;
                mk[b] = ~0u;
                if (b_of(s_) == SR_ZERO_BITS) {
                    // all zeros: its products vanish in any unit
                } else if (s_cur[b] == 0.f) {
                    s_cur[b] = s_min[b] = s_;
                } else if (s_ > s_min[b] * 1.099511627776e12f) {
                    mk[b] = 0u;                              // dropped: this cell's fragments of the group are masked to zero below
                } else {
                    mul[b] = s_ * pow2_inv(s_cur[b]);        // exact; <= 2^40 by the test above, an underflow means the sum so far is negligible
                    s_cur[b] = s_;
                    s_min[b] = fminf(s_min[b], s_);
                }
                any_mul = any_mul || mul[b] != 1.f;
            }
            if (__any(any_mul)) {
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
#pragma unroll
                        for (int i = 0; i < 16; ++i) acc[a][b][i] *= mul[b];
            }
            const bool in1 = gi < ng1;
            next_b += in1 ? p1.gch : p2.gch;
            if (in1 && next_b > nch1) next_b = nch1;
            ++gi;
            if (gi < ngt) {
                s_nx[0] = scale_of(gi, 0);
                s_nx[1] = scale_of(gi, 1);
            }
            if (gi == ng1) next_b = nch1;
        }
        const char* Wb = g_smem + (ch & 1) * G_BUF;
        const char* Xb = Wb + TN * SRB;
        const int wr0 = wm * 128 + l31, xr0 = wn * 64 + l31;   // rows are multiples of 32 apart: (row & 7) = (l31 & 7) for all of them
#pragma unroll
        for (int S = 0; S < 2; ++S) {
            f16x8 xf[2][2];
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    uint4 u = *reinterpret_cast<const uint4*>(Xb + (xr0 + b * 32) * SRB + (((4 * p + 2 * S + h) ^ sw_) << 4));
                    // a dropped group's fragments are zeroed (mk = 0; ~0 otherwise): 16 integer instructions per k-step in the shadow of 24 matrix
                    // instructions -- a second, unmasked copy of the loop behind a branch made the compiler spill 120 registers
                    u.x &= mk[b]; u.y &= mk[b]; u.z &= mk[b]; u.w &= mk[b];
                    xf[b][p] = H8(u);
                }
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                f16x8 wf[2];
#pragma unroll
                for (int p = 0; p < 2; ++p) wf[p] = H8(*reinterpret_cast<const uint4*>(Wb + (wr0 + a * 32) * SRB + (((4 * p + 2 * S + h) ^ sw_) << 4)));
                constexpr int PW[3] = {1, 0, 0}, PX[3] = {0, 1, 0};   // small terms first; the two cell blocks alternate product by product
#pragma unroll
                for (int qq = 0; qq < 3; ++qq)
#pragma unroll
                    for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[PW[qq]], xf[b][PX[qq]], acc[a][b], 0, 0, 0);
            }
        }
    }

    // ---- epilogue: lane = (cell wn * 64 + b * 32 + l31, channels wm * 128 + a * 32 + (r & 3) + 8 (r >> 2) + 4 h)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const float inv_cell = s_cur[b] == 0.f ? 0.f : pow2_inv(s_cur[b]);
        float mx = 0.f;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int c = wm * 128 + a * 32 + 8 * r4 + 4 * h;
                const f32x4_t bb = *reinterpret_cast<const f32x4_t*>(cst + c), sc = *reinterpret_cast<const f32x4_t*>(cst + TN + c),
                              sh = *reinterpret_cast<const f32x4_t*>(cst + 2 * TN + c), iw = *reinterpret_cast<const f32x4_t*>(cst + 3 * TN + c);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float v = __fmaf_rn(acc[a][b][4 * r4 + i], inv_cell * iw[i], bb[i]);
                    v = __fmaf_rn(v, sc[i], sh[i]);
                    if (relu) v = fmaxf(v, 0.f);
                    acc[a][b][4 * r4 + i] = v;
                    mx = fmaxf(mx, fabsf(v));
                }
            }
        const int64_t cell = row0 + wn * 64 + b * 32 + l31;
        if (out.mode == 1) {
            if (cell < M) {
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) {
                        float* o = out.f32 + cell * out.ldo + col0 + wm * 128 + a * 32 + 8 * r4 + 4 * h;
                        *reinterpret_cast<f32x4_t*>(o) = f32x4_t{acc[a][b][4 * r4], acc[a][b][4 * r4 + 1], acc[a][b][4 * r4 + 2], acc[a][b][4 * r4 + 3]};
                    }
            }
            continue;
        }
        // row maximum over the tile's 256 channels: the lane's 64, the partner lane's (other h) 64, the other channel half's 128 through LDS
        uint32_t m = b_of(mx), ma, mb;
        swap32_pair(m, ma, mb);
        m = umax(ma, mb);
        if (h == 0) rmax[wm * TM + wn * 64 + b * 32 + l31] = m;
    }
    if (out.mode == 1) return;
    __syncthreads();
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int tl = wn * 64 + b * 32 + l31;
        const int64_t cell = row0 + tl;
        const uint32_t m = umax(rmax[tl], rmax[TM + tl]);
        float s_store, s_mul;
        sr_scale(m, s_store, s_mul);
        if (cell >= M) continue;
        if (wm == 0 && h == 0) out.scales[cell * out.ng + ct] = s_store;
        char* orow = out.sr + cell * out.row_bytes + (int64_t)(ct * 8 + wm * 4) * SRB + 32 * h;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            uint32_t hi[8], lo[8];
#pragma unroll
            for (int d = 0; d < 8; ++d) split2h(acc[a][b][2 * d] * s_mul, acc[a][b][2 * d + 1] * s_mul, hi[d], lo[d]);
            char* o = orow + a * SRB;
            *reinterpret_cast<uint4*>(o) = make_uint4(hi[0], hi[1], hi[2], hi[3]);
            *reinterpret_cast<uint4*>(o + 16) = make_uint4(hi[4], hi[5], hi[6], hi[7]);
            *reinterpret_cast<uint4*>(o + 64) = make_uint4(lo[0], lo[1], lo[2], lo[3]);
            *reinterpret_cast<uint4*>(o + 80) = make_uint4(lo[4], lo[5], lo[6], lo[7]);
        }
    }
}

// ---- k_gemm_sr2: the same product, second arrangement (default) ----------------------------------------------------------------------------------
// What k_gemm_sr waits for is its operands (one barrier + one DMA round trip per 32-wide K chunk, 39 % of the matrix peak at K = N = 1024).  Here
//   * a wavefront owns 32 CELLS and ALL 256 channels of the tile (8 accumulator blocks): its cell operand does not go through LDS at all -- every lane
//     loads the four 16-byte fragments of its own cell's chunk straight from the split row (the row IS in fragment order) TWO chunks ahead, three
//     register sets rotating;
//   * only the weights go global -> LDS by DMA, three stages of 32 KB, issued two chunks ahead; the barrier per chunk only hands a stage back;
//   * the epilogue needs no LDS exchange: a cell's 256 finished channels sit in two lanes (l, l ^ 32);
//   * mode 2 (the decoder): the finished hidden row is multiplied by W3 [n_proj][n_out] on the spot and only the logits leave the launch
//     (reference learning/surfaceNetStaticEdgeFilters.py:180-187: Linear - BN - ReLU - Linear): partial sums of the column tiles are added
//     atomically into zero-initialised logits -- two addends per logit at n_out = 512, an order-independent sum; tile 0 carries the bias.
// CPS = K chunks per weight stage: 1 = three stages of one chunk (a barrier per chunk), 2 = two stages of two chunks (a barrier every other chunk)
template <int CPS> struct G2 {
    static constexpr int W_STG = CPS == 1 ? 3 : 2;
    static constexpr int STAGE = CPS * TN * SRB;
    static constexpr int CST = W_STG * STAGE;
    static constexpr int SMEM = CST + 6 * TN * 4;      // bias | bn scale | bn shift | 1 / sw | W3 row 0 | W3 row 1
};

template <int CPS>
__global__ void __launch_bounds__(GT, 1) k_gemm_sr2(SrPart p1, SrPart p2, const char* __restrict__ Wp, int64_t w_row_bytes, const float* __restrict__ sw,
                                                    const float* __restrict__ bias, const float* __restrict__ scale, const float* __restrict__ shift, int relu,
                                                    int64_t M, int n_out, SrOut out, const float* __restrict__ W3, const float* __restrict__ b3, int n_proj,
                                                    float* __restrict__ logits, int nt_) {
    const bool nt = nt_ != 0;
    typedef G2<CPS> L;
    extern __shared__ __attribute__((aligned(16))) char g2_smem[];
    float* const cst = reinterpret_cast<float*>(g2_smem + L::CST);
    const int lane = lane_id(), w = wave_id_uniform();
    const int h = lane >> 5, l31 = lane & 31;
    const int ncb = n_out / TN;
    const int64_t rb = (int64_t)((blockIdx.x >> 3) / ncb) * 8 + (blockIdx.x & 7);
    if (rb * TM >= M) return;
    const int64_t row0 = rb * TM;
    const int ct = (int)((blockIdx.x >> 3) % ncb), col0 = ct * TN;
    for (int c = threadIdx.x; c < TN; c += GT) {
        cst[c] = bias ? bias[col0 + c] : 0.f;
        cst[TN + c] = scale ? scale[col0 + c] : 1.f;
        cst[2 * TN + c] = scale ? shift[col0 + c] : 0.f;
        cst[3 * TN + c] = pow2_inv(sw[col0 + c]);
        cst[4 * TN + c] = (out.mode == 2) ? W3[col0 + c] : 0.f;
        cst[5 * TN + c] = (out.mode == 2 && n_proj > 1) ? W3[n_out + col0 + c] : 0.f;
    }
    const int nch1 = p1.nch, nch = nch1 + p2.nch, ng1 = p1.ng, ngt = ng1 + p2.ng;
    // weights: DMA, one instruction = 8 rows x 128 B; wave w takes instructions w, w + 8, w + 16, w + 24 of a chunk's 32
    const int rr = lane >> 3, qs = lane & 7;
    const char* const wbase = Wp + (int64_t)col0 * w_row_bytes;
    uint32_t woff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r_t = (w + 8 * j) * 8 + rr;
        woff[j] = (uint32_t)(r_t * (int)w_row_bytes + (qs ^ ((r_t >> 1) & 7)) * 16);
    }
    auto dma_stage = [&](int stg) {       // the chunks stg * CPS .. of the weights -> stage stg % W_STG
        char* buf = g2_smem + (stg % L::W_STG) * L::STAGE;
#pragma unroll
        for (int c = 0; c < CPS; ++c) {
            const int ch = stg * CPS + c;
            if (ch < nch) {
                const char* wb_ = wbase + (int64_t)ch * SRB;
#pragma unroll
                for (int j = 0; j < 4; ++j) dma16(wb_ + woff[j], buf + c * (TN * SRB) + (w + 8 * j) * 1024);
            }
        }
    };
    // this lane's cell and its split rows: fragment (part p, k-step S) of a chunk = the 16 bytes at (4 p + 2 S + h) * 16
    const int64_t cell = row0 + w * 32 + l31;
    const int64_t cellc = cell < M ? cell : M - 1;
    const char* const xrow1 = p1.base + cellc * p1.row_bytes + h * 16;
    const char* const xrow2 = p2.nch ? p2.base + cellc * p2.row_bytes + h * 16 - (int64_t)nch1 * SRB : xrow1;
    auto load_x = [&](int ch, uint4 (&x)[4]) {
        const char* r_ = (ch < nch1 ? xrow1 : xrow2) + (int64_t)ch * SRB;
#pragma unroll
        for (int k = 0; k < 4; ++k) x[k] = *reinterpret_cast<const uint4*>(r_ + 32 * k);      // k = 2 p + S
    };
    auto scale_of = [&](int gi) -> float { return gi < ng1 ? p1.scales[cellc * ng1 + gi] : p2.scales[cellc * p2.ng + (gi - ng1)]; };

    f32x16 acc[8];
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[a][i] = 0.f;
    float s_cur = 0.f, s_min = 0.f, s_nx = scale_of(0);
    uint32_t mk = ~0u;
    int gi = 0, next_b = 0;
    uint4 xs0[4], xs1[4], xs2[4];
    dma_stage(0);
    load_x(0, xs0);
    if (CPS == 1) dma_stage(1);
    if (nch > 1) load_x(1, xs1);
    // LDS slot of a row's 16-byte piece: piece ^ ((row >> 1) & 7).  ds_read_b128 is served in groups of 16 NON-contiguous lanes ({0-3, 12-15, 20-27},
    // {4-11, 16-19, 28-31}: MI355X_MICROARCH.md); a lane reads row (32 a + l31), i.e. bank row (l31 & 1): with (l31 >> 1) & 7 as the XOR key the 8 even
    // and the 8 odd lanes of either group get 8 distinct slots each (the first version keyed on row & 7: two lanes of every group on one bank,
    // SQ_LDS_BANK_CONFLICT = 47 % of the LDS cycles, profiles/r05_wide.md)
    const int sw_ = (l31 >> 1) & 7;
    auto step = [&](int ch, uint4 (&xc)[4], uint4 (&xn)[4]) {
        const int stg = ch / CPS;
        if (ch % CPS == 0) {
            // stage stg holds W(ch ..) once this wave's share has landed (everything but the 8 most recent operations has: those are cell fragments
            // of the next two chunks and, behind a group boundary, a scale word) and everybody else's has (barrier); the barrier also hands the stage
            // read before this one back to the DMA
            // At the tail the operations counted on are not issued any more (no stage / no cell fragments past the last chunk): behind the last
            // CPS chunks fewer than 8 operations follow the stage's DMA, so the wave drains everything there (nch is uniform: a scalar branch).
            if (ch + CPS >= nch)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else
                asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            __syncthreads();
            dma_stage(stg + (CPS == 1 ? 2 : 1));
        }
        if (ch + 2 < nch) load_x(ch + 2, xn);
        if (ch == next_b) {
            const float s_ = s_nx;
            float mul = 1.f;
            mk = ~0u;
            if (b_of(s_) == SR_ZERO_BITS) {
            } else if (s_cur == 0.f) {
                s_cur = s_min = s_;
            } else if (s_ > s_min * 1.099511627776e12f) {
                mk = 0u;
            } else {
                mul = s_ * pow2_inv(s_cur);
                s_cur = s_;
                s_min = fminf(s_min, s_);
            }
            if (__any(mul != 1.f)) {
#pragma unroll
                for (int a = 0; a < 8; ++a)
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[a][i] *= mul;
            }
            const bool in1 = gi < ng1;
            next_b += in1 ? p1.gch : p2.gch;
            if (in1 && next_b > nch1) next_b = nch1;
            ++gi;
            if (gi < ngt) s_nx = scale_of(gi);
            if (gi == ng1) next_b = nch1;
        }
        const char* Wb = g2_smem + (stg % L::W_STG) * L::STAGE + (ch % CPS) * (TN * SRB);
#pragma unroll
        for (int S = 0; S < 2; ++S) {
            uint4 uh = xc[S], ul = xc[2 + S];
            uh.x &= mk; uh.y &= mk; uh.z &= mk; uh.w &= mk;
            ul.x &= mk; ul.y &= mk; ul.z &= mk; ul.w &= mk;
            const f16x8 xh = H8(uh), xl = H8(ul);
#pragma unroll
            for (int a = 0; a < 8; a += 2) {
                // two accumulator blocks at a time, their three products interleaved: no product waits for its predecessor's result (small terms first)
                f16x8 wh[2], wl[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const char* wr_ = Wb + ((a + u) * 32 + l31) * SRB;
                    wh[u] = H8(*reinterpret_cast<const uint4*>(wr_ + (((2 * S + h) ^ sw_) << 4)));
                    wl[u] = H8(*reinterpret_cast<const uint4*>(wr_ + (((4 + 2 * S + h) ^ sw_) << 4)));
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) acc[a + u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[u], xh, acc[a + u], 0, 0, 0);
#pragma unroll
                for (int u = 0; u < 2; ++u) acc[a + u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[u], xl, acc[a + u], 0, 0, 0);
#pragma unroll
                for (int u = 0; u < 2; ++u) acc[a + u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[u], xh, acc[a + u], 0, 0, 0);
            }
        }
    };
    for (int ch0 = 0; ch0 < nch; ch0 += 3) {
        step(ch0, xs0, xs2);
        if (ch0 + 1 < nch) step(ch0 + 1, xs1, xs0);
        if (ch0 + 2 < nch) step(ch0 + 2, xs2, xs1);
    }

    // ---- epilogue: lane = (cell, channels a * 32 + (r & 3) + 8 (r >> 2) + 4 h of the tile)
    const float inv_cell = s_cur == 0.f ? 0.f : pow2_inv(s_cur);
    float mx = 0.f, pj0 = 0.f, pj1 = 0.f;
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            const int c = a * 32 + 8 * r4 + 4 * h;
            const f32x4_t bb = *reinterpret_cast<const f32x4_t*>(cst + c), sc = *reinterpret_cast<const f32x4_t*>(cst + TN + c),
                          sh = *reinterpret_cast<const f32x4_t*>(cst + 2 * TN + c), iw = *reinterpret_cast<const f32x4_t*>(cst + 3 * TN + c);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float v = __fmaf_rn(acc[a][4 * r4 + i], inv_cell * iw[i], bb[i]);
                v = __fmaf_rn(v, sc[i], sh[i]);
                if (relu) v = fmaxf(v, 0.f);
                acc[a][4 * r4 + i] = v;
                mx = fmaxf(mx, fabsf(v));
            }
            if (out.mode == 2) {
                const f32x4_t w0 = *reinterpret_cast<const f32x4_t*>(cst + 4 * TN + c), w1 = *reinterpret_cast<const f32x4_t*>(cst + 5 * TN + c);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    pj0 = __fmaf_rn(acc[a][4 * r4 + i], w0[i], pj0);
                    pj1 = __fmaf_rn(acc[a][4 * r4 + i], w1[i], pj1);
                }
            }
        }
    if (out.mode == 1) {
        if (cell < M) {
#pragma unroll
            for (int a = 0; a < 8; ++a)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    float* o = out.f32 + cell * out.ldo + col0 + a * 32 + 8 * r4 + 4 * h;
                    *reinterpret_cast<f32x4_t*>(o) = f32x4_t{acc[a][4 * r4], acc[a][4 * r4 + 1], acc[a][4 * r4 + 2], acc[a][4 * r4 + 3]};
                }
        }
        return;
    }
    if (out.mode == 2) {
        // the two lanes of a cell add their halves in one fixed order (h = 0's first), the column tiles through atomics on zeroed logits
        uint32_t oa, ob;
        swap32_pair(b_of(pj0), oa, ob);
        const float other0 = f_of(oa ^ ob ^ b_of(pj0));
        swap32_pair(b_of(pj1), oa, ob);
        const float other1 = f_of(oa ^ ob ^ b_of(pj1));
        if (h == 0 && cell < M) {
            float t0 = pj0 + other0, t1 = pj1 + other1;
            if (ct == 0) {
                t0 += b3 ? b3[0] : 0.f;
                if (n_proj > 1) t1 += b3 ? b3[1] : 0.f;
            }
            if (ncb == 1) {
                logits[cell * n_proj] = t0;
                if (n_proj > 1) logits[cell * n_proj + 1] = t1;
            } else {
                atomicAdd(logits + cell * n_proj, t0);
                if (n_proj > 1) atomicAdd(logits + cell * n_proj + 1, t1);
            }
        }
        return;
    }
    uint32_t m = b_of(mx), ma, mb;
    swap32_pair(m, ma, mb);
    m = umax(ma, mb);
    float s_store, s_mul;
    sr_scale(m, s_store, s_mul);
    if (cell >= M) return;
    if (h == 0) out.scales[cell * out.ng + ct] = s_store;
    char* orow = out.sr + cell * out.row_bytes + (int64_t)(ct * 8) * SRB + 32 * h;
#pragma unroll
    for (int a = 0; a < 8; ++a) {
        uint32_t hi[8], lo[8];
#pragma unroll
        for (int d = 0; d < 8; ++d) split2h(acc[a][2 * d] * s_mul, acc[a][2 * d + 1] * s_mul, hi[d], lo[d]);
        char* o = orow + a * SRB;
        st16(o, hi[0], hi[1], hi[2], hi[3], nt);
        st16(o + 16, hi[4], hi[5], hi[6], hi[7], nt);
        st16(o + 64, lo[0], lo[1], lo[2], lo[3], nt);
        st16(o + 80, lo[4], lo[5], lo[6], lo[7], nt);
    }
}

}  // namespace

// =====================================================================================================================
// C ABI
// =====================================================================================================================
static int sr_nt() {      // DGNN_SR_NT=0: plain stores of the split rows (A/B)
    static const int v = getenv("DGNN_SR_NT") ? atoi(getenv("DGNN_SR_NT")) : 1;
    return v;
}

extern "C" int64_t dgnn_sr_row_bytes(int C) { return C > 0 && C % 32 == 0 ? (int64_t)(C / 32) * SRB : 0; }

// fp32 rows [rows, k1 (+ k2)] -> split rows (dst [rows][row_bytes], scales [rows][ng]); gch chunks of 32 channels per scale group, 0 = one group per
// row (weights: dgnn_linear_sr's `Wp` / `sw` are made by this call on [Wj | Wi] with gch = 0); activations: gch = 8 (ng = ceil(C / 256))
extern "C" int dgnn_sr_pack(const float* A1, int64_t ld1, int k1, const float* A2, int64_t ld2, int k2, int64_t rows, int gch, void* dst,
                            int64_t dst_row_bytes, float* scales, int ng, void* stream) {
    DGNN_REQUIRE(rows >= 0 && k1 > 0 && k1 % 32 == 0 && k2 >= 0 && k2 % 32 == 0 && (k2 == 0) == (A2 == nullptr), DGNN_E_INVALID, "sr_pack: bad sizes (widths must be multiples of 32)");
    if (rows == 0) return DGNN_OK;
    const int nch = (k1 + k2) / 32, g = gch > 0 ? gch : nch;
    DGNN_REQUIRE(A1 && dst && scales && ld1 >= k1 && (!A2 || ld2 >= k2) && dst_row_bytes >= (int64_t)nch * SRB && dst_row_bytes % 16 == 0 && ng == (nch + g - 1) / g &&
                     ((uintptr_t)dst % 16) == 0,
                 DGNN_E_INVALID, "sr_pack: null / short / unaligned buffers or ng != ceil(chunks / gch)");
    hipLaunchKernelGGL(k_sr_pack, dim3((unsigned)dgnn_grid_cap(dgnn_cdiv(rows, 4), 16)), dim3(256), 0, (hipStream_t)stream, A1, ld1, k1, A2, ld2, k2, rows, gch,
                       static_cast<char*>(dst), dst_row_bytes, scales, ng);
    return dgnn_check_launch("sr_pack");
}

extern "C" int dgnn_sr_unpack(const void* src, int64_t row_bytes, const float* scales, int ng, int gch, int C, int64_t rows, float* out, int64_t ldo,
                              void* stream) {
    DGNN_REQUIRE(rows >= 0 && C > 0 && C % 32 == 0 && ng >= 1, DGNN_E_INVALID, "sr_unpack: bad sizes");
    if (rows == 0) return DGNN_OK;
    DGNN_REQUIRE(src && scales && out && row_bytes >= (int64_t)(C / 32) * SRB && ldo >= C, DGNN_E_INVALID, "sr_unpack: null / short buffers");
    hipLaunchKernelGGL(k_sr_unpack, dim3((unsigned)dgnn_grid_cap(dgnn_cdiv(rows * (C / 2), 256))), dim3(256), 0, (hipStream_t)stream, static_cast<const char*>(src),
                       row_bytes, scales, ng, gch, C, rows, out, ldo);
    return dgnn_check_launch("sr_unpack");
}

extern "C" int64_t dgnn_sr_filter_prepared_bytes(int C) { return (C == 128 || C == 256 || C == 512) ? 16 + (int64_t)(C / 16) * 2 * 48 * 16 : 0; }

extern "C" int dgnn_sr_prepare_filter(const float* We, const float* be, int C, void* buf, void* stream) {
    DGNN_REQUIRE(We && be && buf && ((uintptr_t)buf % 16) == 0, DGNN_E_INVALID, "sr_prepare_filter: null / unaligned pointer");
    if (dgnn_sr_filter_prepared_bytes(C) == 0) return DGNN_E_UNSUPPORTED;
    hipLaunchKernelGGL(k_sr_prepare_filter, dim3(1), dim3(256), 0, (hipStream_t)stream, We, be, C, static_cast<char*>(buf));
    return dgnn_check_launch("sr_prepare_filter");
}

// a = mean_j x_j * lin_e(edge_attr_j) over the plan's in-edges (reference :75-80, :89-96) as SPLIT ROWS: a_out [n_dst][C / 32 x 128 B], a_scales
// [n_dst][ceil(C / 256)].  x: the source rows -- split rows (x_is_sr != 0; xs their scales [n_src][ceil(C / 256)]) or fp32 rows with row stride ldx
// (16-byte aligned rows); with fp32 rows and x_out != NULL the destinations' own rows x[:n_dst] are ALSO written as split rows (x_out, x_scales):
// the x_i operand of the layer's dgnn_linear_sr.  edge_attr: fp32 [E, 20] packed rows (lde == 20, 16-byte aligned), gathered by eid (NULL: plan
// order).  prep: dgnn_sr_prepare_filter(We, be, C).  C in {128, 256, 512} (512: two launches, one per 256 channels).
extern "C" int dgnn_sage_aggregate_sr(const int32_t* rowptr, const int32_t* src, const int32_t* eid, int64_t n_dst, const void* x, int x_is_sr, int64_t ldx,
                                      const float* xs, int C, const float* edge_attr, int64_t lde, const float* We, const float* be, const void* prep, void* a_out,
                                      float* a_scales, void* x_out, float* x_scales, void* stream) {
    DGNN_REQUIRE(n_dst >= 0, DGNN_E_INVALID, "sage_aggregate_sr: bad size");
    if (dgnn_sr_filter_prepared_bytes(C) == 0 || lde != 20 || ((uintptr_t)edge_attr % 16) != 0) return DGNN_E_UNSUPPORTED;
    if (!x_is_sr && (ldx % 4 != 0 || ((uintptr_t)x % 16) != 0 || ldx < C || C != 128)) return DGNN_E_UNSUPPORTED;
    if (x_is_sr && C == 128) return DGNN_E_UNSUPPORTED;
    if (n_dst == 0) return DGNN_OK;
    DGNN_REQUIRE(rowptr && src && x && edge_attr && We && be && prep && a_out && a_scales && (!x_is_sr || xs) && ((uintptr_t)a_out % 16) == 0 &&
                     (!x_out || (x_scales && !x_is_sr && ((uintptr_t)x_out % 16) == 0)) && ((uintptr_t)prep % 16) == 0 && (!x_is_sr || ((uintptr_t)x % 16) == 0),
                 DGNN_E_INVALID, "sage_aggregate_sr: null / unaligned pointer");
    const int NBv = agg_nb(C), np = agg_passes(C), ng = (C + 255) / 256;
    const int64_t rowb = (int64_t)(C / 32) * SRB;
    const size_t lds = (size_t)NBv * 2 * 48 * 16;
    static const int wg_per_cu = getenv("DGNN_AGG_SR_WGS") ? atoi(getenv("DGNN_AGG_SR_WGS")) : 2;
    // (the kernel deals the groups of 4 cells to the XCDs' eighths by blockIdx.x & 7: with fewer than 8 workgroups the eighths without one were
    // never computed -- a block of fewer than 256 destination cells, round 6 -- so a launch has at least 8, the surplus ones leave at once)
    int nwg_ = dgnn_grid_cap(dgnn_cdiv(dgnn_cdiv(n_dst, 4), 8), wg_per_cu < 1 ? 1 : wg_per_cu);
    if (nwg_ < 8) nwg_ = 8;
    const dim3 grid((unsigned)nwg_), block(512);
    hipStream_t st = (hipStream_t)stream;
    static const int ilv = getenv("DGNN_AGG_SR_ILV") ? atoi(getenv("DGNN_AGG_SR_ILV")) : 2;   // measured: 1 / 2 / 4 interleaved chains within 1 %.  (Round 6: with the ticket walk's registers the split-row form of ILV = 2 compiles to 137 VGPRs = ONE resident workgroup per CU, 2 wavefronts per SIMD; forced to 128 = two workgroups, 4 per SIMD, the launch is 3 % SLOWER -- profiles/r06_l1_probe.md)
    // the XCDs' group counters of the ticket walk (see the kernel): 8 ints per device, zeroed in stream order before every launch
    static const bool tickets_on = !(getenv("DGNN_AGG_SR_TICKETS") && getenv("DGNN_AGG_SR_TICKETS")[0] == '0');
    static const int tk_g = getenv("DGNN_AGG_SR_TK_G") && atoi(getenv("DGNN_AGG_SR_TK_G")) > 0 ? atoi(getenv("DGNN_AGG_SR_TK_G")) : 1;   // groups of 4 cells per ticket
    int* tickets = nullptr;
    // ... for split-row input (C >= 256).  Measured (profiles/r06_wide.md): C = 256 pass FETCH_SIZE 1.53 M -> 0.85 M KiB, L2 hit rate 52 -> 68 %, 802 -> 805 us;
    // the fp32-row form (C = 128, which also writes the own rows) 0.94 M -> 0.78 M KiB but 581 -> 695 us: it keeps the static walk unless
    // DGNN_AGG_SR_TICKETS=2.
    static const bool tickets_all = getenv("DGNN_AGG_SR_TICKETS") && getenv("DGNN_AGG_SR_TICKETS")[0] == '2';
    if (tickets_on && (x_is_sr || tickets_all)) {
        static int* tk_dev[DGNN_MAX_DEVICES];
        static std::mutex tk_m;
        int dev = 0;
        if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < DGNN_MAX_DEVICES) {
            std::lock_guard<std::mutex> lock(tk_m);
            if (!tk_dev[dev] && hipMalloc((void**)&tk_dev[dev], 16 * 2 * 512 * sizeof(int)) != hipSuccess) {
                tk_dev[dev] = nullptr;
                (void)hipGetLastError();
            }
            static unsigned tk_turn[DGNN_MAX_DEVICES];
            if (tk_dev[dev]) tickets = tk_dev[dev] + 2 * 512 * (tk_turn[dev]++ & 15);    // a ring of 16 counter sets (two passes each): launches in flight on other streams keep theirs
        }
    }
#define DGNN_AGG_SR(NB_, SR_, IL_)                                                                                                                       \
    hipLaunchKernelGGL((k_agg_sr<NB_, SR_, IL_>), grid, block, lds, st, rowptr, src, eid, n_dst, x, ldx, rowb, xs, ng, edge_attr, lde, We, be,            \
                       static_cast<const char*>(prep), pass, static_cast<char*>(a_out), rowb, a_scales, ng, static_cast<char*>(x_out), x_scales, sr_nt() | (tk_g << 8), \
                       tickets ? tickets + 512 * (pass & 1) : nullptr)
    if (tickets) (void)hipMemsetAsync(tickets, 0, 2 * 512 * sizeof(int), st);
    for (int pass = 0; pass < np; ++pass) {
        if (x_is_sr) {
            if (ilv == 1) DGNN_AGG_SR(16, true, 1);
            else if (ilv == 2) DGNN_AGG_SR(16, true, 2);
            else DGNN_AGG_SR(16, true, 4);
        } else {
            if (ilv == 1) DGNN_AGG_SR(8, false, 1);
            else if (ilv == 2) DGNN_AGG_SR(8, false, 2);
            else DGNN_AGG_SR(8, false, 4);
        }
    }
#undef DGNN_AGG_SR
    return dgnn_check_launch("sage_aggregate_sr");
}

// out = act(([A1 | A2] . Wp^T + bias) * scale + shift) on split rows: A1 (and A2, or NULL) split rows of C1 (C2) channels with their scales
// [M][ceil(C / 256)]; Wp / sw = dgnn_sr_pack of the fp32 weights [n_out][C1 (+ C2)] with gch = 0 (one scale per output channel); n_out a multiple of
// 256.  Exactly one output: split rows (out_sr [M][n_out / 32 x 128 B] + out_scales [M][n_out / 256]); fp32 rows out_f32 [M][n_out] (row stride ldo,
// 16-byte aligned); or logits [M][n_proj] = out . W3^T + b3 (W3 fp32 [n_proj][n_out], n_proj 1 or 2: the decoder's output Linear on the finished
// hidden rows, which never leave the launch; with n_out > 256 the logits must be ZERO on entry -- the column tiles add into them).  relu: 0 / 1.
extern "C" int dgnn_linear_sr(const void* A1, int64_t row_bytes1, const float* scales1, int C1, const void* A2, int64_t row_bytes2, const float* scales2, int C2,
                              const void* Wp, const float* sw, const float* bias, const float* scale, const float* shift, int relu, int64_t M, int n_out, void* out_sr,
                              int64_t out_row_bytes, float* out_scales, float* out_f32, int64_t ldo, const float* W3, const float* b3, int n_proj, float* logits,
                              void* stream) {
    DGNN_REQUIRE(M >= 0 && C1 > 0 && C2 >= 0, DGNN_E_INVALID, "linear_sr: bad sizes");
    if (n_out <= 0 || n_out % TN != 0 || C1 % 32 != 0 || C2 % 32 != 0 || (C1 > 256 && C1 % 256 != 0) || (C2 > 256 && C2 % 256 != 0)) return DGNN_E_UNSUPPORTED;
    if (logits && (n_proj < 1 || n_proj > 2 || n_out > 2 * TN)) return DGNN_E_UNSUPPORTED;   // (more than two column tiles: the atomic sum would depend on their order)
    if (M == 0) return DGNN_OK;
    DGNN_REQUIRE(A1 && scales1 && Wp && sw && (C2 == 0) == (A2 == nullptr) && (!A2 || scales2) && (scale == nullptr) == (shift == nullptr) &&
                     ((out_sr != nullptr) + (out_f32 != nullptr) + (logits != nullptr)) == 1 && (!out_sr || out_scales) && (!logits || W3),
                 DGNN_E_INVALID, "linear_sr: null pointer / exactly one kind of output");
    DGNN_REQUIRE(((uintptr_t)A1 % 16) == 0 && ((uintptr_t)A2 % 16) == 0 && ((uintptr_t)Wp % 16) == 0 && row_bytes1 % 16 == 0 && row_bytes2 % 16 == 0 &&
                     row_bytes1 >= (int64_t)(C1 / 32) * SRB && (!A2 || row_bytes2 >= (int64_t)(C2 / 32) * SRB) &&
                     (!out_sr || (((uintptr_t)out_sr % 16) == 0 && out_row_bytes % 16 == 0 && out_row_bytes >= (int64_t)(n_out / 32) * SRB)) &&
                     (!out_f32 || (((uintptr_t)out_f32 % 16) == 0 && ldo % 4 == 0 && ldo >= n_out)),
                 DGNN_E_INVALID, "linear_sr: unaligned / short buffers");
    SrPart p1{static_cast<const char*>(A1), row_bytes1, scales1, C1 / 32, (C1 + 255) / 256, C1 >= 256 ? 8 : C1 / 32};
    SrPart p2{static_cast<const char*>(A2), row_bytes2, scales2, C2 / 32, C2 ? (C2 + 255) / 256 : 0, C2 >= 256 ? 8 : (C2 ? C2 / 32 : 1)};
    SrOut o{static_cast<char*>(out_sr), out_row_bytes, out_scales, n_out / TN, out_f32, ldo, out_sr ? 0 : (out_f32 ? 1 : 2)};
    const int ncb = n_out / TN;
    const int64_t mt = dgnn_cdiv(M, TM);
    const dim3 grid((unsigned)(dgnn_cdiv(mt, 8) * 8 * ncb));
    static const bool v1 = getenv("DGNN_GEMM_SR_V1") && getenv("DGNN_GEMM_SR_V1")[0] == '1';      // the first arrangement (both operands through LDS), for A/B
    if (v1 && !logits) {
        static bool attr_set[DGNN_MAX_DEVICES] = {};
        dgnn_allow_dynamic_lds((const void*)k_gemm_sr, G_SMEM, attr_set);
        hipLaunchKernelGGL(k_gemm_sr, grid, dim3(GT), G_SMEM, (hipStream_t)stream, p1, p2, static_cast<const char*>(Wp), (int64_t)((C1 + C2) / 32) * SRB, sw, bias,
                           scale, shift, relu ? 1 : 0, M, n_out, o);
        return dgnn_check_launch("linear_sr");
    }
    static const int cps = getenv("DGNN_GEMM_SR_CPS") ? atoi(getenv("DGNN_GEMM_SR_CPS")) : 1;   // measured (tools/gpu_wide_ab.sh): three one-chunk stages 2-3 % ahead of two two-chunk stages
    if (cps == 1) {
        static bool attr_a[DGNN_MAX_DEVICES] = {};
        dgnn_allow_dynamic_lds((const void*)k_gemm_sr2<1>, G2<1>::SMEM, attr_a);
        hipLaunchKernelGGL(k_gemm_sr2<1>, grid, dim3(GT), G2<1>::SMEM, (hipStream_t)stream, p1, p2, static_cast<const char*>(Wp), (int64_t)((C1 + C2) / 32) * SRB, sw, bias,
                           scale, shift, relu ? 1 : 0, M, n_out, o, W3, b3, n_proj, logits, sr_nt());
    } else {
        static bool attr_b[DGNN_MAX_DEVICES] = {};
        dgnn_allow_dynamic_lds((const void*)k_gemm_sr2<2>, G2<2>::SMEM, attr_b);
        hipLaunchKernelGGL(k_gemm_sr2<2>, grid, dim3(GT), G2<2>::SMEM, (hipStream_t)stream, p1, p2, static_cast<const char*>(Wp), (int64_t)((C1 + C2) / 32) * SRB, sw, bias,
                           scale, shift, relu ? 1 : 0, M, n_out, o, W3, b3, n_proj, logits, sr_nt());
    }
    return dgnn_check_launch("linear_sr");
}
