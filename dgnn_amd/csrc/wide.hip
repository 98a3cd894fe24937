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
// buffer: 16-byte header (sWe, 1 / sWe, 0, 0), then entries (cb, part, g, j) x 16 bytes: cb < NB = C / 16 (the lane's cb-th position), part hi / lo,
// k-group g < 3 (k = 8 g .. 8 g + 7: attributes 0..19, the bias at k = 20, zeros), j < 16 (channel group): position P = NB j + cb of the row, i.e.
// channel 32 (P / 32) + PI(P % 32)
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
    const int NB = C / 16;
    for (int e = threadIdx.x; e < NB * 48; e += blockDim.x) {
        const int cb = e / 48, gj = e - cb * 48, g = gj >> 4, j = gj & 15;
        const int P = NB * j + cb, c = (P & ~31) + sr_chan(P & 31);
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
        uint4* dst = reinterpret_cast<uint4*>(buf + 16 + ((cb * 2) * 48 + gj) * 16);
        dst[0] = make_uint4(ph[0], ph[1], ph[2], ph[3]);
        dst[48] = make_uint4(pl[0], pl[1], pl[2], pl[3]);
    }
}

// ---- k_agg_sr ----------------------------------------------------------------------------------------------------------------------------------
// Lane (j = lane & 15, t = lane >> 4) of a wavefront step owns NB = C / 16 contiguous POSITIONS [NB j, NB j + NB) of cell t's row (of each of its 4
// neighbour rows, and of the finished row); the filter product's C/D layout puts the 4 in-edges of cell t into the 4 accumulator registers of the
// lanes (., t), so sum_j x_j * phi_j is an in-lane, in-order sum (plan order = the reference's CPU scatter order).
// XSR: the source rows are split rows (scales xs [n_src, ng]); else fp32 rows (ldx floats) -- the layer behind a fused layer -- and, when xo != NULL,
// the cell's OWN row is also written as a split row (the x_i operand of this layer's GEMM).
template <int NB, bool XSR>
__global__ void __launch_bounds__(512, 2) k_agg_sr(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ src, const int32_t* __restrict__ eid,
                                                   int64_t n_dst, const void* __restrict__ x_, int64_t ldx, const float* __restrict__ xs, const float* __restrict__ ea,
                                                   int64_t lde, const float* __restrict__ We, const float* __restrict__ be, const char* __restrict__ prep,
                                                   char* __restrict__ ao, float* __restrict__ as, char* __restrict__ xo, float* __restrict__ xos) {
    constexpr int C = NB * 16, NG = (C + 255) / 256, LPG = 16 / NG;   // lanes of a cell per scale group
    constexpr int NSB = NB / 8;                                        // sub-blocks of 8 positions per lane
    constexpr int ROWB = (C / 32) * SRB;
    extern __shared__ __attribute__((aligned(16))) char agg_smem[];
    char* const bpbuf = agg_smem;                                      // [cb][part][48] x 16 B
    const int lane = lane_id(), w = wave_id_uniform();
    const int jcol = lane & 15, tq = lane >> 4;
    const f32x4_t hd = *reinterpret_cast<const f32x4_t*>(prep);
    const float inv_sWe = hd[1];
    for (int i = threadIdx.x; i < NB * 2 * 48; i += blockDim.x) reinterpret_cast<uint4*>(bpbuf)[i] = reinterpret_cast<const uint4*>(prep + 16)[i];
    __syncthreads();

    const int64_t nq = (n_dst + 3) / 4;
    const int nwg = gridDim.x, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, wg_per_xcd = (nwg + 7 - xcd) >> 3;
    const int64_t per = (nq + 7) / 8, q_lo = xcd * per, q_hi = min(nq, q_lo + per);
    const int wpx = wg_per_xcd * 8;                                    // wavefronts walking this XCD's eighth of the cells
    const int P0 = NB * jcol;                                          // the lane's first position
    const float* const xf = static_cast<const float*>(x_);
    const char* const xb = static_cast<const char*>(x_);

    for (int64_t q = q_lo + slot * 8 + w; q < q_hi; q += wpx) {
        const int64_t i0 = q * 4;
        const int nv = (int)(n_dst - i0 < 4 ? n_dst - i0 : 4);
        const int vb = rowptr[i0 + (lane < nv ? lane : nv)];
        const int b0 = __builtin_amdgcn_readfirstlane(vb);
        const bool regular = __all(vb == b0 + 4 * (lane < nv ? lane : nv)) != 0;
        const int tl = tq < nv ? tq : nv - 1;                          // a short group at the end of the graph: clamped (duplicated) cells
        const int64_t cell = i0 + tl;
        float aout[NB];
        if (regular) {
            const int k_me = b0 + (lane < 4 * nv ? lane : 4 * nv - 1);
            const int vsrc = src[k_me];
            const int veid = eid ? eid[k_me] : k_me;
            // A operand: lane (edge jcol, k-group tq) holds attributes 8 tq .. 8 tq + 7 of its edge; k = 20 is the constant 1 of the bias row
            const float* er = ea + (int64_t)__shfl(veid, jcol) * lde;
            const f32x4_t q0 = *reinterpret_cast<const f32x4_t*>(er + 8 * (tq < 2 ? tq : 2));
            const f32x4_t q1 = *reinterpret_cast<const f32x4_t*>(er + 8 * (tq < 1 ? tq : 1) + 4);
            int sidx[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) sidx[r] = __shfl(vsrc, tl * 4 + r);
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
                const int P = P0 + 8 * sb, ch = P >> 5, p0 = P & 31;
                float xr[4][8];
                if constexpr (XSR) {
                    uint4 hh[4], ll[4];
                    float isx[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const char* rp = xb + (int64_t)sidx[r] * ROWB + ch * SRB + p0 * 2;
                        hh[r] = *reinterpret_cast<const uint4*>(rp);
                        ll[r] = *reinterpret_cast<const uint4*>(rp + 64);
                        const float s_ = xs[(int64_t)sidx[r] * NG + (P >> 8)];
                        isx[r] = b_of(s_) == SR_ZERO_BITS ? 0.f : pow2_inv(s_);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const uint32_t hw[4] = {hh[r].x, hh[r].y, hh[r].z, hh[r].w}, lw[4] = {ll[r].x, ll[r].y, ll[r].z, ll[r].w};
#pragma unroll
                        for (int d = 0; d < 4; ++d) {
                            const f16x2_t h2 = __builtin_bit_cast(f16x2_t, hw[d]), l2 = __builtin_bit_cast(f16x2_t, lw[d]);
                            xr[r][2 * d] = ((float)h2[0] + (float)l2[0]) * isx[r];      // hi + lo is exact in fp32 (two disjoint 11-bit pieces)
                            xr[r][2 * d + 1] = ((float)h2[1] + (float)l2[1]) * isx[r];
                        }
                    }
                } else {
                    // positions p0 .. p0 + 7 of chunk ch = channels 32 ch + PI(p0) .. + 3 and 32 ch + PI(p0 + 4) .. + 3
                    const int ca = ch * 32 + sr_chan(p0), cb_ = ch * 32 + sr_chan(p0 + 4);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float* rp = xf + (int64_t)sidx[r] * ldx;
                        const f32x4_t a4 = *reinterpret_cast<const f32x4_t*>(rp + ca), b4 = *reinterpret_cast<const f32x4_t*>(rp + cb_);
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            xr[r][i] = a4[i];
                            xr[r][4 + i] = b4[i];
                        }
                    }
                }
#pragma unroll
                for (int c8 = 0; c8 < 8; ++c8) {
                    const int cb = 8 * sb + c8;
                    const char* bp = bpbuf + ((cb * 2) * 48 + (tq < 3 ? tq : 0) * 16 + jcol) * 16;   // k-group 3 re-reads group 0: its A operand is zero
                    const f16x8 bh = H8(*reinterpret_cast<const uint4*>(bp)), bl = H8(*reinterpret_cast<const uint4*>(bp + 768));
                    f32x4_t d = {0.f, 0.f, 0.f, 0.f};
                    d = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, d, 0, 0, 0);
                    d = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, d, 0, 0, 0);
                    d = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, d, 0, 0, 0);
#pragma unroll
                    for (int r = 0; r < 4; ++r) d[r] *= inv_e[r];     // exact: powers of two
                    float a = __fmul_rn(xr[0][c8], d[0]);
#pragma unroll
                    for (int r = 1; r < 4; ++r) a = __fmaf_rn(xr[r][c8], d[r], a);
                    aout[cb] = a * (0.25f * inv_sWe);
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
                            const uint16_t* hp = reinterpret_cast<const uint16_t*>(xb + (int64_t)s_ * ROWB + (P >> 5) * SRB) + (P & 31);
                            const float sx = xs[(int64_t)s_ * NG + (P >> 8)];
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
        // the finished row: one scale per group of 256 channels (LPG lanes of the cell), split, stored as the lane's NB positions of the split row
        auto put_row = [&](const float (&v)[NB], char* orow, float* oscale) {
            float mx = 0.f;
#pragma unroll
            for (int i = 0; i < NB; i += 2) mx = fmaxf(fmaxf(mx, fabsf(v[i])), fabsf(v[i + 1]));
            uint32_t m = b_of(mx);
#pragma unroll
            for (int off = 1; off < LPG; off <<= 1) m = umax(m, (uint32_t)__shfl_xor((int)m, off));
            float s_store, s_mul;
            sr_scale(m, s_store, s_mul);
            if (tq < nv) {
                if ((jcol & (LPG - 1)) == 0) oscale[(jcol / LPG)] = s_store;
#pragma unroll
                for (int sb = 0; sb < NSB; ++sb) {
                    const int P = P0 + 8 * sb;
                    uint32_t hi[4], lo[4];
#pragma unroll
                    for (int d = 0; d < 4; ++d) split2h(v[8 * sb + 2 * d] * s_mul, v[8 * sb + 2 * d + 1] * s_mul, hi[d], lo[d]);
                    char* o = orow + (P >> 5) * SRB + (P & 31) * 2;
                    *reinterpret_cast<uint4*>(o) = make_uint4(hi[0], hi[1], hi[2], hi[3]);
                    *reinterpret_cast<uint4*>(o + 64) = make_uint4(lo[0], lo[1], lo[2], lo[3]);
                }
            }
        };
        put_row(aout, ao + cell * ROWB, as + cell * NG);
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
                put_row(xv, xo + cell * ROWB, xos + cell * NG);
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
        const int piece = qs ^ (r_t & 7);
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
    const int sw_ = l31 & 7;
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

}  // namespace

// =====================================================================================================================
// C ABI
// =====================================================================================================================
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
// order).  prep: dgnn_sr_prepare_filter(We, be, C).  C in {128, 256, 512}.
extern "C" int dgnn_sage_aggregate_sr(const int32_t* rowptr, const int32_t* src, const int32_t* eid, int64_t n_dst, const void* x, int x_is_sr, int64_t ldx,
                                      const float* xs, int C, const float* edge_attr, int64_t lde, const float* We, const float* be, const void* prep, void* a_out,
                                      float* a_scales, void* x_out, float* x_scales, void* stream) {
    DGNN_REQUIRE(n_dst >= 0, DGNN_E_INVALID, "sage_aggregate_sr: bad size");
    if (dgnn_sr_filter_prepared_bytes(C) == 0 || lde != 20 || ((uintptr_t)edge_attr % 16) != 0) return DGNN_E_UNSUPPORTED;
    if (!x_is_sr && (ldx % 4 != 0 || ((uintptr_t)x % 16) != 0 || ldx < C)) return DGNN_E_UNSUPPORTED;
    if (n_dst == 0) return DGNN_OK;
    DGNN_REQUIRE(rowptr && src && x && edge_attr && We && be && prep && a_out && a_scales && (!x_is_sr || xs) && ((uintptr_t)a_out % 16) == 0 &&
                     (!x_out || (x_scales && !x_is_sr && ((uintptr_t)x_out % 16) == 0)) && ((uintptr_t)prep % 16) == 0 && (!x_is_sr || ((uintptr_t)x % 16) == 0),
                 DGNN_E_INVALID, "sage_aggregate_sr: null / unaligned pointer");
    const int NBv = C / 16;
    const size_t lds = (size_t)NBv * 2 * 48 * 16;
    const dim3 grid((unsigned)dgnn_grid_cap(dgnn_cdiv(dgnn_cdiv(n_dst, 4), 8), 2)), block(512);
    hipStream_t st = (hipStream_t)stream;
#define DGNN_AGG_SR(NB_, SR_)                                                                                                                                   \
    hipLaunchKernelGGL((k_agg_sr<NB_, SR_>), grid, block, lds, st, rowptr, src, eid, n_dst, x, ldx, xs, edge_attr, lde, We, be, static_cast<const char*>(prep), \
                       static_cast<char*>(a_out), a_scales, static_cast<char*>(x_out), x_scales)
    if (x_is_sr) {
        if (NBv == 8) DGNN_AGG_SR(8, true);
        else if (NBv == 16) DGNN_AGG_SR(16, true);
        else DGNN_AGG_SR(32, true);
    } else {
        if (NBv == 8) DGNN_AGG_SR(8, false);
        else if (NBv == 16) DGNN_AGG_SR(16, false);
        else DGNN_AGG_SR(32, false);
    }
#undef DGNN_AGG_SR
    return dgnn_check_launch("sage_aggregate_sr");
}

// out = act(([A1 | A2] . Wp^T + bias) * scale + shift) on split rows: A1 (and A2, or NULL) split rows of C1 (C2) channels with their scales
// [M][ceil(C / 256)]; Wp / sw = dgnn_sr_pack of the fp32 weights [n_out][C1 (+ C2)] with gch = 0 (one scale per output channel); n_out a multiple of
// 256.  Output: split rows (out_sr [M][n_out / 32 x 128 B] + out_scales [M][n_out / 256]) or, out_sr == NULL, fp32 rows out_f32 [M][n_out] (row
// stride ldo, 16-byte aligned).  relu: 0 / 1.
extern "C" int dgnn_linear_sr(const void* A1, int64_t row_bytes1, const float* scales1, int C1, const void* A2, int64_t row_bytes2, const float* scales2, int C2,
                              const void* Wp, const float* sw, const float* bias, const float* scale, const float* shift, int relu, int64_t M, int n_out, void* out_sr,
                              int64_t out_row_bytes, float* out_scales, float* out_f32, int64_t ldo, void* stream) {
    DGNN_REQUIRE(M >= 0 && C1 > 0 && C2 >= 0, DGNN_E_INVALID, "linear_sr: bad sizes");
    if (n_out <= 0 || n_out % TN != 0 || C1 % 32 != 0 || C2 % 32 != 0 || (C1 > 256 && C1 % 256 != 0) || (C2 > 256 && C2 % 256 != 0)) return DGNN_E_UNSUPPORTED;
    if (M == 0) return DGNN_OK;
    DGNN_REQUIRE(A1 && scales1 && Wp && sw && (C2 == 0) == (A2 == nullptr) && (!A2 || scales2) && (scale == nullptr) == (shift == nullptr) &&
                     ((out_sr != nullptr) != (out_f32 != nullptr)) && (!out_sr || out_scales),
                 DGNN_E_INVALID, "linear_sr: null pointer / exactly one kind of output");
    DGNN_REQUIRE(((uintptr_t)A1 % 16) == 0 && ((uintptr_t)A2 % 16) == 0 && ((uintptr_t)Wp % 16) == 0 && row_bytes1 % 16 == 0 && row_bytes2 % 16 == 0 &&
                     row_bytes1 >= (int64_t)(C1 / 32) * SRB && (!A2 || row_bytes2 >= (int64_t)(C2 / 32) * SRB) &&
                     (!out_sr || (((uintptr_t)out_sr % 16) == 0 && out_row_bytes % 16 == 0 && out_row_bytes >= (int64_t)(n_out / 32) * SRB)) &&
                     (!out_f32 || (((uintptr_t)out_f32 % 16) == 0 && ldo % 4 == 0 && ldo >= n_out)),
                 DGNN_E_INVALID, "linear_sr: unaligned / short buffers");
    SrPart p1{static_cast<const char*>(A1), row_bytes1, scales1, C1 / 32, (C1 + 255) / 256, C1 >= 256 ? 8 : C1 / 32};
    SrPart p2{static_cast<const char*>(A2), row_bytes2, scales2, C2 / 32, C2 ? (C2 + 255) / 256 : 0, C2 >= 256 ? 8 : (C2 ? C2 / 32 : 1)};
    SrOut o{static_cast<char*>(out_sr), out_row_bytes, out_scales, n_out / TN, out_f32, ldo, out_sr ? 0 : 1};
    static bool attr_set[DGNN_MAX_DEVICES] = {};
    dgnn_allow_dynamic_lds((const void*)k_gemm_sr, G_SMEM, attr_set);
    const int ncb = n_out / TN;
    const int64_t mt = dgnn_cdiv(M, TM);
    const dim3 grid((unsigned)(dgnn_cdiv(mt, 8) * 8 * ncb));
    hipLaunchKernelGGL(k_gemm_sr, grid, dim3(GT), G_SMEM, (hipStream_t)stream, p1, p2, static_cast<const char*>(Wp), (int64_t)((C1 + C2) / 32) * SRB, sw, bias, scale,
                       shift, relu ? 1 : 0, M, n_out, o);
    return dgnn_check_launch("linear_sr");
}
