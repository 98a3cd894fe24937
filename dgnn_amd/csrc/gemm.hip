// Dense linears of the SAGE layer on the fp32 matrix cores (v_mfma_f32_32x32x2_f32).
//
//  dgnn_linear_fwd  : out = act((A1.W1^T + A2.W2^T + bias) * scale + shift)   "NN" with torch Linear weights
//  dgnn_linear_wgrad: dW  = A^T . B summed over rows                            "TN", two-stage deterministic
//
// fp32-input MFMA is an exact fp32 fmaf chain at the fp32 vector rate (MI355X guide: 157 TF peak),
// so these run the reference's fp32 arithmetic without a reduced-precision path.
//
// linear_fwd tile: 128 rows x 64 cols per 256-thread block; wave w owns rows 32w..32w+31 and both
// 32-column halves (2 accumulators = 32 VGPRs).  K is walked in chunks of 32 over the concatenation
// [A1 | A2]; both operands are staged in LDS with a 36-float row stride, which makes the 16-byte
// fragment reads (ds_read_b128) bank-conflict free for the 16-lane groups of that instruction.
// k-permutation: within a chunk, half-wave h of MFMA step (S,j) multiplies k = 8S + 4h + j, so one
// b128 read per operand feeds 4 MFMAs; A and B use the same map, hence every k is used exactly once.
#include <cstdlib>

#include "common.h"
#include "reduce_common.h"

namespace {

constexpr int BM = 128, BN = 64, BK = 32, LDT = BK + 4;

// stage a [rows x BK] tile of a row-major matrix into LDS (zero-filled outside [nrows) x [kmax))
template <int ROWS>
__device__ __forceinline__ void stage_tile(float* __restrict__ dst, const float* __restrict__ src, int64_t ld,
                                           int64_t row0, int64_t nrows, int k0, int kmax, bool vec) {
    // 256 threads: 8 threads per row (4 floats each), 32 rows per pass
    const int t = threadIdx.x, r = t >> 3, c = (t & 7) * 4;
#pragma unroll
    for (int p = 0; p < ROWS / 32; ++p) {
        const int rr = r + p * 32;
        const int64_t gr = row0 + rr;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (gr < nrows) {
            const float* g = src + gr * ld + k0 + c;
            if (vec && k0 + c + 3 < kmax) {
                v = *reinterpret_cast<const f32x4*>(g);
            } else {
                if (k0 + c + 0 < kmax) v[0] = g[0];
                if (k0 + c + 1 < kmax) v[1] = g[1];
                if (k0 + c + 2 < kmax) v[2] = g[2];
                if (k0 + c + 3 < kmax) v[3] = g[3];
            }
        }
        *reinterpret_cast<f32x4*>(dst + rr * LDT + c) = v;
    }
}

__global__ void __launch_bounds__(256) k_linear_fwd(const float* __restrict__ A1, int64_t lda1, int k1,
                                                    const float* __restrict__ W1, int64_t ldw1, bool vec1,
                                                    const float* __restrict__ A2, int64_t lda2, int k2,
                                                    const float* __restrict__ W2, int64_t ldw2, bool vec2,
                                                    const float* __restrict__ bias, const float* __restrict__ scale,
                                                    const float* __restrict__ shift, int relu, int64_t M, int n_out,
                                                    float* __restrict__ out, int64_t ldo) {
    __shared__ __attribute__((aligned(16))) float As[BM * LDT];
    __shared__ __attribute__((aligned(16))) float Ws[BN * LDT];
    const int lane = lane_id(), w = wave_id_uniform();
    // 1-D grid, column blocks fastest: the blocks that share a row panel of A run together and read it once from HBM (with the
    // row blocks fastest every column block streamed all of A again: 4x the traffic at n_out = 512)
    const int ncb = (n_out + BN - 1) / BN;
    const int64_t row0 = (int64_t)(blockIdx.x / ncb) * BM;
    const int col0 = (int)(blockIdx.x % ncb) * BN;
    f32x16 acc0, acc1;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc0[i] = acc1[i] = 0.f;
    const int nch1 = (k1 + BK - 1) / BK, nch2 = A2 ? (k2 + BK - 1) / BK : 0;
    const int h = lane >> 5, l31 = lane & 31;
    for (int ch = 0; ch < nch1 + nch2; ++ch) {
        const bool first = ch < nch1;
        const float* A = first ? A1 : A2;
        const float* W = first ? W1 : W2;
        const int64_t lda = first ? lda1 : lda2, ldw = first ? ldw1 : ldw2;
        const int kk = first ? k1 : k2, k0 = (first ? ch : ch - nch1) * BK;
        const bool vec = first ? vec1 : vec2;
        __syncthreads();
        stage_tile<BM>(As, A, lda, row0, M, k0, kk, vec);
        stage_tile<BN>(Ws, W, ldw, col0, n_out, k0, kk, vec);
        __syncthreads();
        const float* ap = As + (w * 32 + l31) * LDT + 4 * h;
        const float* bp0 = Ws + l31 * LDT + 4 * h;
        const float* bp1 = bp0 + 32 * LDT;
#pragma unroll
        for (int S = 0; S < 4; ++S) {
            const f32x4 av = *reinterpret_cast<const f32x4*>(ap + 8 * S);
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(bp0 + 8 * S);
            const f32x4 b1 = *reinterpret_cast<const f32x4*>(bp1 + 8 * S);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], b0[j], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], b1[j], acc1, 0, 0, 0);
            }
        }
    }
    // epilogue: C/D layout col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int col = col0 + half * 32 + l31;
        if (col >= n_out) continue;
        const float bb = bias ? bias[col] : 0.f;
        const float sc = scale ? scale[col] : 1.f;
        const float sh = shift ? shift[col] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t row = row0 + w * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (row >= M) continue;
            float v = (half ? acc1[r] : acc0[r]) + bb;
            if (scale) v = __fmaf_rn(v, sc, sh);
            if (relu & 1) v = fmaxf(v, 0.f);
            if (relu & DGNN_LINEAR_ACCUMULATE) v += out[row * ldo + col];
            out[row * ldo + col] = v;
        }
    }
}

// ---- weight gradient: dW[na, nb] = sum_rows A[r, :]^T B[r, :] -----------------------------------
// block = 256 threads (4 waves) computes a 64 x 64 tile of dW over a slice of rows; wave (wa,wb)
// owns a 32x32 sub-tile.  Row slices of RK=32 rows are staged row-major, so the MFMA operands
// (A-operand lane: [i = column of A][k = row]) are conflict-free ds_read_b32 along a row.
constexpr int RK = 32, WT = 64, LDW_T = WT + 1;
constexpr int WGRAD_SPLITS = 512;

__global__ void __launch_bounds__(256) k_linear_wgrad(const float* __restrict__ A, int64_t lda, int na,
                                                      const float* __restrict__ B, int64_t ldb, int nb, int64_t M,
                                                      int64_t rows_per_split, float* __restrict__ partials) {
    __shared__ float As[RK * LDW_T];
    __shared__ float Bs[RK * LDW_T];
    const int lane = lane_id(), w = wave_id_uniform();
    const int wa = w >> 1, wb = w & 1, h = lane >> 5, l31 = lane & 31;
    const int a0 = blockIdx.x * WT, b0 = blockIdx.y * WT;
    const int64_t r_beg = (int64_t)blockIdx.z * rows_per_split;
    const int64_t r_end = min(M, r_beg + rows_per_split);
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const int t = threadIdx.x, tr = t >> 3, tc = (t & 7) * 8;  // 32 rows x 64 cols, 8 floats per thread
    for (int64_t r0 = r_beg; r0 < r_end; r0 += RK) {
        __syncthreads();
        const int64_t gr = r0 + tr;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int ca = a0 + tc + j, cb = b0 + tc + j;
            As[tr * LDW_T + tc + j] = (gr < r_end && ca < na) ? A[gr * lda + ca] : 0.f;
            Bs[tr * LDW_T + tc + j] = (gr < r_end && cb < nb) ? B[gr * ldb + cb] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < RK / 2; ++s) {
            const float av = As[(2 * s + h) * LDW_T + wa * 32 + l31];
            const float bv = Bs[(2 * s + h) * LDW_T + wb * 32 + l31];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
        }
    }
    float* P = partials + (int64_t)blockIdx.z * na * nb;
    const int col = b0 + wb * 32 + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = a0 + wa * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (row < na && col < nb) P[(int64_t)row * nb + col] = acc[r];
    }
}

// dW (+)= sum over the row splits, 16 outputs x 16 slices per block in a fixed order (see k_reduce_slabs)
__global__ void __launch_bounds__(256) k_wgrad_reduce(const float* __restrict__ partials, int splits, int na, int nb,
                                                      float* __restrict__ dW, int64_t lddw, int accumulate) {
    __shared__ float red[16][17];
    const int o = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int i = blockIdx.x * 16 + o;
    const bool live = i < na * nb;
    float p = 0.f;
    if (live)
        for (int z = sl; z < splits; z += 16) p += partials[(int64_t)z * na * nb + i];
    red[sl][o] = p;
    __syncthreads();
    if (sl != 0 || !live) return;
    float s = 0.f;
    for (int k = 0; k < 16; ++k) s += red[k][o];
    float* out = dW + (int64_t)(i / nb) * lddw + (i % nb);
    *out = accumulate ? *out + s : s;
}

// =====================================================================================================================
// bf16-storage GEMMs (BASELINE config 3): activations bf16 in HBM, weights fp32 in HBM (master copies, rounded to bf16 when a
// block stages them), one v_mfma_f32_32x32x16_bf16 per product, fp32 accumulation, fp32 epilogue, output bf16 or fp32.
//
// k_linear_fwd_b: 128 x 64 output tile per 256-thread block as above; K is walked in chunks of 64 over [A1 | A2].  LDS rows are
// 64 bf16 + 16 B pad = 144 B (an odd number of 16-B slots), so the b128 fragment reads of a 16-lane group never share a bank.
// =====================================================================================================================
constexpr int BKB = 64, LDB = BKB * 2 + 16;  // bytes per LDS row

__device__ __forceinline__ uint32_t pk_bf16(float a, float b) {
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f2{a, b}, bf2));
}

// A tile: ROWS x 64 bf16 from a bf16 matrix; 8 threads per row, 16 B (8 elements) each
template <int ROWS>
__device__ __forceinline__ void stage_a_bf16(char* __restrict__ dst, const uint16_t* __restrict__ src, int64_t ld, int64_t row0, int64_t nrows,
                                             int k0, int kmax, bool vec) {
    const int t = threadIdx.x, r = t >> 3, c = (t & 7) * 8;
#pragma unroll
    for (int p = 0; p < ROWS / 32; ++p) {
        const int rr = r + p * 32;
        const int64_t gr = row0 + rr;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (gr < nrows && k0 + c < kmax) {
            const uint16_t* g = src + gr * ld + k0 + c;
            if (vec && k0 + c + 8 <= kmax) {
                v = *reinterpret_cast<const uint4*>(g);
            } else {
                uint32_t e[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) e[j] = (k0 + c + j < kmax) ? (uint32_t)g[j] : 0u;
                v = make_uint4(e[0] | (e[1] << 16), e[2] | (e[3] << 16), e[4] | (e[5] << 16), e[6] | (e[7] << 16));
            }
        }
        *reinterpret_cast<uint4*>(dst + rr * LDB + c * 2) = v;
    }
}

// W tile: ROWS x 64 from an fp32 matrix, rounded to bf16
template <int ROWS>
__device__ __forceinline__ void stage_w_bf16(char* __restrict__ dst, const float* __restrict__ src, int64_t ld, int64_t row0, int64_t nrows,
                                             int k0, int kmax, bool vec) {
    const int t = threadIdx.x, r = t >> 3, c = (t & 7) * 8;
#pragma unroll
    for (int p = 0; p < ROWS / 32; ++p) {
        const int rr = r + p * 32;
        const int64_t gr = row0 + rr;
        float e[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) e[j] = 0.f;
        if (gr < nrows && k0 + c < kmax) {
            const float* g = src + gr * ld + k0 + c;
            if (vec && k0 + c + 8 <= kmax) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(g), b = *reinterpret_cast<const f32x4*>(g + 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) { e[j] = a[j]; e[4 + j] = b[j]; }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) if (k0 + c + j < kmax) e[j] = g[j];
            }
        }
        *reinterpret_cast<uint4*>(dst + rr * LDB + c * 2) = make_uint4(pk_bf16(e[0], e[1]), pk_bf16(e[2], e[3]), pk_bf16(e[4], e[5]), pk_bf16(e[6], e[7]));
    }
}

typedef short bf16x8_t __attribute__((ext_vector_type(8)));

template <typename TO>
__global__ void __launch_bounds__(256) k_linear_fwd_b(const uint16_t* __restrict__ A1, int64_t lda1, int k1, const float* __restrict__ W1,
                                                      int64_t ldw1, bool va1, bool vw1, const uint16_t* __restrict__ A2, int64_t lda2, int k2,
                                                      const float* __restrict__ W2, int64_t ldw2, bool va2, bool vw2,
                                                      const float* __restrict__ bias, const float* __restrict__ scale,
                                                      const float* __restrict__ shift, int relu, int64_t M, int n_out, TO* __restrict__ out,
                                                      int64_t ldo) {
    __shared__ __attribute__((aligned(16))) char As[BM * LDB];
    __shared__ __attribute__((aligned(16))) char Ws[BN * LDB];
    const int lane = lane_id(), w = wave_id_uniform();
    // 1-D grid, column blocks fastest: the blocks that share a row panel of A run together and read it once from HBM (with the
    // row blocks fastest every column block streamed all of A again: 4x the traffic at n_out = 512)
    const int ncb = (n_out + BN - 1) / BN;
    const int64_t row0 = (int64_t)(blockIdx.x / ncb) * BM;
    const int col0 = (int)(blockIdx.x % ncb) * BN;
    f32x16 acc0, acc1;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc0[i] = acc1[i] = 0.f;
    const int nch1 = (k1 + BKB - 1) / BKB, nch2 = A2 ? (k2 + BKB - 1) / BKB : 0;
    const int h = lane >> 5, l31 = lane & 31;
    for (int ch = 0; ch < nch1 + nch2; ++ch) {
        const bool first = ch < nch1;
        const int kk = first ? k1 : k2, k0 = (first ? ch : ch - nch1) * BKB;
        __syncthreads();
        stage_a_bf16<BM>(As, first ? A1 : A2, first ? lda1 : lda2, row0, M, k0, kk, first ? va1 : va2);
        stage_w_bf16<BN>(Ws, first ? W1 : W2, first ? ldw1 : ldw2, col0, n_out, k0, kk, first ? vw1 : vw2);
        __syncthreads();
        const char* ap = As + (w * 32 + l31) * LDB + h * 16;
        const char* bp0 = Ws + l31 * LDB + h * 16;
        const char* bp1 = bp0 + 32 * LDB;
#pragma unroll
        for (int S = 0; S < BKB / 16; ++S) {
            const bf16x8_t av = *reinterpret_cast<const bf16x8_t*>(ap + 32 * S);
            const bf16x8_t b0 = *reinterpret_cast<const bf16x8_t*>(bp0 + 32 * S);
            const bf16x8_t b1 = *reinterpret_cast<const bf16x8_t*>(bp1 + 32 * S);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, b0, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, b1, acc1, 0, 0, 0);
        }
    }
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int col = col0 + half * 32 + l31;
        if (col >= n_out) continue;
        const float bb = bias ? bias[col] : 0.f;
        const float sc = scale ? scale[col] : 1.f;
        const float sh = scale ? shift[col] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t row = row0 + w * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (row >= M) continue;
            float v = (half ? acc1[r] : acc0[r]) + bb;
            if (scale) v = __fmaf_rn(v, sc, sh);
            if (relu & 1) v = fmaxf(v, 0.f);
            if (relu & DGNN_LINEAR_ACCUMULATE) v += dgnn_ld(out + row * ldo + col);
            dgnn_st(out + row * ldo + col, v);
        }
    }
}

// 64 x 64 output tile with the next chunk's loads in flight under the products (round 6): for the problems between k_linear_fwd_b (128 x 64 tiles,
// loads and products one after the other) and k_linear_fwd_b_small (no LDS: both operands re-read per 32 x 32 block) -- the reference's training widths
// at batch 1024, see k_linear_fwd_x3_mid.  k-steps in k_linear_fwd_b's order: bit-identical results.
// KS = 4: four groups of four wavefronts walk a quarter of the chunks each (see k_linear_fwd_x3_mid<4>).
template <typename TO, int KS>
__global__ void __launch_bounds__(256 * KS, KS == 1 ? 2 : 1) k_linear_fwd_b_mid(const uint16_t* __restrict__ A1, int64_t lda1, int k1, const float* __restrict__ W1,
                                                                            int64_t ldw1, bool va1, bool vw1, const uint16_t* __restrict__ A2, int64_t lda2, int k2,
                                                                            const float* __restrict__ W2, int64_t ldw2, bool va2, bool vw2,
                                                                            const float* __restrict__ bias, const float* __restrict__ scale,
                                                                            const float* __restrict__ shift, int relu, int64_t M, int n_out, TO* __restrict__ out,
                                                                            int64_t ldo) {
    constexpr int TM = 64, TN = 64;
    extern __shared__ __attribute__((aligned(16))) char midb_smem[];     // per group: As [TM][LDB] | Ws [TN][LDB]
    const int lane = lane_id(), w = wave_id_uniform();
    const int g = w >> 2, wq = w & 3;
    const int wr = wq >> 1, wc = wq & 1, h = lane >> 5, l31 = lane & 31;
    char* const As = midb_smem + g * ((TM + TN) * LDB);
    char* const Ws = As + TM * LDB;
    const int ncb = (n_out + TN - 1) / TN;
    const int64_t row0 = (int64_t)(blockIdx.x / ncb) * TM;
    const int col0 = (int)(blockIdx.x % ncb) * TN;
    const int t = (int)threadIdx.x & 255, sr = t >> 3, scol = (t & 7) * 8;      // staging: 8 threads per row, 8 elements each, 32 rows per pass
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const int nch1 = (k1 + BKB - 1) / BKB, nch2 = A2 ? (k2 + BKB - 1) / BKB : 0, nch = nch1 + nch2;
    const int cpg = (nch + KS - 1) / KS;
    const int ch_lo = g * cpg, ch_hi = ch_lo + cpg < nch ? ch_lo + cpg : nch;
    uint4 ra[2];
    float rw[2][8];
    auto load_chunk = [&](int ch) {
        const bool first = ch < nch1;
        const int kk = first ? k1 : k2, k0 = (first ? ch : ch - nch1) * BKB;
        const uint16_t* A = first ? A1 : A2;
        const float* W = first ? W1 : W2;
        const int64_t lda = first ? lda1 : lda2, ldw = first ? ldw1 : ldw2;
        const bool va = first ? va1 : va2, vw = first ? vw1 : vw2;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int rr = sr + p * 32;
            const int64_t gr = row0 + rr;
            ra[p] = make_uint4(0u, 0u, 0u, 0u);
            if (gr < M && k0 + scol < kk) {
                const uint16_t* gp = A + gr * lda + k0 + scol;
                if (va && k0 + scol + 8 <= kk) {
                    ra[p] = *reinterpret_cast<const uint4*>(gp);
                } else {
                    uint32_t e[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) e[j] = (k0 + scol + j < kk) ? (uint32_t)gp[j] : 0u;
                    ra[p] = make_uint4(e[0] | (e[1] << 16), e[2] | (e[3] << 16), e[4] | (e[5] << 16), e[6] | (e[7] << 16));
                }
            }
            const int64_t gc = col0 + rr;
#pragma unroll
            for (int j = 0; j < 8; ++j) rw[p][j] = 0.f;
            if (gc < n_out && k0 + scol < kk) {
                const float* gp = W + gc * ldw + k0 + scol;
                if (vw && k0 + scol + 8 <= kk) {
                    const f32x4 a = *reinterpret_cast<const f32x4*>(gp), b = *reinterpret_cast<const f32x4*>(gp + 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) { rw[p][j] = a[j]; rw[p][4 + j] = b[j]; }
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) if (k0 + scol + j < kk) rw[p][j] = gp[j];
                }
            }
        }
    };
    if (ch_lo < ch_hi) load_chunk(ch_lo);
    for (int i = 0; i < cpg; ++i) {
        const int ch = ch_lo + i;
        const bool live = ch < ch_hi;
        __syncthreads();                 // the previous chunk's fragments have been read
        if (live) {
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int rr = sr + p * 32;
                *reinterpret_cast<uint4*>(As + rr * LDB + scol * 2) = ra[p];
                *reinterpret_cast<uint4*>(Ws + rr * LDB + scol * 2) = make_uint4(pk_bf16(rw[p][0], rw[p][1]), pk_bf16(rw[p][2], rw[p][3]), pk_bf16(rw[p][4], rw[p][5]),
                                                                                 pk_bf16(rw[p][6], rw[p][7]));
            }
        }
        __syncthreads();
        if (ch + 1 < ch_hi) load_chunk(ch + 1);      // in flight under the products below
        if (!live) continue;
        const char* ap = As + (wr * 32 + l31) * LDB + h * 16;
        const char* bp = Ws + (wc * 32 + l31) * LDB + h * 16;
#pragma unroll
        for (int S = 0; S < BKB / 16; ++S) {
            const bf16x8_t av = *reinterpret_cast<const bf16x8_t*>(ap + 32 * S);
            const bf16x8_t bv = *reinterpret_cast<const bf16x8_t*>(bp + 32 * S);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc, 0, 0, 0);
        }
    }
    if (KS > 1) {
        __syncthreads();
        float* red = reinterpret_cast<float*>(midb_smem);
        if (g > 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) red[(((g - 1) * 4 + wq) * 16 + r) * 64 + lane] = acc[r];
        }
        __syncthreads();
        if (g > 0) return;
#pragma unroll
        for (int gg = 0; gg < KS - 1; ++gg)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] += red[((gg * 4 + wq) * 16 + r) * 64 + lane];
    }
    const int col = col0 + wc * 32 + l31;
    if (col < n_out) {
        const float bb = bias ? bias[col] : 0.f;
        const float sc = scale ? scale[col] : 1.f;
        const float sh = scale ? shift[col] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t row = row0 + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (row >= M) continue;
            float v = acc[r] + bb;
            if (scale) v = __fmaf_rn(v, sc, sh);
            if (relu & 1) v = fmaxf(v, 0.f);
            if (relu & DGNN_LINEAR_ACCUMULATE) v += dgnn_ld(out + row * ldo + col);
            dgnn_st(out + row * ldo + col, v);
        }
    }
}

// Small problems (M <= 16384 rows: the inner blocks of a training step, the decoder) in bf16 storage: k_linear_fwd_b's 128-row tiles leave most
// of the chip idle there and walk K behind two barriers per 64 columns (13-20 us per launch for 0.1 GFLOP; the bf16-storage training step spent
// 0.3 ms in 22 such launches).  Like k_linear_fwd_x3_small, ONE WAVEFRONT owns a 32 x 32 output block and takes its fragments straight from
// global memory in the MFMA's layout (A: 8 consecutive bf16 of a row = one 16-byte load; W: 8 fp32 rounded to bf16 in registers), four k-steps
// of loads in flight, no LDS, no barrier.  k-steps in k_linear_fwd_b's order: bit-identical results.
// SPLITK (round 6; K = k1 + k2 >= 1024, the reference's training widths): the four wavefronts of a workgroup share ONE 32 x 32 output block, each walks a
// quarter of every operand pair's k-steps, the three partial blocks cross LDS and wavefront 0 adds them in wavefront order and runs the epilogue.  One
// wavefront walking 64+ k-steps behind four loads in flight was a 25-50 us dependent chain per launch at M <= 16k whatever the chip had free (the
// modelnet-width step spent 280-350 us in these launches).  Another summation order than the tiled kernels' -- fp32 rounding level, used only for K >= 1024.
template <typename TO, bool SPLITK = false>
__global__ void __launch_bounds__(256) k_linear_fwd_b_small(const uint16_t* __restrict__ A1, int64_t lda1, int k1, const float* __restrict__ W1,
                                                            int64_t ldw1, bool va1, bool vw1, const uint16_t* __restrict__ A2, int64_t lda2, int k2,
                                                            const float* __restrict__ W2, int64_t ldw2, bool va2, bool vw2,
                                                            const float* __restrict__ bias, const float* __restrict__ scale,
                                                            const float* __restrict__ shift, int relu, int64_t M, int n_out, TO* __restrict__ out,
                                                            int64_t ldo) {
    __shared__ float splitk_red[SPLITK ? 3 * 16 * 64 : 1];
    const int lane = lane_id(), w = wave_id_uniform();
    const int h = lane >> 5, l31 = lane & 31;
    const int nct = (n_out + 31) / 32;
    const int64_t tile = SPLITK ? (int64_t)blockIdx.x : (int64_t)blockIdx.x * 4 + w;
    const int64_t rt = tile / nct;
    const int ct = (int)(tile - rt * nct);
    if (rt * 32 >= M) return;
    const int64_t row = rt * 32 + l31, rowc = row < M ? row : M - 1;
    const int col = ct * 32 + l31, colc = col < n_out ? col : n_out - 1;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    for (int part = 0; part < 2; ++part) {
        const uint16_t* A = part == 0 ? A1 : A2;
        if (!A) break;
        const float* W = part == 0 ? W1 : W2;
        const int kk = part == 0 ? k1 : k2;
        const bool va = part == 0 ? va1 : va2, vw = part == 0 ? vw1 : vw2;
        const uint16_t* ap = A + rowc * (part == 0 ? lda1 : lda2) + 8 * h;
        const float* wp = W + (int64_t)colc * (part == 0 ? ldw1 : ldw2) + 8 * h;
        const int nst_all = (kk + 15) / 16;
        const int q_ = (nst_all + 3) / 4;
        const int st0 = SPLITK ? w * q_ : 0, nst = SPLITK ? (st0 + q_ < nst_all ? st0 + q_ : nst_all) : nst_all;
        struct Frag {
            uint4 a;
            f32x4 w0, w1;
        };
        auto load_step = [&](Frag& f, int st) {       // k = 16 st + 8 h .. + 7
            const int k0 = 16 * st + 8 * h;
            if (va && k0 + 8 <= kk) {
                f.a = *reinterpret_cast<const uint4*>(ap + 16 * st);
            } else {
                uint32_t e[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) e[j] = (k0 + j < kk) ? (uint32_t)ap[16 * st + j] : 0u;
                f.a = make_uint4(e[0] | (e[1] << 16), e[2] | (e[3] << 16), e[4] | (e[5] << 16), e[6] | (e[7] << 16));
            }
            if (vw && k0 + 8 <= kk) {
                f.w0 = *reinterpret_cast<const f32x4*>(wp + 16 * st);
                f.w1 = *reinterpret_cast<const f32x4*>(wp + 16 * st + 4);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    f.w0[j] = (k0 + j < kk) ? wp[16 * st + j] : 0.f;
                    f.w1[j] = (k0 + 4 + j < kk) ? wp[16 * st + 4 + j] : 0.f;
                }
            }
        };
        auto mul_step = [&](const Frag& f) {
            const bf16x8_t av = __builtin_bit_cast(bf16x8_t, f.a);
            const bf16x8_t bv = __builtin_bit_cast(bf16x8_t, make_uint4(pk_bf16(f.w0[0], f.w0[1]), pk_bf16(f.w0[2], f.w0[3]), pk_bf16(f.w1[0], f.w1[1]),
                                                                        pk_bf16(f.w1[2], f.w1[3])));
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc, 0, 0, 0);
        };
        Frag r0, r1, r2, r3;
        for (int st = st0; st < nst; st += 4) {
            load_step(r0, st);
            if (st + 1 < nst) load_step(r1, st + 1);
            if (st + 2 < nst) load_step(r2, st + 2);
            if (st + 3 < nst) load_step(r3, st + 3);
            mul_step(r0);
            if (st + 1 < nst) mul_step(r1);
            if (st + 2 < nst) mul_step(r2);
            if (st + 3 < nst) mul_step(r3);
        }
    }
    if (SPLITK) {
        if (w > 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) splitk_red[((w - 1) * 16 + r) * 64 + lane] = acc[r];
        }
        __syncthreads();
        if (w > 0) return;
#pragma unroll
        for (int ww = 0; ww < 3; ++ww)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] += splitk_red[(ww * 16 + r) * 64 + lane];
    }
    if (col < n_out) {
        const float bb = bias ? bias[col] : 0.f;
        const float sc = scale ? scale[col] : 1.f;
        const float sh = scale ? shift[col] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t orow = rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (orow >= M) continue;
            float v = acc[r] + bb;
            if (scale) v = __fmaf_rn(v, sc, sh);
            if (relu & 1) v = fmaxf(v, 0.f);
            if (relu & DGNN_LINEAR_ACCUMULATE) v += dgnn_ld(out + orow * ldo + col);
            dgnn_st(out + orow * ldo + col, v);
        }
    }
}

// dW[na, nb] = sum_rows A[r, :]^T B[r, :] with bf16 A and B (TA/TB: uint16_t = bf16 storage, float = fp32 rounded to bf16 here).
// Row slices of 64 rows are staged TRANSPOSED ([column][row], 144-byte rows) so that a lane's MFMA fragment -- 8 consecutive
// rows of one column -- is one 16-byte LDS read.
constexpr int RKB = 64, LDTB = RKB * 2 + 16;

template <typename TA>
__device__ __forceinline__ void stage_t(char* __restrict__ dst, const TA* __restrict__ src, int64_t ld, int64_t r0, int64_t r_end, int c0, int nc) {
    // 256 threads: thread (tr = t>>2 in 0..63 rows, tc = (t&3)*16 columns): 16 elements of one row
    const int t = threadIdx.x, tr = t >> 2, tc = (t & 3) * 16;
    const int64_t gr = r0 + tr;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int cc = c0 + tc + j;
        const float v = (gr < r_end && cc < nc) ? dgnn_ld(src + gr * ld + cc) : 0.f;
        uint16_t b;
        dgnn_st(&b, v);
        *reinterpret_cast<uint16_t*>(dst + (tc + j) * LDTB + tr * 2) = b;
    }
}

template <typename TA, typename TB>
__global__ void __launch_bounds__(256) k_linear_wgrad_b(const TA* __restrict__ A, int64_t lda, int na, const TB* __restrict__ B, int64_t ldb,
                                                        int nb, int64_t M, int64_t rows_per_split, float* __restrict__ partials) {
    __shared__ __attribute__((aligned(16))) char At[WT * LDTB];
    __shared__ __attribute__((aligned(16))) char Bt[WT * LDTB];
    const int lane = lane_id(), w = wave_id_uniform();
    const int wa = w >> 1, wb = w & 1, h = lane >> 5, l31 = lane & 31;
    const int a0 = blockIdx.x * WT, b0 = blockIdx.y * WT;
    const int64_t r_beg = (int64_t)blockIdx.z * rows_per_split;
    const int64_t r_end = min(M, r_beg + rows_per_split);
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    for (int64_t r0 = r_beg; r0 < r_end; r0 += RKB) {
        __syncthreads();
        stage_t<TA>(At, A, lda, r0, r_end, a0, na);
        stage_t<TB>(Bt, B, ldb, r0, r_end, b0, nb);
        __syncthreads();
        const char* ap = At + (wa * 32 + l31) * LDTB + h * 16;
        const char* bp = Bt + (wb * 32 + l31) * LDTB + h * 16;
#pragma unroll
        for (int S = 0; S < RKB / 16; ++S)
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8_t*>(ap + 32 * S),
                                                          *reinterpret_cast<const bf16x8_t*>(bp + 32 * S), acc, 0, 0, 0);
    }
    float* P = partials + (int64_t)blockIdx.z * na * nb;
    const int col = b0 + wb * 32 + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = a0 + wa * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (row < na && col < nb) P[(int64_t)row * nb + col] = acc[r];
    }
}

// ---- merged form for the bf16-storage training step (the Updated variant's conv: dWl = dz^T a, dWr = dz^T x, dbl = sum dz; dWe = dphi^T ea,
// dbe = sum dphi): k_linear_wgrad_b's products on the same row splits (bit-identical dW), both B matrices and the column sums of A from one
// launch.  Staging by 16-byte loads: a thread takes 8 consecutive columns of a ROW PAIR and writes eight 4-byte (row pair) entries of the
// transposed image (k_linear_wgrad_b: sixteen 2-byte loads and stores per thread); the next 64-row slice is loaded under the matrix instructions.
template <typename T>
__device__ __forceinline__ void load_t_b(float (&va)[8], float (&vb)[8], const T* __restrict__ src, int64_t ld, bool vec, int64_t r0, int64_t r_end, int c0,
                                         int nc) {
    const int t = threadIdx.x, tp = t >> 3, tc = (t & 7) * 8;
    const int64_t ra = r0 + 2 * tp, rb = ra + 1;
    if (vec && c0 + tc + 8 <= nc) {
        if constexpr (sizeof(T) == 2) {
            uint4 a = make_uint4(0, 0, 0, 0), b = a;
            if (ra < r_end) a = *reinterpret_cast<const uint4*>(src + ra * ld + c0 + tc);
            if (rb < r_end) b = *reinterpret_cast<const uint4*>(src + rb * ld + c0 + tc);
            const uint32_t wa[4] = {a.x, a.y, a.z, a.w}, wb[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                va[2 * j] = __builtin_bit_cast(float, wa[j] << 16), va[2 * j + 1] = __builtin_bit_cast(float, wa[j] & 0xFFFF0000u);
                vb[2 * j] = __builtin_bit_cast(float, wb[j] << 16), vb[2 * j + 1] = __builtin_bit_cast(float, wb[j] & 0xFFFF0000u);
            }
        } else {
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            const float* pa = reinterpret_cast<const float*>(src) + ra * ld + c0 + tc;
            const float* pb = reinterpret_cast<const float*>(src) + rb * ld + c0 + tc;
            const f32x4 a0 = ra < r_end ? *reinterpret_cast<const f32x4*>(pa) : z, a1 = ra < r_end ? *reinterpret_cast<const f32x4*>(pa + 4) : z;
            const f32x4 b0 = rb < r_end ? *reinterpret_cast<const f32x4*>(pb) : z, b1 = rb < r_end ? *reinterpret_cast<const f32x4*>(pb + 4) : z;
#pragma unroll
            for (int j = 0; j < 4; ++j) va[j] = a0[j], va[4 + j] = a1[j], vb[j] = b0[j], vb[4 + j] = b1[j];
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int cc = c0 + tc + j;
        va[j] = (ra < r_end && cc < nc) ? dgnn_ld(src + ra * ld + cc) : 0.f;
        vb[j] = (rb < r_end && cc < nc) ? dgnn_ld(src + rb * ld + cc) : 0.f;
    }
}
__device__ __forceinline__ void store_t_b(char* __restrict__ dst, const float (&va)[8], const float (&vb)[8]) {
    const int t = threadIdx.x, tp = t >> 3, tc = (t & 7) * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        uint16_t lo, hi;
        dgnn_st(&lo, va[j]);      // round to nearest even like stage_t (exact for bf16 storage)
        dgnn_st(&hi, vb[j]);
        *reinterpret_cast<uint32_t*>(dst + (tc + j) * LDTB + tp * 4) = (uint32_t)lo | ((uint32_t)hi << 16);
    }
}

struct WgradCatB {
    const void* B[2];
    int64_t ldb[2];
    int nb[2];
};

template <typename TA, typename TB>
__global__ void __launch_bounds__(256) k_linear_wgrad_b_cat(const TA* __restrict__ A, int64_t lda, int na, WgradCatB c, int nby1, int64_t M,
                                                            int64_t rows_per_split, float* __restrict__ partials, double* __restrict__ bias_partials) {
    __shared__ __attribute__((aligned(16))) char At[WT * LDTB];
    __shared__ __attribute__((aligned(16))) char Bt[WT * LDTB];
    const int lane = lane_id(), w = wave_id_uniform();
    const int wa = w >> 1, wb = w & 1, h = lane >> 5, l31 = lane & 31;
    const int which = (int)blockIdx.y >= nby1 ? 1 : 0;
    const TB* __restrict__ B = reinterpret_cast<const TB*>(c.B[which]);
    const int64_t ldb = c.ldb[which];
    const int nb = c.nb[which], nbt = c.nb[0] + c.nb[1], colbase = which ? c.nb[0] : 0;
    const int a0 = blockIdx.x * WT, b0 = ((int)blockIdx.y - (which ? nby1 : 0)) * WT;
    const int64_t r_beg = (int64_t)blockIdx.z * rows_per_split;
    const int64_t r_end = min(M, r_beg + rows_per_split);
    const bool do_bias = bias_partials != nullptr && blockIdx.y == 0;
    const int t = threadIdx.x, tp = t >> 3, tc = (t & 7) * 8;
    const bool vecA = ((uintptr_t)A % 16 == 0) && (lda % (16 / sizeof(TA)) == 0), vecB = ((uintptr_t)B % 16 == 0) && (ldb % (16 / sizeof(TB)) == 0);
    double bs[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) bs[j] = 0.0;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    float aa[8], ab[8], ba[8], bb[8];
    if (r_beg < r_end) {
        load_t_b<TA>(aa, ab, A, lda, vecA, r_beg, r_end, a0, na);
        load_t_b<TB>(ba, bb, B, ldb, vecB, r_beg, r_end, b0, nb);
    }
    for (int64_t r0 = r_beg; r0 < r_end; r0 += RKB) {
        __syncthreads();
        store_t_b(At, aa, ab);
        store_t_b(Bt, ba, bb);
        if (do_bias) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                bs[j] += (double)aa[j];
                bs[j] += (double)ab[j];
            }
        }
        __syncthreads();
        if (r0 + RKB < r_end) {
            load_t_b<TA>(aa, ab, A, lda, vecA, r0 + RKB, r_end, a0, na);
            load_t_b<TB>(ba, bb, B, ldb, vecB, r0 + RKB, r_end, b0, nb);
        }
        const char* ap = At + (wa * 32 + l31) * LDTB + h * 16;
        const char* bp = Bt + (wb * 32 + l31) * LDTB + h * 16;
#pragma unroll
        for (int S = 0; S < RKB / 16; ++S)
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8_t*>(ap + 32 * S),
                                                          *reinterpret_cast<const bf16x8_t*>(bp + 32 * S), acc, 0, 0, 0);
    }
    float* P = partials + (int64_t)blockIdx.z * na * nbt;
    const int col = b0 + wb * 32 + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = a0 + wa * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (row < na && col < nb) P[(int64_t)row * nbt + colbase + col] = acc[r];
    }
    if (do_bias) {
        __syncthreads();
        double* red = reinterpret_cast<double*>(At);     // [32 row pairs][64 columns] doubles = 16 KB > the 9 KB tile: two halves of 16 row pairs
        double* red2 = reinterpret_cast<double*>(Bt);
        double* mine = (tp < 16 ? red : red2) + (tp & 15) * 64;
        static_assert(16 * 64 * 8 <= WT * LDTB, "bias scratch fits the tiles");
#pragma unroll
        for (int j = 0; j < 8; ++j) mine[tc + j] = bs[j];
        __syncthreads();
        if (t < 64 && a0 + t < na) {
            double sum = 0.0;
            for (int k = 0; k < 16; ++k) sum += red[k * 64 + t];
            for (int k = 0; k < 16; ++k) sum += red2[k * 64 + t];
            bias_partials[(int64_t)blockIdx.z * na + a0 + t] = sum;
        }
    }
}

// =====================================================================================================================
// fp32-CLASS GEMMs on the bf16 matrix cores ("x3": the arithmetic of the fused layers, fused_common.h split3): both fp32
// operands are split exactly into 3 bf16 parts while they are staged into LDS, the 6 partial products of weight >= 2^-18 run on
// v_mfma_f32_32x32x16_bf16 with fp32 accumulation (dropped terms <= 2^-25 relative, below fp32 rounding) -- 6/16 of the time of
// the bit-faithful fp32 MFMA above.  Used for the wide conv layers the fused kernel does not cover and for the training GEMMs
// when ops.GEMM_MODE is not "f32".
//
// k_linear_fwd_x3: 128 x 128 output tile per 256-thread block, wave (wr, wc) owns a 64 x 64 quadrant = 2 x 2 MFMA blocks
// (4 accumulators); K is walked in chunks of 32 over [A1 | A2].  What-if builds (-DDGNN_WHATIF_NO_SPLIT / _NO_GLOBAL / _NO_MFMA,
// tools/dbg_gemm2.py; M = 1M, K = 256+256, N = 512, 3.95 ms): without the split and LDS stores 3.92 ms, without global loads 2.86,
// without fragment reads + MFMAs 2.75 -- the split VALU is free, loads and MFMAs each cost ~1.1 ms and do NOT overlap, ~1.6 ms is
// neither (epilogue stores, barriers, per-block ramp); a W pre-split variant was 40 % slower (more L2 traffic, same stalls).  LDS row = [hi | mid | lo] x 32 bf16 + 16 B pad = 208 B (an odd
// number of 16-B slots): conflict-free b128 fragment reads.  Per chunk and wave: 24 b128 reads feed 48 MFMAs.
// =====================================================================================================================
constexpr int XM = 128, XN = 128, XK = 32, XLD = 3 * XK * 2 + 16;

__device__ __forceinline__ void x3_split(float x0, float x1, uint32_t& hi, uint32_t& mid, uint32_t& lo) {
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    hi = __builtin_bit_cast(uint32_t, __builtin_convertvector(f2{x0, x1}, bf2));
    const float r0 = x0 - __builtin_bit_cast(float, hi << 16), r1 = x1 - __builtin_bit_cast(float, hi & 0xFFFF0000u);
    mid = __builtin_bit_cast(uint32_t, __builtin_convertvector(f2{r0, r1}, bf2));
    const float s0 = r0 - __builtin_bit_cast(float, mid << 16), s1 = r1 - __builtin_bit_cast(float, mid & 0xFFFF0000u);
    lo = __builtin_bit_cast(uint32_t, __builtin_convertvector(f2{s0, s1}, bf2));
}

// [128 x 32] fp32 tile -> 3 bf16 parts in LDS; 8 threads per row (4 floats each), 32 rows per pass.  Two steps, so that the next
// chunk's global loads are in flight while the current chunk is multiplied (register staging, guide T14): x3_load issues the
// loads into 4 x float4, x3_store splits them and writes the LDS image after the barrier.
template <int NP = 4>
__device__ __forceinline__ void x3_load(f32x4 (&v)[NP], const float* __restrict__ src, int64_t ld, int64_t row0, int64_t nrows, int k0, int kmax,
                                        bool vec, int tid = -1) {
    const int t = tid < 0 ? (int)threadIdx.x : tid, r = t >> 3, c = (t & 7) * 4;
    if (vec && k0 + XK <= kmax) {
        // whole chunk inside K and 16-byte aligned rows (uniform per chunk): four unconditional loads.  Rows past the end are
        // clamped to the last row -- they only feed output rows / columns that are never stored.
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int64_t gr = row0 + r + p * 32;
            v[p] = *reinterpret_cast<const f32x4*>(src + (gr < nrows ? gr : nrows - 1) * ld + k0 + c);
        }
        return;
    }
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int64_t gr = row0 + r + p * 32;
        const float* g = src + (gr < nrows ? gr : nrows - 1) * ld;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = k0 + c + j;
            const float val = g[k < kmax ? k : kmax - 1];
            v[p][j] = k < kmax ? val : 0.f;
        }
    }
}
template <int NP = 4>
__device__ __forceinline__ void x3_store(char* __restrict__ dst, const f32x4 (&v)[NP], int tid = -1) {
    const int t = tid < 0 ? (int)threadIdx.x : tid, r = t >> 3, c = (t & 7) * 4;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        uint32_t h0, m0, l0, h1, m1, l1;
        x3_split(v[p][0], v[p][1], h0, m0, l0);
        x3_split(v[p][2], v[p][3], h1, m1, l1);
        char* d = dst + (r + p * 32) * XLD + c * 2;
        *reinterpret_cast<uint2*>(d) = make_uint2(h0, h1);
        *reinterpret_cast<uint2*>(d + 64) = make_uint2(m0, m1);
        *reinterpret_cast<uint2*>(d + 128) = make_uint2(l0, l1);
    }
}

// BatchNorm batch statistics from the GEMM that produces z (training forward): every wavefront adds the 32 x 32 accumulator block it is
// about to store (bias included) into fp64 column sums / sums of squares and writes them as partial row `row / 32` of
// colstats[ceil(M / 32)][2][n_out] -- the layout k_stats_finalize reads; the launch that used to read z back for them is gone.
__device__ __forceinline__ double shfl_xor32_f64(double v) {
    const uint64_t b = __builtin_bit_cast(uint64_t, v);
    const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)b, 32), hi = (uint32_t)__shfl_xor((int)(uint32_t)(b >> 32), 32);
    return __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ void stats_block_store(double* __restrict__ colstats, int64_t row_first, int64_t M, int n_out, int col, int h, double s, double q) {
    s += shfl_xor32_f64(s);     // rows 4h .. of the block sit in the two half-waves
    q += shfl_xor32_f64(q);
    if (h == 0 && col < n_out && row_first < M) {
        const int64_t rbk = row_first >> 5;
        colstats[(rbk * 2 + 0) * n_out + col] = s;
        colstats[(rbk * 2 + 1) * n_out + col] = q;
    }
}

__global__ void __launch_bounds__(256, 2) k_linear_fwd_x3(const float* __restrict__ A1, int64_t lda1, int k1, const float* __restrict__ W1,
                                                          int64_t ldw1, bool vec1, const float* __restrict__ A2, int64_t lda2, int k2,
                                                          const float* __restrict__ W2, int64_t ldw2, bool vec2,
                                                          const float* __restrict__ bias, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, int relu, int64_t M, int n_out,
                                                          float* __restrict__ out, int64_t ldo, double* __restrict__ colstats) {
    __shared__ __attribute__((aligned(16))) char As[XM * XLD];
    __shared__ __attribute__((aligned(16))) char Ws[XN * XLD];
    const int lane = lane_id(), w = wave_id_uniform();
    const int wr = w >> 1, wc = w & 1, h = lane >> 5, l31 = lane & 31;
    // 1-D grid with an XCD-aware tile map: workgroups are dealt to the 8 XCDs round-robin (blockIdx % 8) and every XCD has its own L2,
    // so the column blocks of one row panel sit on ONE XCD, next to each other in its queue, and the panel of A comes through the fabric
    // once (FETCH_SIZE at M = 1M, K = 512, N = 512: 2.5 GB; with plain "column blocks fastest" the four column blocks landed on four
    // XCDs: 8.4 GB; with row blocks fastest every column block streamed all of A again).  The grid is padded to 8 x ceil(row blocks / 8).
    const int ncb = (n_out + XN - 1) / XN;
    const int64_t rb = (int64_t)((blockIdx.x >> 3) / ncb) * 8 + (blockIdx.x & 7);
    if (rb * XM >= M) return;
    const int64_t row0 = rb * XM;
    const int col0 = (int)((blockIdx.x >> 3) % ncb) * XN;
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
    const int nch1 = (k1 + XK - 1) / XK, nch2 = A2 ? (k2 + XK - 1) / XK : 0, nch = nch1 + nch2;
    f32x4 ra[4], rw[4];
    auto load_chunk = [&](int ch) {
        const bool first = ch < nch1;
        const int kk = first ? k1 : k2, k0 = (first ? ch : ch - nch1) * XK;
        x3_load(ra, first ? A1 : A2, first ? lda1 : lda2, row0, M, k0, kk, first ? vec1 : vec2);
        x3_load(rw, first ? W1 : W2, first ? ldw1 : ldw2, col0, n_out, k0, kk, first ? vec1 : vec2);
    };
    load_chunk(0);
    for (int ch = 0; ch < nch; ++ch) {
        __syncthreads();                 // the previous chunk's fragments have been read
#ifndef DGNN_WHATIF_NO_SPLIT
        x3_store(As, ra);
        x3_store(Ws, rw);
#else
        if (ch == 0) { x3_store(As, ra); x3_store(Ws, rw); }
        else asm volatile("" :: "v"(ra[0][0]), "v"(ra[1][0]), "v"(ra[2][0]), "v"(ra[3][0]), "v"(rw[0][0]), "v"(rw[1][0]), "v"(rw[2][0]), "v"(rw[3][0]));
#endif
        __syncthreads();
#ifndef DGNN_WHATIF_NO_GLOBAL
        if (ch + 1 < nch) load_chunk(ch + 1);   // in flight under the 48 MFMAs below
#endif
#ifdef DGNN_WHATIF_NO_MFMA
        if (ch + 1 < nch) continue;
#endif
        const char* ap = As + (wr * 64 + l31) * XLD + h * 16;
        const char* bp = Ws + (wc * 64 + l31) * XLD + h * 16;
#pragma unroll
        for (int S = 0; S < XK / 16; ++S) {
            bf16x8_t af[2][3], bf[2][3];
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    af[m][p] = *reinterpret_cast<const bf16x8_t*>(ap + m * 32 * XLD + p * 64 + S * 32);
                    bf[m][p] = *reinterpret_cast<const bf16x8_t*>(bp + m * 32 * XLD + p * 64 + S * 32);
                }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    f32x16 c = acc[a][b];
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][2], bf[b][0], c, 0, 0, 0);   // small terms first
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][0], bf[b][2], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][1], bf[b][1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][1], bf[b][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][0], bf[b][1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][0], bf[b][0], c, 0, 0, 0);
                    acc[a][b] = c;
                }
        }
    }
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int col = col0 + wc * 64 + b * 32 + l31;
        const int colc = col < n_out ? col : n_out - 1;
        const float bb = bias ? bias[colc] : 0.f;
        const float sc = scale ? scale[colc] : 1.f;
        const float sh = scale ? shift[colc] : 0.f;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            double s = 0.0, q = 0.0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t row = row0 + wr * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (row >= M || col >= n_out) continue;
                float v = acc[a][b][r] + bb;
                if (colstats) {
                    s += v;
                    q += (double)v * v;
                }
                if (scale) v = __fmaf_rn(v, sc, sh);
                if (relu & 1) v = fmaxf(v, 0.f);
                if (relu & DGNN_LINEAR_ACCUMULATE) v += out[row * ldo + col];
                out[row * ldo + col] = v;
            }
            if (colstats) stats_block_store(colstats, row0 + wr * 64 + a * 32, M, n_out, col, h, s, q);
        }
    }
}

// 64 x 64 output tile (round 6): k_linear_fwd_x3's loop -- operands split while staged, the next chunk's loads in flight under the products -- with
// the 2 x 2 wave grid on 32 x 32 blocks (one accumulator each).  For the problems between the two existing forms: the reference's training widths at
// batch 1024 (M = 1 024 .. 6 000 rows, K and N 512 .. 1 024; configs/modelnet.yaml:44,56) give the 128 x 128 tiles a quarter of the chip (M = 1 024,
// K = N = 1 024: 66 us) and make the LDS-free small-problem kernel re-read both operands once per 32 x 32 block (39 us, L2-bound: 268 MB for 8 MB of
// operands); four times as many tiles as the former, half the operand traffic of the latter.  Same chunk and product order per output element as
// k_linear_fwd_x3: bit-identical results.
// KS = 4 (K >= 1 024 and fewer tiles than 2 per CU): the workgroup is 16 wavefronts, four groups of four, every group walks a quarter of the chunks
// of the SAME tile through its own LDS buffers; the partial blocks are added in group order by group 0, which runs the epilogue.  With one 4-wavefront
// tile per CU (M = 1 024, K = N = 1 024) every barrier, LDS round trip and load of the chunk loop was exposed -- one wavefront per SIMD: 34 us, and two
// chunks of loads in flight instead of one changed nothing; four per SIMD hide them behind each other.  (Another summation order than the KS = 1
// form: fp32 rounding level.)
template <int KS>
__global__ void __launch_bounds__(256 * KS, KS == 1 ? 2 : 1) k_linear_fwd_x3_mid(const float* __restrict__ A1, int64_t lda1, int k1, const float* __restrict__ W1,
                                                                             int64_t ldw1, bool vec1, const float* __restrict__ A2, int64_t lda2, int k2,
                                                                             const float* __restrict__ W2, int64_t ldw2, bool vec2,
                                                                             const float* __restrict__ bias, const float* __restrict__ scale,
                                                                             const float* __restrict__ shift, int relu, int64_t M, int n_out,
                                                                             float* __restrict__ out, int64_t ldo, double* __restrict__ colstats) {
    constexpr int TM = 64, TN = 64;
    extern __shared__ __attribute__((aligned(16))) char mid_smem[];      // per group: As [TM][XLD] | Ws [TN][XLD]
    const int lane = lane_id(), w = wave_id_uniform();
    const int g = w >> 2, wq = w & 3, tid = (int)threadIdx.x & 255;
    const int wr = wq >> 1, wc = wq & 1, h = lane >> 5, l31 = lane & 31;
    char* const As = mid_smem + g * ((TM + TN) * XLD);
    char* const Ws = As + TM * XLD;
    const int ncb = (n_out + TN - 1) / TN;
    const int64_t rb = (int64_t)((blockIdx.x >> 3) / ncb) * 8 + (blockIdx.x & 7);      // XCD-aware tile map as in k_linear_fwd_x3
    if (rb * TM >= M) return;
    const int64_t row0 = rb * TM;
    const int col0 = (int)((blockIdx.x >> 3) % ncb) * TN;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const int nch1 = (k1 + XK - 1) / XK, nch2 = A2 ? (k2 + XK - 1) / XK : 0, nch = nch1 + nch2;
    const int cpg = (nch + KS - 1) / KS;                       // chunks per group
    const int ch_lo = g * cpg, ch_hi = ch_lo + cpg < nch ? ch_lo + cpg : nch;
    f32x4 ra[2], rw[2];
    auto load_chunk = [&](int ch) {
        const bool first = ch < nch1;
        const int kk = first ? k1 : k2, k0 = (first ? ch : ch - nch1) * XK;
        x3_load<2>(ra, first ? A1 : A2, first ? lda1 : lda2, row0, M, k0, kk, first ? vec1 : vec2, tid);
        x3_load<2>(rw, first ? W1 : W2, first ? ldw1 : ldw2, col0, n_out, k0, kk, first ? vec1 : vec2, tid);
    };
    if (ch_lo < ch_hi) load_chunk(ch_lo);
    for (int i = 0; i < cpg; ++i) {
        const int ch = ch_lo + i;
        const bool live = ch < ch_hi;
        __syncthreads();
        if (live) {
            x3_store<2>(As, ra, tid);
            x3_store<2>(Ws, rw, tid);
        }
        __syncthreads();
        if (ch + 1 < ch_hi) load_chunk(ch + 1);
        if (!live) continue;
        const char* ap = As + (wr * 32 + l31) * XLD + h * 16;
        const char* bp = Ws + (wc * 32 + l31) * XLD + h * 16;
#pragma unroll
        for (int S = 0; S < XK / 16; ++S) {
            bf16x8_t af[3], bf[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                af[p] = *reinterpret_cast<const bf16x8_t*>(ap + p * 64 + S * 32);
                bf[p] = *reinterpret_cast<const bf16x8_t*>(bp + p * 64 + S * 32);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[2], bf[0], acc, 0, 0, 0);   // small terms first
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], bf[1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], bf[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[0], acc, 0, 0, 0);
        }
    }
    if (KS > 1) {      // the groups' partial blocks -> group 0, in group order (16 KB per group: [wavefront][register][lane], over the chunk buffers)
        __syncthreads();
        float* red = reinterpret_cast<float*>(mid_smem);
        if (g > 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) red[(((g - 1) * 4 + wq) * 16 + r) * 64 + lane] = acc[r];
        }
        __syncthreads();
        if (g > 0) return;
#pragma unroll
        for (int gg = 0; gg < KS - 1; ++gg)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] += red[((gg * 4 + wq) * 16 + r) * 64 + lane];
    }
    {
        const int col = col0 + wc * 32 + l31;
        const int colc = col < n_out ? col : n_out - 1;
        const float bb = bias ? bias[colc] : 0.f;
        const float sc = scale ? scale[colc] : 1.f;
        const float sh = scale ? shift[colc] : 0.f;
        double s_ = 0.0, q_ = 0.0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t row = row0 + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (row >= M || col >= n_out) continue;
            float v = acc[r] + bb;
            if (colstats) {
                s_ += v;
                q_ += (double)v * v;
            }
            if (scale) v = __fmaf_rn(v, sc, sh);
            if (relu & 1) v = fmaxf(v, 0.f);
            if (relu & DGNN_LINEAR_ACCUMULATE) v += out[row * ldo + col];
            out[row * ldo + col] = v;
        }
        if (colstats) stats_block_store(colstats, row0 + wr * 32, M, n_out, col, h, s_, q_);
    }
}

// n_out <= 64 (first conv layer, input gradients of the narrow layers, the decoder's hidden layer): 128 x 64 tile, the 2 x 2 wave grid
// owns 64 x 32 blocks (2 accumulators each) -- the 128-wide tile computes 64..100 % padding columns there.  Same chunk and product
// order per output element as k_linear_fwd_x3.
constexpr int ZN = 64;
__global__ void __launch_bounds__(256, 2) k_linear_fwd_x3_n64(const float* __restrict__ A1, int64_t lda1, int k1, const float* __restrict__ W1,
                                                              int64_t ldw1, bool vec1, const float* __restrict__ A2, int64_t lda2, int k2,
                                                              const float* __restrict__ W2, int64_t ldw2, bool vec2,
                                                              const float* __restrict__ bias, const float* __restrict__ scale,
                                                              const float* __restrict__ shift, int relu, int64_t M, int n_out,
                                                              float* __restrict__ out, int64_t ldo, double* __restrict__ colstats) {
    __shared__ __attribute__((aligned(16))) char As[XM * XLD];
    __shared__ __attribute__((aligned(16))) char Ws[ZN * XLD];
    const int lane = lane_id(), w = wave_id_uniform();
    const int wr = w >> 1, wc = w & 1, h = lane >> 5, l31 = lane & 31;
    const int64_t row0 = (int64_t)blockIdx.x * XM;
    f32x16 acc[2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[a][i] = 0.f;
    const int nch1 = (k1 + XK - 1) / XK, nch2 = A2 ? (k2 + XK - 1) / XK : 0, nch = nch1 + nch2;
    f32x4 ra[4], rw[2];
    // W tile: 64 rows x 32 floats: 8 threads per row, 32 rows per pass, 2 passes
    auto load_w = [&](const float* __restrict__ src, int64_t ld, int k0, int kmax, bool vec) {
        const int t = threadIdx.x, r = t >> 3, c = (t & 7) * 4;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int gr = r + p * 32;
            const float* g = src + (int64_t)(gr < n_out ? gr : n_out - 1) * ld;
            if (vec && k0 + XK <= kmax) {
                rw[p] = *reinterpret_cast<const f32x4*>(g + k0 + c);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int k = k0 + c + j;
                    const float val = g[k < kmax ? k : kmax - 1];
                    rw[p][j] = k < kmax ? val : 0.f;
                }
            }
        }
    };
    auto store_w = [&]() {
        const int t = threadIdx.x, r = t >> 3, c = (t & 7) * 4;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            uint32_t h0, m0, l0, h1, m1, l1;
            x3_split(rw[p][0], rw[p][1], h0, m0, l0);
            x3_split(rw[p][2], rw[p][3], h1, m1, l1);
            char* d = Ws + (r + p * 32) * XLD + c * 2;
            *reinterpret_cast<uint2*>(d) = make_uint2(h0, h1);
            *reinterpret_cast<uint2*>(d + 64) = make_uint2(m0, m1);
            *reinterpret_cast<uint2*>(d + 128) = make_uint2(l0, l1);
        }
    };
    auto load_chunk = [&](int ch) {
        const bool first = ch < nch1;
        const int kk = first ? k1 : k2, k0 = (first ? ch : ch - nch1) * XK;
        x3_load(ra, first ? A1 : A2, first ? lda1 : lda2, row0, M, k0, kk, first ? vec1 : vec2);
        load_w(first ? W1 : W2, first ? ldw1 : ldw2, k0, kk, first ? vec1 : vec2);
    };
    load_chunk(0);
    for (int ch = 0; ch < nch; ++ch) {
        __syncthreads();
        x3_store(As, ra);
        store_w();
        __syncthreads();
        if (ch + 1 < nch) load_chunk(ch + 1);
        const char* ap = As + (wr * 64 + l31) * XLD + h * 16;
        const char* bp = Ws + (wc * 32 + l31) * XLD + h * 16;
#pragma unroll
        for (int S = 0; S < XK / 16; ++S) {
            bf16x8_t af[2][3], bf[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                bf[p] = *reinterpret_cast<const bf16x8_t*>(bp + p * 64 + S * 32);
#pragma unroll
                for (int m = 0; m < 2; ++m) af[m][p] = *reinterpret_cast<const bf16x8_t*>(ap + m * 32 * XLD + p * 64 + S * 32);
            }
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                f32x16 c = acc[a];
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][2], bf[0], c, 0, 0, 0);   // small terms first
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][0], bf[2], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][1], bf[1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][1], bf[0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][0], bf[1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][0], bf[0], c, 0, 0, 0);
                acc[a] = c;
            }
        }
    }
    const int col = wc * 32 + l31;
    {
        const int colc = col < n_out ? col : n_out - 1;
        const float bb = bias ? bias[colc] : 0.f;
        const float sc = scale ? scale[colc] : 1.f;
        const float sh = scale ? shift[colc] : 0.f;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            double s = 0.0, q = 0.0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t row = row0 + wr * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (row >= M || col >= n_out) continue;
                float v = acc[a][r] + bb;
                if (colstats) {
                    s += v;
                    q += (double)v * v;
                }
                if (scale) v = __fmaf_rn(v, sc, sh);
                if (relu & 1) v = fmaxf(v, 0.f);
                if (relu & DGNN_LINEAR_ACCUMULATE) v += out[row * ldo + col];
                out[row * ldo + col] = v;
            }
            if (colstats) stats_block_store(colstats, row0 + wr * 64 + a * 32, M, n_out, col, h, s, q);
        }
    }
}

// Small problems (M <= 16384 rows: the innermost training blocks, the decoder): a 128-row tile leaves most of the chip idle (M = 2048,
// n_out = 128: 16 workgroups) and walks K chunk by chunk behind two barriers each -- 20-33 us for 0.1 GFLOP.  Here ONE WAVEFRONT owns a
// 32 x 32 output block and takes both operands straight from global memory in the MFMA's own layout (a lane's fragment is 8 consecutive
// k of one row: two 16-byte loads), splitting them in registers: no LDS, no barrier, four k-steps of loads in flight, 4 x as many
// independent workgroup slots.  k-steps and products in the order of k_linear_fwd_x3: bit-identical results.
// SPLITK: as k_linear_fwd_b_small<., true> -- the workgroup's four wavefronts share one output block, a quarter of the k-steps each (K >= 1024 only)
template <bool SPLITK>
__global__ void __launch_bounds__(256) k_linear_fwd_x3_small(const float* __restrict__ A1, int64_t lda1, int k1, const float* __restrict__ W1,
                                                             int64_t ldw1, bool vec1, const float* __restrict__ A2, int64_t lda2, int k2,
                                                             const float* __restrict__ W2, int64_t ldw2, bool vec2,
                                                             const float* __restrict__ bias, const float* __restrict__ scale,
                                                             const float* __restrict__ shift, int relu, int64_t M, int n_out,
                                                             float* __restrict__ out, int64_t ldo, double* __restrict__ colstats) {
    __shared__ float splitk_red[SPLITK ? 3 * 16 * 64 : 1];
    const int lane = lane_id(), w = wave_id_uniform();
    const int h = lane >> 5, l31 = lane & 31;
    const int nct = (n_out + 31) / 32;                      // column tiles; a workgroup = 4 consecutive (row tile, column tile) pairs
    const int64_t tile = SPLITK ? (int64_t)blockIdx.x : (int64_t)blockIdx.x * 4 + w;
    const int64_t rt = tile / nct;
    const int ct = (int)(tile - rt * nct);
    if (rt * 32 >= M) return;
    const int64_t row = rt * 32 + l31, rowc = row < M ? row : M - 1;
    const int col = ct * 32 + l31, colc = col < n_out ? col : n_out - 1;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    // one operand pair after the other ([A1 | A2] against [W1 | W2]), k-steps of 16
    for (int part = 0; part < 2; ++part) {
        const float* A = part == 0 ? A1 : A2;
        if (!A) break;
        const float* W = part == 0 ? W1 : W2;
        const int kk = part == 0 ? k1 : k2;
        const bool vec = part == 0 ? vec1 : vec2;
        const float* ap = A + rowc * (part == 0 ? lda1 : lda2) + 8 * h;
        const float* wp = W + (int64_t)colc * (part == 0 ? ldw1 : ldw2) + 8 * h;
        const int nst_all = (kk + 15) / 16;
        const int q_ = (nst_all + 3) / 4;
        const int st0 = SPLITK ? w * q_ : 0, nst = SPLITK ? (st0 + q_ < nst_all ? st0 + q_ : nst_all) : nst_all;
        auto load_step = [&](f32x4 (&v)[4], int st) {      // v[0..1]: A k .. k+7, v[2..3]: W k .. k+7 (k = 16 st + 8 h)
            const int k0 = 16 * st;
            if (vec && k0 + 16 <= kk) {
                v[0] = *reinterpret_cast<const f32x4*>(ap + k0);
                v[1] = *reinterpret_cast<const f32x4*>(ap + k0 + 4);
                v[2] = *reinterpret_cast<const f32x4*>(wp + k0);
                v[3] = *reinterpret_cast<const f32x4*>(wp + k0 + 4);
            } else {
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int k = k0 + 8 * h + 4 * q + j;
                        const int kc = k < kk ? k : kk - 1;
                        const float av = A[rowc * (part == 0 ? lda1 : lda2) + kc], wv = W[(int64_t)colc * (part == 0 ? ldw1 : ldw2) + kc];
                        v[q][j] = k < kk ? av : 0.f;
                        v[2 + q][j] = k < kk ? wv : 0.f;
                    }
            }
        };
        auto mul_step = [&](const f32x4 (&v)[4]) {
            uint32_t ph[4], pm[4], pl[4], qh[4], qm[4], ql[4];
            x3_split(v[0][0], v[0][1], ph[0], pm[0], pl[0]);
            x3_split(v[0][2], v[0][3], ph[1], pm[1], pl[1]);
            x3_split(v[1][0], v[1][1], ph[2], pm[2], pl[2]);
            x3_split(v[1][2], v[1][3], ph[3], pm[3], pl[3]);
            x3_split(v[2][0], v[2][1], qh[0], qm[0], ql[0]);
            x3_split(v[2][2], v[2][3], qh[1], qm[1], ql[1]);
            x3_split(v[3][0], v[3][1], qh[2], qm[2], ql[2]);
            x3_split(v[3][2], v[3][3], qh[3], qm[3], ql[3]);
            const bf16x8_t a0 = __builtin_bit_cast(bf16x8_t, make_uint4(ph[0], ph[1], ph[2], ph[3])),
                           a1 = __builtin_bit_cast(bf16x8_t, make_uint4(pm[0], pm[1], pm[2], pm[3])),
                           a2 = __builtin_bit_cast(bf16x8_t, make_uint4(pl[0], pl[1], pl[2], pl[3])),
                           b0 = __builtin_bit_cast(bf16x8_t, make_uint4(qh[0], qh[1], qh[2], qh[3])),
                           b1 = __builtin_bit_cast(bf16x8_t, make_uint4(qm[0], qm[1], qm[2], qm[3])),
                           b2 = __builtin_bit_cast(bf16x8_t, make_uint4(ql[0], ql[1], ql[2], ql[3]));
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b0, acc, 0, 0, 0);   // small terms first
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b2, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc, 0, 0, 0);
        };
        // four k-steps of loads in flight
        f32x4 r0[4], r1[4], r2[4], r3[4];
        for (int st = st0; st < nst; st += 4) {
            load_step(r0, st);
            if (st + 1 < nst) load_step(r1, st + 1);
            if (st + 2 < nst) load_step(r2, st + 2);
            if (st + 3 < nst) load_step(r3, st + 3);
            mul_step(r0);
            if (st + 1 < nst) mul_step(r1);
            if (st + 2 < nst) mul_step(r2);
            if (st + 3 < nst) mul_step(r3);
        }
    }
    if (SPLITK) {
        if (w > 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) splitk_red[((w - 1) * 16 + r) * 64 + lane] = acc[r];
        }
        __syncthreads();
        if (w > 0) return;
#pragma unroll
        for (int ww = 0; ww < 3; ++ww)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] += splitk_red[(ww * 16 + r) * 64 + lane];
    }
    {
        const float bb = bias ? bias[colc] : 0.f;
        const float sc = scale ? scale[colc] : 1.f;
        const float sh = scale ? shift[colc] : 0.f;
        double s = 0.0, q = 0.0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t orow = rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (orow >= M || col >= n_out) continue;
            float v = acc[r] + bb;
            if (colstats) {
                s += v;
                q += (double)v * v;
            }
            if (scale) v = __fmaf_rn(v, sc, sh);
            if (relu & 1) v = fmaxf(v, 0.f);
            if (relu & DGNN_LINEAR_ACCUMULATE) v += out[orow * ldo + col];
            out[orow * ldo + col] = v;
        }
        if (colstats) stats_block_store(colstats, rt * 32, M, n_out, col, h, s, q);
    }
}

// Large problems (M >= 8192 rows, n_out > 128): 256 x 256 output tile per 512-thread block, wave (wr, wc) of the 2 x 4 wave grid owns
// a 128 x 64 block = 4 x 2 MFMA blocks (8 accumulators).  Splitting the operands costs VALU issue slots (about 12 per element
// pair) that share the SIMD's issue port with the MFMAs: at 128 x 128 a chunk is 32 elements per thread for 48 MFMAs per wave,
// about as many VALU as can hide behind them; the 256 x 256 tile stages the same 32 elements per thread for 96 MFMAs, and
// fetches A and W half as often.  LDS: 512 rows x 208 B = 104 KB, one block per CU, 2 waves per SIMD.
constexpr int YM = 256, YN = 256, YT = 512;

template <int NP>
__device__ __forceinline__ void y3_load(f32x4 (&v)[NP], const float* __restrict__ src, int64_t ld, int64_t row0, int64_t nrows, int k0, int kmax,
                                        bool vec) {
    constexpr int RPP = YT / 8;   // rows per pass
    const int t = threadIdx.x, r = t >> 3, c = (t & 7) * 4;
    if (vec && k0 + XK <= kmax) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int64_t gr = row0 + r + p * RPP;
            v[p] = *reinterpret_cast<const f32x4*>(src + (gr < nrows ? gr : nrows - 1) * ld + k0 + c);
        }
        return;
    }
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int64_t gr = row0 + r + p * RPP;
        const float* g = src + (gr < nrows ? gr : nrows - 1) * ld;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = k0 + c + j;
            const float val = g[k < kmax ? k : kmax - 1];
            v[p][j] = k < kmax ? val : 0.f;
        }
    }
}
template <int NP>
__device__ __forceinline__ void y3_store(char* __restrict__ dst, const f32x4 (&v)[NP]) {
    constexpr int RPP = YT / 8;
    const int t = threadIdx.x, r = t >> 3, c = (t & 7) * 4;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        uint32_t h0, m0, l0, h1, m1, l1;
        x3_split(v[p][0], v[p][1], h0, m0, l0);
        x3_split(v[p][2], v[p][3], h1, m1, l1);
        char* d = dst + (r + p * RPP) * XLD + c * 2;
        *reinterpret_cast<uint2*>(d) = make_uint2(h0, h1);
        *reinterpret_cast<uint2*>(d + 64) = make_uint2(m0, m1);
        *reinterpret_cast<uint2*>(d + 128) = make_uint2(l0, l1);
    }
}

__global__ void __launch_bounds__(YT, 1) k_linear_fwd_x3_big(const float* __restrict__ A1, int64_t lda1, int k1, const float* __restrict__ W1,
                                                             int64_t ldw1, bool vec1, const float* __restrict__ A2, int64_t lda2, int k2,
                                                             const float* __restrict__ W2, int64_t ldw2, bool vec2,
                                                             const float* __restrict__ bias, const float* __restrict__ scale,
                                                             const float* __restrict__ shift, int relu, int64_t M, int n_out,
                                                             float* __restrict__ out, int64_t ldo) {
    extern __shared__ __attribute__((aligned(16))) char y3_smem[];
    char* const As = y3_smem;
    char* const Ws = y3_smem + YM * XLD;
    const int lane = lane_id(), w = wave_id_uniform();
    const int wr = w >> 2, wc = w & 3, h = lane >> 5, l31 = lane & 31;
    const int ncb = (n_out + YN - 1) / YN;   // XCD-aware tile map (see k_linear_fwd_x3)
    const int64_t rb = (int64_t)((blockIdx.x >> 3) / ncb) * 8 + (blockIdx.x & 7);
    if (rb * YM >= M) return;
    const int64_t row0 = rb * YM;
    const int col0 = (int)((blockIdx.x >> 3) % ncb) * YN;
    f32x16 acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
    const int nch1 = (k1 + XK - 1) / XK, nch2 = A2 ? (k2 + XK - 1) / XK : 0, nch = nch1 + nch2;
    constexpr int NP = YM / (YT / 8);
    f32x4 ra[NP], rw[NP];
    auto load_chunk = [&](int ch) {
        const bool first = ch < nch1;
        const int kk = first ? k1 : k2, k0 = (first ? ch : ch - nch1) * XK;
        y3_load<NP>(ra, first ? A1 : A2, first ? lda1 : lda2, row0, M, k0, kk, first ? vec1 : vec2);
        y3_load<NP>(rw, first ? W1 : W2, first ? ldw1 : ldw2, col0, n_out, k0, kk, first ? vec1 : vec2);
    };
    load_chunk(0);
    for (int ch = 0; ch < nch; ++ch) {
        __syncthreads();
        y3_store<NP>(As, ra);
        y3_store<NP>(Ws, rw);
        __syncthreads();
        if (ch + 1 < nch) load_chunk(ch + 1);   // in flight under the 96 MFMAs below
        const char* ap = As + (wr * 128 + l31) * XLD + h * 16;
        const char* bp = Ws + (wc * 64 + l31) * XLD + h * 16;
#pragma unroll
        for (int S = 0; S < XK / 16; ++S) {
            bf16x8_t bf[2][3];
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int p = 0; p < 3; ++p) bf[m][p] = *reinterpret_cast<const bf16x8_t*>(bp + m * 32 * XLD + p * 64 + S * 32);
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                bf16x8_t af[3];
#pragma unroll
                for (int p = 0; p < 3; ++p) af[p] = *reinterpret_cast<const bf16x8_t*>(ap + a * 32 * XLD + p * 64 + S * 32);
                // the two column blocks alternate product by product: consecutive MFMAs never wait for each other's result, and every
                // accumulator still sees its six products in the same order (small terms first)
                constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
                for (int q = 0; q < 6; ++q)
#pragma unroll
                    for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[PA[q]], bf[b][PB[q]], acc[a][b], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int col = col0 + wc * 64 + b * 32 + l31;
        if (col >= n_out) continue;
        const float bb = bias ? bias[col] : 0.f;
        const float sc = scale ? scale[col] : 1.f;
        const float sh = scale ? shift[col] : 0.f;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t row = row0 + wr * 128 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (row >= M) continue;
                float v = acc[a][b][r] + bb;
                if (scale) v = __fmaf_rn(v, sc, sh);
                if (relu & 1) v = fmaxf(v, 0.f);
                if (relu & DGNN_LINEAR_ACCUMULATE) v += out[row * ldo + col];
                out[row * ldo + col] = v;
            }
    }
}

// =====================================================================================================================
// "x2h": the same GEMM on the fp16 matrix cores in the fused layers' fp16 form (fused_common.h): every ROW of [A1 | A2] and every
// row of [W1 | W2] (= output column) is multiplied by its own power of two so that its largest magnitude lies in [2^14, 2^15), split
// into (hi, lo) fp16 parts while staged into LDS, 3 products per fp32 product (lo.lo dropped, <= 2^-22 relative), fp32 accumulation,
// the two inverse powers of two applied to the accumulator in the epilogue (exact).  Half the matrix instructions and a third of the
// split instructions of x3; a first pass over the operands writes the row scales (one extra read of A).
// Large problems only (the 256 x 256 tile of k_linear_fwd_x3_big; M >= 8192, n_out > 128) -- the wide conv layers.
// =====================================================================================================================
#ifndef DGNN_X2H_K
#define DGNN_X2H_K 32
#endif
constexpr int HK = DGNN_X2H_K;          // K chunk per barrier pair.  Measured (tools/bench_gemm_x2h.py, M = 1M): 64 halves the barriers but needs 64 staging
                                        // registers next to the 128 accumulators -- 52 spilled, 4.26 ms against 2.67 ms with 32 (K = 512, N = 512)
constexpr int HLD = 2 * HK * 2 + 16;   // LDS row: [hi | lo] x HK fp16 + 16 B pad (an odd number of 16-byte slots)
constexpr int HTPR = HK / 4;           // threads per staged row (4 floats each)

template <int NP>
__device__ __forceinline__ void y2h_load(f32x4 (&v)[NP], const float* __restrict__ src, int64_t ld, int64_t row0, int64_t nrows, int k0, int kmax,
                                         bool vec) {
    constexpr int RPP = YT / HTPR;   // rows per pass
    const int t = threadIdx.x, r = t / HTPR, c = (t % HTPR) * 4;
    if (vec && k0 + HK <= kmax) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int64_t gr = row0 + r + p * RPP;
            v[p] = *reinterpret_cast<const f32x4*>(src + (gr < nrows ? gr : nrows - 1) * ld + k0 + c);
        }
        return;
    }
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int64_t gr = row0 + r + p * RPP;
        const float* g = src + (gr < nrows ? gr : nrows - 1) * ld;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = k0 + c + j;
            const float val = g[k < kmax ? k : kmax - 1];
            v[p][j] = k < kmax ? val : 0.f;
        }
    }
}

typedef _Float16 h16x8_t __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void x2h_split(float x0, float x1, uint32_t& hi, uint32_t& lo) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    hi = __builtin_bit_cast(uint32_t, __builtin_convertvector(f2{x0, x1}, h2));
    float r0, r1;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hi), "v"(x0));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hi), "v"(x1));
    lo = __builtin_bit_cast(uint32_t, __builtin_convertvector(f2{r0, r1}, h2));
}
// s = 2^(141 - E) for the row maximum's biased exponent E (clamped to [14, 254]); 1/s from s without a division
__device__ __forceinline__ float x2h_scale_of(float rowmax) {
    uint32_t E = (__builtin_bit_cast(uint32_t, rowmax) & 0x7FFFFFFFu) >> 23;
    E = E < 14u ? 14u : (E > 254u ? 254u : E);
    return __builtin_bit_cast(float, (268u - E) << 23);
}
__device__ __forceinline__ float x2h_inv(float s) { return __builtin_bit_cast(float, (254u << 23) - __builtin_bit_cast(uint32_t, s)); }

// one wavefront per row of [A1 | A2] (rows 0 .. M) and of [W1 | W2] (rows M .. M + n_out): scales[row] = its power-of-two scale
__global__ void __launch_bounds__(256) k_x2h_row_scales(const float* __restrict__ A1, int64_t lda1, int k1, const float* __restrict__ A2, int64_t lda2,
                                                        int k2, int64_t M, const float* __restrict__ W1, int64_t ldw1, const float* __restrict__ W2,
                                                        int64_t ldw2, int n_out, float* __restrict__ scales) {
    const int lane = lane_id();
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    for (int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < M + n_out; row += nwaves) {
        const bool isw = row >= M;
        const float* p1 = isw ? W1 + (row - M) * ldw1 : A1 + row * lda1;
        const float* p2 = isw ? (W2 ? W2 + (row - M) * ldw2 : nullptr) : (A2 ? A2 + row * lda2 : nullptr);
        float m = 0.f;
        if ((((uintptr_t)p1) & 15) == 0 && (k1 & 3) == 0) {
            for (int k = lane * 4; k < k1; k += 256) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(p1 + k);
                m = fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
            }
        } else {
            for (int k = lane; k < k1; k += 64) m = fmaxf(m, fabsf(p1[k]));
        }
        if (p2) {
            if ((((uintptr_t)p2) & 15) == 0 && (k2 & 3) == 0) {
                for (int k = lane * 4; k < k2; k += 256) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(p2 + k);
                    m = fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
                }
            } else {
                for (int k = lane; k < k2; k += 64) m = fmaxf(m, fabsf(p2[k]));
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
        if (lane == 0) scales[row] = x2h_scale_of(m);
    }
}

template <int NP>
__device__ __forceinline__ void y2h_store(char* __restrict__ dst, const f32x4 (&v)[NP], const float (&s)[NP]) {
    constexpr int RPP = YT / HTPR;
    const int t = threadIdx.x, r = t / HTPR, c = (t % HTPR) * 4;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        uint32_t h0, l0, h1, l1;
        x2h_split(v[p][0] * s[p], v[p][1] * s[p], h0, l0);
        x2h_split(v[p][2] * s[p], v[p][3] * s[p], h1, l1);
        char* d = dst + (r + p * RPP) * HLD + c * 2;
        *reinterpret_cast<uint2*>(d) = make_uint2(h0, h1);
        *reinterpret_cast<uint2*>(d + 2 * HK) = make_uint2(l0, l1);
    }
}

__global__ void __launch_bounds__(YT, 1) k_linear_fwd_x2h_big(const float* __restrict__ A1, int64_t lda1, int k1, const float* __restrict__ W1,
                                                              int64_t ldw1, bool vec1, const float* __restrict__ A2, int64_t lda2, int k2,
                                                              const float* __restrict__ W2, int64_t ldw2, bool vec2,
                                                              const float* __restrict__ bias, const float* __restrict__ scale,
                                                              const float* __restrict__ shift, int relu, int64_t M, int n_out,
                                                              float* __restrict__ out, int64_t ldo, const float* __restrict__ scales) {
    extern __shared__ __attribute__((aligned(16))) char y2h_smem[];
    char* const As = y2h_smem;
    char* const Ws = y2h_smem + YM * HLD;
    const int lane = lane_id(), w = wave_id_uniform();
    const int wr = w >> 2, wc = w & 3, h = lane >> 5, l31 = lane & 31;
    const int ncb = (n_out + YN - 1) / YN;   // XCD-aware tile map (see k_linear_fwd_x3)
    const int64_t rb = (int64_t)((blockIdx.x >> 3) / ncb) * 8 + (blockIdx.x & 7);
    if (rb * YM >= M) return;
    const int64_t row0 = rb * YM;
    const int col0 = (int)((blockIdx.x >> 3) % ncb) * YN;
    f32x16 acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
    const int nch1 = (k1 + HK - 1) / HK, nch2 = A2 ? (k2 + HK - 1) / HK : 0, nch = nch1 + nch2;
    constexpr int NP = YM / (YT / HTPR);
    f32x4 ra[NP], rw[NP];
    float sa[NP], sw[NP];   // scales of the rows this thread stages (fixed for the whole K walk)
    {
        const int r = threadIdx.x / HTPR;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int64_t ga = row0 + r + p * (YT / HTPR);
            const int gw = col0 + r + p * (YT / HTPR);
            sa[p] = scales[ga < M ? ga : M - 1];
            sw[p] = scales[M + (gw < n_out ? gw : n_out - 1)];
        }
    }
    auto load_chunk = [&](int ch) {
        const bool first = ch < nch1;
        const int kk = first ? k1 : k2, k0 = (first ? ch : ch - nch1) * HK;
        y2h_load<NP>(ra, first ? A1 : A2, first ? lda1 : lda2, row0, M, k0, kk, first ? vec1 : vec2);
        y2h_load<NP>(rw, first ? W1 : W2, first ? ldw1 : ldw2, col0, n_out, k0, kk, first ? vec1 : vec2);
    };
    load_chunk(0);
    for (int ch = 0; ch < nch; ++ch) {
        __syncthreads();
        y2h_store<NP>(As, ra, sa);
        y2h_store<NP>(Ws, rw, sw);
        __syncthreads();
        if (ch + 1 < nch) load_chunk(ch + 1);   // in flight under the MFMAs below
        const char* ap = As + (wr * 128 + l31) * HLD + h * 16;
        const char* bp = Ws + (wc * 64 + l31) * HLD + h * 16;
#pragma unroll
        for (int S = 0; S < HK / 16; ++S) {
            h16x8_t bf[2][2];
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int p = 0; p < 2; ++p) bf[m][p] = *reinterpret_cast<const h16x8_t*>(bp + m * 32 * HLD + p * 2 * HK + S * 32);
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                h16x8_t af[2];
#pragma unroll
                for (int p = 0; p < 2; ++p) af[p] = *reinterpret_cast<const h16x8_t*>(ap + a * 32 * HLD + p * 2 * HK + S * 32);
                // small terms first; the two column blocks alternate product by product
                constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};
#pragma unroll
                for (int q = 0; q < 3; ++q)
#pragma unroll
                    for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[PA[q]], bf[b][PB[q]], acc[a][b], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int col = col0 + wc * 64 + b * 32 + l31;
        if (col >= n_out) continue;
        const float bb = bias ? bias[col] : 0.f;
        const float sc = scale ? scale[col] : 1.f;
        const float sh = scale ? shift[col] : 0.f;
        const float iw = x2h_inv(scales[M + col]);
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t row = row0 + wr * 128 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (row >= M) continue;
                float v = __fmaf_rn(acc[a][b][r], x2h_inv(scales[row]) * iw, bb);
                if (scale) v = __fmaf_rn(v, sc, sh);
                if (relu & 1) v = fmaxf(v, 0.f);
                if (relu & DGNN_LINEAR_ACCUMULATE) v += out[row * ldo + col];
                out[row * ldo + col] = v;
            }
    }
}

// =====================================================================================================================
// "x2hp": the x2h GEMM with operands split ONCE, by a pass of their own, instead of by every workgroup that stages them.
//   pass 1 (k_x2hp_presplit): one wavefront per row of [A1 | A2] and of [W1 | W2]: row maximum -> power-of-two scale (as k_x2h_row_scales), then
//       the scaled row as (hi, lo) fp16 in the GEMM's own staging format: per 32-wide K chunk 128 bytes = [hi x 32 | lo x 32]; k1 and k2 are
//       padded to whole chunks with zeros.  Costs one more write + read of the operand (4 B per element, what the fp32 operand weighs).
//   pass 2 (k_linear_fwd_x2hp_big): the 256 x 256 tile of x2h; a K chunk of both operands goes global -> LDS by DMA (global_load_lds, 16 bytes per
//       lane, no registers, no VALU), double-buffered: ONE barrier per chunk, the next chunk's DMA in flight under the 48 products of the current one.
//       LDS rows are 128 bytes without padding; the 16-byte pieces of a row are stored at slot (piece ^ (row & 7)) -- chosen through the DMA's
//       per-lane SOURCE address -- so that the 32 rows a fragment read touches spread over all banks.
// Same scaling groups, same chunk / k-step / product order as x2h: bit-identical results (tested).  What it removes is the ~220 split / address
// VALU instructions per chunk that stand between the matrix instructions of x2h / x3 and the second barrier (DESIGN 7 "Next 2").
// =====================================================================================================================
constexpr int PCH = 128;   // bytes of one row's K chunk: [hi x 32 | lo x 32] fp16

__global__ void __launch_bounds__(256) k_x2hp_presplit(const float* __restrict__ A1, int64_t lda1, int k1, const float* __restrict__ A2, int64_t lda2, int k2,
                                                       int64_t M, const float* __restrict__ W1, int64_t ldw1, const float* __restrict__ W2, int64_t ldw2,
                                                       int n_out, float* __restrict__ scales, char* __restrict__ As, char* __restrict__ Ws) {
    const int lane = lane_id();
    const int nch1 = (k1 + HK - 1) / HK, nch2 = A2 ? (k2 + HK - 1) / HK : 0;
    const int64_t rowb = (int64_t)(nch1 + nch2) * PCH;
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    for (int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < M + n_out; row += nwaves) {
        const bool isw = row >= M;
        const float* p1 = isw ? W1 + (row - M) * ldw1 : A1 + row * lda1;
        const float* p2 = isw ? (W2 ? W2 + (row - M) * ldw2 : nullptr) : (A2 ? A2 + row * lda2 : nullptr);
        float m = 0.f;
        for (int k = lane; k < k1; k += 64) m = fmaxf(m, fabsf(p1[k]));
        if (p2)
            for (int k = lane; k < k2; k += 64) m = fmaxf(m, fabsf(p2[k]));
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
        const float s_ = x2h_scale_of(m);
        if (lane == 0) scales[row] = s_;
        char* dst = (isw ? Ws + (row - M) * rowb : As + row * rowb);
        // 4 chunks per sweep: lane -> (chunk, pair of consecutive k); the row was read a moment ago and comes from the caches
        const int pr = lane & 15, cq = lane >> 4;
        for (int ch = cq; ch < nch1 + nch2; ch += 4) {
            const bool first = ch < nch1;
            const float* src_ = first ? p1 : p2;
            const int kk = first ? k1 : k2, k = (first ? ch : ch - nch1) * HK + 2 * pr;
            const float v0 = k < kk ? src_[k] : 0.f, v1 = k + 1 < kk ? src_[k + 1] : 0.f;
            uint32_t hi, lo;
            x2h_split(v0 * s_, v1 * s_, hi, lo);
            uint32_t* d = reinterpret_cast<uint32_t*>(dst + (int64_t)ch * PCH) + pr;
            d[0] = hi;
            d[16] = lo;
        }
    }
}

__device__ __forceinline__ void x2hp_dma16(const char* g, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

__global__ void __launch_bounds__(YT, 1) k_linear_fwd_x2hp_big(const char* __restrict__ As, const char* __restrict__ Ws, int nch,
                                                               const float* __restrict__ bias, const float* __restrict__ scale,
                                                               const float* __restrict__ shift, int relu, int64_t M, int n_out,
                                                               float* __restrict__ out, int64_t ldo, const float* __restrict__ scales) {
    extern __shared__ __attribute__((aligned(16))) char y2p_smem[];   // [2 buffers][A 256 rows | W 256 rows] x 128 B
    constexpr int BUF = (YM + YN) * PCH;
    const int lane = lane_id(), w = wave_id_uniform();
    const int wr = w >> 2, wc = w & 3, h = lane >> 5, l31 = lane & 31;
    const int ncb = (n_out + YN - 1) / YN;   // XCD-aware tile map (see k_linear_fwd_x3)
    const int64_t rb = (int64_t)((blockIdx.x >> 3) / ncb) * 8 + (blockIdx.x & 7);
    if (rb * YM >= M) return;
    const int64_t row0 = rb * YM;
    const int col0 = (int)((blockIdx.x >> 3) % ncb) * YN;
    const int64_t rowb = (int64_t)nch * PCH;
    f32x16 acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
    // DMA roles: one instruction moves 8 rows x 128 B (1 KB of LDS, contiguous); this wave takes instructions w, w + 8, ... of the 32 (A) + 32 (W).
    // lane -> (row of the group rr = lane >> 3, LDS slot q = lane & 7); slot q of row r holds logical piece q ^ (r & 7)
    const int rr = lane >> 3, q = lane & 7;
    const char* gsrc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int ins = w + 8 * j;                 // 0..31: A rows, 32..63: W rows
        const int r_t = (ins & 31) * 8 + rr;       // row within the tile
        const int piece = q ^ (r_t & 7);
        if (ins < 32) {
            const int64_t gr = row0 + r_t;
            gsrc[j] = As + (gr < M ? gr : M - 1) * rowb + piece * 16;
        } else {
            const int gc = col0 + r_t;
            gsrc[j] = Ws + (int64_t)(gc < n_out ? gc : n_out - 1) * rowb + piece * 16;
        }
    }
    auto dma_chunk = [&](int ch) {
        char* buf = y2p_smem + (ch & 1) * BUF;
#pragma unroll
        for (int j = 0; j < 8; ++j) x2hp_dma16(gsrc[j] + (int64_t)ch * PCH, buf + (w + 8 * j) * 1024);
    };
    dma_chunk(0);
    for (int ch = 0; ch < nch; ++ch) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of chunk ch has landed
        __syncthreads();                                     // everybody's has; everybody is done with the other buffer (chunk ch - 1)
        if (ch + 1 < nch) dma_chunk(ch + 1);                 // in flight under the products below
        const char* Ab = y2p_smem + (ch & 1) * BUF;
        const char* Wb = Ab + YM * PCH;
        const int ar0 = wr * 128 + l31, br0 = wc * 64 + l31;   // rows are multiples of 32 apart: (row & 7) = (l31 & 7) for all of them
        const int sw = l31 & 7;
#pragma unroll
        for (int S = 0; S < HK / 16; ++S) {
            h16x8_t bf[2][2];
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int p = 0; p < 2; ++p) bf[m][p] = *reinterpret_cast<const h16x8_t*>(Wb + (br0 + m * 32) * PCH + (((4 * p + 2 * S + h) ^ sw) << 4));
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                h16x8_t af[2];
#pragma unroll
                for (int p = 0; p < 2; ++p) af[p] = *reinterpret_cast<const h16x8_t*>(Ab + (ar0 + a * 32) * PCH + (((4 * p + 2 * S + h) ^ sw) << 4));
                constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};   // small terms first; the two column blocks alternate product by product (as x2h)
#pragma unroll
                for (int qq = 0; qq < 3; ++qq)
#pragma unroll
                    for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[PA[qq]], bf[b][PB[qq]], acc[a][b], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int col = col0 + wc * 64 + b * 32 + l31;
        if (col >= n_out) continue;
        const float bb = bias ? bias[col] : 0.f;
        const float sc = scale ? scale[col] : 1.f;
        const float sh = scale ? shift[col] : 0.f;
        const float iw = x2h_inv(scales[M + col]);
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t row = row0 + wr * 128 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (row >= M) continue;
                float v = __fmaf_rn(acc[a][b][r], x2h_inv(scales[row]) * iw, bb);
                if (scale) v = __fmaf_rn(v, sc, sh);
                if (relu & 1) v = fmaxf(v, 0.f);
                if (relu & DGNN_LINEAR_ACCUMULATE) v += out[row * ldo + col];
                out[row * ldo + col] = v;
            }
    }
}

// dW[na, nb] = sum_rows A[r, :]^T B[r, :], fp32 operands split in 3 bf16 parts while they are staged TRANSPOSED ([column][row]):
// a thread takes 2 consecutive rows x 4 columns, so that a packed bf16 pair is two consecutive k (= rows) of one column.
// 64 x 64 tile per block, wave (wa, wb) owns a 32 x 32 block; row slices of 32.
constexpr int XRK = 32, XLDT = 3 * XRK * 2 + 16;

__device__ __forceinline__ void stage_t_x3(char* __restrict__ dst, const float* __restrict__ src, int64_t ld, int64_t r0, int64_t r_end, int c0,
                                           int nc) {
    // 256 threads: (row pair tp = t >> 4 in 0..15, column quad tc = (t & 15) * 4)
    const int t = threadIdx.x, tp = t >> 4, tc = (t & 15) * 4;
    const int64_t ra = r0 + 2 * tp, rb = ra + 1;
    float va[4], vb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int cc = c0 + tc + j;
        va[j] = (ra < r_end && cc < nc) ? src[ra * ld + cc] : 0.f;
        vb[j] = (rb < r_end && cc < nc) ? src[rb * ld + cc] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        uint32_t hh, mm, ll;
        x3_split(va[j], vb[j], hh, mm, ll);
        char* d = dst + (tc + j) * XLDT + tp * 4;
        *reinterpret_cast<uint32_t*>(d) = hh;
        *reinterpret_cast<uint32_t*>(d + 64) = mm;
        *reinterpret_cast<uint32_t*>(d + 128) = ll;
    }
}

__global__ void __launch_bounds__(256) k_linear_wgrad_x3(const float* __restrict__ A, int64_t lda, int na, const float* __restrict__ B, int64_t ldb,
                                                         int nb, int64_t M, int64_t rows_per_split, float* __restrict__ partials) {
    __shared__ __attribute__((aligned(16))) char At[WT * XLDT];
    __shared__ __attribute__((aligned(16))) char Bt[WT * XLDT];
    const int lane = lane_id(), w = wave_id_uniform();
    const int wa = w >> 1, wb = w & 1, h = lane >> 5, l31 = lane & 31;
    const int a0 = blockIdx.x * WT, b0 = blockIdx.y * WT;
    const int64_t r_beg = (int64_t)blockIdx.z * rows_per_split;
    const int64_t r_end = min(M, r_beg + rows_per_split);
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    for (int64_t r0 = r_beg; r0 < r_end; r0 += XRK) {
        __syncthreads();
        stage_t_x3(At, A, lda, r0, r_end, a0, na);
        stage_t_x3(Bt, B, ldb, r0, r_end, b0, nb);
        __syncthreads();
        const char* ap = At + (wa * 32 + l31) * XLDT + h * 16;
        const char* bp = Bt + (wb * 32 + l31) * XLDT + h * 16;
#pragma unroll
        for (int S = 0; S < XRK / 16; ++S) {
            bf16x8_t af[3], bf[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                af[p] = *reinterpret_cast<const bf16x8_t*>(ap + p * 64 + S * 32);
                bf[p] = *reinterpret_cast<const bf16x8_t*>(bp + p * 64 + S * 32);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[2], bf[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], bf[1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], bf[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[0], acc, 0, 0, 0);
        }
    }
    float* P = partials + (int64_t)blockIdx.z * na * nb;
    const int col = b0 + wb * 32 + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = a0 + wa * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (row < na && col < nb) P[(int64_t)row * nb + col] = acc[r];
    }
}

// stage_t_x3 in two halves, so that the next row slice's values travel while the matrix instructions of the current one run
__device__ __forceinline__ void load_t_x3(float (&va)[4], float (&vb)[4], const float* __restrict__ src, int64_t ld, bool vec, int64_t r0, int64_t r_end,
                                          int c0, int nc) {
    const int t = threadIdx.x, tp = t >> 4, tc = (t & 15) * 4;
    const int64_t ra = r0 + 2 * tp, rb = ra + 1;
    if (vec && c0 + tc + 4 <= nc) {
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        const f32x4 a = ra < r_end ? *reinterpret_cast<const f32x4*>(src + ra * ld + c0 + tc) : z;
        const f32x4 b = rb < r_end ? *reinterpret_cast<const f32x4*>(src + rb * ld + c0 + tc) : z;
#pragma unroll
        for (int j = 0; j < 4; ++j) va[j] = a[j], vb[j] = b[j];
        return;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int cc = c0 + tc + j;
        va[j] = (ra < r_end && cc < nc) ? src[ra * ld + cc] : 0.f;
        vb[j] = (rb < r_end && cc < nc) ? src[rb * ld + cc] : 0.f;
    }
}
__device__ __forceinline__ void store_t_x3(char* __restrict__ dst, const float (&va)[4], const float (&vb)[4]) {
    const int t = threadIdx.x, tp = t >> 4, tc = (t & 15) * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        uint32_t hh, mm, ll;
        x3_split(va[j], vb[j], hh, mm, ll);
        char* d = dst + (tc + j) * XLDT + tp * 4;
        *reinterpret_cast<uint32_t*>(d) = hh;
        *reinterpret_cast<uint32_t*>(d + 64) = mm;
        *reinterpret_cast<uint32_t*>(d + 128) = ll;
    }
}

// Both weight gradients of a conv layer and the bias gradient in one launch: dW1 = A^T B1, dW2 = A^T B2 (B2 optional), dbias = column
// sums of A.  Block column y takes its 64 columns from B1 (y < nby1) or B2; every output element runs k_linear_wgrad_x3's loop on the
// same row splits (bit-identical dW1 / dW2).  The bias sums ride on the staging of A in the blocks of column 0: every thread adds the
// values it stages (fp64), the 16 row-pair threads of a column are summed in a fixed order through LDS, one partial per row split.
struct WgradCat {
    const float* B[2];
    int64_t ldb[2];
    int nb[2];
};

__global__ void __launch_bounds__(256) k_linear_wgrad_x3_cat(const float* __restrict__ A, int64_t lda, int na, WgradCat c, int nby1, int64_t M,
                                                             int64_t rows_per_split, float* __restrict__ partials, double* __restrict__ bias_partials) {
    __shared__ __attribute__((aligned(16))) char At[WT * XLDT];
    __shared__ __attribute__((aligned(16))) char Bt[WT * XLDT];
    const int lane = lane_id(), w = wave_id_uniform();
    const int wa = w >> 1, wb = w & 1, h = lane >> 5, l31 = lane & 31;
    const int which = (int)blockIdx.y >= nby1 ? 1 : 0;
    const float* __restrict__ B = c.B[which];
    const int64_t ldb = c.ldb[which];
    const int nb = c.nb[which], nbt = c.nb[0] + c.nb[1], colbase = which ? c.nb[0] : 0;
    const int a0 = blockIdx.x * WT, b0 = ((int)blockIdx.y - (which ? nby1 : 0)) * WT;
    const int64_t r_beg = (int64_t)blockIdx.z * rows_per_split;
    const int64_t r_end = min(M, r_beg + rows_per_split);
    const bool do_bias = bias_partials != nullptr && blockIdx.y == 0;
    const int t = threadIdx.x, tp = t >> 4, tc = (t & 15) * 4;
    const bool vecA = ((uintptr_t)A % 16 == 0) && (lda % 4 == 0), vecB = ((uintptr_t)B % 16 == 0) && (ldb % 4 == 0);
    double bs[4] = {0.0, 0.0, 0.0, 0.0};
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    float aa[4], ab[4], ba[4], bb[4];   // the slice in flight: rows r0 + 2 tp (.a) and + 1 (.b), columns tc .. tc + 3 of the A and B tiles
    if (r_beg < r_end) {
        load_t_x3(aa, ab, A, lda, vecA, r_beg, r_end, a0, na);
        load_t_x3(ba, bb, B, ldb, vecB, r_beg, r_end, b0, nb);
    }
    for (int64_t r0 = r_beg; r0 < r_end; r0 += XRK) {
        __syncthreads();
        store_t_x3(At, aa, ab);
        store_t_x3(Bt, ba, bb);
        if (do_bias) {   // rows beyond the split and columns beyond na arrive as zeros
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                bs[j] += (double)aa[j];
                bs[j] += (double)ab[j];
            }
        }
        __syncthreads();
        if (r0 + XRK < r_end) {   // in flight under the matrix instructions below
            load_t_x3(aa, ab, A, lda, vecA, r0 + XRK, r_end, a0, na);
            load_t_x3(ba, bb, B, ldb, vecB, r0 + XRK, r_end, b0, nb);
        }
        const char* ap = At + (wa * 32 + l31) * XLDT + h * 16;
        const char* bp = Bt + (wb * 32 + l31) * XLDT + h * 16;
#pragma unroll
        for (int S = 0; S < XRK / 16; ++S) {
            bf16x8_t af[3], bf[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                af[p] = *reinterpret_cast<const bf16x8_t*>(ap + p * 64 + S * 32);
                bf[p] = *reinterpret_cast<const bf16x8_t*>(bp + p * 64 + S * 32);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[2], bf[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], bf[1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], bf[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[0], acc, 0, 0, 0);
        }
    }
    float* P = partials + (int64_t)blockIdx.z * na * nbt;
    const int col = b0 + wb * 32 + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = a0 + wa * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (row < na && col < nb) P[(int64_t)row * nbt + colbase + col] = acc[r];
    }
    if (do_bias) {
        __syncthreads();                                 // the fragments of the last slice have been read
        double* red = reinterpret_cast<double*>(At);     // [16 row pairs][64 columns] = 8 KB of the 13 KB tile
#pragma unroll
        for (int j = 0; j < 4; ++j) red[tp * 64 + tc + j] = bs[j];
        __syncthreads();
        if (t < 64 && a0 + t < na) {
            double sum = 0.0;
            for (int k = 0; k < 16; ++k) sum += red[k * 64 + t];
            bias_partials[(int64_t)blockIdx.z * na + a0 + t] = sum;
        }
    }
}

// dW1 / dW2 / dbias = sums over the row splits in k_wgrad_reduce's order; the blocks behind the matrix elements take the bias columns
__global__ void __launch_bounds__(256) k_wgrad_reduce_cat(WgradReduceDesc d) {
    __shared__ float redf[16][17];
    __shared__ double redd[16][17];
    wgrad_reduce_cat_phase(d, blockIdx.x, threadIdx.x, redf, redd, 0);
    __syncthreads();
    wgrad_reduce_cat_phase(d, blockIdx.x, threadIdx.x, redf, redd, 1);
}

// Row splits of a weight gradient A^T B (A [M, n_a]).  n_a = 0: the upper bound the scratch sizes are computed from.
// Round 6: the count follows the number of 64-row tiles of the OUTPUT along n_a -- every split writes an [n_a, n_b] partial matrix, and at the
// reference's training widths 512 splits of a [1024, 2048] gradient were 4 GB of partials written and read back per layer (k_linear_wgrad_x3_cat
// 165 us + k_reduce_layer 90 us of a 1.18 ms step); a gradient with many tiles fills the chip with far fewer splits.  It depends on (M, n_a) only, so
// the separate and the merged launches of a layer's gradients still walk the same row splits (bit-identical results).
int wgrad_splits(int64_t M, int n_a = 0) {
    int64_t s = dgnn_cdiv(M, 4 * RK);  // at least 128 rows per split
    int64_t cap = WGRAD_SPLITS;
    if (n_a > 0) {
        static const bool by_tiles = !(getenv("DGNN_WGRAD_SPLITS_BY_TILES") && getenv("DGNN_WGRAD_SPLITS_BY_TILES")[0] == '0');
        if (by_tiles) {
            static const int base = getenv("DGNN_WGRAD_SPLIT_CAP") && atoi(getenv("DGNN_WGRAD_SPLIT_CAP")) > 0 ? atoi(getenv("DGNN_WGRAD_SPLIT_CAP")) : WGRAD_SPLITS;
            cap = base / dgnn_cdiv(n_a, WT);
            if (cap < 32) cap = 32;
        }
    }
    if (s > cap) s = cap;
    return (int)(s < 1 ? 1 : s);
}

bool vec_ok(const float* p, int64_t ld) { return ((uintptr_t)p % 16 == 0) && (ld % 4 == 0); }

}  // namespace

extern "C" int dgnn_linear_fwd(const float* A1, int64_t lda1, int k1, const float* W1, int64_t ldw1, const float* A2,
                               int64_t lda2, int k2, const float* W2, int64_t ldw2, const float* bias, const float* scale,
                               const float* shift, int relu, int64_t M, int n_out, float* out, int64_t ldo, void* stream) {
    DGNN_REQUIRE(M >= 0 && n_out > 0 && k1 > 0, DGNN_E_INVALID, "linear_fwd: bad sizes M=%lld n_out=%d k1=%d", (long long)M, n_out, k1);
    if (M == 0) return DGNN_OK;
    DGNN_REQUIRE(A1 && W1 && out, DGNN_E_INVALID, "linear_fwd: null pointer");
    DGNN_REQUIRE((A2 == nullptr) == (W2 == nullptr) && (!A2 || k2 > 0), DGNN_E_INVALID, "linear_fwd: A2/W2 must come together");
    DGNN_REQUIRE((scale == nullptr) == (shift == nullptr), DGNN_E_INVALID, "linear_fwd: scale/shift must come together");
    const bool v1 = vec_ok(A1, lda1) && vec_ok(W1, ldw1);
    const bool v2 = A2 && vec_ok(A2, lda2) && vec_ok(W2, ldw2);
    dim3 grid((unsigned)(dgnn_cdiv(M, BM) * dgnn_cdiv(n_out, BN)));
    hipLaunchKernelGGL(k_linear_fwd, grid, dim3(256), 0, (hipStream_t)stream, A1, lda1, k1, W1, ldw1, v1, A2, lda2, k2, W2,
                       ldw2, v2, bias, scale, shift, relu, M, n_out, out, ldo);
    return dgnn_check_launch("linear_fwd");
}

extern "C" int64_t dgnn_linear_wgrad_scratch_elems(int64_t M, int n_a, int n_b) {
    if (M < 0 || n_a <= 0 || n_b <= 0) return 1;
    return (int64_t)wgrad_splits(M, n_a) * n_a * n_b;
}

extern "C" int dgnn_linear_wgrad(const float* A, int64_t lda, int n_a, const float* B, int64_t ldb, int n_b, int64_t M,
                                 float* dW, int64_t lddw, int accumulate, float* partials, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DGNN_REQUIRE(M >= 0 && n_a > 0 && n_b > 0, DGNN_E_INVALID, "linear_wgrad: bad sizes");
    DGNN_REQUIRE(dW && partials && (M == 0 || (A && B)), DGNN_E_INVALID, "linear_wgrad: null pointer");
    const int splits = wgrad_splits(M, n_a);
    const int64_t rps = dgnn_cdiv(dgnn_cdiv(M, splits), RK) * RK;
    dim3 grid((unsigned)dgnn_cdiv(n_a, WT), (unsigned)dgnn_cdiv(n_b, WT), splits);
    hipLaunchKernelGGL(k_linear_wgrad, grid, dim3(256), 0, stream, A, lda, n_a, B, ldb, n_b, M, rps < RK ? RK : rps, partials);
    hipLaunchKernelGGL(k_wgrad_reduce, dim3((unsigned)dgnn_cdiv((int64_t)n_a * n_b, 16)), dim3(256), 0, stream, partials, splits,
                       n_a, n_b, dW, lddw, accumulate);
    return dgnn_check_launch("linear_wgrad");
}


// ---- bf16 storage entry points ----------------------------------------------------------------------------------------------
static bool vec16(const void* p, int64_t ld_elems, size_t elem) { return ((uintptr_t)p % 16 == 0) && ((ld_elems * (int64_t)elem) % 16 == 0); }

extern "C" int dgnn_linear_fwd_bf16(const uint16_t* A1, int64_t lda1, int k1, const float* W1, int64_t ldw1, const uint16_t* A2, int64_t lda2,
                                    int k2, const float* W2, int64_t ldw2, const float* bias, const float* scale, const float* shift, int relu,
                                    int64_t M, int n_out, void* out, int64_t ldo, int out_f32, void* stream) {
    DGNN_REQUIRE(M >= 0 && n_out > 0 && k1 > 0, DGNN_E_INVALID, "linear_fwd_bf16: bad sizes M=%lld n_out=%d k1=%d", (long long)M, n_out, k1);
    if (M == 0) return DGNN_OK;
    DGNN_REQUIRE(A1 && W1 && out, DGNN_E_INVALID, "linear_fwd_bf16: null pointer");
    DGNN_REQUIRE((A2 == nullptr) == (W2 == nullptr) && (!A2 || k2 > 0), DGNN_E_INVALID, "linear_fwd_bf16: A2/W2 must come together");
    DGNN_REQUIRE((scale == nullptr) == (shift == nullptr), DGNN_E_INVALID, "linear_fwd_bf16: scale/shift must come together");
    const bool va1 = vec16(A1, lda1, 2), vw1 = vec16(W1, ldw1, 4);
    const bool va2 = A2 && vec16(A2, lda2, 2), vw2 = W2 && vec16(W2, ldw2, 4);
    static const bool small_ok = !(getenv("DGNN_BF16_SMALL") && getenv("DGNN_BF16_SMALL")[0] == '0');
    static const bool splitk_ok = !(getenv("DGNN_SMALL_SPLITK") && getenv("DGNN_SMALL_SPLITK")[0] == '0');
    // Round 6: the small-problem kernel re-reads its operands once per 32 x 32 output block -- fine for the shipped widths' inner blocks, not for the
    // reference's training widths (M = 6 000, K = N = 512: 38 us against the tiled kernel's 28; M = 12 000, K = N = 1024: 249 against 100,
    // tools/bench_gemm_train_shapes.py).  It keeps the problems whose 128 x 64 tiles would leave most of the chip idle (fewer than 150 of them).
    static const bool by_tiles = !(getenv("DGNN_SMALL_BY_TILES") && getenv("DGNN_SMALL_BY_TILES")[0] == '0');
    const bool tiles_fill = by_tiles && k1 + (A2 ? k2 : 0) >= 128 && dgnn_cdiv(M, BM) * dgnn_cdiv(n_out, BN) >= 150;
    static const bool mid_ok = !(getenv("DGNN_GEMM_MID") && getenv("DGNN_GEMM_MID")[0] == '0');
    if (mid_ok && !tiles_fill && M <= 16384 && k1 + (A2 ? k2 : 0) >= 512 && dgnn_cdiv(M, 64) * dgnn_cdiv(n_out, 64) >= 128) {   // 64 x 64 tiles (see the kernel)
        dim3 mgrid((unsigned)(dgnn_cdiv(M, 64) * dgnn_cdiv(n_out, 64)));
        constexpr size_t lds1 = (size_t)(64 + 64) * LDB;
        static const bool ks_ok = !(getenv("DGNN_GEMM_MID_KS") && getenv("DGNN_GEMM_MID_KS")[0] == '0');
        const bool ks4 = ks_ok && k1 + (A2 ? k2 : 0) >= 1024 && dgnn_cdiv(M, 64) * dgnn_cdiv(n_out, 64) <= 2 * DGNN_NUM_CU;
        static_assert(4 * lds1 >= 3 * 4 * 16 * 64 * sizeof(float), "the groups' partial blocks fit the chunk buffers");
#define DGNN_MIDB(TO_, KS_)                                                                                                                                  \
        hipLaunchKernelGGL((k_linear_fwd_b_mid<TO_, KS_>), mgrid, dim3(256 * KS_), KS_ * lds1, (hipStream_t)stream, A1, lda1, k1, W1, ldw1, va1, vw1, A2, lda2, k2, \
                           W2, ldw2, va2, vw2, bias, scale, shift, relu, M, n_out, (TO_*)out, ldo)
        if (ks4) {
            static bool attr_f[DGNN_MAX_DEVICES], attr_h[DGNN_MAX_DEVICES];
            if (out_f32) {
                dgnn_allow_dynamic_lds((const void*)k_linear_fwd_b_mid<float, 4>, 4 * lds1, attr_f);
                DGNN_MIDB(float, 4);
            } else {
                dgnn_allow_dynamic_lds((const void*)k_linear_fwd_b_mid<uint16_t, 4>, 4 * lds1, attr_h);
                DGNN_MIDB(uint16_t, 4);
            }
        } else {
            if (out_f32) DGNN_MIDB(float, 1); else DGNN_MIDB(uint16_t, 1);
        }
#undef DGNN_MIDB
        return dgnn_check_launch("linear_fwd_bf16");
    }
    if (small_ok && !tiles_fill && splitk_ok && M <= 16384 && k1 + (A2 ? k2 : 0) >= 1024) {   // four wavefronts per output block, a quarter of K each (see the kernel)
        dim3 sgrid((unsigned)(dgnn_cdiv(M, 32) * dgnn_cdiv(n_out, 32)));
        if (out_f32)
            hipLaunchKernelGGL((k_linear_fwd_b_small<float, true>), sgrid, dim3(256), 0, (hipStream_t)stream, A1, lda1, k1, W1, ldw1, va1, vw1, A2, lda2, k2, W2,
                               ldw2, va2, vw2, bias, scale, shift, relu, M, n_out, (float*)out, ldo);
        else
            hipLaunchKernelGGL((k_linear_fwd_b_small<uint16_t, true>), sgrid, dim3(256), 0, (hipStream_t)stream, A1, lda1, k1, W1, ldw1, va1, vw1, A2, lda2, k2,
                               W2, ldw2, va2, vw2, bias, scale, shift, relu, M, n_out, (uint16_t*)out, ldo);
        return dgnn_check_launch("linear_fwd_bf16");
    }
    if (small_ok && !tiles_fill && M <= 16384) {   // same k order per output element: identical results
        dim3 sgrid((unsigned)dgnn_cdiv(dgnn_cdiv(M, 32) * dgnn_cdiv(n_out, 32), 4));
        if (out_f32)
            hipLaunchKernelGGL((k_linear_fwd_b_small<float>), sgrid, dim3(256), 0, (hipStream_t)stream, A1, lda1, k1, W1, ldw1, va1, vw1, A2, lda2, k2, W2,
                               ldw2, va2, vw2, bias, scale, shift, relu, M, n_out, (float*)out, ldo);
        else
            hipLaunchKernelGGL((k_linear_fwd_b_small<uint16_t>), sgrid, dim3(256), 0, (hipStream_t)stream, A1, lda1, k1, W1, ldw1, va1, vw1, A2, lda2, k2,
                               W2, ldw2, va2, vw2, bias, scale, shift, relu, M, n_out, (uint16_t*)out, ldo);
        return dgnn_check_launch("linear_fwd_bf16");
    }
    dim3 grid((unsigned)(dgnn_cdiv(M, BM) * dgnn_cdiv(n_out, BN)));
    if (out_f32)
        hipLaunchKernelGGL((k_linear_fwd_b<float>), grid, dim3(256), 0, (hipStream_t)stream, A1, lda1, k1, W1, ldw1, va1, vw1, A2, lda2, k2, W2, ldw2,
                           va2, vw2, bias, scale, shift, relu, M, n_out, (float*)out, ldo);
    else
        hipLaunchKernelGGL((k_linear_fwd_b<uint16_t>), grid, dim3(256), 0, (hipStream_t)stream, A1, lda1, k1, W1, ldw1, va1, vw1, A2, lda2, k2, W2,
                           ldw2, va2, vw2, bias, scale, shift, relu, M, n_out, (uint16_t*)out, ldo);
    return dgnn_check_launch("linear_fwd_bf16");
}

extern "C" int dgnn_linear_wgrad_bf16(const void* A, int a_f32, int64_t lda, int n_a, const void* B, int b_f32, int64_t ldb, int n_b, int64_t M,
                                      float* dW, int64_t lddw, int accumulate, float* partials, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DGNN_REQUIRE(M >= 0 && n_a > 0 && n_b > 0, DGNN_E_INVALID, "linear_wgrad_bf16: bad sizes");
    DGNN_REQUIRE(dW && partials && (M == 0 || (A && B)), DGNN_E_INVALID, "linear_wgrad_bf16: null pointer");
    const int splits = wgrad_splits(M, n_a);
    int64_t rps = dgnn_cdiv(dgnn_cdiv(M, splits), RKB) * RKB;
    if (rps < RKB) rps = RKB;
    dim3 grid((unsigned)dgnn_cdiv(n_a, WT), (unsigned)dgnn_cdiv(n_b, WT), splits);
#define WG(TA, TB) hipLaunchKernelGGL((k_linear_wgrad_b<TA, TB>), grid, dim3(256), 0, stream, (const TA*)A, lda, n_a, (const TB*)B, ldb, n_b, M, rps, partials)
    if (a_f32 && b_f32) WG(float, float);
    else if (a_f32) WG(float, uint16_t);
    else if (b_f32) WG(uint16_t, float);
    else WG(uint16_t, uint16_t);
#undef WG
    hipLaunchKernelGGL(k_wgrad_reduce, dim3((unsigned)dgnn_cdiv((int64_t)n_a * n_b, 16)), dim3(256), 0, stream, partials, splits, n_a, n_b, dW, lddw,
                       accumulate);
    return dgnn_check_launch("linear_wgrad_bf16");
}


// ---- fp32-class GEMMs on the bf16 matrix cores (3-way exact split, 6 products) ---------------------------------------------
namespace {
int linear_fwd_x3_impl(const float* A1, int64_t lda1, int k1, const float* W1, int64_t ldw1, const float* A2, int64_t lda2, int k2, const float* W2,
                       int64_t ldw2, const float* bias, const float* scale, const float* shift, int relu, int64_t M, int n_out, float* out, int64_t ldo,
                       double* colstats, void* stream) {
    DGNN_REQUIRE(M >= 0 && n_out > 0 && k1 > 0, DGNN_E_INVALID, "linear_fwd_x3: bad sizes M=%lld n_out=%d k1=%d", (long long)M, n_out, k1);
    if (M == 0) return DGNN_OK;
    DGNN_REQUIRE(A1 && W1 && out, DGNN_E_INVALID, "linear_fwd_x3: null pointer");
    DGNN_REQUIRE((A2 == nullptr) == (W2 == nullptr) && (!A2 || k2 > 0), DGNN_E_INVALID, "linear_fwd_x3: A2/W2 must come together");
    DGNN_REQUIRE((scale == nullptr) == (shift == nullptr), DGNN_E_INVALID, "linear_fwd_x3: scale/shift must come together");
    const bool v1 = vec_ok(A1, lda1) && vec_ok(W1, ldw1);
    const bool v2 = A2 && vec_ok(A2, lda2) && vec_ok(W2, ldw2);
    static const bool big_ok = !(getenv("DGNN_X3_BIG") && getenv("DGNN_X3_BIG")[0] == '0');
    // the 256 x 256 tile pays once its grid fills most of the chip (192 tiles; the merged input-gradient GEMM of a training block, M = 10k,
    // n_out = 256, has 40: 42 us there against 29 us for the small tiles); same arithmetic per output element (chunk order, product
    // order) in every variant: identical results
    if (big_ok && M >= 8192 && n_out > XN && dgnn_cdiv(M, YM) * dgnn_cdiv(n_out, YN) >= 192) {
        if (colstats) return DGNN_E_UNSUPPORTED;   // the wide tile has no statistics epilogue: the caller reduces z in a launch of its own
        static bool attr_set[DGNN_MAX_DEVICES];
        constexpr size_t lds = (size_t)(YM + YN) * XLD;
        dgnn_allow_dynamic_lds((const void*)k_linear_fwd_x3_big, lds, attr_set);
        dim3 grid((unsigned)(dgnn_cdiv(dgnn_cdiv(M, YM), 8) * 8 * dgnn_cdiv(n_out, YN)));
        hipLaunchKernelGGL(k_linear_fwd_x3_big, grid, dim3(YT), lds, (hipStream_t)stream, A1, lda1, k1, W1, ldw1, v1, A2, lda2, k2, W2, ldw2, v2,
                           bias, scale, shift, relu, M, n_out, out, ldo);
        return dgnn_check_launch("linear_fwd_x3");
    }
    static const bool small_ok = !(getenv("DGNN_X3_SMALL") && getenv("DGNN_X3_SMALL")[0] == '0');
    static const int64_t small_m = getenv("DGNN_X3_SMALL_M") ? atoll(getenv("DGNN_X3_SMALL_M")) : 16384;
    // (round 6: ... unless the 128 x 128 tiles of the tiled kernels would fill the chip anyway -- 150 of them -- where the small-problem kernel's operand
    // re-reads cost more than its barrier-free walk saves: M = 6 000, K = N = 512 57 -> 44 us, M = 12 000, K = N = 1024 356 -> 172 us)
    static const bool by_tiles = !(getenv("DGNN_SMALL_BY_TILES") && getenv("DGNN_SMALL_BY_TILES")[0] == '0');
    const bool tiles_fill = by_tiles && k1 + (A2 ? k2 : 0) >= 128 && n_out > ZN && dgnn_cdiv(M, XM) * dgnn_cdiv(n_out, XN) >= 150;
    // ... and between the two: 64 x 64 tiles when THEY fill the chip and the inner dimension is long enough for the operand re-reads of the
    // small-problem kernel to matter (k_linear_fwd_x3_mid)
    static const bool mid_ok = !(getenv("DGNN_GEMM_MID") && getenv("DGNN_GEMM_MID")[0] == '0');
    if (mid_ok && !tiles_fill && M <= small_m && n_out > ZN && k1 + (A2 ? k2 : 0) >= 512 && dgnn_cdiv(M, 64) * dgnn_cdiv(n_out, 64) >= 128) {
        dim3 grid((unsigned)(dgnn_cdiv(dgnn_cdiv(M, 64), 8) * 8 * dgnn_cdiv(n_out, 64)));
        constexpr size_t lds1 = (size_t)(64 + 64) * XLD;
        static const bool ks_ok = !(getenv("DGNN_GEMM_MID_KS") && getenv("DGNN_GEMM_MID_KS")[0] == '0');
        if (ks_ok && k1 + (A2 ? k2 : 0) >= 1024 && dgnn_cdiv(M, 64) * dgnn_cdiv(n_out, 64) <= 2 * DGNN_NUM_CU) {
            static bool attr_set[DGNN_MAX_DEVICES];
            dgnn_allow_dynamic_lds((const void*)k_linear_fwd_x3_mid<4>, 4 * lds1, attr_set);
            hipLaunchKernelGGL(k_linear_fwd_x3_mid<4>, grid, dim3(1024), 4 * lds1, (hipStream_t)stream, A1, lda1, k1, W1, ldw1, v1, A2, lda2, k2, W2, ldw2, v2, bias,
                               scale, shift, relu, M, n_out, out, ldo, colstats);
        } else {
            hipLaunchKernelGGL(k_linear_fwd_x3_mid<1>, grid, dim3(256), lds1, (hipStream_t)stream, A1, lda1, k1, W1, ldw1, v1, A2, lda2, k2, W2, ldw2, v2, bias, scale,
                               shift, relu, M, n_out, out, ldo, colstats);
        }
        return dgnn_check_launch("linear_fwd_x3");
    }
    if (small_ok && !tiles_fill && M <= small_m) {
        const int64_t tiles = dgnn_cdiv(M, 32) * dgnn_cdiv(n_out, 32);
        static const bool splitk_ok = !(getenv("DGNN_SMALL_SPLITK") && getenv("DGNN_SMALL_SPLITK")[0] == '0');
        if (splitk_ok && k1 + (A2 ? k2 : 0) >= 1024)
            hipLaunchKernelGGL(k_linear_fwd_x3_small<true>, dim3((unsigned)tiles), dim3(256), 0, (hipStream_t)stream, A1, lda1, k1, W1, ldw1, v1, A2, lda2,
                               k2, W2, ldw2, v2, bias, scale, shift, relu, M, n_out, out, ldo, colstats);
        else
            hipLaunchKernelGGL(k_linear_fwd_x3_small<false>, dim3((unsigned)dgnn_cdiv(tiles, 4)), dim3(256), 0, (hipStream_t)stream, A1, lda1, k1, W1, ldw1, v1, A2,
                               lda2, k2, W2, ldw2, v2, bias, scale, shift, relu, M, n_out, out, ldo, colstats);
        return dgnn_check_launch("linear_fwd_x3");
    }
    static const bool n64_ok = !(getenv("DGNN_X3_N64") && getenv("DGNN_X3_N64")[0] == '0');
    if (n64_ok && n_out <= ZN) {
        hipLaunchKernelGGL(k_linear_fwd_x3_n64, dim3((unsigned)dgnn_cdiv(M, XM)), dim3(256), 0, (hipStream_t)stream, A1, lda1, k1, W1, ldw1, v1, A2, lda2, k2,
                           W2, ldw2, v2, bias, scale, shift, relu, M, n_out, out, ldo, colstats);
        return dgnn_check_launch("linear_fwd_x3");
    }
    dim3 grid((unsigned)(dgnn_cdiv(dgnn_cdiv(M, XM), 8) * 8 * dgnn_cdiv(n_out, XN)));
    hipLaunchKernelGGL(k_linear_fwd_x3, grid, dim3(256), 0, (hipStream_t)stream, A1, lda1, k1, W1, ldw1, v1, A2, lda2, k2, W2, ldw2, v2, bias,
                       scale, shift, relu, M, n_out, out, ldo, colstats);
    return dgnn_check_launch("linear_fwd_x3");
}
}  // namespace

extern "C" int dgnn_linear_fwd_x3(const float* A1, int64_t lda1, int k1, const float* W1, int64_t ldw1, const float* A2, int64_t lda2, int k2,
                                  const float* W2, int64_t ldw2, const float* bias, const float* scale, const float* shift, int relu,
                                  int64_t M, int n_out, float* out, int64_t ldo, void* stream) {
    return linear_fwd_x3_impl(A1, lda1, k1, W1, ldw1, A2, lda2, k2, W2, ldw2, bias, scale, shift, relu, M, n_out, out, ldo, nullptr, stream);
}

// dgnn_linear_fwd_x3 that also leaves the fp64 column sums and sums of squares of `out` per block of 32 rows in
// colstats[ceil(M / 32)][2][n_out] (8-byte aligned) for dgnn_bn_stats_finalize_fold.  DGNN_E_UNSUPPORTED (nothing launched) for the
// shapes that take the 256 x 256 tile (n_out > 128 and at least 192 such tiles).
extern "C" int dgnn_linear_fwd_x3_stats(const float* A1, int64_t lda1, int k1, const float* W1, int64_t ldw1, const float* A2, int64_t lda2, int k2,
                                        const float* W2, int64_t ldw2, const float* bias, int64_t M, int n_out, float* out, int64_t ldo,
                                        double* colstats, void* stream) {
    DGNN_REQUIRE(colstats && ((uintptr_t)colstats & 7) == 0, DGNN_E_INVALID, "linear_fwd_x3_stats: colstats missing or unaligned");
    return linear_fwd_x3_impl(A1, lda1, k1, W1, ldw1, A2, lda2, k2, W2, ldw2, bias, nullptr, nullptr, 0, M, n_out, out, ldo, colstats, stream);
}

extern "C" int64_t dgnn_linear_fwd_x2h_scratch_elems(int64_t M, int n_out) { return M + n_out; }

// fp16 two-part form of dgnn_linear_fwd_x3 (see k_linear_fwd_x2h_big).  Covers M >= 8192 with n_out > 128 -- the wide conv layers; other shapes
// return DGNN_E_UNSUPPORTED (callers use dgnn_linear_fwd_x3).  scratch: dgnn_linear_fwd_x2h_scratch_elems(M, n_out) floats.
extern "C" int dgnn_linear_fwd_x2h(const float* A1, int64_t lda1, int k1, const float* W1, int64_t ldw1, const float* A2, int64_t lda2, int k2,
                                   const float* W2, int64_t ldw2, const float* bias, const float* scale, const float* shift, int relu,
                                   int64_t M, int n_out, float* out, int64_t ldo, float* scratch, void* stream) {
    DGNN_REQUIRE(M >= 0 && n_out > 0 && k1 > 0, DGNN_E_INVALID, "linear_fwd_x2h: bad sizes M=%lld n_out=%d k1=%d", (long long)M, n_out, k1);
    if (M == 0) return DGNN_OK;
    DGNN_REQUIRE(A1 && W1 && out && scratch, DGNN_E_INVALID, "linear_fwd_x2h: null pointer");
    DGNN_REQUIRE((A2 == nullptr) == (W2 == nullptr) && (!A2 || k2 > 0), DGNN_E_INVALID, "linear_fwd_x2h: A2/W2 must come together");
    DGNN_REQUIRE((scale == nullptr) == (shift == nullptr), DGNN_E_INVALID, "linear_fwd_x2h: scale/shift must come together");
    if (!(M >= 8192 && n_out > XN)) return DGNN_E_UNSUPPORTED;
    const bool v1 = vec_ok(A1, lda1) && vec_ok(W1, ldw1);
    const bool v2 = A2 && vec_ok(A2, lda2) && vec_ok(W2, ldw2);
    hipLaunchKernelGGL(k_x2h_row_scales, dim3((unsigned)dgnn_grid_cap(dgnn_cdiv(M + n_out, 4), 16)), dim3(256), 0, (hipStream_t)stream, A1, lda1, k1, A2,
                       lda2, k2, M, W1, ldw1, W2, ldw2, n_out, scratch);
    static bool attr_set[DGNN_MAX_DEVICES];
    constexpr size_t lds = (size_t)(YM + YN) * HLD;
    dgnn_allow_dynamic_lds((const void*)k_linear_fwd_x2h_big, lds, attr_set);
    dim3 grid((unsigned)(dgnn_cdiv(dgnn_cdiv(M, YM), 8) * 8 * dgnn_cdiv(n_out, YN)));
    hipLaunchKernelGGL(k_linear_fwd_x2h_big, grid, dim3(YT), lds, (hipStream_t)stream, A1, lda1, k1, W1, ldw1, v1, A2, lda2, k2, W2, ldw2, v2, bias,
                       scale, shift, relu, M, n_out, out, ldo, (const float*)scratch);
    return dgnn_check_launch("linear_fwd_x2h");
}

// scratch of dgnn_linear_fwd_x2hp, in floats: row scales [M + n_out] (padded to 16 bytes), then the pre-split operands
// (M + n_out) rows x (ceil(k1/32) + ceil(k2/32)) chunks x 128 bytes
extern "C" int64_t dgnn_linear_fwd_x2hp_scratch_elems(int64_t M, int n_out, int k1, int k2) {
    if (M < 0 || n_out <= 0 || k1 <= 0 || k2 < 0) return 0;
    const int64_t nch = (k1 + HK - 1) / HK + (k2 + HK - 1) / HK;
    return ((M + n_out + 3) / 4) * 4 + (M + n_out) * nch * (PCH / 4);
}

// dgnn_linear_fwd_x2h with the operands split once by a pass of their own and staged by DMA (see k_x2hp_presplit / k_linear_fwd_x2hp_big): same
// arguments, same results bit for bit, scratch = dgnn_linear_fwd_x2hp_scratch_elems(M, n_out, k1, k2) floats (16-byte aligned).
extern "C" int dgnn_linear_fwd_x2hp(const float* A1, int64_t lda1, int k1, const float* W1, int64_t ldw1, const float* A2, int64_t lda2, int k2,
                                    const float* W2, int64_t ldw2, const float* bias, const float* scale, const float* shift, int relu,
                                    int64_t M, int n_out, float* out, int64_t ldo, float* scratch, void* stream) {
    DGNN_REQUIRE(M >= 0 && n_out > 0 && k1 > 0, DGNN_E_INVALID, "linear_fwd_x2hp: bad sizes M=%lld n_out=%d k1=%d", (long long)M, n_out, k1);
    if (M == 0) return DGNN_OK;
    DGNN_REQUIRE(A1 && W1 && out && scratch && ((uintptr_t)scratch % 16) == 0, DGNN_E_INVALID, "linear_fwd_x2hp: null / unaligned pointer");
    DGNN_REQUIRE((A2 == nullptr) == (W2 == nullptr) && (!A2 || k2 > 0), DGNN_E_INVALID, "linear_fwd_x2hp: A2/W2 must come together");
    DGNN_REQUIRE((scale == nullptr) == (shift == nullptr), DGNN_E_INVALID, "linear_fwd_x2hp: scale/shift must come together");
    if (!(M >= 8192 && n_out > XN)) return DGNN_E_UNSUPPORTED;
    const int nch = (k1 + HK - 1) / HK + (A2 ? (k2 + HK - 1) / HK : 0);
    float* scales = scratch;
    char* As = reinterpret_cast<char*>(scratch + ((M + n_out + 3) / 4) * 4);
    char* Ws = As + M * (int64_t)nch * PCH;
    hipLaunchKernelGGL(k_x2hp_presplit, dim3((unsigned)dgnn_grid_cap(dgnn_cdiv(M + n_out, 4), 16)), dim3(256), 0, (hipStream_t)stream, A1, lda1, k1, A2, lda2,
                       A2 ? k2 : 0, M, W1, ldw1, W2, ldw2, n_out, scales, As, Ws);
    static bool attr_set[DGNN_MAX_DEVICES];
    constexpr size_t lds = (size_t)2 * (YM + YN) * PCH;
    dgnn_allow_dynamic_lds((const void*)k_linear_fwd_x2hp_big, lds, attr_set);
    dim3 grid((unsigned)(dgnn_cdiv(dgnn_cdiv(M, YM), 8) * 8 * dgnn_cdiv(n_out, YN)));
    hipLaunchKernelGGL(k_linear_fwd_x2hp_big, grid, dim3(YT), lds, (hipStream_t)stream, (const char*)As, (const char*)Ws, nch, bias, scale, shift, relu, M, n_out,
                       out, ldo, (const float*)scales);
    return dgnn_check_launch("linear_fwd_x2hp");
}

extern "C" int dgnn_linear_wgrad_x3(const float* A, int64_t lda, int n_a, const float* B, int64_t ldb, int n_b, int64_t M, float* dW,
                                    int64_t lddw, int accumulate, float* partials, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DGNN_REQUIRE(M >= 0 && n_a > 0 && n_b > 0, DGNN_E_INVALID, "linear_wgrad_x3: bad sizes");
    DGNN_REQUIRE(dW && partials && (M == 0 || (A && B)), DGNN_E_INVALID, "linear_wgrad_x3: null pointer");
    const int splits = wgrad_splits(M, n_a);
    const int64_t rps = dgnn_cdiv(dgnn_cdiv(M, splits), XRK) * XRK;
    dim3 grid((unsigned)dgnn_cdiv(n_a, WT), (unsigned)dgnn_cdiv(n_b, WT), splits);
    hipLaunchKernelGGL(k_linear_wgrad_x3, grid, dim3(256), 0, stream, A, lda, n_a, B, ldb, n_b, M, rps < XRK ? XRK : rps, partials);
    hipLaunchKernelGGL(k_wgrad_reduce, dim3((unsigned)dgnn_cdiv((int64_t)n_a * n_b, 16)), dim3(256), 0, stream, partials, splits, n_a, n_b, dW, lddw,
                       accumulate);
    return dgnn_check_launch("linear_wgrad_x3");
}

// fp32 partials [splits][n_a][n_b1 + n_b2], then fp64 bias partials [splits][n_a] (8-byte aligned inside the float scratch)
extern "C" int64_t dgnn_linear_wgrad_cat_scratch_elems(int64_t M, int n_a, int n_b1, int n_b2) {
    if (M < 0 || n_a <= 0 || n_b1 <= 0 || n_b2 < 0) return 4;
    const int64_t splits = wgrad_splits(M, n_a);
    return splits * n_a * (n_b1 + n_b2) + 2 * splits * n_a + 4;
}

int dgnn_linear_wgrad_x3_cat_deferred(const float* A, int64_t lda, int n_a, const float* B1, int64_t ldb1, int n_b1, const float* B2, int64_t ldb2, int n_b2,
                                      int64_t M, float* dW1, float* dW2, float* dbias, float* scratch, void* stream_, WgradReduceDesc* desc) {
    hipStream_t stream = (hipStream_t)stream_;
    DGNN_REQUIRE(M >= 0 && n_a > 0 && n_b1 > 0 && n_b2 >= 0, DGNN_E_INVALID, "linear_wgrad_x3_cat: bad sizes");
    DGNN_REQUIRE(dW1 && scratch && (M == 0 || (A && B1)) && ((n_b2 == 0) == (B2 == nullptr)) && (n_b2 == 0 || dW2), DGNN_E_INVALID,
                 "linear_wgrad_x3_cat: null pointer");
    const int splits = wgrad_splits(M, n_a);
    const int64_t rps = dgnn_cdiv(dgnn_cdiv(M, splits), XRK) * XRK;
    const int nby1 = (int)dgnn_cdiv(n_b1, WT), nby2 = (int)dgnn_cdiv(n_b2, WT);
    float* partials = scratch;
    double* bias_partials = reinterpret_cast<double*>(((uintptr_t)(scratch + (int64_t)splits * n_a * (n_b1 + n_b2)) + 7) & ~(uintptr_t)7);
    WgradCat c;
    c.B[0] = B1, c.B[1] = B2, c.ldb[0] = ldb1, c.ldb[1] = ldb2, c.nb[0] = n_b1, c.nb[1] = n_b2;
    dim3 grid((unsigned)dgnn_cdiv(n_a, WT), (unsigned)(nby1 + nby2), splits);
    hipLaunchKernelGGL(k_linear_wgrad_x3_cat, grid, dim3(256), 0, stream, A, lda, n_a, c, nby1, M, rps < XRK ? XRK : rps, partials,
                       dbias ? bias_partials : nullptr);
    WgradReduceDesc d;
    d.partials = partials, d.bias_partials = bias_partials, d.splits = splits, d.na = n_a, d.nb1 = n_b1, d.nb2 = n_b2, d.dW1 = dW1, d.dW2 = dW2, d.dbias = dbias;
    if (desc)
        *desc = d;
    else
        hipLaunchKernelGGL(k_wgrad_reduce_cat, dim3((unsigned)wgrad_reduce_blocks(d)), dim3(256), 0, stream, d);
    return dgnn_check_launch("linear_wgrad_x3_cat");
}

extern "C" int dgnn_linear_wgrad_x3_cat(const float* A, int64_t lda, int n_a, const float* B1, int64_t ldb1, int n_b1, const float* B2, int64_t ldb2,
                                        int n_b2, int64_t M, float* dW1, float* dW2, float* dbias, float* scratch, void* stream) {
    return dgnn_linear_wgrad_x3_cat_deferred(A, lda, n_a, B1, ldb1, n_b1, B2, ldb2, n_b2, M, dW1, dW2, dbias, scratch, stream, nullptr);
}

// bf16-storage form: A / B1 / B2 are bf16 (`*_f32` = 0) or fp32 rounded to bf16 when staged (1); dW1 / dW2 bit-identical to dgnn_linear_wgrad_bf16
int dgnn_linear_wgrad_bf16_cat_deferred(const void* A, int a_f32, int64_t lda, int n_a, const void* B1, int64_t ldb1, int n_b1, const void* B2, int64_t ldb2,
                                        int n_b2, int b_f32, int64_t M, float* dW1, float* dW2, float* dbias, float* scratch, void* stream_,
                                        WgradReduceDesc* desc) {
    hipStream_t stream = (hipStream_t)stream_;
    DGNN_REQUIRE(M >= 0 && n_a > 0 && n_b1 > 0 && n_b2 >= 0, DGNN_E_INVALID, "linear_wgrad_bf16_cat: bad sizes");
    DGNN_REQUIRE(dW1 && scratch && (M == 0 || (A && B1)) && ((n_b2 == 0) == (B2 == nullptr)) && (n_b2 == 0 || dW2), DGNN_E_INVALID,
                 "linear_wgrad_bf16_cat: null pointer");
    const int splits = wgrad_splits(M, n_a);
    int64_t rps = dgnn_cdiv(dgnn_cdiv(M, splits), RKB) * RKB;
    if (rps < RKB) rps = RKB;
    const int nby1 = (int)dgnn_cdiv(n_b1, WT), nby2 = (int)dgnn_cdiv(n_b2, WT);
    float* partials = scratch;
    double* bias_partials = reinterpret_cast<double*>(((uintptr_t)(scratch + (int64_t)splits * n_a * (n_b1 + n_b2)) + 7) & ~(uintptr_t)7);
    WgradCatB c;
    c.B[0] = B1, c.B[1] = B2, c.ldb[0] = ldb1, c.ldb[1] = ldb2, c.nb[0] = n_b1, c.nb[1] = n_b2;
    dim3 grid((unsigned)dgnn_cdiv(n_a, WT), (unsigned)(nby1 + nby2), splits);
#define WG(TA, TB) hipLaunchKernelGGL((k_linear_wgrad_b_cat<TA, TB>), grid, dim3(256), 0, stream, (const TA*)A, lda, n_a, c, nby1, M, rps, partials, \
                                      dbias ? bias_partials : nullptr)
    if (a_f32 && b_f32) WG(float, float);
    else if (a_f32) WG(float, uint16_t);
    else if (b_f32) WG(uint16_t, float);
    else WG(uint16_t, uint16_t);
#undef WG
    WgradReduceDesc d;
    d.partials = partials, d.bias_partials = bias_partials, d.splits = splits, d.na = n_a, d.nb1 = n_b1, d.nb2 = n_b2, d.dW1 = dW1, d.dW2 = dW2, d.dbias = dbias;
    if (desc)
        *desc = d;
    else
        hipLaunchKernelGGL(k_wgrad_reduce_cat, dim3((unsigned)wgrad_reduce_blocks(d)), dim3(256), 0, stream, d);
    return dgnn_check_launch("linear_wgrad_bf16_cat");
}

extern "C" int dgnn_linear_wgrad_bf16_cat(const void* A, int a_f32, int64_t lda, int n_a, const void* B1, int64_t ldb1, int n_b1, const void* B2, int64_t ldb2,
                                          int n_b2, int b_f32, int64_t M, float* dW1, float* dW2, float* dbias, float* scratch, void* stream) {
    return dgnn_linear_wgrad_bf16_cat_deferred(A, a_f32, lda, n_a, B1, ldb1, n_b1, B2, ldb2, n_b2, b_f32, M, dW1, dW2, dbias, scratch, stream, nullptr);
}
