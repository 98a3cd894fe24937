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
#include "common.h"

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
    const int64_t row0 = (int64_t)blockIdx.x * BM;
    const int col0 = blockIdx.y * BN;
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
            if (relu) v = fmaxf(v, 0.f);
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

int wgrad_splits(int64_t M) {
    int64_t s = dgnn_cdiv(M, 4 * RK);  // at least 128 rows per split
    if (s > WGRAD_SPLITS) s = WGRAD_SPLITS;
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
    dim3 grid((unsigned)dgnn_cdiv(M, BM), (unsigned)dgnn_cdiv(n_out, BN));
    hipLaunchKernelGGL(k_linear_fwd, grid, dim3(256), 0, (hipStream_t)stream, A1, lda1, k1, W1, ldw1, v1, A2, lda2, k2, W2,
                       ldw2, v2, bias, scale, shift, relu, M, n_out, out, ldo);
    return dgnn_check_launch("linear_fwd");
}

extern "C" int64_t dgnn_linear_wgrad_scratch_elems(int64_t M, int n_a, int n_b) {
    if (M < 0 || n_a <= 0 || n_b <= 0) return 1;
    return (int64_t)wgrad_splits(M) * n_a * n_b;
}

extern "C" int dgnn_linear_wgrad(const float* A, int64_t lda, int n_a, const float* B, int64_t ldb, int n_b, int64_t M,
                                 float* dW, int64_t lddw, int accumulate, float* partials, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DGNN_REQUIRE(M >= 0 && n_a > 0 && n_b > 0, DGNN_E_INVALID, "linear_wgrad: bad sizes");
    DGNN_REQUIRE(dW && partials && (M == 0 || (A && B)), DGNN_E_INVALID, "linear_wgrad: null pointer");
    const int splits = wgrad_splits(M);
    const int64_t rps = dgnn_cdiv(dgnn_cdiv(M, splits), RK) * RK;
    dim3 grid((unsigned)dgnn_cdiv(n_a, WT), (unsigned)dgnn_cdiv(n_b, WT), splits);
    hipLaunchKernelGGL(k_linear_wgrad, grid, dim3(256), 0, stream, A, lda, n_a, B, ldb, n_b, M, rps < RK ? RK : rps, partials);
    hipLaunchKernelGGL(k_wgrad_reduce, dim3((unsigned)dgnn_cdiv((int64_t)n_a * n_b, 16)), dim3(256), 0, stream, partials, splits,
                       n_a, n_b, dW, lddw, accumulate);
    return dgnn_check_launch("linear_wgrad");
}
