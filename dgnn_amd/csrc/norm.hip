// BatchNorm1d pieces and column reductions (HBM-bound streaming kernels).
//
// Reference: torch_geometric.nn.norm.BatchNorm -> torch.nn.BatchNorm1d(eps=1e-5, momentum=0.1)
// (surfaceNetStaticEdgeFilters.py:116-123, applied :218,263,305,345).
//   eval : y = x*scale + shift, scale = gamma/sqrt(running_var+eps), shift = beta - running_mean*scale
//   train: batch mean / biased batch variance normalise; running stats take momentum*(mean, unbiased var)
// Column reductions run in two deterministic stages (per-block fp64 partials, then a fixed-order sum).
#include "common.h"

namespace {

constexpr int RED_BLOCKS = 1024;

// MODE 0: (x, x*x)   MODE 1: (g, g*xhat) with g = dy*[y>0]   MODE 2: (x, 0)
template <int MODE, typename T>
__global__ void __launch_bounds__(256) k_colreduce(const T* __restrict__ x, int64_t ldx, const T* __restrict__ y,
                                                   int64_t ldy, const T* __restrict__ dy, int64_t lddy,
                                                   const float* __restrict__ mean, const float* __restrict__ var, float eps,
                                                   int relu, int64_t M, int c, int64_t rows_per_block,
                                                   double* __restrict__ partials) {
    __shared__ double red[2][4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = min(M, r0 + rows_per_block);
    for (int cb = 0; cb < c; cb += 64) {
        const int col = cb + tx;
        double s0 = 0.0, s1 = 0.0;
        if (col < c) {
            float mu = 0.f, is = 1.f;
            if (MODE == 1) {
                mu = mean[col];
                is = 1.0f / sqrtf(var[col] + eps);
            }
            for (int64_t r = r0 + ty; r < r1; r += 4) {
                if (MODE == 0) {
                    const float v = dgnn_ld(x + r * ldx + col);
                    s0 += v;
                    s1 += (double)v * v;
                } else if (MODE == 1) {
                    float g = dgnn_ld(dy + r * lddy + col);
                    if (relu && !(dgnn_ld(y + r * ldy + col) > 0.f)) g = 0.f;
                    const float xh = (dgnn_ld(x + r * ldx + col) - mu) * is;
                    s0 += g;
                    s1 += (double)g * xh;
                } else {
                    s0 += dgnn_ld(x + r * ldx + col);
                }
            }
        }
        red[0][ty][tx] = s0;
        red[1][ty][tx] = s1;
        __syncthreads();
        if (ty == 0 && col < c) {
            partials[((int64_t)blockIdx.x * 2 + 0) * c + col] = ((red[0][0][tx] + red[0][1][tx]) + red[0][2][tx]) + red[0][3][tx];
            partials[((int64_t)blockIdx.x * 2 + 1) * c + col] = ((red[1][0][tx] + red[1][1][tx]) + red[1][2][tx]) + red[1][3][tx];
        }
        __syncthreads();
    }
}

// Rows of a power-of-two number of channels (fp32: 4..256, bf16: 8..512) or a multiple of that window (round 4: 512, 1024, ...) with 16-byte aligned rows: every thread walks V = 16 / sizeof(T)
// columns with 16-byte loads (256 threads = c/V column groups x 256 V / c row lanes).  The scalar kernel above keeps one element load per
// row in flight per lane and ran the training step's 15 column reductions at 1.4-2.7 TB/s.  Same partials layout; sums are fp64, so the
// different association order is invisible after the rounding to fp32 except at exact ties.
template <typename T>
struct RowPiece {
    static constexpr int V = 16 / sizeof(T);
    alignas(16) T v[V];
    __device__ __forceinline__ void load(const T* p) { *reinterpret_cast<uint4*>(v) = *reinterpret_cast<const uint4*>(p); }
    __device__ __forceinline__ float get(int j) const { return dgnn_ld(&v[j]); }
};

// zscale / zshift (MODE 1, round 6): the ReLU mask is taken from x itself -- [fma(x, zscale, zshift) > 0], the very expression scale_shift_act stored
// y = max(., 0) of -- instead of from y: one row read less per pass (the training step's BatchNorm backward read dy, y AND z twice per layer).
template <int MODE, typename T>
__global__ void __launch_bounds__(256) k_colreduce4(const T* __restrict__ x, int64_t ldx, const T* __restrict__ y, int64_t ldy,
                                                    const T* __restrict__ dy, int64_t lddy, const float* __restrict__ mean,
                                                    const float* __restrict__ var, float eps, int relu, int64_t M, int c,
                                                    int64_t rows_per_block, double* __restrict__ partials,
                                                    const float* __restrict__ zscale = nullptr, const float* __restrict__ zshift = nullptr) {
    constexpr int V = RowPiece<T>::V;
    constexpr int CW = 64 * V;            // widest column window: 64 column groups x 4 row lanes (fp32: 256 columns, bf16: 512)
    __shared__ double red[2][256 * V];   // [quantity][row lane][column of the window], nrl * cw = 256 V
    // rows wider than the window (the reference's training widths 512 / 1024): one window = a 1 KB piece of the row per blockIdx.y
    const int cw = c < CW ? c : CW;
    const int ng = cw / V, nrl = 256 / ng;
    const int tg = threadIdx.x % ng, ty = threadIdx.x / ng;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = min(M, r0 + rows_per_block);
    {
        const int cb = blockIdx.y * cw;      // one window per blockIdx.y: a [1024 x 1024] matrix (the last layers of a small batch) still fills the chip
        const int col = cb + V * tg;
        double s0[V], s1[V];
        float mu[V], is[V], zs[V], zh[V];
        const bool zmask = MODE == 1 && relu && zscale != nullptr;
#pragma unroll
        for (int j = 0; j < V; ++j) {
            s0[j] = s1[j] = 0.0;
            mu[j] = 0.f;
            is[j] = 1.f;
            zs[j] = zh[j] = 0.f;
            if (MODE == 1) {
                mu[j] = mean[col + j];
                is[j] = 1.0f / sqrtf(var[col + j] + eps);
                if (zmask) {
                    zs[j] = zscale[col + j];
                    zh[j] = zshift[col + j];
                }
            }
        }
        for (int64_t r = r0 + ty; r < r1; r += nrl) {
            RowPiece<T> xv;
            xv.load(x + r * ldx + col);
            if (MODE == 0) {
#pragma unroll
                for (int j = 0; j < V; ++j) {
                    const float v = xv.get(j);
                    s0[j] += v;
                    s1[j] += (double)v * v;
                }
            } else if (MODE == 1) {
                RowPiece<T> gv, yv;
                gv.load(dy + r * lddy + col);
                if (relu && !zmask) yv.load(y + r * ldy + col);
#pragma unroll
                for (int j = 0; j < V; ++j) {
                    float g = gv.get(j);
                    if (zmask) {
                        if (!(__fmaf_rn(xv.get(j), zs[j], zh[j]) > 0.f)) g = 0.f;
                    } else if (relu && !(yv.get(j) > 0.f)) g = 0.f;
                    const float xh = (xv.get(j) - mu[j]) * is[j];
                    s0[j] += g;
                    s1[j] += (double)g * xh;
                }
            } else {
#pragma unroll
                for (int j = 0; j < V; ++j) s0[j] += xv.get(j);
            }
        }
#pragma unroll
        for (int j = 0; j < V; ++j) {
            red[0][ty * cw + V * tg + j] = s0[j];
            red[1][ty * cw + V * tg + j] = s1[j];
        }
        __syncthreads();
        for (int o = threadIdx.x; o < 2 * cw; o += 256) {
            const int q = o / cw, cc = o - q * cw;
            double t = 0.0;
            for (int k = 0; k < nrl; ++k) t += red[q][k * cw + cc];
            partials[((int64_t)blockIdx.x * 2 + q) * c + cb + cc] = t;
        }
    }
}

template <typename T>
inline bool colreduce4_ok(int c) {
    constexpr int V = 16 / sizeof(T);
    return c >= V && ((c <= 64 * V && (c & (c - 1)) == 0) || c % (64 * V) == 0);
}

// Deterministic two-level finalisers: 1024 threads = COLS columns x (1024 / COLS) slices; slice s sums partial blocks s, s + slices, ... in
// ascending order, then the slice sums are added in slice order.  (A single thread per column walking all ~1000 partial
// blocks was a 80-90 us dependent-load chain -- 30 % of a training step; 16 slices still left a 64-long chain, 10-20 us.)
// COLS = 16 (64 slices): the form of rounds 2-5.  COLS = 4 (256 slices), round 6, for many partial rows: a 64-channel layer's 4 400 partial rows (the
// GEMM epilogue leaves one per 32 rows of a 140k-cell block) were read by FOUR workgroups, 15 us; with 4 columns per workgroup sixteen read them.
constexpr int FIN_THREADS = 1024;
template <int COLS>
__device__ __forceinline__ bool finalize_pair(const double* __restrict__ partials, int nblk, int c, double& s, double& q) {
    constexpr int SLICES = FIN_THREADS / COLS;
    __shared__ double red[2][SLICES][COLS + 1];
    const int o = threadIdx.x % COLS, sl = threadIdx.x / COLS, col = blockIdx.x * COLS + o;
    double ps = 0.0, pq = 0.0;
    if (col < c) {
        // four partial rows requested at a time, added in ascending order (the sums' order is unchanged; a load -> add chain per row was ~600 ns
        // a row: 13 us for the 4 400 partial rows of a 140k-cell layer)
        int b = sl;
        for (; b + 3 * SLICES < nblk; b += 4 * SLICES) {
            double vs[4], vq[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                vs[u] = partials[((int64_t)(b + u * SLICES) * 2 + 0) * c + col];
                vq[u] = partials[((int64_t)(b + u * SLICES) * 2 + 1) * c + col];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                ps += vs[u];
                pq += vq[u];
            }
        }
        for (; b < nblk; b += SLICES) {
            ps += partials[((int64_t)b * 2 + 0) * c + col];
            pq += partials[((int64_t)b * 2 + 1) * c + col];
        }
    }
    red[0][sl][o] = ps;
    red[1][sl][o] = pq;
    __syncthreads();
    if (sl != 0 || col >= c) return false;
    s = 0.0;
    q = 0.0;
    for (int k = 0; k < SLICES; ++k) {
        s += red[0][k][o];
        q += red[1][k][o];
    }
    return true;
}
inline int fin_cols(int nblk) { return nblk >= 512 ? 4 : 16; }

template <int COLS>
__global__ void __launch_bounds__(FIN_THREADS) k_stats_finalize(const double* __restrict__ partials, int nblk, int64_t M, int c,
                                                        float* __restrict__ mean, float* __restrict__ var,
                                                        float* __restrict__ rmean, float* __restrict__ rvar, float momentum,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                        float* __restrict__ scale, float* __restrict__ shift, int64_t* __restrict__ nbt) {
    if (nbt && blockIdx.x == 0 && threadIdx.x == 0) *nbt += 1;      // num_batches_tracked of this BatchNorm (was a launch of its own per forward)
    double s, q;
    if (!finalize_pair<COLS>(partials, nblk, c, s, q)) return;
    const int col = blockIdx.x * COLS + (threadIdx.x % COLS);
    const double m = s / (double)M;
    double v = q / (double)M - m * m;
    if (v < 0.0) v = 0.0;
    mean[col] = (float)m;
    var[col] = (float)v;
    if (scale) {   // the fold of k_bn_fold on the values just stored (same fp32 arithmetic): one launch less per training layer
        const float invstd = 1.0f / sqrtf((float)v + eps);
        const float sc = (gamma ? gamma[col] : 1.f) * invstd;
        scale[col] = sc;
        shift[col] = (beta ? beta[col] : 0.f) - (float)m * sc;
    }
    if (rmean) rmean[col] = (1.f - momentum) * rmean[col] + momentum * (float)m;
    if (rvar) {
        const double unb = M > 1 ? v * (double)M / (double)(M - 1) : v;
        rvar[col] = (1.f - momentum) * rvar[col] + momentum * (float)unb;
    }
}

// sums[0][c] = first quantity, sums[1][c] = second (float), optional accumulate into out0/out1
template <int COLS>
__global__ void __launch_bounds__(FIN_THREADS) k_sum_finalize(const double* __restrict__ partials, int nblk, int c, float* __restrict__ out0,
                                                      float* __restrict__ out1, int accumulate, float* __restrict__ copy0,
                                                      float* __restrict__ copy1) {
    double s, q;
    if (!finalize_pair<COLS>(partials, nblk, c, s, q)) return;
    const int col = blockIdx.x * COLS + (threadIdx.x % COLS);
    if (out0) out0[col] = accumulate ? out0[col] + (float)s : (float)s;
    if (out1) out1[col] = accumulate ? out1[col] + (float)q : (float)q;
    if (copy0) copy0[col] = (float)s;  // second destination (dbeta / dgamma of the caller) instead of two memcpy launches
    if (copy1) copy1[col] = (float)q;
}

static void launch_stats_finalize(hipStream_t stream, const double* P, int nblk, int64_t M, int c, float* mean, float* var, float* rmean, float* rvar,
                                  float momentum, const float* gamma, const float* beta, float eps, float* scale, float* shift, int64_t* nbt) {
    if (fin_cols(nblk) == 4)
        hipLaunchKernelGGL(k_stats_finalize<4>, dim3((c + 3) / 4), dim3(FIN_THREADS), 0, stream, P, nblk, M, c, mean, var, rmean, rvar, momentum, gamma, beta, eps,
                           scale, shift, nbt);
    else
        hipLaunchKernelGGL(k_stats_finalize<16>, dim3((c + 15) / 16), dim3(FIN_THREADS), 0, stream, P, nblk, M, c, mean, var, rmean, rvar, momentum, gamma, beta,
                           eps, scale, shift, nbt);
}
static void launch_sum_finalize(hipStream_t stream, const double* P, int nblk, int c, float* out0, float* out1, int accumulate, float* copy0, float* copy1) {
    if (fin_cols(nblk) == 4)
        hipLaunchKernelGGL(k_sum_finalize<4>, dim3((c + 3) / 4), dim3(FIN_THREADS), 0, stream, P, nblk, c, out0, out1, accumulate, copy0, copy1);
    else
        hipLaunchKernelGGL(k_sum_finalize<16>, dim3((c + 15) / 16), dim3(FIN_THREADS), 0, stream, P, nblk, c, out0, out1, accumulate, copy0, copy1);
}

__global__ void k_bn_fold(const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ mean,
                          const float* __restrict__ var, float eps, int c, float* __restrict__ scale,
                          float* __restrict__ shift) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= c) return;
    const float invstd = 1.0f / sqrtf(var[i] + eps);
    const float s = (gamma ? gamma[i] : 1.f) * invstd;
    scale[i] = s;
    shift[i] = (beta ? beta[i] : 0.f) - mean[i] * s;
}

template <typename T>
__global__ void k_scale_shift_act(const T* __restrict__ x, int64_t ldx, const float* __restrict__ scale,
                                  const float* __restrict__ shift, int relu, int64_t M, int c, T* __restrict__ y,
                                  int64_t ldy) {
    const int64_t total = M * c;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = t / c;
        const int col = (int)(t - r * c);
        float v = __fmaf_rn(dgnn_ld(x + r * ldx + col), scale[col], shift[col]);
        if (relu) v = fmaxf(v, 0.f);
        dgnn_st(y + r * ldy + col, v);
    }
}

// dx for y = relu(bn(x)).  sums: [0][c] = sum g, [1][c] = sum g*xhat (already final floats in dbeta/dgamma)
template <typename T>
__global__ void k_bn_relu_bwd_apply(const T* __restrict__ x, int64_t ldx, const T* __restrict__ y, int64_t ldy,
                                    const T* __restrict__ dy, int64_t lddy, const float* __restrict__ gamma,
                                    const float* __restrict__ mean, const float* __restrict__ var, float eps, int train,
                                    int relu, int64_t M, int c, const float* __restrict__ sum_g,
                                    const float* __restrict__ sum_gx, T* __restrict__ dx, int64_t lddx, float invM,
                                    const float* __restrict__ zscale = nullptr, const float* __restrict__ zshift = nullptr) {
    const int64_t total = M * c;      // invM = 1 / (rows the sums were taken over): M here, the whole scene's rows when the batch spans several ranks
    const bool zmask = relu && zscale != nullptr;      // the mask from x (see k_colreduce4)
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = t / c;
        const int col = (int)(t - r * c);
        float g = dgnn_ld(dy + r * lddy + col);
        if (zmask) {
            if (!(__fmaf_rn(dgnn_ld(x + r * ldx + col), zscale[col], zshift[col]) > 0.f)) g = 0.f;
        } else if (relu && !(dgnn_ld(y + r * ldy + col) > 0.f)) g = 0.f;
        const float is = 1.0f / sqrtf(var[col] + eps);
        const float gs = (gamma ? gamma[col] : 1.f) * is;
        float o;
        if (train) {
            const float xh = (dgnn_ld(x + r * ldx + col) - mean[col]) * is;
            o = gs * (g - invM * sum_g[col] - xh * invM * sum_gx[col]);
        } else {
            o = g * gs;
        }
        dgnn_st(dx + r * lddx + col, o);
    }
}

int red_blocks(int64_t M) {
    int64_t b = dgnn_cdiv(M, 64);
    if (b > RED_BLOCKS) b = RED_BLOCKS;
    return (int)(b < 1 ? 1 : b);
}

}  // namespace

// scratch (floats): partials as doubles [blocks][2][c] -> 4*blocks*c floats, + 2*c floats of sums
extern "C" int64_t dgnn_colstats_scratch_elems(int64_t M, int c) {
    if (c <= 0) return 2;
    const int64_t own = red_blocks(M < 0 ? 0 : M), gemm = M > 0 ? (M + 31) / 32 : 0;   // partial rows of the reducing kernels | of a GEMM's epilogue
    return (own > gemm ? own : gemm) * 4 * c + 2 * c + 2;
}

static double* as_f64(float* scratch) { return reinterpret_cast<double*>(((uintptr_t)scratch + 7) & ~(uintptr_t)7); }

extern "C" int dgnn_bn_fold(const float* gamma, const float* beta, const float* mean, const float* var, float eps, int c,
                            float* scale, float* shift, void* stream) {
    DGNN_REQUIRE(c > 0 && mean && var && scale && shift, DGNN_E_INVALID, "bn_fold: bad args");
    hipLaunchKernelGGL(k_bn_fold, dim3((c + 255) / 256), dim3(256), 0, (hipStream_t)stream, gamma, beta, mean, var, eps, c, scale, shift);
    return dgnn_check_launch("bn_fold");
}

// MODE-templated launch: the vectorised kernel for the rows it covers, the general one otherwise
template <int MODE, typename T>
static bool launch_colreduce(int nblk, hipStream_t stream, const T* x, int64_t ldx, const T* y, int64_t ldy, const T* dy, int64_t lddy, const float* mean,
                             const float* var, float eps, int relu, int64_t M, int c, int64_t rpb, double* P, const float* zscale = nullptr,
                             const float* zshift = nullptr) {
    constexpr int V = 16 / sizeof(T);
    auto al = [](const void* p, int64_t ld) { return p == nullptr || ((((uintptr_t)p) & 15) == 0 && ld % V == 0); };
    if (colreduce4_ok<T>(c) && al(x, ldx) && al(y, ldy) && al(dy, lddy)) {
        hipLaunchKernelGGL((k_colreduce4<MODE, T>), dim3(nblk, c > 64 * V ? c / (64 * V) : 1), dim3(256), 0, stream, x, ldx, y, ldy, dy, lddy, mean, var, eps, relu, M, c, rpb, P,
                           zscale, zshift);
        return true;      // (took zscale / zshift: the caller's apply pass must use the same mask)
    }
    hipLaunchKernelGGL((k_colreduce<MODE, T>), dim3(nblk), dim3(256), 0, stream, x, ldx, y, ldy, dy, lddy, mean, var, eps, relu, M, c, rpb, P);
    return false;
}

template <typename T>
static int bn_batch_stats_t(const T* x, int64_t ldx, int64_t M, int c, float* mean, float* var, float* running_mean, float* running_var,
                            float momentum, float* scratch, hipStream_t stream, const float* gamma = nullptr, const float* beta = nullptr,
                            float eps = 0.f, float* scale = nullptr, float* shift = nullptr) {
    DGNN_REQUIRE(M > 0 && c > 0 && x && mean && var && scratch, DGNN_E_INVALID, "bn_batch_stats: bad args (M=%lld c=%d)", (long long)M, c);
    const int nblk = red_blocks(M);
    const int64_t rpb = dgnn_cdiv(M, nblk);
    double* P = as_f64(scratch);
    launch_colreduce<0, T>(nblk, stream, x, ldx, (const T*)nullptr, (int64_t)0, (const T*)nullptr, (int64_t)0, nullptr, nullptr, 0.f, 0, M, c, rpb, P);
    launch_stats_finalize(stream, P, nblk, M, c, mean, var, running_mean, running_var, momentum, gamma, beta, eps, scale, shift, nullptr);
    return dgnn_check_launch("bn_batch_stats");
}

template <typename T>
static int scale_shift_act_t(const T* x, int64_t ldx, const float* scale, const float* shift, int relu, int64_t M, int c, T* y, int64_t ldy,
                             hipStream_t stream) {
    DGNN_REQUIRE(M >= 0 && c > 0, DGNN_E_INVALID, "scale_shift_act: bad sizes");
    if (M == 0) return DGNN_OK;
    DGNN_REQUIRE(x && scale && shift && y, DGNN_E_INVALID, "scale_shift_act: null pointer");
    hipLaunchKernelGGL((k_scale_shift_act<T>), dim3(dgnn_grid_cap(dgnn_cdiv(M * c, 256))), dim3(256), 0, stream, x, ldx, scale, shift, relu, M,
                       c, y, ldy);
    return dgnn_check_launch("scale_shift_act");
}

template <typename T>
static int bn_relu_bwd_t(const T* x, int64_t ldx, const T* y, int64_t ldy, const T* dy, int64_t lddy, const float* gamma, const float* mean,
                         const float* var, float eps, int train, int relu, int64_t M, int c, T* dx, int64_t lddx, float* dgamma, float* dbeta,
                         float* scratch, hipStream_t stream, const float* zscale = nullptr, const float* zshift = nullptr) {
    DGNN_REQUIRE(M > 0 && c > 0 && x && dy && mean && var && dx && scratch && (!relu || y), DGNN_E_INVALID, "bn_relu_bwd: bad args");
    const int nblk = red_blocks(M);
    const int64_t rpb = dgnn_cdiv(M, nblk);
    double* P = as_f64(scratch);
    float* sums = reinterpret_cast<float*>(P + (int64_t)nblk * 2 * c);
    const bool zm = launch_colreduce<1, T>(nblk, stream, x, ldx, y, ldy, dy, lddy, mean, var, eps, relu, M, c, rpb, P, zscale, zshift) && zscale != nullptr;
    launch_sum_finalize(stream, P, nblk, c, sums, sums + c, 0, dbeta, dgamma);
    hipLaunchKernelGGL((k_bn_relu_bwd_apply<T>), dim3(dgnn_grid_cap(dgnn_cdiv(M * c, 256))), dim3(256), 0, stream, x, ldx, y, ldy, dy, lddy,
                       gamma, mean, var, eps, train, relu, M, c, sums, sums + c, dx, lddx, 1.0f / (float)M, zm ? zscale : nullptr, zm ? zshift : nullptr);
    return dgnn_check_launch("bn_relu_bwd");
}

// The two halves of bn_relu_bwd_t for batch statistics that span several ranks (a scene cut across GPUs, dgnn_amd/partition.py): the local sums
// [sum g | sum g * x_hat] (g = dy behind the ReLU mask) -- which the host all-reduces --, then dx from sums over `count` rows.
static int bn_relu_bwd_sums_f32(const float* x, int64_t ldx, const float* y, int64_t ldy, const float* dy, int64_t lddy, const float* mean, const float* var,
                                float eps, int relu, int64_t M, int c, float* sums, float* scratch, hipStream_t stream) {
    DGNN_REQUIRE(M > 0 && c > 0 && x && dy && mean && var && sums && scratch && (!relu || y), DGNN_E_INVALID, "bn_relu_bwd_sums: bad args");
    const int nblk = red_blocks(M);
    const int64_t rpb = dgnn_cdiv(M, nblk);
    double* P = as_f64(scratch);
    launch_colreduce<1, float>(nblk, stream, x, ldx, y, ldy, dy, lddy, mean, var, eps, relu, M, c, rpb, P);
    launch_sum_finalize(stream, P, nblk, c, sums, sums + c, 0, (float*)nullptr, (float*)nullptr);
    return dgnn_check_launch("bn_relu_bwd_sums");
}

template <typename T>
static int colsum_t(const T* x, int64_t ldx, int64_t M, int c, float* out, int accumulate, float* scratch, hipStream_t stream) {
    DGNN_REQUIRE(M > 0 && c > 0 && x && out && scratch, DGNN_E_INVALID, "colsum: bad args");
    const int nblk = red_blocks(M);
    const int64_t rpb = dgnn_cdiv(M, nblk);
    double* P = as_f64(scratch);
    launch_colreduce<2, T>(nblk, stream, x, ldx, (const T*)nullptr, (int64_t)0, (const T*)nullptr, (int64_t)0, nullptr, nullptr, 0.f, 0, M, c, rpb, P);
    launch_sum_finalize(stream, P, nblk, c, out, (float*)nullptr, accumulate, (float*)nullptr, (float*)nullptr);
    return dgnn_check_launch("colsum");
}

extern "C" int dgnn_bn_batch_stats(const float* x, int64_t ldx, int64_t M, int c, float* mean, float* var,
                                   float* running_mean, float* running_var, float momentum, float* scratch, void* stream) {
    return bn_batch_stats_t<float>(x, ldx, M, c, mean, var, running_mean, running_var, momentum, scratch, (hipStream_t)stream);
}
// batch statistics AND the scale / shift fold from them in one pass (training-mode forward): = dgnn_bn_batch_stats + dgnn_bn_fold
extern "C" int dgnn_bn_batch_stats_fold(const float* x, int64_t ldx, int64_t M, int c, float* mean, float* var, float* running_mean,
                                        float* running_var, float momentum, const float* gamma, const float* beta, float eps, float* scale,
                                        float* shift, float* scratch, void* stream) {
    DGNN_REQUIRE(scale && shift, DGNN_E_INVALID, "bn_batch_stats_fold: scale / shift missing");
    return bn_batch_stats_t<float>(x, ldx, M, c, mean, var, running_mean, running_var, momentum, scratch, (hipStream_t)stream, gamma, beta, eps, scale, shift);
}
// The finalising half of dgnn_bn_batch_stats_fold on partial sums that are already there: colstats[nblk][2][c] doubles (sums, sums of
// squares per block of rows), as dgnn_linear_fwd_x3_stats leaves them with nblk = ceil(M / 32).
extern "C" int dgnn_bn_stats_finalize_fold(const double* colstats, int64_t nblk, int64_t M, int c, float* mean, float* var, float* running_mean,
                                           float* running_var, float momentum, const float* gamma, const float* beta, float eps, float* scale,
                                           float* shift, void* stream) {
    DGNN_REQUIRE(colstats && nblk > 0 && M > 0 && c > 0 && mean && var, DGNN_E_INVALID, "bn_stats_finalize_fold: bad arguments");
    DGNN_REQUIRE((scale == nullptr) == (shift == nullptr), DGNN_E_INVALID, "bn_stats_finalize_fold: scale / shift must come together");
    launch_stats_finalize((hipStream_t)stream, colstats, (int)nblk, M, c, mean, var, running_mean, running_var, momentum, gamma, beta, eps, scale, shift, nullptr);
    return dgnn_check_launch("bn_stats_finalize_fold");
}
// ... that also counts the batch: *num_batches_tracked += 1 (NULL: no counter) -- library-internal (csrc/train.hip): the whole-model training forward
// no longer launches a kernel of its own for the counters
int dgnn_bn_stats_finalize_fold_nbt(const double* colstats, int64_t nblk, int64_t M, int c, float* mean, float* var, float* running_mean, float* running_var,
                                    float momentum, const float* gamma, const float* beta, float eps, float* scale, float* shift, int64_t* nbt, void* stream) {
    DGNN_REQUIRE(colstats && nblk > 0 && M > 0 && c > 0 && mean && var, DGNN_E_INVALID, "bn_stats_finalize_fold: bad arguments");
    DGNN_REQUIRE((scale == nullptr) == (shift == nullptr), DGNN_E_INVALID, "bn_stats_finalize_fold: scale / shift must come together");
    launch_stats_finalize((hipStream_t)stream, colstats, (int)nblk, M, c, mean, var, running_mean, running_var, momentum, gamma, beta, eps, scale, shift, nbt);
    return dgnn_check_launch("bn_stats_finalize_fold");
}
extern "C" int dgnn_bn_batch_stats_bf16(const uint16_t* x, int64_t ldx, int64_t M, int c, float* mean, float* var,
                                        float* running_mean, float* running_var, float momentum, float* scratch, void* stream) {
    return bn_batch_stats_t<uint16_t>(x, ldx, M, c, mean, var, running_mean, running_var, momentum, scratch, (hipStream_t)stream);
}
extern "C" int dgnn_scale_shift_act(const float* x, int64_t ldx, const float* scale, const float* shift, int relu, int64_t M,
                                    int c, float* y, int64_t ldy, void* stream) {
    return scale_shift_act_t<float>(x, ldx, scale, shift, relu, M, c, y, ldy, (hipStream_t)stream);
}
extern "C" int dgnn_scale_shift_act_bf16(const uint16_t* x, int64_t ldx, const float* scale, const float* shift, int relu, int64_t M,
                                         int c, uint16_t* y, int64_t ldy, void* stream) {
    return scale_shift_act_t<uint16_t>(x, ldx, scale, shift, relu, M, c, y, ldy, (hipStream_t)stream);
}
extern "C" int dgnn_bn_relu_bwd(const float* x, int64_t ldx, const float* y, int64_t ldy, const float* dy, int64_t lddy,
                                const float* gamma, const float* mean, const float* var, float eps, int train, int relu,
                                int64_t M, int c, float* dx, int64_t lddx, float* dgamma, float* dbeta, float* scratch,
                                void* stream) {
    return bn_relu_bwd_t<float>(x, ldx, y, ldy, dy, lddy, gamma, mean, var, eps, train, relu, M, c, dx, lddx, dgamma, dbeta, scratch,
                                (hipStream_t)stream);
}
// library-internal (csrc/train.hip): dgnn_bn_relu_bwd whose ReLU mask comes from x and the forward's (scale, shift) -- y is not read (see k_colreduce4)
int dgnn_bn_relu_bwd_zmask(const float* x, int64_t ldx, const float* y, int64_t ldy, const float* dy, int64_t lddy, const float* gamma, const float* mean,
                           const float* var, float eps, int train, int relu, int64_t M, int c, float* dx, int64_t lddx, float* dgamma, float* dbeta,
                           float* scratch, const float* zscale, const float* zshift, void* stream) {
    return bn_relu_bwd_t<float>(x, ldx, y, ldy, dy, lddy, gamma, mean, var, eps, train, relu, M, c, dx, lddx, dgamma, dbeta, scratch, (hipStream_t)stream, zscale,
                                zshift);
}

extern "C" int dgnn_bn_relu_bwd_sums(const float* x, int64_t ldx, const float* y, int64_t ldy, const float* dy, int64_t lddy, const float* mean,
                                     const float* var, float eps, int relu, int64_t M, int c, float* sums, float* scratch, void* stream) {
    return bn_relu_bwd_sums_f32(x, ldx, y, ldy, dy, lddy, mean, var, eps, relu, M, c, sums, scratch, (hipStream_t)stream);
}
extern "C" int dgnn_bn_relu_bwd_apply(const float* x, int64_t ldx, const float* y, int64_t ldy, const float* dy, int64_t lddy, const float* gamma,
                                      const float* mean, const float* var, float eps, int relu, int64_t M, int c, const float* sums, double count,
                                      float* dx, int64_t lddx, void* stream) {
    DGNN_REQUIRE(M >= 0 && c > 0 && count >= 1.0, DGNN_E_INVALID, "bn_relu_bwd_apply: bad sizes");
    if (M == 0) return DGNN_OK;
    DGNN_REQUIRE(x && dy && mean && var && sums && dx && (!relu || y), DGNN_E_INVALID, "bn_relu_bwd_apply: null pointer");
    hipLaunchKernelGGL((k_bn_relu_bwd_apply<float>), dim3(dgnn_grid_cap(dgnn_cdiv(M * c, 256))), dim3(256), 0, (hipStream_t)stream, x, ldx, y, ldy, dy, lddy,
                       gamma, mean, var, eps, 1, relu, M, c, sums, sums + c, dx, lddx, (float)(1.0 / count));
    return dgnn_check_launch("bn_relu_bwd_apply");
}
extern "C" int dgnn_bn_relu_bwd_bf16(const uint16_t* x, int64_t ldx, const uint16_t* y, int64_t ldy, const uint16_t* dy, int64_t lddy,
                                     const float* gamma, const float* mean, const float* var, float eps, int train, int relu,
                                     int64_t M, int c, uint16_t* dx, int64_t lddx, float* dgamma, float* dbeta, float* scratch,
                                     void* stream) {
    return bn_relu_bwd_t<uint16_t>(x, ldx, y, ldy, dy, lddy, gamma, mean, var, eps, train, relu, M, c, dx, lddx, dgamma, dbeta, scratch,
                                   (hipStream_t)stream);
}
extern "C" int dgnn_colsum(const float* x, int64_t ldx, int64_t M, int c, float* out, int accumulate, float* scratch, void* stream) {
    return colsum_t<float>(x, ldx, M, c, out, accumulate, scratch, (hipStream_t)stream);
}
extern "C" int dgnn_colsum_bf16(const uint16_t* x, int64_t ldx, int64_t M, int c, float* out, int accumulate, float* scratch, void* stream) {
    return colsum_t<uint16_t>(x, ldx, M, c, out, accumulate, scratch, (hipStream_t)stream);
}
