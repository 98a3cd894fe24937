// Per-scene feature standardisation on the device: what processing/data.py:444-506 does with
// sklearn.preprocessing.StandardScaler on pandas float64 frames, then toTorch() casts to float32 (:512-519).
//   mean_c, var_c (population) over the scene's rows, columns [c_first, C); scale_c = sqrt(var_c), 0 -> 1;
//   out[i,c] = float((x[i,c] - mean_c) / scale_c);  columns < c_first (the un-scaled loss-weight copy, :485-488) are cast.
// Two passes in fp64 (mean, then centred sum of squares), per-block partials summed in a fixed order.
#include "common.h"

namespace {

constexpr int ING_BLOCKS = 512;

// MODE 0: sum x.  MODE 1: sum (x - mean)^2
template <int MODE>
__global__ void __launch_bounds__(256) k_ing_colreduce(const double* __restrict__ x, int64_t ld, int64_t n, int c, int64_t rpb,
                                                       const double* __restrict__ mean, double* __restrict__ partials) {
    __shared__ double red[4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int64_t r0 = (int64_t)blockIdx.x * rpb, r1 = min(n, r0 + rpb);
    for (int cb = 0; cb < c; cb += 64) {
        const int col = cb + tx;
        double s = 0.0;
        if (col < c) {
            const double mu = MODE ? mean[col] : 0.0;
            for (int64_t r = r0 + ty; r < r1; r += 4) {
                const double v = x[r * ld + col] - mu;
                s += MODE ? v * v : v;
            }
        }
        red[ty][tx] = s;
        __syncthreads();
        if (ty == 0 && col < c) partials[(int64_t)blockIdx.x * c + col] = ((red[0][tx] + red[1][tx]) + red[2][tx]) + red[3][tx];
        __syncthreads();
    }
}

__global__ void k_ing_finalize(const double* __restrict__ partials, int nblk, int64_t n, int c, int sqrt_it, double* __restrict__ out) {
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= c) return;
    double s = 0.0;
    for (int b = 0; b < nblk; ++b) s += partials[(int64_t)b * c + col];
    s /= (double)n;
    if (sqrt_it) {
        s = sqrt(s);
        if (s < 10.0 * 2.220446049250313e-16) s = 1.0;  // sklearn _handle_zeros_in_scale
    }
    out[col] = s;
}

__global__ void k_ing_apply(const double* __restrict__ x, int64_t ld, int64_t n, int c, int c_first, const double* __restrict__ mean,
                            const double* __restrict__ scale, float* __restrict__ out, int64_t ldo) {
    const int64_t total = n * c;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = t / c;
        const int col = (int)(t - r * c);
        const double v = x[r * ld + col];
        out[r * ldo + col] = col < c_first ? (float)v : (float)((v - mean[col]) / scale[col]);
    }
}

}  // namespace

// scratch: doubles: partials [ING_BLOCKS][c] + mean [c] + scale [c]
extern "C" int64_t dgnn_standardize_scratch_doubles(int c) { return (int64_t)ING_BLOCKS * c + 2 * c; }

extern "C" int dgnn_standardize_f64(const double* x, int64_t ld, int64_t n, int c, int c_first, float* out, int64_t ldo,
                                    double* scratch, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DGNN_REQUIRE(n > 0 && c > 0 && c_first >= 0 && c_first <= c && x && out && scratch, DGNN_E_INVALID, "standardize_f64: bad args");
    double* partials = scratch;
    double* mean = scratch + (int64_t)ING_BLOCKS * c;
    double* scale = mean + c;
    int nblk = (int)(dgnn_cdiv(n, 64) < ING_BLOCKS ? dgnn_cdiv(n, 64) : ING_BLOCKS);
    const int64_t rpb = dgnn_cdiv(n, nblk);
    nblk = (int)dgnn_cdiv(n, rpb);
    hipLaunchKernelGGL((k_ing_colreduce<0>), dim3(nblk), dim3(256), 0, stream, x, ld, n, c, rpb, nullptr, partials);
    hipLaunchKernelGGL(k_ing_finalize, dim3((c + 255) / 256), dim3(256), 0, stream, partials, nblk, n, c, 0, mean);
    hipLaunchKernelGGL((k_ing_colreduce<1>), dim3(nblk), dim3(256), 0, stream, x, ld, n, c, rpb, mean, partials);
    hipLaunchKernelGGL(k_ing_finalize, dim3((c + 255) / 256), dim3(256), 0, stream, partials, nblk, n, c, 1, scale);
    hipLaunchKernelGGL(k_ing_apply, dim3(dgnn_grid_cap(dgnn_cdiv(n * c, 256))), dim3(256), 0, stream, x, ld, n, c, c_first, mean, scale,
                       out, ldo);
    return dgnn_check_launch("standardize_f64");
}
