// Fused decoder (reference learning/surfaceNetStaticEdgeFilters.py:180-187, applied :350-351):
//     logits = W3 . relu( (W0 . y + b0) * scale + shift ) + b3          y [N,K] -> [N,n_out], n_out <= 2
// One launch instead of two GEMMs and a [N,64] round trip through HBM: reads 4K B/tet, writes 4*n_out.
//
// 256-thread workgroup, tile = 64 rows.  The y tile is staged through LDS (coalesced 16-B loads, row stride
// K+4 floats -> conflict-free ds_read_b128).  The 4 waves split the hidden layer as (32-column block) x
// (K half) and run v_mfma_f32_32x32x2_f32 with their K/4 weight VGPRs resident; K halves are summed through
// LDS, bias/BatchNorm(eval)/ReLU applied, the 64-wide hidden tile parked in LDS and the final n_out x 64
// projection done by 64*n_out threads as dot products (W3 rows in registers, hidden rows broadcast).
#include "common.h"

namespace {

constexpr int HID = 64, DTILE = 64;

template <int K>
__global__ void __launch_bounds__(256) k_decoder_fused(const float* __restrict__ y, int64_t ldy, int64_t M,
                                                       const float* __restrict__ W0, const float* __restrict__ b0,
                                                       const float* __restrict__ scale, const float* __restrict__ shift,
                                                       const float* __restrict__ W3, const float* __restrict__ b3, int n_out,
                                                       float* __restrict__ out, int64_t ldo, bool vec) {
    constexpr int LDA = K + 4, KH = K / 2, LDH = HID + 1;
    __shared__ __attribute__((aligned(16))) float As[DTILE * LDA];
    __shared__ float red[2 * 2 * 16 * 64];  // [cb][rb][16][64]
    __shared__ float Hs[DTILE * LDH];
    const int lane = lane_id(), w = wave_id_uniform();
    const int h = lane >> 5, l31 = lane & 31;
    const int cb = w & 1, kh = w >> 1;
    const int col = cb * 32 + l31;
    float wr[KH / 2];
#pragma unroll
    for (int S = 0; S < KH / 8; ++S)
#pragma unroll
        for (int j = 0; j < 4; ++j) wr[S * 4 + j] = W0[(int64_t)col * K + kh * KH + 8 * S + 4 * h + j];
    const float bb = b0 ? b0[col] : 0.f;
    const float sc = scale ? scale[col] : 1.f, sh = scale ? shift[col] : 0.f;
    // final projection: thread t < 64*n_out owns (row = t / n_out, o = t % n_out)
    const int t = threadIdx.x;
    const bool proj = t < DTILE * n_out;
    const int prow = proj ? t / n_out : 0, po = proj ? t % n_out : 0;
    float w3[HID];
#pragma unroll
    for (int c = 0; c < HID; ++c) w3[c] = W3[po * HID + c];
    const float b3v = b3 ? b3[po] : 0.f;

    const int64_t ntiles = (M + DTILE - 1) / DTILE;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t row0 = tile * DTILE;
        __syncthreads();  // previous tile's As/Hs readers are done
        // ---- stage y tile: 64 rows x K floats, 16 B per thread per pass
        constexpr int V4_PER_ROW = K / 4;
        for (int i = t; i < DTILE * V4_PER_ROW; i += 256) {
            const int r = i / V4_PER_ROW, c4 = (i - r * V4_PER_ROW) * 4;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (row0 + r < M) {
                const float* g = y + (row0 + r) * ldy + c4;
                if (vec) v = *reinterpret_cast<const f32x4*>(g);
                else { v[0] = g[0]; v[1] = g[1]; v[2] = g[2]; v[3] = g[3]; }
            }
            *reinterpret_cast<f32x4*>(As + r * LDA + c4) = v;
        }
        __syncthreads();
        // ---- hidden layer on the matrix cores: 2 row blocks x (my column block, my K half)
        f32x16 acc[2];
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[rb][i] = 0.f;
            const float* A = As + (rb * 32 + l31) * LDA + kh * KH + 4 * h;
#pragma unroll
            for (int S = 0; S < KH / 8; ++S) {
                const f32x4 av = *reinterpret_cast<const f32x4*>(A + 8 * S);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[rb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], wr[S * 4 + j], acc[rb], 0, 0, 0);
            }
        }
        if (kh == 1) {
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int r = 0; r < 16; ++r) red[((cb * 2 + rb) * 16 + r) * 64 + lane] = acc[rb][r];
        }
        __syncthreads();
        if (kh == 0) {
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = (acc[rb][r] + red[((cb * 2 + rb) * 16 + r) * 64 + lane]) + bb;
                    if (scale) v = __fmaf_rn(v, sc, sh);
                    v = fmaxf(v, 0.f);
                    Hs[(rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * LDH + col] = v;
                }
        }
        __syncthreads();
        // ---- output projection
        if (proj && row0 + prow < M) {
            const float* hr = Hs + prow * LDH;
            float s = b3v;
#pragma unroll
            for (int c = 0; c < HID; ++c) s = __fmaf_rn(hr[c], w3[c], s);
            out[(row0 + prow) * ldo + po] = s;
        }
    }
}

}  // namespace

extern "C" int dgnn_decoder_fused_fwd(const float* y, int64_t ldy, int64_t M, int k, const float* W0, const float* b0,
                                      const float* scale, const float* shift, int hidden, const float* W3, const float* b3,
                                      int n_out, float* out, int64_t ldo, void* stream) {
    DGNN_REQUIRE(M >= 0 && k > 0 && hidden > 0 && n_out > 0, DGNN_E_INVALID, "decoder_fused_fwd: bad sizes");
    if (M == 0) return DGNN_OK;
    DGNN_REQUIRE(y && W0 && W3 && out, DGNN_E_INVALID, "decoder_fused_fwd: null pointer");
    DGNN_REQUIRE((scale == nullptr) == (shift == nullptr), DGNN_E_INVALID, "decoder_fused_fwd: scale/shift must come together");
    DGNN_REQUIRE(k == 128 && hidden == HID && n_out <= 2, DGNN_E_UNSUPPORTED,
                 "decoder_fused_fwd: supports 128 -> 64 -> {1,2} (got %d -> %d -> %d); use dgnn_linear_fwd twice", k, hidden, n_out);
    const bool vec = ((uintptr_t)y % 16 == 0) && (ldy % 4 == 0);
    const int64_t ntiles = dgnn_cdiv(M, DTILE);
    const int grid = (int)(ntiles < 2 * DGNN_NUM_CU ? ntiles : 2 * DGNN_NUM_CU);
    hipLaunchKernelGGL((k_decoder_fused<128>), dim3(grid), dim3(256), 0, (hipStream_t)stream, y, ldy, M, W0, b0, scale, shift, W3,
                       b3, n_out, out, ldo, vec);
    return dgnn_check_launch("decoder_fused_fwd");
}
