// Fused decoder (reference learning/surfaceNetStaticEdgeFilters.py:180-187, applied :350-351):
//     logits = W3 . relu( (W0 . y + b0) * scale + shift ) + b3          y [N,K] -> [N,n_out], n_out <= 2
// One launch instead of two GEMMs and a [N,64] round trip through HBM: reads 4K B/tet, writes 4*n_out.
//
// 256-thread workgroup, tile = 64 rows.  The y tile is staged through LDS (coalesced 16-B loads, row stride
// K+4 floats -> conflict-free ds_read_b128).  The 4 waves split the hidden layer as (32-column block) x
// (K half) and run v_mfma_f32_32x32x2_f32 with their K/4 weight VGPRs resident; K halves are summed through
// LDS, bias/BatchNorm(eval)/ReLU applied, the 64-wide hidden tile parked in LDS and the final n_out x 64
// projection done by 64*n_out threads as dot products (W3 rows in registers, hidden rows broadcast).
//
// Two kernels: k_decoder_rows (default; split-bf16 matrix cores, streaming, no barrier in the loop) and the
// original fp32-MFMA tile kernel k_decoder_fused kept for rows that are not 16-byte aligned.
#include "fused_common.h"

namespace {

using fused::bf16x8;
using fused::pack8;
using fused::split3;

constexpr int HID = 64, DTILE = 64;

// ---------------------------------------------------------------------------------------------------------------
// k_decoder_rows<128>: 512 threads = 8 independent wavefronts, each streams its own 32-row tiles.
//   * y rows go global -> registers directly in the MFMA A-operand layout (lane = (row i, k-group g) reads the 32
//     contiguous bytes y[i][16S+8g .. +7]); the two K halves are separate register sets so that the next tile's
//     half is in flight while the other half is being multiplied -- no LDS staging, no barrier.
//   * W0 is split exactly into 3 bf16 parts once per workgroup and parked in LDS in B-operand order (48 KB);
//     every fp32 y value is split the same way in registers; 6 partial products (hh hm mh mm hl lh, small terms
//     first) on v_mfma_f32_32x32x16_bf16 with fp32 accumulation -- the same arithmetic as the fused conv layers.
//   * epilogue: bias / BatchNorm(eval) / ReLU on the accumulators, the 32x64 hidden tile goes to a wave-private
//     LDS strip (row stride 68 floats -> conflict-free ds_read_b128) and lane (row, o) finishes logit o of its row
//     with 64 fp32 FMAs in channel order; one 8-byte-per-row coalesced store.
// Algorithmic bytes: 512 read + 4*n_out written per row.
// ---------------------------------------------------------------------------------------------------------------
template <int K>
__global__ void __launch_bounds__(512) k_decoder_rows(const float* __restrict__ y, int64_t ldy, int64_t M,
                                                      const float* __restrict__ W0, const float* __restrict__ b0,
                                                      const float* __restrict__ scale, const float* __restrict__ shift,
                                                      const float* __restrict__ W3, const float* __restrict__ b3, int n_out,
                                                      float* __restrict__ out, int64_t ldo) {
    constexpr int NS = K / 16, LDH = HID + 4;
    extern __shared__ __attribute__((aligned(16))) char dsm[];
    uint4* const Bs = reinterpret_cast<uint4*>(dsm);                        // [3 parts][2 col blocks][NS][64 lanes] x 16 B
    float* const W3s = reinterpret_cast<float*>(dsm + 3 * 2 * NS * 64 * 16);  // [2][HID]
    float* const Hall = W3s + 2 * HID;                                      // [8 waves][32][LDH]
    const int lane = lane_id(), w = wave_id_uniform();
    const int g = lane >> 5, l31 = lane & 31;

    for (int e = threadIdx.x; e < 2 * NS * 64; e += blockDim.x) {
        const int ln = e & 63, S = (e >> 6) % NS, cblk = e / (64 * NS);
        const float* wr = W0 + (int64_t)(cblk * 32 + (ln & 31)) * K + 16 * S + 8 * (ln >> 5);
        uint32_t ph[4], pm[4], pl[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) split3(wr[2 * d], wr[2 * d + 1], ph[d], pm[d], pl[d]);
        Bs[((0 * 2 + cblk) * NS + S) * 64 + ln] = make_uint4(ph[0], ph[1], ph[2], ph[3]);
        Bs[((1 * 2 + cblk) * NS + S) * 64 + ln] = make_uint4(pm[0], pm[1], pm[2], pm[3]);
        Bs[((2 * 2 + cblk) * NS + S) * 64 + ln] = make_uint4(pl[0], pl[1], pl[2], pl[3]);
    }
    for (int e = threadIdx.x; e < 2 * HID; e += blockDim.x) W3s[e] = (e / HID) < n_out ? W3[e] : 0.f;
    float bb[2], sc[2], sh[2];
#pragma unroll
    for (int cblk = 0; cblk < 2; ++cblk) {
        const int col = cblk * 32 + l31;
        bb[cblk] = b0 ? b0[col] : 0.f;
        sc[cblk] = scale ? scale[col] : 1.f;
        sh[cblk] = scale ? shift[col] : 0.f;
    }
    const bool has_scale = scale != nullptr;
    const int po = g;  // epilogue role: lane (row l31, output g)
    const float b3v = (b3 && po < n_out) ? b3[po] : 0.f;
    float* const Hs = Hall + w * 32 * LDH;
    __syncthreads();

    const int64_t ntiles = (M + 31) / 32, stride = (int64_t)gridDim.x * 8;
    f32x4 a0[NS], a1[NS];  // a0: k-steps 0..NS/2-1, a1: the rest; two 16-byte loads per k-step
    auto row_ptr = [&](int64_t tile) {
        const int64_t r = tile * 32 + l31;
        return y + (r < M ? r : M - 1) * ldy + 8 * g;
    };
    auto load_half = [&](f32x4 (&a)[NS], const float* p, int half) {
#pragma unroll
        for (int s = 0; s < NS / 2; ++s) {
            a[2 * s] = *reinterpret_cast<const f32x4*>(p + 16 * (half * (NS / 2) + s));
            a[2 * s + 1] = *reinterpret_cast<const f32x4*>(p + 16 * (half * (NS / 2) + s) + 4);
        }
    };
    f32x16 acc[2];
    auto mul_half = [&](const f32x4 (&a)[NS], int half) {
#pragma unroll
        for (int s = 0; s < NS / 2; ++s) {
            const int S = half * (NS / 2) + s;
            uint32_t ph[4], pm[4], pl[4];
            split3(a[2 * s][0], a[2 * s][1], ph[0], pm[0], pl[0]);
            split3(a[2 * s][2], a[2 * s][3], ph[1], pm[1], pl[1]);
            split3(a[2 * s + 1][0], a[2 * s + 1][1], ph[2], pm[2], pl[2]);
            split3(a[2 * s + 1][2], a[2 * s + 1][3], ph[3], pm[3], pl[3]);
            const bf16x8 ah = pack8(ph), am = pack8(pm), al = pack8(pl);
#pragma unroll
            for (int cblk = 0; cblk < 2; ++cblk) {
                const bf16x8 bh = __builtin_bit_cast(bf16x8, Bs[((0 * 2 + cblk) * NS + S) * 64 + lane]);
                const bf16x8 bm = __builtin_bit_cast(bf16x8, Bs[((1 * 2 + cblk) * NS + S) * 64 + lane]);
                const bf16x8 bl = __builtin_bit_cast(bf16x8, Bs[((2 * 2 + cblk) * NS + S) * 64 + lane]);
                acc[cblk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[cblk], 0, 0, 0);
                acc[cblk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[cblk], 0, 0, 0);
                acc[cblk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc[cblk], 0, 0, 0);
                acc[cblk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc[cblk], 0, 0, 0);
                acc[cblk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc[cblk], 0, 0, 0);
                acc[cblk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[cblk], 0, 0, 0);
            }
        }
    };

    int64_t tile = (int64_t)blockIdx.x * 8 + w;
    if (tile < ntiles) {
        const float* p = row_ptr(tile);
        load_half(a0, p, 0);
        load_half(a1, p, 1);
    }
    for (; tile < ntiles; tile += stride) {
        const bool more = tile + stride < ntiles;
        const float* pn = row_ptr(more ? tile + stride : tile);
#pragma unroll
        for (int cblk = 0; cblk < 2; ++cblk)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[cblk][i] = 0.f;
        mul_half(a0, 0);
        if (more) load_half(a0, pn, 0);   // flies under the second half and the epilogue
        mul_half(a1, 1);
        if (more) load_half(a1, pn, 1);
        // hidden tile -> wave-private LDS strip
#pragma unroll
        for (int cblk = 0; cblk < 2; ++cblk)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[cblk][r] + bb[cblk];
                if (has_scale) v = __fmaf_rn(v, sc[cblk], sh[cblk]);
                Hs[((r & 3) + 8 * (r >> 2) + 4 * g) * LDH + cblk * 32 + l31] = fmaxf(v, 0.f);
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // wave-private strip: own writes landed, no barrier needed
        float sacc = b3v;
        const float* hr = Hs + l31 * LDH;
        const float* w3 = W3s + po * HID;
#pragma unroll
        for (int c = 0; c < HID; c += 4) {
            const f32x4 hv = *reinterpret_cast<const f32x4*>(hr + c);
            const f32x4 wv = *reinterpret_cast<const f32x4*>(w3 + c);
#pragma unroll
            for (int j = 0; j < 4; ++j) sacc = __fmaf_rn(hv[j], wv[j], sacc);
        }
        const int64_t row = tile * 32 + l31;
        if (row < M && po < n_out) out[row * ldo + po] = sacc;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // strip reads done before the next tile overwrites it
    }
}

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
    if (vec) {
        constexpr int K = 128, LDH = HID + 4;
        const size_t smem = 3 * 2 * (K / 16) * 64 * 16 + 2 * HID * 4 + 8 * 32 * LDH * 4;
        static bool attr_set[DGNN_MAX_DEVICES] = {};
        dgnn_allow_dynamic_lds(reinterpret_cast<const void*>(&k_decoder_rows<128>), smem, attr_set);
        const int64_t nt = dgnn_cdiv(M, 32);
        const int grid = (int)(dgnn_cdiv(nt, 8) < DGNN_NUM_CU ? dgnn_cdiv(nt, 8) : DGNN_NUM_CU);
        hipLaunchKernelGGL((k_decoder_rows<128>), dim3(grid), dim3(512), smem, (hipStream_t)stream, y, ldy, M, W0, b0, scale, shift,
                           W3, b3, n_out, out, ldo);
        return dgnn_check_launch("decoder_fused_fwd");
    }
    const int64_t ntiles = dgnn_cdiv(M, DTILE);
    const int grid = (int)(ntiles < 2 * DGNN_NUM_CU ? ntiles : 2 * DGNN_NUM_CU);
    hipLaunchKernelGGL((k_decoder_fused<128>), dim3(grid), dim3(256), 0, (hipStream_t)stream, y, ldy, M, W0, b0, scale, shift, W3,
                       b3, n_out, out, ldo, vec);
    return dgnn_check_launch("decoder_fused_fwd");
}
