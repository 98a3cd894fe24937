// Does v_cvt_pk_bf16_f32 in one wave return wrong results while another wave on the same SIMD streams bf16 MFMAs?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

__device__ inline uint32_t rn_bf16(float x) {
    uint32_t b = __builtin_bit_cast(uint32_t, x);
    b += 0x7FFFu + ((b >> 16) & 1u);
    return b >> 16;
}

template <bool WITH_MFMA>
__global__ void __launch_bounds__(512, 2) k(unsigned long long* bad, float* sink, int iters) {
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    if (w < 4) {
        if (!WITH_MFMA) return;
        f32x16 acc;
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        bf16x8 ab, bb;
        for (int i = 0; i < 8; ++i) { ab[i] = (short)(0x3f80 + lane); bb[i] = (short)(0x3f00 + i); }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 16; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc, 0, 0, 0);
        }
        float s = 0.f;
        for (int i = 0; i < 16; ++i) s += acc[i];
        sink[blockIdx.x * 512 + threadIdx.x] = s;
    } else {
        unsigned long long nbad = 0;
        uint32_t st = 1234567u * (blockIdx.x * 512 + threadIdx.x + 1);
        for (int it = 0; it < iters * 8; ++it) {
            st = st * 1664525u + 1013904223u;
            const float x0 = __builtin_bit_cast(float, (st & 0x007FFFFFu) | 0x3F000000u) * (float)(1 + (st >> 28));
            st = st * 1664525u + 1013904223u;
            const float x1 = -__builtin_bit_cast(float, (st & 0x007FFFFFu) | 0x40000000u);
            const bf16x2_t h = __builtin_convertvector(f32x2_t{x0, x1}, bf16x2_t);
            const uint32_t got = __builtin_bit_cast(uint32_t, h);
            const uint32_t want = rn_bf16(x0) | (rn_bf16(x1) << 16);
            nbad += (got != want);
        }
        if (nbad) atomicAdd(bad, nbad);
    }
}

int main() {
    unsigned long long* bad; float* sink;
    (void)hipMalloc(&bad, 8); (void)hipMalloc(&sink, 256 * 512 * 4);
    for (int with = 0; with < 2; ++with) {
        (void)hipMemset(bad, 0, 8);
        if (with) hipLaunchKernelGGL(k<true>, dim3(256), dim3(512), 0, 0, bad, sink, 20000);
        else hipLaunchKernelGGL(k<false>, dim3(256), dim3(512), 0, 0, bad, sink, 20000);
        (void)hipDeviceSynchronize();
        unsigned long long h = 0; (void)hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost);
        printf("%s MFMA neighbours: %llu wrong conversions out of %.3g\n", with ? "with" : "without", h, 256.0 * 256 * 20000 * 8);
    }
    return 0;
}
