// Micro-benchmark: do fp32 MFMA (32x32x2) chains and fp32 VALU FMA work overlap on one SIMD of gfx950?
// Variants (768-thread blocks, 1 per CU, 256 blocks):
//   0: waves 0..3 run a dependent MFMA chain; waves 4..11 idle
//   1: waves 4..11 run independent v_fma chains; waves 0..3 idle
//   2: both
//   3: both, MFMA waves at s_setprio 3
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE, bool BF>
__global__ void __launch_bounds__(768, 3) k(float* out, int iters) {
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    if (w < 4) {
        if (MODE == 1) return;
        if (MODE == 3) __builtin_amdgcn_s_setprio(3);
        f32x16 acc;
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        float a = lane * 0.001f, b = 1.0f - lane * 0.002f;
        bf16x8 ab, bb;
        for (int i = 0; i < 8; ++i) { ab[i] = (short)(0x3f80 + lane); bb[i] = (short)(0x3f00 + i); }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                if (BF) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc, 0, 0, 0);
                else acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
            }
        }
        float s = 0.f;
        for (int i = 0; i < 16; ++i) s += acc[i];
        out[blockIdx.x * 768 + threadIdx.x] = s;
    } else {
        if (MODE == 0) return;
        float x[8];
        for (int i = 0; i < 8; ++i) x[i] = lane * 0.01f + i;
        const float m = 1.0001f, c = 0.5f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {  // 8 x 8 = 64 independent-ish FMAs per iteration
#pragma unroll
                for (int i = 0; i < 8; ++i) x[i] = __builtin_fmaf(x[i], m, c);
            }
        }
        float s = 0.f;
        for (int i = 0; i < 8; ++i) s += x[i];
        out[blockIdx.x * 768 + threadIdx.x] = s;
    }
}

template <int MODE, bool BF>
float run(float* d, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, BF>), dim3(256), dim3(768), 0, 0, d, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, BF>), dim3(256), dim3(768), 0, 0, d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    float* d; hipMalloc(&d, 256 * 768 * 4);
    const int iters = 20000;
    // per SIMD: MFMA wave: iters*16 MFMAs * 64 cyc; VALU: 2 waves * iters*64 FMAs
    float t0 = run<0, false>(d, iters), t1 = run<1, false>(d, iters), t2 = run<2, false>(d, iters), t3 = run<3, false>(d, iters);
    float b0 = run<0, true>(d, iters), b2 = run<2, true>(d, iters), b3 = run<3, true>(d, iters);
    double mf = (double)iters * 16, vf = (double)iters * 64 * 2;
    printf("mfma only  : %.3f ms  (%.1f cyc/MFMA @2.4GHz)\n", t0, t0 * 1e-3 * 2.4e9 / mf);
    printf("valu only  : %.3f ms  (%.2f cyc/FMA-instr per SIMD @2.4GHz)\n", t1, t1 * 1e-3 * 2.4e9 / vf);
    printf("both       : %.3f ms  (sum would be %.3f, max %.3f)\n", t2, t0 + t1, t0 > t1 ? t0 : t1);
    printf("both+prio  : %.3f ms\n", t3);
    printf("bf16 mfma only: %.3f ms (%.1f cyc/MFMA)\n", b0, b0 * 1e-3 * 2.4e9 / mf);
    printf("bf16 both     : %.3f ms (sum %.3f, max %.3f)\n", b2, b0 + t1, b0 > t1 ? b0 : t1);
    printf("bf16 both+prio: %.3f ms\n", b3);
    return 0;
}
