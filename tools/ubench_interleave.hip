// Micro-benchmark 2: within ONE wave per SIMD, how many independent v_fma_f32 hide behind each MFMA?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

template <int NV, bool BF, int WAVES>
__global__ void __launch_bounds__(64 * WAVES) k(float* out, int iters) {
    const int lane = threadIdx.x & 63;
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    float a = lane * 0.001f, b = 1.0f - lane * 0.002f;
    bf16x8 ab, bb;
    for (int i = 0; i < 8; ++i) { ab[i] = (short)(0x3f80 + lane); bb[i] = (short)(0x3f00 + i); }
    float x[16];
    for (int i = 0; i < 16; ++i) x[i] = lane * 0.01f + i;
    const float m = 1.0001f, c = 0.5f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (BF) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc, 0, 0, 0);
            else acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
#pragma unroll
            for (int v = 0; v < NV; ++v) x[v & 15] = __builtin_fmaf(x[v & 15], m, c);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc[i] + x[i];
    out[blockIdx.x * 64 * WAVES + threadIdx.x] = s;
}

template <int NV, bool BF, int WAVES>
void run(float* d, int iters) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NV, BF, WAVES>), dim3(256), dim3(64 * WAVES), 0, 0, d, iters);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<NV, BF, WAVES>), dim3(256), dim3(64 * WAVES), 0, 0, d, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%s waves/SIMD=%d NV=%2d : %.3f ms  -> %.1f cyc per (MFMA + NV fma) per wave @2.4GHz\n", BF ? "bf16" : "f32 ", WAVES / 4, NV, ms,
           ms * 1e-3 * 2.4e9 / ((double)iters * 8));
}

int main() {
    float* d; (void)hipMalloc(&d, 256 * 1024 * 4);
    const int iters = 20000;
    run<0, false, 4>(d, iters); run<4, false, 4>(d, iters); run<8, false, 4>(d, iters); run<16, false, 4>(d, iters); run<32, false, 4>(d, iters);
    run<0, true, 4>(d, iters); run<4, true, 4>(d, iters); run<8, true, 4>(d, iters); run<16, true, 4>(d, iters);
    run<8, false, 8>(d, iters); run<16, false, 8>(d, iters); run<8, true, 8>(d, iters);
    return 0;
}
