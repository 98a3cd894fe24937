// Micro-benchmark 5: which kinds of VALU instructions hide behind an MFMA of the same wave / of the sibling wave?
// One MFMA (32x32x16 bf16, 8 passes = 32 cycles) followed by NV independent VALU ops of one kind, 1 or 2 waves per SIMD.
// If a kind overlapped with the matrix pipe, time per (MFMA + NV ops) would stay ~32 cycles until NV*4 > 32.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

template <int KIND, int NV, int WAVES, bool MFMA>
__global__ void __launch_bounds__(64 * WAVES) k(float* out, int iters) {
    const int lane = threadIdx.x & 63;
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    bf16x8 ab, bb;
    for (int i = 0; i < 8; ++i) { ab[i] = (short)(0x3f80 + lane); bb[i] = (short)(0x3f00 + i); }
    float x[16];
    unsigned u[16];
    for (int i = 0; i < 16; ++i) { x[i] = lane * 0.01f + i; u[i] = lane * 77u + i; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (MFMA) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc, 0, 0, 0);
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const int i = v & 15;
                if (KIND == 0) x[i] = __builtin_fmaf(x[i], 1.0001f, 0.5f);                 // v_fma_f32
                if (KIND == 1) asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(u[i]));   // integer bit op
                if (KIND == 2) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(u[i]));        // shift
                if (KIND == 3) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %1" : "=v"(u[i]) : "v"(x[i]));
                if (KIND == 4) asm volatile("v_pk_add_f32 %0, %0, %0" : "+v"(*reinterpret_cast<f32x2*>(&x[2 * (v & 7)])));
                if (KIND == 5) asm volatile("v_mov_b32 %0, %1" : "=v"(u[i]) : "v"(u[(i + 1) & 15]));
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc[i] + x[i] + (float)u[i];
    out[blockIdx.x * 64 * WAVES + threadIdx.x] = s;
}

template <int KIND, int NV, int WAVES, bool MFMA>
double run(float* d, int iters) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<KIND, NV, WAVES, MFMA>), dim3(256), dim3(64 * WAVES), 0, 0, d, iters);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<KIND, NV, WAVES, MFMA>), dim3(256), dim3(64 * WAVES), 0, 0, d, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

template <int KIND>
void kind(const char* name, float* d) {
    const int iters = 20000;
    const double m0 = run<KIND, 0, 4, true>(d, iters);       // MFMA only
    const double v8 = run<KIND, 8, 4, false>(d, iters);      // 8 VALU only
    const double b8 = run<KIND, 8, 4, true>(d, iters);       // both, same wave
    const double v8w = run<KIND, 8, 8, false>(d, iters);     // 2 waves/SIMD VALU only
    const double b8w = run<KIND, 8, 8, true>(d, iters);      // 2 waves/SIMD both
    const double m0w = run<KIND, 0, 8, true>(d, iters);
    printf("%-18s 1 wave/SIMD: mfma %.2f  valu %.2f  both %.2f (sum %.2f)   2 waves/SIMD: mfma %.2f valu %.2f both %.2f (sum %.2f) ms\n", name, m0, v8, b8,
           m0 + v8, m0w, v8w, b8w, m0w + v8w);
}

int main() {
    float* d; (void)hipMalloc(&d, 256 * 1024 * 4);
    kind<0>("v_fma_f32", d);
    kind<1>("v_and_b32", d);
    kind<2>("v_lshlrev_b32", d);
    kind<3>("v_cvt_pk_bf16_f32", d);
    kind<4>("v_pk_add_f32", d);
    kind<5>("v_mov_b32", d);
    return 0;
}
