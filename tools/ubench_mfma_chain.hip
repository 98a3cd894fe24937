// Micro-benchmark 4: cost of DEPENDENT v_mfma_f32_16x16x32_bf16 chains (the filter phase issues 6 dependent products
// per channel block).  Variants: one in-place chain, one chain that hops registers (what hipcc emitted), 2 and 4
// interleaved independent chains; 1 or 2 waves per SIMD.  Cycles per MFMA per SIMD (4 passes = 16 cycles is the pipe).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

template <int CHAINS, bool HOP, int WAVES>
__global__ void __launch_bounds__(64 * WAVES) k(float* out, int iters) {
    const int lane = threadIdx.x & 63;
    bf16x8 ab, bb;
    for (int i = 0; i < 8; ++i) { ab[i] = (short)(0x3f80 + lane); bb[i] = (short)(0x3f00 + i); }
    f32x4 d[4];
    for (int c = 0; c < 4; ++c) d[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
        if (HOP) {
            // chain of 6 where the accumulator changes registers (C = previous D, new D)
            f32x4 t0, t1;
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %2, %3, 0\n"
                         "v_mfma_f32_16x16x32_bf16 %1, %2, %3, %0\n"
                         "v_mfma_f32_16x16x32_bf16 %0, %2, %3, %1\n"
                         "v_mfma_f32_16x16x32_bf16 %1, %2, %3, %0\n"
                         "v_mfma_f32_16x16x32_bf16 %0, %2, %3, %1\n"
                         "v_mfma_f32_16x16x32_bf16 %1, %2, %3, %0\n"
                         "s_nop 7\n s_nop 7\n"
                         : "=&v"(t0), "=&v"(t1) : "v"(ab), "v"(bb));
            d[0] += t1;
        } else {
#pragma unroll
            for (int j = 0; j < 12 / CHAINS; ++j)
#pragma unroll
                for (int c = 0; c < CHAINS; ++c) d[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, d[c], 0, 0, 0);
        }
    }
    f32x4 s = d[0] + d[1] + d[2] + d[3];
    out[blockIdx.x * 64 * WAVES + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}

template <int CHAINS, bool HOP, int WAVES>
void run(float* d, int iters) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<CHAINS, HOP, WAVES>), dim3(256), dim3(64 * WAVES), 0, 0, d, iters);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<CHAINS, HOP, WAVES>), dim3(256), dim3(64 * WAVES), 0, 0, d, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double n = (double)iters * (HOP ? 6 : 12) * (WAVES / 4);
    printf("%s chains=%d waves/SIMD=%d : %.3f ms -> %.1f cycles per MFMA per SIMD @2.4GHz\n", HOP ? "hop     " : "in-place", CHAINS, WAVES / 4, ms,
           ms * 1e-3 * 2.4e9 / n);
}

int main() {
    float* d; (void)hipMalloc(&d, 256 * 1024 * 4);
    const int iters = 20000;
    run<1, false, 4>(d, iters); run<2, false, 4>(d, iters); run<4, false, 4>(d, iters); run<1, true, 4>(d, iters);
    run<1, false, 8>(d, iters); run<2, false, 8>(d, iters); run<4, false, 8>(d, iters); run<1, true, 8>(d, iters);
    return 0;
}
