// Does v_mfma_f32_32x32x16_f16 honour fp16 subnormal inputs on gfx950, and does the fp16 two-way split (hi + lo, 3 products) reach
// the accuracy the design assumes?  One wavefront, A[32x16] . B[16x32]; host compares against fp64.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_f16_denorm.hip -o tools/ubench_f16_denorm
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f16v __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void split2(float x0, float x1, uint32_t& hi, uint32_t& lo) {
    const h2 h = __builtin_convertvector(f2{x0, x1}, h2);
    hi = __builtin_bit_cast(uint32_t, h);
    float r0, r1;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hi), "v"(x0));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hi), "v"(x1));
    const h2 l = __builtin_convertvector(f2{r0, r1}, h2);
    lo = __builtin_bit_cast(uint32_t, l);
}
// A: [32][16] row-major, B: [16][32] row-major (k, n); lane l: row/col l&31, k = 8*(l>>5) .. +7
__global__ void k(const float* A, const float* B, float* D, uint32_t* parts) {
    const int l = threadIdx.x, rc = l & 31, kg = l >> 5;
    uint32_t ah[4], al[4], bh[4], bl[4];
    for (int d = 0; d < 4; ++d) {
        const int kk = 8 * kg + 2 * d;
        split2(A[rc * 16 + kk], A[rc * 16 + kk + 1], ah[d], al[d]);
        split2(B[kk * 32 + rc], B[(kk + 1) * 32 + rc], bh[d], bl[d]);
    }
    if (parts) { parts[l * 2] = ah[0]; parts[l * 2 + 1] = al[0]; }
    h8 AH, AL, BH, BL;
    uint32_t* p;
    p = (uint32_t*)&AH; for (int d = 0; d < 4; ++d) p[d] = ah[d];
    p = (uint32_t*)&AL; for (int d = 0; d < 4; ++d) p[d] = al[d];
    p = (uint32_t*)&BH; for (int d = 0; d < 4; ++d) p[d] = bh[d];
    p = (uint32_t*)&BL; for (int d = 0; d < 4; ++d) p[d] = bl[d];
    f16v acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(AL, BH, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(AH, BL, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(AH, BH, acc, 0, 0, 0);
    // D[row][col]: lane holds col = l&31, rows (r&3) + 8*(r>>2) + 4*(l>>5)
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * kg) * 32 + rc] = acc[r];
}

static double urand() { return (rand() + 0.5) / ((double)RAND_MAX + 1); }
static double nrand() { return sqrt(-2 * log(urand())) * cos(6.283185307179586 * urand()); }

int main() {
    float hA[512], hB[512], hD[1024], *dA, *dB, *dD;
    uint32_t hp[128], *dp;
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, sizeof hD); hipMalloc(&dp, sizeof hp);
    const char* names[] = {"normal(0,1) x normal(0,0.1)", "A ~ 1e-3 (lo parts subnormal)", "A ~ 2^-20 (hi parts subnormal) x B ~ 2^10",
                           "A ~ 1e-6 (below fp16 subnormals) x B ~ 1e3", "A lognormal sigma 4"};
    for (int t = 0; t < 5; ++t) {
        srand(1 + t);
        for (int i = 0; i < 512; ++i) {
            double a = nrand(), b = 0.1 * nrand();
            if (t == 1) a *= 1e-3;
            if (t == 2) { a *= ldexp(1.0, -20); b = ldexp(nrand(), 10); }
            if (t == 3) { a *= 1e-6; b = 1e3 * nrand(); }
            if (t == 4) a = exp(4 * nrand()) * (rand() & 1 ? 1 : -1);
            hA[i] = (float)a; hB[i] = (float)b;
        }
        hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
        k<<<1, 64>>>(dA, dB, dD, dp);
        hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost); hipMemcpy(hp, dp, sizeof hp, hipMemcpyDeviceToHost);
        double emax = 0, e32max = 0, scale = 0;
        for (int r = 0; r < 32; ++r) for (int c = 0; c < 32; ++c) {
            double ref = 0, mag = 0; float f = 0.f;
            for (int kk = 0; kk < 16; ++kk) { ref += (double)hA[r * 16 + kk] * hB[kk * 32 + c]; mag += fabs((double)hA[r * 16 + kk] * hB[kk * 32 + c]); f = fmaf(hA[r * 16 + kk], hB[kk * 32 + c], f); }
            emax = fmax(emax, fabs(hD[r * 32 + c] - ref) / mag); e32max = fmax(e32max, fabs((double)f - ref) / mag); scale = fmax(scale, mag);
        }
        printf("%-48s  max |err| / sum|a.b|: f16x2 %.3e   fp32 fmaf chain %.3e   (2^-22 = %.3e)\n", names[t], emax, e32max, ldexp(1.0, -22));
    }
    return 0;
}
