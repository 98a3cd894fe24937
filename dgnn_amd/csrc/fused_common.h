// Helpers shared by the fused-layer kernels (fused.hip, fused_mfma.hip).
#pragma once
#include "common.h"

// Optional phase tracing (debug): workgroup 0 stamps wall_clock64() at phase boundaries into a caller
// buffer registered with dgnn_debug_trace_buffer(); NULL (default) disables it.
extern int64_t* g_dgnn_trace_buf;
extern int64_t g_dgnn_trace_cap;

namespace fused {

constexpr int FE = 20;
constexpr int NWAVE = 8;

// slot layout: trace[(it * 12 + wave) * 8 + phase]
__device__ __forceinline__ void stamp(int64_t* trace, int64_t cap, int64_t it, int w, int phase) {
    if (trace && blockIdx.x == 0 && (threadIdx.x & 63) == 0) {
        const int64_t i = (it * 12 + w) * 8 + phase;
        if (i < cap) trace[i] = (int64_t)wall_clock64();
    }
}

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

// Exact 3-way split of two fp32 values into packed bf16 pairs: x = hi + mid + lo with each part a bf16
// (8 significant bits, same exponent range), rounding to nearest at every step (v_cvt_pk_bf16_f32).
// Residuals are exact in fp32, so the three parts carry all 24 significand bits.
__device__ __forceinline__ void split3(float x0, float x1, uint32_t& hi, uint32_t& mid, uint32_t& lo) {
    const bf16x2_t h = __builtin_convertvector(f32x2_t{x0, x1}, bf16x2_t);
    hi = __builtin_bit_cast(uint32_t, h);
    const float r0 = x0 - __builtin_bit_cast(float, hi << 16), r1 = x1 - __builtin_bit_cast(float, hi & 0xFFFF0000u);
    const bf16x2_t m = __builtin_convertvector(f32x2_t{r0, r1}, bf16x2_t);
    mid = __builtin_bit_cast(uint32_t, m);
    const float s0 = r0 - __builtin_bit_cast(float, mid << 16), s1 = r1 - __builtin_bit_cast(float, mid & 0xFFFF0000u);
    const bf16x2_t l = __builtin_convertvector(f32x2_t{s0, s1}, bf16x2_t);
    lo = __builtin_bit_cast(uint32_t, l);
}

// Row fragment load.  The caller passes an address that is valid for every lane (inactive lanes are
// clamped to channel 0), so the load is unconditional -- no exec-mask branch around it -- and inactive
// lanes are zeroed by a select at the point of use.
template <int CPL>
__device__ __forceinline__ void ld_row(float (&v)[CPL], const float* p) {
    if (CPL == 2) {
        const float2 t = *reinterpret_cast<const float2*>(p);
        v[0] = t.x;
        v[CPL - 1] = t.y;
    } else {
        v[0] = *p;
    }
}

// 16 bytes per lane, global -> LDS without a VGPR landing (LDS destination = wave-uniform base + lane*16).
// The issuing wave must cover it with s_waitcnt vmcnt before reading the strip (hipcc does not track it).
__device__ __forceinline__ void glds16(const float* g, float* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// workgroup barrier that waits for this wave's LDS traffic only: __syncthreads() would also drain vmcnt,
// i.e. the next tile's loads that are meant to fly across the barrier (LDS-DMA counts as a pending LDS write).
__device__ __forceinline__ void tile_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}


__device__ __forceinline__ void glds4(const float* g, float* lds_wave_base) {  // 4 bytes per lane (256 B per wave)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 4, 0, 0);
}

__device__ __forceinline__ bf16x8 pack8(const uint32_t (&d)[4]) {
    return __builtin_bit_cast(bf16x8, f32x4{__builtin_bit_cast(float, d[0]), __builtin_bit_cast(float, d[1]),
                                            __builtin_bit_cast(float, d[2]), __builtin_bit_cast(float, d[3])});
}

}  // namespace fused
