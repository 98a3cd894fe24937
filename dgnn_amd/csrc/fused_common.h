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

// ---- fp16 two-way split (gemm_mode DGNN_GEMM_F16X2) ------------------------------------------------------------------
// x*s = hi + lo with hi = RN16(x*s), lo = RN16(x*s - hi): 22 significand bits in two fp16 values (the residual is exact in fp32).
// Three products hi.hi + hi.lo + lo.hi on v_mfma_f32_*_f16 with fp32 accumulation drop only lo.lo (<= 2^-22 relative) -- the
// "3xTF32" scheme (TF32 and fp16 both carry 11 significant bits) at half the matrix work and a third of the split instructions of
// the bf16 x 3 form.  fp16 has 5 exponent bits, so every operand is first multiplied by a power of two s (exact) that puts the
// largest magnitude of its scaling group (one row of an operand, or one weight matrix) into [2^14, 2^15): elements down to 2^-15 of that maximum keep all 22 bits AND split
// identically under any other s with the same property; smaller ones are quantised to 2^-24 (fp16 subnormals, honoured by the
// MFMA: tools/ubench_f16_denorm.hip), i.e. to <= 2^-38 of the group maximum.  The inverse powers of two are applied to the fp32
// accumulator (exact).
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void split2h(float x0, float x1, uint32_t& hi, uint32_t& lo) {
    const f16x2_t h = __builtin_convertvector(f32x2_t{x0, x1}, f16x2_t);  // v_cvt_pk_f16_f32 (round to nearest even)
    hi = __builtin_bit_cast(uint32_t, h);
    float r0, r1;  // x - hi in ONE instruction each (fp16 source operand widened by the mixed-precision FMA)
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hi), "v"(x0));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hi), "v"(x1));
    const f16x2_t l = __builtin_convertvector(f32x2_t{r0, r1}, f16x2_t);
    lo = __builtin_bit_cast(uint32_t, l);
}

// maxbits = bit pattern of the largest |value| of a scaling group (sign cleared).  s = 2^(141-E) moves it into [2^14, 2^15);
// inv = 1/s.  E is clamped to [14, 254]: groups below 2^-113 (all-zero rows included) come out as 0, inf/nan stay inf/nan.
__device__ __forceinline__ void pow2_scales(uint32_t maxbits, float& s, float& inv) {
    uint32_t E = maxbits >> 23;
    E = E < 14u ? 14u : (E > 254u ? 254u : E);
    s = __builtin_bit_cast(float, (268u - E) << 23);
    inv = __builtin_bit_cast(float, (E - 14u) << 23);
}
__device__ __forceinline__ uint32_t absbits(float v) { return __builtin_bit_cast(uint32_t, v) & 0x7FFFFFFFu; }
__device__ __forceinline__ uint32_t umax(uint32_t a, uint32_t b) { return a > b ? a : b; }
// maximum over each aligned group of 16 lanes, left in every lane of the group (four v_max_u32_dpp row_ror)
__device__ __forceinline__ uint32_t row16_umax(uint32_t m) {
    m = umax(m, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)m, 0x128, 0xf, 0xf, false));
    m = umax(m, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)m, 0x124, 0xf, 0xf, false));
    m = umax(m, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)m, 0x122, 0xf, 0xf, false));
    m = umax(m, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)m, 0x121, 0xf, 0xf, false));
    return m;
}
// maximum over the wavefront as a wave-uniform (SGPR) value
__device__ __forceinline__ uint32_t wave_umax(uint32_t m) {
    m = row16_umax(m);
    const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)m, 0), b = (uint32_t)__builtin_amdgcn_readlane((int)m, 16),
                   c = (uint32_t)__builtin_amdgcn_readlane((int)m, 32), d = (uint32_t)__builtin_amdgcn_readlane((int)m, 48);
    return umax(umax(a, b), umax(c, d));
}
// maximum over the four lanes l, l^16, l^32, l^48 (the k-groups of one row of a 16x16x32 A operand), left in all four:
// v_permlane16_swap / v_permlane32_swap exchange whole 16- / 32-lane rows between two registers without going through LDS
// The swap instructions on two copies of ONE value: (a, b) = {own value, value of lane l ^ 16 (l ^ 32)} -- which of the two is which depends on the
// lane's row.  Both copies and both results pass through empty asm statements: with plain copies LLVM's machine copy propagation treated the two
// results as the same register and compiled `r[0] + r[1]` of the bf16 decoder stage as `r[0] + r[0]` (round 4, ROCm 7.2; found by the parity test).
__device__ __forceinline__ void swap16_pair(uint32_t v, uint32_t& a, uint32_t& b) {
    uint32_t c = v;
    asm volatile("" : "+v"(v), "+v"(c));
    const auto r = __builtin_amdgcn_permlane16_swap(v, c, false, false);
    a = r[0];
    b = r[1];
    asm volatile("" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void swap32_pair(uint32_t v, uint32_t& a, uint32_t& b) {
    uint32_t c = v;
    asm volatile("" : "+v"(v), "+v"(c));
    const auto r = __builtin_amdgcn_permlane32_swap(v, c, false, false);
    a = r[0];
    b = r[1];
    asm volatile("" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ uint32_t cross_row_umax(uint32_t m) {
    uint32_t a, b;
    swap16_pair(m, a, b);
    m = umax(a, b);
    swap32_pair(m, a, b);
    return umax(a, b);
}
// sum over the four lanes l, l ^ 16, l ^ 32, l ^ 48, the same value (bit for bit) in all four: (own + partner16) + (the other pair's sum)
__device__ __forceinline__ float cross_row_sum(float v) {
    uint32_t a, b;
    swap16_pair(__builtin_bit_cast(uint32_t, v), a, b);
    const float s = __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
    swap32_pair(__builtin_bit_cast(uint32_t, s), a, b);
    return __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
}
// value of lane l ^ 16
__device__ __forceinline__ float partner16(float v) {
    uint32_t a, b;
    const uint32_t vb = __builtin_bit_cast(uint32_t, v);
    swap16_pair(vb, a, b);
    return __builtin_bit_cast(float, a ^ b ^ vb);     // {a, b} = {own, partner}
}
__device__ __forceinline__ f16x8 pack8h(const uint32_t (&d)[4]) {
    return __builtin_bit_cast(f16x8, f32x4{__builtin_bit_cast(float, d[0]), __builtin_bit_cast(float, d[1]),
                                           __builtin_bit_cast(float, d[2]), __builtin_bit_cast(float, d[3])});
}

}  // namespace fused
