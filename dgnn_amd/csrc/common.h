// Shared helpers for the gfx950 kernels of libdgnn_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/dgnn_hip.h"

#define DGNN_WAVE 64

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// thread-local error text returned by dgnn_last_error_string()
void dgnn_set_error(const char* fmt, ...);

#define DGNN_REQUIRE(cond, code, ...) \
    do {                              \
        if (!(cond)) {                \
            dgnn_set_error(__VA_ARGS__); \
            return (code);            \
        }                             \
    } while (0)

static inline int dgnn_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        dgnn_set_error("%s: %s", what, hipGetErrorString(e));
        return DGNN_E_LAUNCH;
    }
    return DGNN_OK;
}

// Raises a kernel's dynamic-LDS limit once PER DEVICE (the attribute belongs to the device's code object: a process that
// addresses its GPU as cuda:<n> with n != 0, as the reference does, would otherwise run device n with the default 64 KB cap).
#define DGNN_MAX_DEVICES 64
static inline void dgnn_allow_dynamic_lds(const void* kernel, size_t bytes, bool (&done)[DGNN_MAX_DEVICES]) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= DGNN_MAX_DEVICES) {
        (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        return;
    }
    if (!__atomic_load_n(&done[dev], __ATOMIC_ACQUIRE)) {   // two threads may both set it (idempotent); neither may see `done` before the attribute is set
        (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        __atomic_store_n(&done[dev], true, __ATOMIC_RELEASE);
    }
}

// Asynchronous device-side error word (pinned host memory mapped into every GPU).  Kernels that meet data no launch-time
// check can see -- an edge_index entry outside [0, n) -- stay memory-safe (the edge is skipped) and OR a DGNN_ASYNC_* bit
// into it; dgnn_poll_async_error() (called by the Python binding after every entry point, and by anyone after a stream
// sync) turns it into DGNN_E_INDEX.  Like HIP's own asynchronous errors it surfaces at a later call, not at the launch.
#define DGNN_ASYNC_KEY_RANGE 1   /* plan_build: sort-key endpoint outside [0, n_key) */
#define DGNN_ASYNC_OTHER_RANGE 2 /* plan_build: other endpoint outside [0, n_other) */
#define DGNN_ASYNC_DUPLICATE 4   /* edge_chain: an edge id occurs twice in e_id_cur / e_id_next (the one-to-one maps of chain.hip need unique ids) */
int32_t* dgnn_async_flag_dev();  // device-visible pointer, NULL if pinned memory is unavailable (then errors are only skipped)
__device__ __forceinline__ void dgnn_raise_async(int32_t* flag, int32_t bits) {
    if (flag) atomicOr(flag, bits);
}

static inline int64_t dgnn_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// MI355X: 256 CUs in 8 XCDs.  Grid-stride kernels cap their grid at 8 blocks per CU.
#define DGNN_NUM_CU 256
static inline int dgnn_grid_cap(int64_t blocks, int per_cu = 8) {
    int64_t cap = (int64_t)DGNN_NUM_CU * per_cu;
    return (int)(blocks < cap ? (blocks < 1 ? 1 : blocks) : cap);
}

__device__ __forceinline__ int wave_id_uniform() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

// ---- element access for the two storage types: float (fp32) and uint16_t (bf16 bit patterns) --------------------------------
// bf16 storage: widen on load, round to nearest even on store; everything in between is fp32 arithmetic.
__device__ __forceinline__ float dgnn_ld(const float* p) { return *p; }
__device__ __forceinline__ float dgnn_ld(const uint16_t* p) { return __builtin_bit_cast(float, (uint32_t)(*p) << 16); }
__device__ __forceinline__ void dgnn_st(float* p, float v) { *p = v; }
__device__ __forceinline__ void dgnn_st(uint16_t* p, float v) {
    typedef __bf16 dgnn_bf2 __attribute__((ext_vector_type(2)));
    typedef float dgnn_f2 __attribute__((ext_vector_type(2)));
    const dgnn_bf2 h = __builtin_convertvector(dgnn_f2{v, 0.f}, dgnn_bf2);
    *p = (uint16_t)(__builtin_bit_cast(uint32_t, h) & 0xFFFFu);
}
