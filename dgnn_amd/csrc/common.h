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

static inline int64_t dgnn_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// MI355X: 256 CUs in 8 XCDs.  Grid-stride kernels cap their grid at 8 blocks per CU.
#define DGNN_NUM_CU 256
static inline int dgnn_grid_cap(int64_t blocks, int per_cu = 8) {
    int64_t cap = (int64_t)DGNN_NUM_CU * per_cu;
    return (int)(blocks < cap ? (blocks < 1 ? 1 : blocks) : cap);
}

__device__ __forceinline__ int wave_id_uniform() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }
