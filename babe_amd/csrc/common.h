// Shared helpers for the babe_hip C-ABI library (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cstdint>

#define BABE_OK 0
#define BABE_ERR_ARG -1
#define BABE_ERR_HIP -2
#define BABE_ERR_UNSUPPORTED -3

void babe_set_error(const char* fmt, ...);

#define BABE_CHECK_ARG(cond, ...)                      \
    do {                                               \
        if (!(cond)) {                                 \
            babe_set_error(__VA_ARGS__);               \
            return BABE_ERR_ARG;                       \
        }                                              \
    } while (0)

#define BABE_LAUNCH_CHECK()                                                       \
    do {                                                                          \
        hipError_t e__ = hipGetLastError();                                       \
        if (e__ != hipSuccess) {                                                  \
            babe_set_error("%s:%d HIP launch error: %s", __FILE__, __LINE__,      \
                           hipGetErrorString(e__));                               \
            return BABE_ERR_HIP;                                                  \
        }                                                                         \
    } while (0)

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// wave64 all-lane sum via DPP-free shuffles
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_sumf(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
