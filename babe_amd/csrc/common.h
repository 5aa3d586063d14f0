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

// Opt-in for more than 64 KB of dynamic LDS.  hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE attribute of the
// kernel: `done` is a bitmask owned by the call site, bit = current device index, so a process that drives several GPUs
// sets it on each of them (one process per GPU sets it once).  Returns the first HIP error, hipSuccess otherwise.
#include <atomic>
#include <initializer_list>
static inline hipError_t babe_lds_optin(std::atomic<unsigned long long>& done, std::initializer_list<const void*> fns,
                                        int bytes) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const unsigned long long bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return hipSuccess;
    for (const void* f : fns) {
        e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e != hipSuccess) return e;
    }
    done.fetch_or(bit, std::memory_order_release);
    return hipSuccess;
}

// Channel addressing of the (optionally two-source) conv input with bases and strides pinned in scalar registers.
// Written as a per-lane select between kernarg fields (ci < split ? a.in_cs : a.in2_cs) the compiler emits a
// dependent global load of the selected field plus s_waitcnt vmcnt(0) in every K-chunk, draining the prefetch queue.
__device__ __forceinline__ long sgpr_pin(long v) {
    asm volatile("" : "+s"(v));
    return v;
}
struct ChanSrc {
    const float* p1;      // batch-b base of the first source (stays a kernarg-derived global pointer)
    long d2, cs1, cs2;    // second source as an element offset from p1; channel strides
    int split;
    __device__ __forceinline__ void init(const float* in, long bs, long cs, const float* in2, long bs2, long cs2_,
                                         int split_, int b) {
        p1 = in + (long)b * bs;
        d2 = sgpr_pin(in2 ? (long)((in2 + (long)b * bs2) - p1) : 0);
        cs1 = sgpr_pin(cs);
        cs2 = sgpr_pin(in2 ? cs2_ : cs);
        split = split_;
    }
    __device__ __forceinline__ const float* operator()(int ci) const {
        const bool first = ci < split;
        const long cs = first ? cs1 : cs2;
        const int c = first ? ci : ci - split;
        return p1 + ((first ? 0 : d2) + (long)c * cs);
    }
};

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
