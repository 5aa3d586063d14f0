// Power-of-two complex FFT held entirely in LDS (one workgroup, N <= 4096 points).
// In-place radix-2 decimation-in-time on bit-reversed input: the caller stores element i at
// bitrev(i) (free, because the CQT fold / STFT framing already scatter on load), the result is
// in natural order.  Twiddles come from a 2048-entry table exp(-2*pi*i*q/4096) in global memory
// (L1/L2 resident).  Used by the CQT band transforms (cqt.hip) and the STFT (stft.hip).
#pragma once
#include <hip/hip_runtime.h>

__device__ __forceinline__ unsigned bitrev_n(unsigned i, int log2n) { return __brev(i) >> (32 - log2n); }

// sign = -1: forward (e^{-i...}), +1: inverse (unnormalised).  All threads of the block must call.
__device__ __forceinline__ void fft_lds_inplace(float2* a, int log2n, const float2* __restrict__ tw4096, int sign) {
    const int n = 1 << log2n;
    const int half_n = n >> 1;
    for (int s = 1; s <= log2n; ++s) {
        __syncthreads();
        const int hm = 1 << (s - 1);
        const int tstep = 4096 >> s;                 // table stride: exp(-2 pi i j / 2^s) = tw[j * 4096/2^s]
        for (int k = threadIdx.x; k < half_n; k += blockDim.x) {
            const int j = k & (hm - 1);
            const int base = ((k >> (s - 1)) << s) + j;
            float2 w = tw4096[j * tstep];
            if (sign > 0) w.y = -w.y;
            const float2 u = a[base];
            const float2 x = a[base + hm];
            const float2 v = make_float2(x.x * w.x - x.y * w.y, x.x * w.y + x.y * w.x);
            a[base] = make_float2(u.x + v.x, u.y + v.y);
            a[base + hm] = make_float2(u.x - v.x, u.y - v.y);
        }
    }
    __syncthreads();
}
