// Power-of-two complex FFT held entirely in LDS (one workgroup, N <= 4096 points).
// In-place decimation-in-time on bit-reversed input: the caller stores element i at
// fft_at(bitrev(i)) (free, because the CQT fold / STFT framing already scatter on load), the result is
// in natural order (element k at fft_at(k)).  Twiddles come from a 2048-entry table exp(-2*pi*i*q/4096) in global memory
// (L1/L2 resident).  Used by the CQT band transforms (cqt.hip) and the STFT (stft.hip).
#pragma once
#include <hip/hip_runtime.h>

__device__ __forceinline__ unsigned bitrev_n(unsigned i, int log2n) { return __brev(i) >> (32 - log2n); }

// LDS image: element i lives at fft_at(i) = i + i/32 (one pad slot per 32 elements).  The bit-reversed scatter of the
// callers puts consecutive lanes 2^(log2n-6) elements apart - on ONE bank without the padding (a 32- to 64-way conflict
// on every store); with it strides 1..32 are conflict-free and stride 64 is 2-way.  Arrays: float2 a[FFT_LDS_LEN(n)].
#define FFT_LDS_LEN(n) ((n) + ((n) >> 5))
__device__ __forceinline__ int fft_at(int i) { return i + (i >> 5); }

__device__ __forceinline__ float2 fft_cmul(float2 x, float2 w) {
    return make_float2(x.x * w.x - x.y * w.y, x.x * w.y + x.y * w.x);
}

// sign = -1: forward (e^{-i...}), +1: inverse (unnormalised).  All threads of the block must call.
// Two radix-2 stages are fused per pass (a thread owns the 4 elements base + {0,1,2,3}*hm that the stages s and s+1
// connect), so the data crosses LDS log2(n)/2 times with one barrier per pass; an odd log2(n) starts with a single
// radix-2 stage.
__device__ __forceinline__ void fft_lds_inplace(float2* a, int log2n, const float2* __restrict__ tw4096, int sign) {
    const int n = 1 << log2n;
    int s = 1;
    if (log2n & 1) {
        __syncthreads();
        for (int k = threadIdx.x; k < (n >> 1); k += blockDim.x) {
            const float2 u = a[fft_at(2 * k)], x = a[fft_at(2 * k + 1)];                    // stage 1: twiddle 1
            a[fft_at(2 * k)] = make_float2(u.x + x.x, u.y + x.y);
            a[fft_at(2 * k + 1)] = make_float2(u.x - x.x, u.y - x.y);
        }
        s = 2;
    }
    for (; s < log2n; s += 2) {
        __syncthreads();
        const int hm = 1 << (s - 1);
        const int t1 = 4096 >> s, t2 = 4096 >> (s + 1);                      // table strides of stages s and s+1
        for (int k = threadIdx.x; k < (n >> 2); k += blockDim.x) {
            const int j = k & (hm - 1);
            const int base = ((k >> (s - 1)) << (s + 1)) + j;
            float2 w1 = tw4096[j * t1], w2 = tw4096[j * t2], w3 = tw4096[(j + hm) * t2];
            if (sign > 0) {
                w1.y = -w1.y;
                w2.y = -w2.y;
                w3.y = -w3.y;
            }
            const int i0 = fft_at(base), i1 = fft_at(base + hm), i2 = fft_at(base + 2 * hm), i3 = fft_at(base + 3 * hm);
            const float2 x0 = a[i0], x1 = a[i1], x2 = a[i2], x3 = a[i3];
            const float2 v1 = fft_cmul(x1, w1), v3 = fft_cmul(x3, w1);
            const float2 y0 = make_float2(x0.x + v1.x, x0.y + v1.y), y1 = make_float2(x0.x - v1.x, x0.y - v1.y);
            const float2 y2 = make_float2(x2.x + v3.x, x2.y + v3.y), y3 = make_float2(x2.x - v3.x, x2.y - v3.y);
            const float2 u2 = fft_cmul(y2, w2), u3 = fft_cmul(y3, w3);
            a[i0] = make_float2(y0.x + u2.x, y0.y + u2.y);
            a[i2] = make_float2(y0.x - u2.x, y0.y - u2.y);
            a[i1] = make_float2(y1.x + u3.x, y1.y + u3.y);
            a[i3] = make_float2(y1.x - u3.x, y1.y - u3.y);
        }
    }
    __syncthreads();
}
