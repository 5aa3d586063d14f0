// Time-axis x1/2 and x2 resamplers of the CQTDiff+ UNet as depth-wise 8-tap polyphase FIRs, plus
// their exact adjoints.  Reference: /root/reference/networks/cqtdiff+.py:549-580 (UpDownResample,
// 'cubic' kernel :513-515, reflect padding) which runs them as dense FxFx8 convs with a diagonal
// weight.  Index forms (checked against the reference in tests/golden/blocks.npz):
//   down : y[n]  = sum_k h[k] x[refl(2n+k-3)]                      n in [0,T/2)
//   up   : y[n]  = sum_{k: (n+7-k) even} h[k] x[refl((n+7-k)/2-2)]  n in [0,2T)
// HBM-bound: one thread per output sample, rows are contiguous in T.
#include "common.h"
#include "../../include/babe_hip.h"

namespace {
__constant__ float kH[8] = {-0.01171875f, -0.03515625f, 0.11328125f, 0.43359375f,
                            0.43359375f,  0.11328125f,  -0.03515625f, -0.01171875f};

__device__ __forceinline__ int refl(int j, int T) { return j < 0 ? -j : (j >= T ? 2 * (T - 1) - j : j); }

// gradient w.r.t. the reflect-3-padded input of `down`, at padded index i (0..T+5)
__device__ __forceinline__ float down_gpad(const float* gy, int i, int Th) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int d = i - k;
        if (d >= 0 && !(d & 1) && (d >> 1) < Th) s += kH[k] * gy[d >> 1];
    }
    return s;
}
// gradient w.r.t. the reflect-2-padded input of `up`, at padded index j (0..T+3)
__device__ __forceinline__ float up_gpad(const float* gy, int j, int T2) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int n = 2 * j + k - 7;
        if (n >= 0 && n < T2) s += kH[k] * gy[n];
    }
    return s;
}

// grid: (ceil(Tout/256), F, B*C)
__global__ __launch_bounds__(256) void resample_kernel(const float* __restrict__ in, long in_bs, long in_cs,
                                                       float* __restrict__ out, long out_bs, long out_cs, int C,
                                                       int T, int mode, float alpha, float beta) {
    const int f = blockIdx.y;
    const int b = blockIdx.z / C, c = blockIdx.z % C;
    const int Tin = (mode == 0 || mode == 1) ? T : (mode == 2 ? T / 2 : 2 * T);
    const int Tout = (mode == 0) ? T / 2 : (mode == 1 ? 2 * T : T);
    const float* x = in + (long)b * in_bs + (long)c * in_cs + (long)f * Tin;
    float* y = out + (long)b * out_bs + (long)c * out_cs + (long)f * Tout;
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= Tout) return;
    float s = 0.f;
    if (mode == 0) {
#pragma unroll
        for (int k = 0; k < 8; ++k) s += kH[k] * x[refl(2 * n + k - 3, T)];
    } else if (mode == 1) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int d = n + 7 - k;
            if (!(d & 1)) {
                const int j = d >> 1;                    // index into the padded signal, 0..T+3
                if (j < T + 4) s += kH[k] * x[refl(j - 2, T)];
            }
        }
    } else if (mode == 2) {
        const int Th = T / 2;
        s = down_gpad(x, n + 3, Th);
        if (n >= 1 && n <= 3) s += down_gpad(x, 3 - n, Th);
        if (n >= T - 4 && n <= T - 2) s += down_gpad(x, 2 * T + 1 - n, Th);
    } else {
        const int T2 = 2 * T;
        s = up_gpad(x, n + 2, T2);
        if (n >= 1 && n <= 2) s += up_gpad(x, 2 - n, T2);
        if (n >= T - 3 && n <= T - 2) s += up_gpad(x, 2 * T - n, T2);
    }
    y[n] = (beta != 0.f) ? alpha * s + beta * y[n] : alpha * s;
}
}  // namespace

extern "C" int babe_resample(const float* in, long in_bs, long in_cs, float* out, long out_bs, long out_cs, int B,
                             int C, int F, int T, int mode, float alpha, float beta, void* stream) {
    BABE_CHECK_ARG(in && out && B > 0 && C > 0 && F > 0, "resample: bad arguments");
    BABE_CHECK_ARG(mode >= 0 && mode <= 3, "resample: bad mode %d", mode);
    BABE_CHECK_ARG(T >= 8 && (T % 2) == 0, "resample: T=%d unsupported (need even T >= 8)", T);
    BABE_CHECK_ARG((long)B * C <= 65535 && F <= 65535, "resample: grid too large");
    const int Tout = (mode == 0) ? T / 2 : (mode == 1 ? 2 * T : T);
    hipLaunchKernelGGL(resample_kernel, dim3(cdiv(Tout, 256), F, B * C), dim3(256), 0, (hipStream_t)stream, in, in_bs,
                       in_cs, out, out_bs, out_cs, C, T, mode, alpha, beta);
    BABE_LAUNCH_CHECK();
    return BABE_OK;
}
