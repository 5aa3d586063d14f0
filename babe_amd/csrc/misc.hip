// Small helpers: library info / errors, strided axpby (cat / slice / residual merges), Linear, RFF.
#include "common.h"
#include "../../include/babe_hip.h"
#include <cstdarg>

static thread_local char g_err[512] = "";

void babe_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* babe_last_error(void) { return g_err; }
extern "C" const char* babe_version(void) { return "babe_hip 0.1 (gfx950)"; }

namespace {
// grid: (blocks over T, F, B*C)
__global__ __launch_bounds__(256) void axpby4d_kernel(const float* __restrict__ in, long in_bs, long in_cs,
                                                      float* __restrict__ out, long out_bs, long out_cs, int C,
                                                      int T, float alpha, float beta) {
    const int f = blockIdx.y;
    const int b = blockIdx.z / C, c = blockIdx.z % C;
    const float* x = in + (long)b * in_bs + (long)c * in_cs + (long)f * T;
    float* y = out + (long)b * out_bs + (long)c * out_cs + (long)f * T;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < T; t += gridDim.x * blockDim.x) {
        const float v = alpha * x[t];
        y[t] = (beta != 0.f) ? v + beta * y[t] : v;
    }
}

// one wave per output row j, all batches
__global__ __launch_bounds__(256) void linear_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                     const float* __restrict__ bias, float* __restrict__ out, int B,
                                                     int K, int J, int relu) {
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (j >= J) return;
    for (int b = 0; b < B; ++b) {
        float s = 0.f;
        for (int k = lane; k < K; k += 64) s += x[(long)b * K + k] * W[(long)j * K + k];
        s = wave_sumf(s);
        if (lane == 0) {
            s += bias ? bias[j] : 0.f;
            if (relu) s = s > 0.f ? s : 0.f;
            out[(long)b * J + j] = s;
        }
    }
}

__global__ void rff_kernel(const float* __restrict__ cnoise, const float* __restrict__ freq, float* __restrict__ out,
                           int B, int R) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * R) return;
    const int b = i / R, r = i % R;
    // same fp32 op order as the reference: ((2*pi) * sigma) * freq
    const float tab = (6.283185307179586f * cnoise[b]) * freq[r];
    out[(long)b * 2 * R + r] = sinf(tab);
    out[(long)b * 2 * R + R + r] = cosf(tab);
}
}  // namespace

extern "C" int babe_axpby4d(const float* in, long in_bs, long in_cs, float* out, long out_bs, long out_cs, int B,
                            int C, int F, int T, float alpha, float beta, void* stream) {
    BABE_CHECK_ARG(in && out && B > 0 && C > 0 && F > 0 && T > 0, "axpby4d: bad arguments");
    BABE_CHECK_ARG((long)B * C <= 65535 && F <= 65535, "axpby4d: grid too large");
    int bx = cdiv(T, 256);
    if (bx > 16) bx = 16;
    hipLaunchKernelGGL(axpby4d_kernel, dim3(bx, F, B * C), dim3(256), 0, (hipStream_t)stream, in, in_bs, in_cs, out,
                       out_bs, out_cs, C, T, alpha, beta);
    BABE_LAUNCH_CHECK();
    return BABE_OK;
}

extern "C" int babe_linear(const float* x, const float* W, const float* bias, float* out, int B, int K, int J,
                           int relu, void* stream) {
    BABE_CHECK_ARG(x && W && out && B > 0 && K > 0 && J > 0, "linear: bad arguments");
    hipLaunchKernelGGL(linear_kernel, dim3(cdiv(J, 4)), dim3(256), 0, (hipStream_t)stream, x, W, bias, out, B, K, J,
                       relu);
    BABE_LAUNCH_CHECK();
    return BABE_OK;
}

extern "C" int babe_rff(const float* cnoise, const float* freq, float* out, int B, int R, void* stream) {
    BABE_CHECK_ARG(cnoise && freq && out && B > 0 && R > 0, "rff: bad arguments");
    hipLaunchKernelGGL(rff_kernel, dim3(cdiv(B * R, 64)), dim3(64), 0, (hipStream_t)stream, cnoise, freq, out, B, R);
    BABE_LAUNCH_CHECK();
    return BABE_OK;
}
