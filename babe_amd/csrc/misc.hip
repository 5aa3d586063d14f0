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
// y = alpha*x + beta*y over [B][C] planes of n = F*T contiguous floats (rows are contiguous, so a frequency sub-view
// is still one contiguous run per plane).  grid: (blocks over n/V, B*C); V = 4 when every plane keeps 16-byte alignment.
template <int V>
__global__ __launch_bounds__(256) void axpby4d_kernel(const float* __restrict__ in, long in_bs, long in_cs,
                                                      float* __restrict__ out, long out_bs, long out_cs, int C, long n,
                                                      float alpha, float beta) {
    const int b = blockIdx.y / C, c = blockIdx.y % C;
    const float* x = in + (long)b * in_bs + (long)c * in_cs;
    float* y = out + (long)b * out_bs + (long)c * out_cs;
    const long stride = (long)gridDim.x * blockDim.x * V;
    for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * V; i < n; i += stride) {
        if constexpr (V == 4) {
            typedef float f32x4 __attribute__((ext_vector_type(4)));
            f32x4 v = alpha * *reinterpret_cast<const f32x4*>(x + i);
            if (beta != 0.f) v += beta * *reinterpret_cast<const f32x4*>(y + i);
            *reinterpret_cast<f32x4*>(y + i) = v;
        } else {
            const float v = alpha * x[i];
            y[i] = (beta != 0.f) ? v + beta * y[i] : v;
        }
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
    const long n = (long)F * T;
    auto al16 = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
    const bool v4 = (n % 4 == 0) && al16(in) && al16(out) && in_bs % 4 == 0 && in_cs % 4 == 0 && out_bs % 4 == 0 &&
                    out_cs % 4 == 0;
    BABE_CHECK_ARG((long)B * C <= 65535, "axpby4d: grid too large");
    if (v4) {
        int bx = cdiv(n / 4, 256);
        if (bx > 64) bx = 64;
        hipLaunchKernelGGL(axpby4d_kernel<4>, dim3(bx, B * C), dim3(256), 0, (hipStream_t)stream, in, in_bs, in_cs, out,
                           out_bs, out_cs, C, n, alpha, beta);
    } else {
        int bx = cdiv(n, 256);
        if (bx > 64) bx = 64;
        hipLaunchKernelGGL(axpby4d_kernel<1>, dim3(bx, B * C), dim3(256), 0, (hipStream_t)stream, in, in_bs, in_cs, out,
                           out_bs, out_cs, C, n, alpha, beta);
    }
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
