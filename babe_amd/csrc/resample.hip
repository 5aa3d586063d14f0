// Time-axis x1/2 and x2 resamplers of the CQTDiff+ UNet as depth-wise 8-tap polyphase FIRs, plus
// their exact adjoints.  Reference: /root/reference/networks/cqtdiff+.py:549-580 (UpDownResample,
// 'cubic' kernel :513-515, reflect padding) which runs them as dense FxFx8 convs with a diagonal
// weight.  Index forms (checked against the reference in tests/golden/blocks.npz):
//   down : y[n]  = sum_k h[k] x[refl(2n+k-3)]                      n in [0,T/2)
//   up   : y[n]  = sum_{k: (n+7-k) even} h[k] x[refl((n+7-k)/2-2)]  n in [0,2T)
// HBM-bound: one thread per output sample, rows are contiguous in T.
#include "common.h"
#include "../../include/babe_hip.h"
#include "prof.h"

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

__device__ __forceinline__ float resample_one(const float* __restrict__ x, int n, int T, int mode) {
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
    return s;
}

// One thread = V consecutive outputs of one row; (row, position) come from a flat index over the F*Tout outputs of a
// (b, c) plane, so the short rows of the deep UNet levels (T = 64) still fill whole workgroups.  grid: (blocks, B*C)
// VI = 1: source rows are 16-byte aligned (T % 4 == 0, aligned base and strides): the interior fast paths read their
// register window with 16- / 8-byte loads (4 instead of 14, 3 instead of 6 load instructions per 4 outputs).
template <int V, int VI>
__global__ __launch_bounds__(256) void resample_kernel(const float* __restrict__ in, long in_bs, long in_cs,
                                                       float* __restrict__ out, long out_bs, long out_cs, int C, int F,
                                                       int T, int mode, float alpha, float beta, int tout_shift,
                                                       const float* res, long res_bs, long res_cs) {
    const int b = blockIdx.y / C, c = blockIdx.y % C;
    const int Tin = (mode == 0 || mode == 1) ? T : (mode == 2 ? T / 2 : 2 * T);
    const int Tout = (mode == 0) ? T / 2 : (mode == 1 ? 2 * T : T);
    // (row, position) from the flat index: 32-bit, and a shift when the row length is a power of two (every level of the 44.1 /
    // 22.05 kHz UNets) - the 64-bit division this replaces was most of the kernel's instructions (0.31 of the HBM peak, round 3)
    const unsigned i = (blockIdx.x * blockDim.x + threadIdx.x) * (unsigned)V;
    if (i >= (unsigned)(F * Tout)) return;
    const int f = tout_shift >= 0 ? (int)(i >> tout_shift) : (int)(i / (unsigned)Tout);
    const int n0 = (int)(i - (unsigned)f * (unsigned)Tout);
    const float* x = in + (long)b * in_bs + (long)c * in_cs + (long)f * Tin;
    float* y = out + (long)b * out_bs + (long)c * out_cs + (long)f * Tout + n0;
    // beta * (what is accumulated onto): `out` itself, or a separate tensor `res` (babe_resample_res: saves the copy res -> out)
    const float* ry = res ? res + (long)b * res_bs + (long)c * res_cs + (long)f * Tout + n0 : y;
    float s[V];
    if constexpr (V == 4) {
        // The 4 outputs share one register window of the source row.  Interior threads read it with 16- / 8-byte loads;
        // the first and last thread of a row build the same window element by element - reflected (modes 0, 1: the forward
        // resamplers pad by reflection) or zero-padded (modes 2, 3: the adjoints see a gradient that ends) - and add the few
        // taps the reflection folds back onto the first / last samples.  Same tap order as resample_one (V = 1 path), so
        // the results are identical; the deep UNet levels (T = 64, 128: 2 of 16 / 32 threads of every row are border threads,
        // i.e. every wave has some) no longer run the generic per-output loops beside the fast path (mode 2 at T = 128:
        // 1.6 TB/s, round 3).
        if (mode == 0 || mode == 3) {                       // y[n] = sum_k h[k] src[2n - 3 + k]
            float w[14];
            const int base = 2 * n0 - 3;
            const bool inner = base >= 0 && base + 13 <= Tin - 1;
            if (inner) {
                if (VI && base - 1 >= 0 && base + 14 <= Tin - 1) {
                    typedef float f32x4 __attribute__((ext_vector_type(4)));
                    const f32x4* p4 = reinterpret_cast<const f32x4*>(x + base - 1);        // n0 % 4 == 0: 16-byte aligned
                    const f32x4 q0 = p4[0], q1 = p4[1], q2 = p4[2], q3 = p4[3];
                    const float X[16] = {q0[0], q0[1], q0[2], q0[3], q1[0], q1[1], q1[2], q1[3],
                                         q2[0], q2[1], q2[2], q2[3], q3[0], q3[1], q3[2], q3[3]};
#pragma unroll
                    for (int j = 0; j < 14; ++j) w[j] = X[j + 1];
                } else {
#pragma unroll
                    for (int j = 0; j < 14; ++j) w[j] = x[base + j];
                }
            } else {
#pragma unroll
                for (int j = 0; j < 14; ++j) {
                    const int idx = base + j;
                    w[j] = mode == 0 ? x[refl(idx, T)] : ((idx >= 0 && idx < Tin) ? x[idx] : 0.f);
                }
            }
            float raw[4];
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                float acc = 0.f;
#pragma unroll
                for (int k = 0; k < 8; ++k) acc += kH[k] * w[2 * v + k];
                raw[v] = acc;
            }
            if (mode == 3) {
                // n = 1, 2: + up_gpad(2 - n);  n = T-3, T-2: + up_gpad(2T - n)   (w[j] = gy[2 n0 - 3 + j])
                if (n0 == 0) {
                    float e = 0.f;
                    e += kH[5] * w[3];
                    e += kH[6] * w[4];
                    e += kH[7] * w[5];
                    raw[1] += e;
                    float e2 = 0.f;
                    e2 += kH[7] * w[3];
                    raw[2] += e2;
                }
                if (n0 == T - 4) {
                    float e = 0.f;
                    e += kH[0] * w[10];
                    raw[1] += e;
                    float e2 = 0.f;
                    e2 += kH[0] * w[8];
                    e2 += kH[1] * w[9];
                    e2 += kH[2] * w[10];
                    raw[2] += e2;
                }
            }
#pragma unroll
            for (int v = 0; v < 4; ++v) s[v] = alpha * raw[v];
        } else {                                            // 4-tap polyphase: even n taps (1,3,5,7), odd n taps (0,2,4,6)
            const int m = n0 >> 1;
            const int len = mode == 1 ? T : T / 2;          // source row length
            const bool inner = m - 2 >= 0 && m + 3 <= len - 1;
            float w[6];
            if (inner) {
                if (VI) {
                    typedef float f32x2 __attribute__((ext_vector_type(2)));
                    const f32x2* p2 = reinterpret_cast<const f32x2*>(x + m - 2);           // m even: 8-byte aligned
                    const f32x2 q0 = p2[0], q1 = p2[1], q2 = p2[2];
                    w[0] = q0[0], w[1] = q0[1], w[2] = q1[0], w[3] = q1[1], w[4] = q2[0], w[5] = q2[1];
                } else {
#pragma unroll
                    for (int j = 0; j < 6; ++j) w[j] = x[m - 2 + j];     // w[j] = src[m - 2 + j]
                }
            } else {
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    const int idx = m - 2 + j;
                    w[j] = mode == 1 ? x[refl(idx, T)] : ((idx >= 0 && idx < len) ? x[idx] : 0.f);
                }
            }
            float raw[4];
            raw[0] = kH[1] * w[3] + kH[3] * w[2] + kH[5] * w[1] + kH[7] * w[0];
            raw[1] = kH[0] * w[4] + kH[2] * w[3] + kH[4] * w[2] + kH[6] * w[1];
            raw[2] = kH[1] * w[4] + kH[3] * w[3] + kH[5] * w[2] + kH[7] * w[1];
            raw[3] = kH[0] * w[5] + kH[2] * w[4] + kH[4] * w[3] + kH[6] * w[2];
            if (mode == 2) {
                // n = 1, 2, 3: + down_gpad(3 - n);  n = T-4 .. T-2: + down_gpad(2T + 1 - n)   (w[j] = gy[m - 2 + j])
                if (n0 == 0) {
                    float e = 0.f;
                    e += kH[0] * w[3];
                    e += kH[2] * w[2];
                    raw[1] += e;                             // gpad[2] = h0 gy[1] + h2 gy[0]
                    float e2 = 0.f;
                    e2 += kH[1] * w[2];
                    raw[2] += e2;                            // gpad[1] = h1 gy[0]
                    float e3 = 0.f;
                    e3 += kH[0] * w[2];
                    raw[3] += e3;                            // gpad[0] = h0 gy[0]
                }
                if (n0 == T - 4) {
                    float e = 0.f;
                    e += kH[7] * w[3];
                    raw[0] += e;                             // gpad[T+5] = h7 gy[Th-1]
                    float e2 = 0.f;
                    e2 += kH[6] * w[3];
                    raw[1] += e2;                            // gpad[T+4] = h6 gy[Th-1]
                    float e3 = 0.f;
                    e3 += kH[5] * w[3];
                    e3 += kH[7] * w[2];
                    raw[2] += e3;                            // gpad[T+3] = h5 gy[Th-1] + h7 gy[Th-2]
                }
            }
#pragma unroll
            for (int v = 0; v < 4; ++v) s[v] = alpha * raw[v];
        }
    } else {
#pragma unroll
        for (int v = 0; v < V; ++v) s[v] = alpha * resample_one(x, n0 + v, T, mode);
    }
    if constexpr (V == 4) {
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        f32x4 o = {s[0], s[1], s[2], s[3]};
        if (beta != 0.f) o += beta * *reinterpret_cast<const f32x4*>(ry);
        *reinterpret_cast<f32x4*>(y) = o;
    } else {
        y[0] = (beta != 0.f) ? s[0] + beta * ry[0] : s[0];
    }
}
}  // namespace

static int resample_launch(const float* in, long in_bs, long in_cs, float* out, long out_bs, long out_cs, const float* res,
                           long res_bs, long res_cs, int B, int C, int F, int T, int mode, float alpha, float beta, void* stream) {
    BABE_CHECK_ARG(in && out && B > 0 && C > 0 && F > 0, "resample: bad arguments");
    BABE_CHECK_ARG(!res || ((((uintptr_t)res & 15) == 0) && res_bs % 4 == 0 && res_cs % 4 == 0 && (((uintptr_t)out & 15) == 0) &&
                            out_bs % 4 == 0 && out_cs % 4 == 0),
                   "resample_res: res / out must be 16-byte aligned views");
    BABE_CHECK_ARG(mode >= 0 && mode <= 3, "resample: bad mode %d", mode);
    BABE_CHECK_ARG(T >= 8 && (T % 2) == 0, "resample: T=%d unsupported (need even T >= 8)", T);
    BABE_CHECK_ARG((long)B * C <= 65535, "resample: grid too large");
    const int Tout = (mode == 0) ? T / 2 : (mode == 1 ? 2 * T : T);
    // 4 outputs per thread (16-byte stores) when the output rows keep 16-byte alignment
    const bool v4 = (Tout % 4 == 0) && (((uintptr_t)out & 15) == 0) && (out_bs % 4 == 0) && (out_cs % 4 == 0);
    const long total = (long)F * Tout;
    BABE_CHECK_ARG(total < (1L << 31), "resample: plane too large");
    int tsh = -1;
    for (int k = 0; k < 31; ++k)
        if ((1 << k) == Tout) tsh = k;
    BabeProfScope prof(BABE_SLOT_RESAMPLE, 4.0 * B * C * (double)F * ((mode == 0 || mode == 1 ? T : (mode == 2 ? T / 2 : 2 * T)) + (beta != 0.f ? 2 : 1) * (double)Tout), 0, 0, stream);
    const int Tin = (mode == 0 || mode == 1) ? T : (mode == 2 ? T / 2 : 2 * T);
    const bool vin = (Tin % 4 == 0) && (((uintptr_t)in & 15) == 0) && (in_bs % 4 == 0) && (in_cs % 4 == 0);
    if (v4 && vin)
        hipLaunchKernelGGL((resample_kernel<4, 1>), dim3(cdiv(total / 4, 256), B * C), dim3(256), 0, (hipStream_t)stream, in,
                           in_bs, in_cs, out, out_bs, out_cs, C, F, T, mode, alpha, beta, tsh, res, res_bs, res_cs);
    else if (v4)
        hipLaunchKernelGGL((resample_kernel<4, 0>), dim3(cdiv(total / 4, 256), B * C), dim3(256), 0, (hipStream_t)stream, in,
                           in_bs, in_cs, out, out_bs, out_cs, C, F, T, mode, alpha, beta, tsh, res, res_bs, res_cs);
    else
        hipLaunchKernelGGL((resample_kernel<1, 0>), dim3(cdiv(total, 256), B * C), dim3(256), 0, (hipStream_t)stream, in,
                           in_bs, in_cs, out, out_bs, out_cs, C, F, T, mode, alpha, beta, tsh, res, res_bs, res_cs);
    BABE_LAUNCH_CHECK();
    return BABE_OK;
}

extern "C" int babe_resample(const float* in, long in_bs, long in_cs, float* out, long out_bs, long out_cs, int B,
                             int C, int F, int T, int mode, float alpha, float beta, void* stream) {
    return resample_launch(in, in_bs, in_cs, out, out_bs, out_cs, nullptr, 0, 0, B, C, F, T, mode, alpha, beta, stream);
}

extern "C" int babe_resample_res(const float* in, long in_bs, long in_cs, const float* res, long res_bs, long res_cs, float* out,
                                 long out_bs, long out_cs, int B, int C, int F, int T, int mode, float alpha, float beta,
                                 void* stream) {
    BABE_CHECK_ARG(res, "resample_res: null res");
    return resample_launch(in, in_bs, in_cs, out, out_bs, out_cs, res, res_bs, res_cs, B, C, F, T, mode, alpha, beta, stream);
}
