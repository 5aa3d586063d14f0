// (5,3) / (KH,3) dilated Conv2d with a HANDFUL of output channels (<= 4) on the vector ALU.
//
// The pyramid projections of the CQTDiff+ UNet (networks/cqtdiff+.py:676, 794: Conv2d(2 -> N, (5,3))) have 2 input
// channels; their input-VJP therefore has 2 OUTPUT channels and N = 64..256 input channels.  On the MFMA kernels that is
// one 32-row tile with 30 of 32 rows idle (140 us per launch on the F(2,3) kernel, 1.2 % of the benchmark); as a plain
// FMA loop it is 30 FMAs per input sample: each thread owns 4 consecutive time steps of one row for all output channels,
// walks the input channels and frequency taps, and reads the weights (co x ci x KH x 3, reference layout, flipped and
// transposed on the fly when computing the input-VJP) from an LDS copy by broadcast.  HBM-side it reads the input once
// (the KH row re-reads hit L2).  out = alpha * conv + rbeta * res like every conv entry point.
#include "common.h"
#include "../../include/babe_hip.h"
#include "prof.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

// CS = channel splits inside the workgroup: the 256 threads are 256/CS position quads x CS channel slices, partial sums
// combined through LDS in a fixed order.  Small planes (the deep UNet levels: 7168 quads x 256 channels) otherwise fill 28
// of 256 CUs with threads that each walk all 256 channels (189 us per launch; 42 us with CS = 8).
template <int CO, int CS>
__global__ __launch_bounds__(256) void conv_fewco_kernel(babe_conv_args a, const float* __restrict__ w, int tf) {
    extern __shared__ __attribute__((aligned(16))) float wl[];         // [ci][kh][co][4] (kw 0..2, pad)
    const int KH = a.KH;
    const int nw = a.Cin * KH * CO;
    for (int i = threadIdx.x; i < nw; i += 256) {
        const int co = i % CO, kh = (i / CO) % KH, ci = i / (CO * KH);
        float t0 = 0.f, t1 = 0.f, t2 = 0.f;
        if (co < a.Cout) {
            if (!tf) {                  // w[co][ci][kh][kw]
                const float* p = w + (((long)co * a.Cin + ci) * KH + kh) * 3;
                t0 = p[0]; t1 = p[1]; t2 = p[2];
            } else {                    // input-VJP of a conv with weights w[ci][co][kh][kw]: flipped in both axes
                const float* p = w + (((long)ci * a.Cout + co) * KH + (KH - 1 - kh)) * 3;
                t0 = p[2]; t1 = p[1]; t2 = p[0];
            }
        }
        *reinterpret_cast<f32x4*>(wl + (long)i * 4) = f32x4{t0, t1, t2, 0.f};
    }
    __syncthreads();
    constexpr int NQ = 256 / CS;
    const int q4 = a.T >> 2;
    const int pq = threadIdx.x % NQ, cs = threadIdx.x / NQ;
    const long q = (long)blockIdx.x * NQ + pq;
    const int b = blockIdx.y;
    const bool live = q < (long)a.F * q4;
    if (CS == 1 && !live) return;
    const long qq = live ? q : 0;
    const int f = (int)(qq / q4), t = (int)(qq % q4) * 4;
    const int cpc = (a.Cin + CS - 1) / CS;                   // channels per slice
    const int ci_lo = cs * cpc, ci_hi = ci_lo + cpc < a.Cin ? ci_lo + cpc : a.Cin;
    const int khc = KH >> 1;
    f32x4 acc[CO];
#pragma unroll
    for (int c = 0; c < CO; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* xb = a.in + (long)b * a.in_bs;
    const bool hl = t > 0, hr = t + 4 < a.T;
    // row offsets of the KH taps (invalid rows point at this thread's own row and get a zero weight multiplier), so that
    // the channel loop below is branch-free and keeps KH x 3 independent loads in flight per channel
    constexpr int KHM = 7;
    long roff[KHM];
    float rmul[KHM];
#pragma unroll
    for (int kh = 0; kh < KHM; ++kh) {
        const int fr = f + (kh - khc) * a.dil;
        const bool ok = kh < KH && fr >= 0 && fr < a.F;
        roff[kh] = (long)(ok ? fr : f) * a.T + t;
        rmul[kh] = ok ? 1.f : 0.f;
    }
    if (KH == 5) {
#pragma unroll 2
        for (int ci = ci_lo; ci < ci_hi; ++ci) {
            const float* pc = xb + (long)ci * a.in_cs;
            const float* wc = wl + (long)ci * 5 * CO * 4;
            f32x4 v[5];
            float l[5], r[5];
#pragma unroll
            for (int kh = 0; kh < 5; ++kh) {
                const float* p = pc + roff[kh];
                v[kh] = *reinterpret_cast<const f32x4*>(p);
                l[kh] = hl ? p[-1] : 0.f;
                r[kh] = hr ? p[4] : 0.f;
            }
#pragma unroll
            for (int kh = 0; kh < 5; ++kh) {
                const f32x4 vv = v[kh] * rmul[kh];
                const float ll = l[kh] * rmul[kh], rr = r[kh] * rmul[kh];
#pragma unroll
                for (int c = 0; c < CO; ++c) {
                    const f32x4 ww = *reinterpret_cast<const f32x4*>(wc + (kh * CO + c) * 4);
                    acc[c][0] += ww[0] * ll + ww[1] * vv[0] + ww[2] * vv[1];
                    acc[c][1] += ww[0] * vv[0] + ww[1] * vv[1] + ww[2] * vv[2];
                    acc[c][2] += ww[0] * vv[1] + ww[1] * vv[2] + ww[2] * vv[3];
                    acc[c][3] += ww[0] * vv[2] + ww[1] * vv[3] + ww[2] * rr;
                }
            }
        }
    } else {
        for (int kh = 0; kh < KH; ++kh) {
            if (rmul[kh] == 0.f) continue;
            const float* xr = xb + roff[kh];
            const float* wk = wl + (long)kh * CO * 4;
            for (int ci = ci_lo; ci < ci_hi; ++ci) {
                const float* p = xr + (long)ci * a.in_cs;
                const f32x4 v = *reinterpret_cast<const f32x4*>(p);
                const float l = hl ? p[-1] : 0.f, r = hr ? p[4] : 0.f;
                const float* wc = wk + (long)ci * KH * CO * 4;
#pragma unroll
                for (int c = 0; c < CO; ++c) {
                    const f32x4 ww = *reinterpret_cast<const f32x4*>(wc + c * 4);
                    acc[c][0] += ww[0] * l + ww[1] * v[0] + ww[2] * v[1];
                    acc[c][1] += ww[0] * v[0] + ww[1] * v[1] + ww[2] * v[2];
                    acc[c][2] += ww[0] * v[1] + ww[1] * v[2] + ww[2] * v[3];
                    acc[c][3] += ww[0] * v[2] + ww[1] * v[3] + ww[2] * r;
                }
            }
        }
    }
    if (CS > 1) {                                            // combine the channel slices: slice 0 sums 1..CS-1 in order
        __syncthreads();                                     // (all reads of the weight image are done: reuse its LDS)
        f32x4* red = reinterpret_cast<f32x4*>(wl);
        if (cs > 0) {
#pragma unroll
            for (int c = 0; c < CO; ++c) red[((cs - 1) * NQ + pq) * CO + c] = acc[c];
        }
        __syncthreads();
        if (cs > 0 || !live) return;
        for (int s2 = 0; s2 < CS - 1; ++s2)
#pragma unroll
            for (int c = 0; c < CO; ++c) acc[c] += red[(s2 * NQ + pq) * CO + c];
    }
    const long sp = (long)f * a.T + t;
#pragma unroll
    for (int c = 0; c < CO; ++c) {
        if (c < a.Cout) {
            float os = a.oscale ? a.oscale[b * a.Cout + c] : 1.f;
            f32x4 y = acc[c] * (a.alpha * os);
            if (a.res) y += a.rbeta * *reinterpret_cast<const f32x4*>(a.res + (long)b * a.res_bs + (long)c * a.res_cs + sp);
            *reinterpret_cast<f32x4*>(a.out + (long)b * a.out_bs + (long)c * a.out_cs + sp) = y;
        }
    }
}

}  // namespace

/* 1 if babe_conv2d_fewco takes this problem */
extern "C" int babe_conv2d_fewco_supported(const babe_conv_args* ap) {
    if (!ap) return 0;
    const babe_conv_args& a = *ap;
    auto al16 = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
    if (a.Cout < 1 || a.Cout > 4 || a.KW != 3 || a.KH < 1 || a.KH > 7 || a.T % 4 || a.in2 || a.in_scale) return 0;
    if (!al16(a.in) || a.in_bs % 4 || a.in_cs % 4 || !al16(a.out) || a.out_bs % 4 || a.out_cs % 4) return 0;
    if (a.res && (!al16(a.res) || a.res_bs % 4 || a.res_cs % 4)) return 0;
    if ((long)a.Cin * a.KH * 4 * 16 > 120 * 1024) return 0;          // weights must fit LDS
    return 1;
}

/* w: the conv's weights in the REFERENCE layout: [Cout][Cin][KH][3] for transpose_flip = 0; for transpose_flip = 1 the
 * weights [Cin][Cout][KH][3] of the conv whose input-VJP this is (a.Cin / a.Cout describe the op being executed) */
extern "C" int babe_conv2d_fewco(const babe_conv_args* ap, const float* w, int transpose_flip, void* stream) {
    BABE_CHECK_ARG(ap && w, "conv2d_fewco: null args");
    BABE_CHECK_ARG(babe_conv2d_fewco_supported(ap), "conv2d_fewco: unsupported problem (Cout <= 4, KW == 3, T %% 4 == 0, one source)");
    const babe_conv_args& a = *ap;
    const double flops = babe_conv_flops(a);
    BabeProfScope prof(BABE_SLOT_CONV53_FEWCO, babe_conv_bytes(a), flops, 0, stream);
    const int co = a.Cout <= 2 ? 2 : 4;
    size_t lds = (size_t)a.Cin * a.KH * co * 16;
    const long nq = (long)a.F * (a.T / 4);
    // few position quads and many channels: split the channels inside the workgroup (8 slices) for 8x the workgroups
    const bool split = nq * a.B < 256L * 1024 && a.Cin >= 64;
    if (split && lds < (size_t)7 * 32 * co * 16) lds = (size_t)7 * 32 * co * 16;      // the reduction reuses the weight image
#define FEWCO_LAUNCH(COv, CSv)                                                                                      \
    {                                                                                                               \
        static std::atomic<unsigned long long> once{0};                                                             \
        (void)babe_lds_optin(once, {reinterpret_cast<const void*>(&conv_fewco_kernel<COv, CSv>)}, 120 * 1024);      \
        hipLaunchKernelGGL((conv_fewco_kernel<COv, CSv>), dim3(cdiv(nq, 256 / CSv), a.B), dim3(256), lds,           \
                           (hipStream_t)stream, a, w, transpose_flip);                                              \
    }
    if (co == 2) {
        if (split) FEWCO_LAUNCH(2, 8) else FEWCO_LAUNCH(2, 1)
    } else {
        if (split) FEWCO_LAUNCH(4, 8) else FEWCO_LAUNCH(4, 1)
    }
#undef FEWCO_LAUNCH
    BABE_LAUNCH_CHECK();
    return BABE_OK;
}
