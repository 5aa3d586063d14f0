// Constant-Q (NSGT, "oct") analysis / synthesis kernels.  Definition: oracle/nsgt.py header and
// babe_amd/cqt_plan.py; call sites in the reference: networks/cqtdiff+.py:743,841,
// testing/blind_bwe_sampler.py:156 (cqt_nsgt_pytorch.CQT_nsgt.fwd/.bwd/.apply_hpf_DC).
// HBM/latency-bound: one workgroup per (band, clip); the band's T<=4096 complex points live in LDS.
#include "common.h"
#include "fft_lds.h"
#include "../../include/babe_hip.h"
#include "prof.h"

namespace {

// grid (nbands, B), 256 threads
__global__ __launch_bounds__(512) void band_analysis_kernel(babe_cqt_bands bd, const float* __restrict__ spec,
                                                            const float* __restrict__ win) {
    __shared__ float2 a[FFT_LDS_LEN(4096)];
    const int k = blockIdx.x, b = blockIdx.y;
    const int c = bd.c[k], M = bd.M[k], woff = bd.woff[k], lt = bd.log2T[k];
    const int T = 1 << lt;
    const float* sre = spec + (long)b * 2 * bd.KX;
    const float* sim = sre + bd.KX;
    for (int i = threadIdx.x; i < FFT_LDS_LEN(T); i += blockDim.x) a[i] = make_float2(0.f, 0.f);
    __syncthreads();
    const int half = M >> 1;
    for (int mi = threadIdx.x; mi < M; mi += blockDim.x) {
        const int m = mi - half;
        int n = c + m;
        if (n < 0) n += bd.L;
        if (n >= bd.L) n -= bd.L;
        float2 v;
        if (n <= bd.L / 2) v = make_float2(sre[n], sim[n]);
        else v = make_float2(sre[bd.L - n], -sim[bd.L - n]);
        const float w = win[woff + mi];
        const unsigned pos = (unsigned)(m & (T - 1));
        a[fft_at(bitrev_n(pos, lt))] = make_float2(v.x * w, v.y * w);
    }
    fft_lds_inplace(a, lt, reinterpret_cast<const float2*>(bd.tw4096), +1);
    float* out = bd.coef[bd.oct[k]] + ((long)b * 2 * bd.binsoct + bd.binoct[k]) * T;
    const long imoff = (long)bd.binsoct * T;
    for (int i = threadIdx.x; i < T; i += blockDim.x) {
        const float2 v = a[fft_at(i)];
        out[i] = v.x;
        out[imoff + i] = v.y;
    }
}

__global__ __launch_bounds__(512) void band_synthesis_kernel(babe_cqt_bands bd, float* __restrict__ bs,
                                                             const float* __restrict__ win, long bs_stride) {
    __shared__ float2 a[FFT_LDS_LEN(4096)];
    const int k = blockIdx.x, b = blockIdx.y;
    const int M = bd.M[k], woff = bd.woff[k], lt = bd.log2T[k];
    const int T = 1 << lt;
    const float* in = bd.coef[bd.oct[k]] + ((long)b * 2 * bd.binsoct + bd.binoct[k]) * T;
    const long imoff = (long)bd.binsoct * T;
    for (int i = threadIdx.x; i < T; i += blockDim.x) a[fft_at(bitrev_n(i, lt))] = make_float2(in[i], in[imoff + i]);
    fft_lds_inplace(a, lt, reinterpret_cast<const float2*>(bd.tw4096), -1);
    float2* o = reinterpret_cast<float2*>(bs) + (long)b * bs_stride + woff;
    const int half = M >> 1;
    for (int mi = threadIdx.x; mi < M; mi += blockDim.x) {
        const int m = mi - half;
        const float2 v = a[fft_at(m & (T - 1))];
        const float w = win[woff + mi];
        o[mi] = make_float2(v.x * w, v.y * w);
    }
}

__global__ __launch_bounds__(256) void gather_kernel(const float* __restrict__ bs, long bs_stride,
                                                     const int* __restrict__ rowptr, const int* __restrict__ src,
                                                     float* __restrict__ spec, int KX, int L, float scale,
                                                     const float* __restrict__ mul) {
    const int b = blockIdx.y;
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= KX) return;
    float re = 0.f, im = 0.f;
    if (n <= L / 2) {
        const float2* p = reinterpret_cast<const float2*>(bs) + (long)b * bs_stride;
        const int e0 = rowptr[n], e1 = rowptr[n + 1];
        for (int e = e0; e < e1; ++e) {
            const int s = src[e];
            const float2 v = p[s & 0x7fffffff];
            re += v.x;
            im += (s < 0) ? -v.y : v.y;
        }
        float sc = scale;
        if (mul) sc *= mul[n];
        re *= sc;
        im *= sc;
    }
    spec[(long)b * 2 * KX + n] = re;
    spec[(long)b * 2 * KX + KX + n] = im;
}

__global__ __launch_bounds__(256) void spec_scale_kernel(const float* __restrict__ s1, const float* __restrict__ s2,
                                                         float* __restrict__ out, const float* __restrict__ mul,
                                                         int KX, int L, float sc1, float sc2) {
    const int b = blockIdx.y;
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= KX) return;
    float re = 0.f, im = 0.f;
    if (n <= L / 2) {
        const float m = mul ? mul[n] : 1.f;
        const long o = (long)b * 2 * KX + n;
        re = s1[o] * m * sc1;
        im = s1[o + KX] * m * sc1;
        if (s2) {
            re += s2[o] * m * sc2;
            im += s2[o + KX] * m * sc2;
        }
    }
    out[(long)b * 2 * KX + n] = re;
    out[(long)b * 2 * KX + KX + n] = im;
}

// 32x32 tiled transpose with complex twiddle.  grid (ceil(N2/32), ceil(N1/32), B)
__global__ __launch_bounds__(256) void twiddle_transpose_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                                const float2* __restrict__ tw, int N1, int N2,
                                                                int adjoint) {
    __shared__ float tr[32][33], ti[32][33];
    const int b = blockIdx.z;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;       // 32 x 8
    if (!adjoint) {
        // in [2*N1][N2] -> out [2*N2][N1]
        const float* ire = in + (long)b * 2 * N1 * N2;
        const float* iim = ire + (long)N1 * N2;
        float* ore = out + (long)b * 2 * N2 * N1;
        float* oim = ore + (long)N2 * N1;
        const int n2 = blockIdx.x * 32 + tx;
        for (int r = ty; r < 32; r += 8) {
            const int k1 = blockIdx.y * 32 + r;
            float vr = 0.f, vi = 0.f;
            if (k1 < N1 && n2 < N2) {
                const float ar = ire[(long)k1 * N2 + n2], ai = iim[(long)k1 * N2 + n2];
                const float2 w = tw[(long)k1 * N2 + n2];
                vr = ar * w.x - ai * w.y;
                vi = ar * w.y + ai * w.x;
            }
            tr[r][tx] = vr;
            ti[r][tx] = vi;
        }
        __syncthreads();
        const int k1 = blockIdx.y * 32 + tx;
        for (int r = ty; r < 32; r += 8) {
            const int n2o = blockIdx.x * 32 + r;
            if (k1 < N1 && n2o < N2) {
                ore[(long)n2o * N1 + k1] = tr[tx][r];
                oim[(long)n2o * N1 + k1] = ti[tx][r];
            }
        }
    } else {
        // in [2*N2][N1] -> out [2*N1][N2], multiply by conj(tw[k1][n2])
        const float* ire = in + (long)b * 2 * N2 * N1;
        const float* iim = ire + (long)N2 * N1;
        float* ore = out + (long)b * 2 * N1 * N2;
        float* oim = ore + (long)N1 * N2;
        const int k1 = blockIdx.y * 32 + tx;
        for (int r = ty; r < 32; r += 8) {
            const int n2 = blockIdx.x * 32 + r;
            float vr = 0.f, vi = 0.f;
            if (k1 < N1 && n2 < N2) {
                vr = ire[(long)n2 * N1 + k1];
                vi = iim[(long)n2 * N1 + k1];
            }
            tr[r][tx] = vr;
            ti[r][tx] = vi;
        }
        __syncthreads();
        const int n2 = blockIdx.x * 32 + tx;
        for (int r = ty; r < 32; r += 8) {
            const int k1o = blockIdx.y * 32 + r;
            if (k1o < N1 && n2 < N2) {
                const float ar = tr[tx][r], ai = ti[tx][r];
                const float2 w = tw[(long)k1o * N2 + n2];
                ore[(long)k1o * N2 + n2] = ar * w.x + ai * w.y;
                oim[(long)k1o * N2 + n2] = ai * w.x - ar * w.y;
            }
        }
    }
}

}  // namespace

extern "C" int babe_fft_twiddle_transpose(const float* in, float* out, const float* tw, int B, int N1, int N2,
                                          int adjoint, void* stream) {
    BABE_CHECK_ARG(in && out && tw && B > 0 && N1 > 0 && N2 > 0, "fft_twiddle_transpose: bad arguments");
    BabeProfScope prof(BABE_SLOT_CQT_GATHER, 24.0 * B * (double)N1 * N2, 0, 0, stream);
    hipLaunchKernelGGL(twiddle_transpose_kernel, dim3(cdiv(N2, 32), cdiv(N1, 32), B), dim3(256), 0,
                       (hipStream_t)stream, in, out, reinterpret_cast<const float2*>(tw), N1, N2, adjoint);
    BABE_LAUNCH_CHECK();
    return BABE_OK;
}

static int check_bands(const babe_cqt_bands* bd) {
    BABE_CHECK_ARG(bd && bd->nbands > 0 && bd->c && bd->M && bd->woff && bd->log2T && bd->oct && bd->binoct &&
                       bd->tw4096 && bd->nocts <= 8,
                   "cqt: bad band table");
    return 0;
}

extern "C" int babe_cqt_band_analysis(const babe_cqt_bands* bd, const float* spec, const float* win, int B,
                                      void* stream) {
    if (check_bands(bd)) return BABE_ERR_ARG;
    BABE_CHECK_ARG(spec && win && B > 0, "cqt_band_analysis: bad arguments");
    BabeProfScope prof(BABE_SLOT_CQT_ANALYSIS, (double)B * (8.0 * (bd->L / 2 + 1) + 8.0 * bd->sum_T + 4.0 * bd->sum_M), 5.0 * B * bd->sum_TlogT, 0, stream);
    hipLaunchKernelGGL(band_analysis_kernel, dim3(bd->nbands, B), dim3(512), 0, (hipStream_t)stream, *bd, spec, win);
    BABE_LAUNCH_CHECK();
    return BABE_OK;
}

extern "C" int babe_cqt_band_synthesis(const babe_cqt_bands* bd, float* bs, const float* win, long bs_stride, int B,
                                       void* stream) {
    if (check_bands(bd)) return BABE_ERR_ARG;
    BABE_CHECK_ARG(bs && win && B > 0, "cqt_band_synthesis: bad arguments");
    BabeProfScope prof(BABE_SLOT_CQT_SYNTHESIS, (double)B * (8.0 * bd->sum_T + 12.0 * bd->sum_M), 5.0 * B * bd->sum_TlogT, 0, stream);
    hipLaunchKernelGGL(band_synthesis_kernel, dim3(bd->nbands, B), dim3(512), 0, (hipStream_t)stream, *bd, bs, win,
                       bs_stride);
    BABE_LAUNCH_CHECK();
    return BABE_OK;
}

extern "C" int babe_cqt_gather(const float* bs, long bs_stride, const int* rowptr, const int* src, float* spec, int KX,
                               int L, float scale, const float* mul, int B, void* stream) {
    BABE_CHECK_ARG(bs && rowptr && src && spec && KX > L / 2 && B > 0, "cqt_gather: bad arguments");
    BabeProfScope prof(BABE_SLOT_CQT_GATHER, (double)B * 8.0 * (bs_stride + KX), 0, 0, stream);
    hipLaunchKernelGGL(gather_kernel, dim3(cdiv(KX, 256), B), dim3(256), 0, (hipStream_t)stream, bs, bs_stride, rowptr,
                       src, spec, KX, L, scale, mul);
    BABE_LAUNCH_CHECK();
    return BABE_OK;
}

extern "C" int babe_spec_scale(const float* s1, const float* s2, float* out, const float* mul, int KX, int L,
                               float sc1, float sc2, int B, void* stream) {
    BABE_CHECK_ARG(s1 && out && KX > L / 2 && B > 0, "spec_scale: bad arguments");
    BabeProfScope prof(BABE_SLOT_CQT_GATHER, (double)B * 8.0 * KX * (s2 ? 3 : 2), 0, 0, stream);
    hipLaunchKernelGGL(spec_scale_kernel, dim3(cdiv(KX, 256), B), dim3(256), 0, (hipStream_t)stream, s1, s2, out, mul,
                       KX, L, sc1, sc2);
    BABE_LAUNCH_CHECK();
    return BABE_OK;
}
