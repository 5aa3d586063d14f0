// Constant-Q (NSGT, "oct") analysis / synthesis kernels.  Definition: oracle/nsgt.py header and
// babe_amd/cqt_plan.py; call sites in the reference: networks/cqtdiff+.py:743,841,
// testing/blind_bwe_sampler.py:156 (cqt_nsgt_pytorch.CQT_nsgt.fwd/.bwd/.apply_hpf_DC).
// HBM/latency-bound: a workgroup owns 4096 complex points of one octave (4096/T bands) in LDS.
#include "common.h"
#include "../../include/babe_hip.h"
#include "prof.h"

namespace {

// ---- per-band FFTs: Stockham radix-16 passes with register butterflies --------------------------------------------------
// A workgroup of 256 threads owns 4096 complex points of ONE octave: 4096/T bands of T points each (64 bands when
// T <= 64), so every workgroup of the launch has the same amount of work whatever the octave - the round-1 kernel gave a
// 512-thread workgroup to every band, and 192 of the 448 bands used an eighth of it.  A band of T = R0 * TB points
// (R0 = min(16, T)) is transformed by TB threads in ceil(log16 T) passes: each thread gathers the R0 inputs of its
// butterflies from LDS (stride T/R), multiplies by the inter-pass twiddles (table exp(-2 pi i q/4096) copied to LDS once
// per workgroup), runs the radix-R DFT in registers (R = 16 as 4 x 4, 8 as 4 x 2) and scatters the results autosorted, so
// input and output are both in natural order and nothing is bit-reversed.  One LDS buffer: read - barrier - write - barrier.
// Element i of the workgroup's image lives at i + (i >> 4) (the first pass writes with stride 16: without the pad all lanes
// of a wave hit two banks).
#define CQ_AT(i) ((i) + ((i) >> 4))
constexpr int CQ_PTS = 4096;                           // points per workgroup
constexpr int CQ_LDS = CQ_PTS + (CQ_PTS >> 4);         // float2 elements

__device__ __forceinline__ float2 cq_mul(float2 x, float2 w) { return make_float2(x.x * w.x - x.y * w.y, x.x * w.y + x.y * w.x); }
__device__ __forceinline__ float2 cq_add(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 cq_sub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }

// 4-point DFT, sign S (exp(S 2 pi i nk/4)), in place, natural order
template <int S>
__device__ __forceinline__ void cq_fft4(float2& x0, float2& x1, float2& x2, float2& x3) {
    const float2 a0 = cq_add(x0, x2), a1 = cq_sub(x0, x2), a2 = cq_add(x1, x3), d = cq_sub(x1, x3);
    const float2 a3 = S > 0 ? make_float2(-d.y, d.x) : make_float2(d.y, -d.x);        // (S i) * d
    x0 = cq_add(a0, a2);
    x1 = cq_add(a1, a3);
    x2 = cq_sub(a0, a2);
    x3 = cq_sub(a1, a3);
}
// R-point DFT of v[0..R-1] in registers, R in {2, 4, 8, 16}.  R = 4 * R2: v[R2*k1 + k2] receives X[k1 + 4*k2]; the
// caller un-permutes with cq_perm<R>(r) when it stores.
template <int R, int S>
__device__ __forceinline__ void cq_fft(float2* v) {
    if constexpr (R == 2) {
        const float2 a = v[0], b = v[1];
        v[0] = cq_add(a, b);
        v[1] = cq_sub(a, b);
    } else if constexpr (R == 4) {
        cq_fft4<S>(v[0], v[1], v[2], v[3]);
    } else {
        constexpr int R2 = R / 4;
#pragma unroll
        for (int n2 = 0; n2 < R2; ++n2) cq_fft4<S>(v[n2], v[R2 + n2], v[2 * R2 + n2], v[3 * R2 + n2]);
        // twiddles W_R^(n2*k1), W_R = exp(S 2 pi i / R)
        constexpr float C16[10] = {1.f, 0.92387953251128674f, 0.70710678118654752f, 0.38268343236508977f, 0.f,
                                   -0.38268343236508977f, -0.70710678118654752f, -0.92387953251128674f, -1.f,
                                   -0.92387953251128674f};
        constexpr float S16[10] = {0.f, 0.38268343236508977f, 0.70710678118654752f, 0.92387953251128674f, 1.f,
                                   0.92387953251128674f, 0.70710678118654752f, 0.38268343236508977f, 0.f,
                                   -0.38268343236508977f};
#pragma unroll
        for (int k1 = 1; k1 < 4; ++k1)
#pragma unroll
            for (int n2 = 1; n2 < R2; ++n2) {
                constexpr int unit = 16 / R;                    // index into the 16th-root table
                const int m = n2 * k1 * unit;                   // <= 9
                v[R2 * k1 + n2] = cq_mul(v[R2 * k1 + n2], make_float2(C16[m], S > 0 ? S16[m] : -S16[m]));
            }
#pragma unroll
        for (int k1 = 0; k1 < 4; ++k1) cq_fft<R2, S>(v + R2 * k1);
    }
}
template <int R>
__device__ __forceinline__ constexpr int cq_perm(int r) {       // register r of cq_fft<R> holds output bin cq_perm(r)
    if constexpr (R <= 4) return r;
    else {
        constexpr int R2 = R / 4;
        const int k1 = r / R2, k2 = r % R2;
        return k1 + 4 * (R2 <= 4 ? k2 : cq_perm<R2>(k2));
    }
}

// One Stockham pass of radix R over every band of the workgroup.  LT = log2 T, Ns = product of the earlier radices.
// The band of this thread starts at element `base` of the image; u = its index among the band's TB threads.
template <int LT, int R, int NSL, int S>
__device__ __forceinline__ void cq_pass(float2* a, const float2* twl, int base, int u, bool active) {
    constexpr int N = 1 << LT;
    constexpr int R0 = N < 16 ? N : 16;
    constexpr int TB = N / R0;
    constexpr int CNT = R0 / R;                        // butterflies per thread in this pass
    constexpr int NR = N / R;
    constexpr int Ns = 1 << NSL;
    float2 v[CNT][R];
    if (active) {
#pragma unroll
        for (int c = 0; c < CNT; ++c) {
            const int j = u + c * TB;
#pragma unroll
            for (int t = 0; t < R; ++t) v[c][t] = a[CQ_AT(base + j + t * NR)];
            if constexpr (NSL > 0) {
                const int k = j & (Ns - 1);
                constexpr int stride = 4096 / (Ns * R);
#pragma unroll
                for (int t = 1; t < R; ++t) {
                    const int q = k * t * stride;       // < 4096
                    float2 w = twl[q & 2047];
                    if (q & 2048) w = make_float2(-w.x, -w.y);
                    if (S > 0) w.y = -w.y;
                    v[c][t] = cq_mul(v[c][t], w);
                }
            }
            cq_fft<R, S>(v[c]);
        }
    }
    __syncthreads();
    if (active) {
#pragma unroll
        for (int c = 0; c < CNT; ++c) {
            const int j = u + c * TB;
            const int d = ((j >> NSL) << (NSL + __builtin_ctz(R))) + (j & (Ns - 1));
#pragma unroll
            for (int r = 0; r < R; ++r) a[CQ_AT(base + d + cq_perm<R>(r) * Ns)] = v[c][r];
        }
    }
    __syncthreads();
}
// the whole transform of a T = 2^LT point band held at a[base ...]: radices 16,16,16 / 16,16,8 / 16,16,4 / 16,8,4 / 16,16 /
// 16,8 / 16,4 / 16,2 / 16 / 8
template <int LT, int S>
__device__ __forceinline__ void cq_band_fft(float2* a, const float2* twl, int base, int u, bool active) {
    if constexpr (LT == 12) { cq_pass<LT, 16, 0, S>(a, twl, base, u, active); cq_pass<LT, 16, 4, S>(a, twl, base, u, active); cq_pass<LT, 16, 8, S>(a, twl, base, u, active); }
    else if constexpr (LT == 11) { cq_pass<LT, 16, 0, S>(a, twl, base, u, active); cq_pass<LT, 16, 4, S>(a, twl, base, u, active); cq_pass<LT, 8, 8, S>(a, twl, base, u, active); }
    else if constexpr (LT == 10) { cq_pass<LT, 16, 0, S>(a, twl, base, u, active); cq_pass<LT, 16, 4, S>(a, twl, base, u, active); cq_pass<LT, 4, 8, S>(a, twl, base, u, active); }
    else if constexpr (LT == 9) { cq_pass<LT, 16, 0, S>(a, twl, base, u, active); cq_pass<LT, 8, 4, S>(a, twl, base, u, active); cq_pass<LT, 4, 7, S>(a, twl, base, u, active); }
    else if constexpr (LT == 8) { cq_pass<LT, 16, 0, S>(a, twl, base, u, active); cq_pass<LT, 16, 4, S>(a, twl, base, u, active); }
    else if constexpr (LT == 7) { cq_pass<LT, 16, 0, S>(a, twl, base, u, active); cq_pass<LT, 8, 4, S>(a, twl, base, u, active); }
    else if constexpr (LT == 6) { cq_pass<LT, 16, 0, S>(a, twl, base, u, active); cq_pass<LT, 4, 4, S>(a, twl, base, u, active); }
    else if constexpr (LT == 5) { cq_pass<LT, 16, 0, S>(a, twl, base, u, active); cq_pass<LT, 2, 4, S>(a, twl, base, u, active); }
    else if constexpr (LT == 4) { cq_pass<LT, 16, 0, S>(a, twl, base, u, active); }
    else if constexpr (LT == 3) { cq_pass<LT, 8, 0, S>(a, twl, base, u, active); }
    else { cq_pass<LT, 4, 0, S>(a, twl, base, u, active); }
}

// ANALYSIS (MODE 0): a[pos] = spec[(c + m) mod L] * win[m], m = pos for pos < M - M/2, pos - T for pos >= T - M/2, 0 between;
// IFFT_T (unnormalised; win carries 1/T); coefficients out planar.
// SYNTHESIS (MODE 1): a = coefficients; FFT_T; bs[woff + mi] = A[(mi - M/2) mod T] * win[mi].
template <int LT, int MODE>
__device__ __forceinline__ void cq_run(const babe_cqt_bands& bd, float2* a, const float2* twl, const int* bt, int k0,
                                       int nb, int b, const float* __restrict__ spec, float* __restrict__ bs,
                                       long bs_stride, const float* __restrict__ win, int abl) {
    constexpr int T = 1 << LT;
    constexpr int R0 = T < 16 ? T : 16;
    constexpr int TB = T / R0;
    constexpr int ITER = (T >= 256) ? 16 : 4096 / 256;        // sweeps of 256 threads over the workgroup's points
    const int tid = threadIdx.x;
    const int npts = nb * T;
    // bt: the band table of this workgroup in LDS: bt[3*s] = centre bin, bt[3*s+1] = window length, bt[3*s+2] = window offset
    // ---- load: all 256 threads sweep the points; consecutive threads = consecutive samples of a band.  Addresses first,
    // then all loads back to back (16 independent loads in flight per thread), then the LDS writes.
    if (MODE == 0) {
        const float* sre = spec + (long)b * 2 * bd.KX;
        float vr[ITER], vi[ITER], vw[ITER];
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int i = tid + it * 256;
            const int pos = i & (T - 1);
            const int s = (i >> LT) < nb ? (i >> LT) : nb - 1;
            const int c = bt[3 * s], M = bt[3 * s + 1], wo = bt[3 * s + 2];
            const int half = M >> 1;
            const int m = pos >= T - half ? pos - T : pos;
            const bool in = (i < npts) && (pos < M - half || pos >= T - half);
            int n = c + m;
            n = n < 0 ? n + bd.L : n;
            n = n >= bd.L ? n - bd.L : n;
            const bool mir = n > bd.L / 2;
            const int nn = mir ? bd.L - n : n;
            const int ns = in ? nn : 0;
            vr[it] = sre[ns];
            vi[it] = sre[bd.KX + ns];
            vw[it] = in ? win[wo + m + half] : 0.f;
            if (mir) vi[it] = -vi[it];
        }
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int i = tid + it * 256;
            if (i < npts) a[CQ_AT(i)] = make_float2(vr[it] * vw[it], vi[it] * vw[it]);
        }
    } else {
        float vr[ITER], vi[ITER];
        const long imoff = (long)bd.binsoct * T;
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int i = tid + it * 256;
            const int s = (i >> LT) < nb ? (i >> LT) : nb - 1, pos = i & (T - 1);
            const int k = k0 + s;       // bands of a workgroup belong to one octave: consecutive bins
            const float* in = bd.coef[bd.oct[k0]] + ((long)b * 2 * bd.binsoct + (bd.binoct[k0] + s)) * T;
            vr[it] = in[pos];
            vi[it] = in[imoff + pos];
            (void)k;
        }
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int i = tid + it * 256;
            if (i < npts) a[CQ_AT(i)] = make_float2(vr[it], vi[it]);
        }
    }
    __syncthreads();
    const int s = tid / TB, u = tid % TB;
    if (!(abl & 1)) cq_band_fft<LT, (MODE == 0 ? +1 : -1)>(a, twl, s << LT, u, s < nb);
    // ---- store
    if (MODE == 0) {
        float* out0 = bd.coef[bd.oct[k0]] + ((long)b * 2 * bd.binsoct + bd.binoct[k0]) * T;
        const long imoff = (long)bd.binsoct * T;
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int i = tid + it * 256;              // bands of the workgroup are consecutive bins: out0 + i
            if (i < npts) {
                const float2 v = a[CQ_AT(i)];
                out0[i] = v.x;
                out0[imoff + i] = v.y;
            }
        }
    } else {
        // window samples of the workgroup's bands are contiguous in the band-spectrum buffer
        const int w0 = bt[2];
        const int w1 = bt[3 * (nb - 1) + 2] + bt[3 * (nb - 1) + 1];
        float2* o = reinterpret_cast<float2*>(bs) + (long)b * bs_stride;
        int s2 = 0;
        for (int wi = w0 + tid; wi < w1; wi += 256) {
            while (s2 + 1 < nb && wi >= bt[3 * (s2 + 1) + 2]) ++s2;
            const int mi = wi - bt[3 * s2 + 2];
            const int m = mi - (bt[3 * s2 + 1] >> 1);
            const float2 v = a[CQ_AT((s2 << LT) + (m & (T - 1)))];
            const float w = win[wi];
            o[wi] = make_float2(v.x * w, v.y * w);
        }
    }
}

template <int MODE>
__global__ __launch_bounds__(256) void band_fft_kernel(babe_cqt_bands bd, const float* __restrict__ spec,
                                                       float* __restrict__ bs, long bs_stride,
                                                       const float* __restrict__ win) {
    __shared__ float2 a[CQ_LDS];
    __shared__ float2 twl[2048];
    __shared__ int bt[3 * 64];
    const int wg = blockIdx.x, b = blockIdx.y;
    const int k0 = bd.wg_first[wg], nb = bd.wg_count[wg], lt = bd.log2T[k0];
    if (nb > 64 || (nb << lt) > 4096) __builtin_trap();      // a table that contradicts its own summary fields: fail loudly
    // twiddle table -> LDS: loads issued first, written after (they are not needed before the second pass; the barrier at
    // the end of the load phase covers them)
    const float2* tw = reinterpret_cast<const float2*>(bd.tw4096);
    float2 twr[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) twr[i] = tw[threadIdx.x + i * 256];
    if (threadIdx.x < nb) {
        bt[3 * threadIdx.x] = bd.c[k0 + threadIdx.x];
        bt[3 * threadIdx.x + 1] = bd.M[k0 + threadIdx.x];
        bt[3 * threadIdx.x + 2] = bd.woff[k0 + threadIdx.x];
    }
    if (MODE == 0) __syncthreads();            // (synthesis needs the table only after its load-phase barrier)
#pragma unroll
    for (int i = 0; i < 8; ++i) twl[threadIdx.x + i * 256] = twr[i];
#ifdef BABE_CQT_ABL
    const int abl = bd.abl;
#else
    constexpr int abl = 0;
#endif
    switch (lt) {
#define CQ_CASE(L_) case L_: cq_run<L_, MODE>(bd, a, twl, bt, k0, nb, b, spec, bs, bs_stride, win, abl); break;
        CQ_CASE(12) CQ_CASE(11) CQ_CASE(10) CQ_CASE(9) CQ_CASE(8) CQ_CASE(7) CQ_CASE(6) CQ_CASE(5) CQ_CASE(4) CQ_CASE(3) CQ_CASE(2)
#undef CQ_CASE
        default: __builtin_trap();
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
                       bd->tw4096 && bd->nocts <= 8 && bd->wg_first && bd->wg_count && bd->nwg > 0,
                   "cqt: bad band table");
    // what the band-FFT kernel relies on: <= 64 bands per workgroup (its LDS band table), T in 4..4096 (its dispatch)
    BABE_CHECK_ARG(bd->max_wg_count >= 1 && bd->max_wg_count <= 64, "cqt: wg_count must be in 1..64 (got %d)", bd->max_wg_count);
    BABE_CHECK_ARG(bd->min_log2T >= 2 && bd->max_log2T <= 12 && bd->min_log2T <= bd->max_log2T,
                   "cqt: band lengths must be 2^2..2^12 (got 2^%d..2^%d)", bd->min_log2T, bd->max_log2T);
    BABE_CHECK_ARG(bd->binsoct >= 1 && bd->nwg <= bd->nbands, "cqt: workgroup table inconsistent with the band table");
    return 0;
}

extern "C" int babe_cqt_band_analysis(const babe_cqt_bands* bd, const float* spec, const float* win, int B,
                                      void* stream) {
    if (check_bands(bd)) return BABE_ERR_ARG;
    BABE_CHECK_ARG(spec && win && B > 0, "cqt_band_analysis: bad arguments");
    BabeProfScope prof(BABE_SLOT_CQT_ANALYSIS, (double)B * (8.0 * (bd->L / 2 + 1) + 8.0 * bd->sum_T + 4.0 * bd->sum_M), 5.0 * B * bd->sum_TlogT, 0, stream);
    hipLaunchKernelGGL(band_fft_kernel<0>, dim3(bd->nwg, B), dim3(256), 0, (hipStream_t)stream, *bd, spec, (float*)nullptr,
                       0L, win);
    BABE_LAUNCH_CHECK();
    return BABE_OK;
}

extern "C" int babe_cqt_band_synthesis(const babe_cqt_bands* bd, float* bs, const float* win, long bs_stride, int B,
                                       void* stream) {
    if (check_bands(bd)) return BABE_ERR_ARG;
    BABE_CHECK_ARG(bs && win && B > 0, "cqt_band_synthesis: bad arguments");
    BabeProfScope prof(BABE_SLOT_CQT_SYNTHESIS, (double)B * (8.0 * bd->sum_T + 12.0 * bd->sum_M), 5.0 * B * bd->sum_TlogT, 0, stream);
    hipLaunchKernelGGL(band_fft_kernel<1>, dim3(bd->nwg, B), dim3(256), 0, (hipStream_t)stream, *bd, (const float*)nullptr,
                       bs, bs_stride, win);
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
