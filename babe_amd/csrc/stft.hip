// STFT-domain degradation model of blind BWE on gfx950: STFT, filter + iSTFT, overlap-add / residual,
// per-bin magnitude statistics, piecewise-log low-pass design and the projected-GD filter fit.
// Reference: /root/reference/utils/blind_bwe_utils.py:6-39, :82-119, :250-296 and
// /root/reference/testing/blind_bwe_sampler.py:518-595.  HBM/latency-bound; the 4096-point FFT of
// a frame lives in LDS (fft_lds.h); the fit runs as ONE workgroup per clip on 3x2049 doubles.
#include "common.h"
#include "fft_lds.h"
#include "../../include/babe_hip.h"
#include "prof.h"
#include <cstdlib>

namespace {

__device__ __forceinline__ float hamming_periodic(int i, int n) {
    // torch.hamming_window(n) (periodic): 0.54 - 0.46 cos(2 pi i / n)
    return 0.54f - 0.46f * cospif(2.0f * (float)i / (float)n);
}

// grid (frames, B), 256 threads
__global__ __launch_bounds__(256) void stft_fwd_kernel(const float* __restrict__ x, long x_bs, int L,
                                                       const float* __restrict__ pre, float* __restrict__ spec,
                                                       int log2n, int frames, const float2* __restrict__ tw) {
    __shared__ float2 a[FFT_LDS_LEN(4096)];
    const int n = 1 << log2n, hop = n >> 1;
    const int t = blockIdx.x, b = blockIdx.y;
    const float* xb = x + (long)b * x_bs;
    const int s0 = t * hop;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int s = s0 + i;
        float v = (s < L) ? xb[s] : 0.f;
        if (pre) v *= pre[s];
        a[fft_at(bitrev_n(i, log2n))] = make_float2(v * hamming_periodic(i, n), 0.f);
    }
    fft_lds_inplace(a, log2n, tw, -1);
    const int nb = hop + 1;
    float2* o = reinterpret_cast<float2*>(spec) + ((long)b * frames + t) * nb;
    for (int k = threadIdx.x; k < nb; k += blockDim.x) o[k] = a[fft_at(k)];
}

__global__ __launch_bounds__(256) void spec_filter_istft_kernel(const float* __restrict__ spec,
                                                                const float* __restrict__ H, long H_bs,
                                                                float* __restrict__ fr, int log2n, int frames,
                                                                const float2* __restrict__ tw) {
    __shared__ float2 a[FFT_LDS_LEN(4096)];
    const int n = 1 << log2n, hop = n >> 1;
    const int t = blockIdx.x, b = blockIdx.y;
    const int nb = hop + 1;
    const float2* s = reinterpret_cast<const float2*>(spec) + ((long)b * frames + t) * nb;
    const float* Hb = H + (long)b * H_bs;
    for (int k = threadIdx.x; k < nb; k += blockDim.x) {
        float2 v = s[k];
        const float h = Hb[k];
        v.x *= h;
        v.y *= h;
        if (k == 0 || k == hop) {
            a[fft_at(bitrev_n(k, log2n))] = make_float2(v.x, 0.f);          // irfft ignores Im at DC / Nyquist
        } else {
            a[fft_at(bitrev_n(k, log2n))] = v;
            a[fft_at(bitrev_n(n - k, log2n))] = make_float2(v.x, -v.y);
        }
    }
    fft_lds_inplace(a, log2n, tw, +1);
    float* o = fr + ((long)b * frames + t) * n;
    const float inv = 1.f / (float)n;
    for (int i = threadIdx.x; i < n; i += blockDim.x) o[i] = a[fft_at(i)].x * inv * hamming_periodic(i, n);
}

// grid (nblk, B)
__global__ __launch_bounds__(256) void ola_kernel(const float* __restrict__ fr, const float* __restrict__ post,
                                                  const float* __restrict__ y, long y_bs, float* __restrict__ out,
                                                  long out_bs, double* __restrict__ part, int nblk, int L, int n,
                                                  int frames) {
    __shared__ double sh[8];
    const int b = blockIdx.y;
    const int hop = n >> 1;
    const float* f = fr + (long)b * frames * n;
    double acc = 0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < L; i += (long)gridDim.x * blockDim.x) {
        const int t1 = (int)(i / hop);                 // latest frame containing sample i
        float v = 0.f;
        if (t1 < frames) v += f[(long)t1 * n + (i - (long)t1 * hop)];
        if (t1 >= 1 && t1 - 1 < frames) v += f[(long)(t1 - 1) * n + (i - (long)(t1 - 1) * hop)];
        if (post) v *= post[i];
        if (y) {
            v = y[(long)b * y_bs + i] - v;
            acc += (double)v * v;
        }
        out[(long)b * out_bs + i] = v;
    }
    if (part) {
        acc = wave_sum(acc);
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        if (lane == 0) sh[wave] = acc;
        __syncthreads();
        if (threadIdx.x == 0) part[(long)b * nblk + blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
    }
}

__global__ __launch_bounds__(256) void residual_seed_kernel(const float* __restrict__ r, long r_bs,
                                                            const double* __restrict__ part, int nblk,
                                                            const float* __restrict__ post, float* __restrict__ out,
                                                            long out_bs, int L) {
    const int b = blockIdx.y;
    double s = 0;
    for (int i = 0; i < nblk; ++i) s += part[(long)b * nblk + i];
    const float inv = (s > 0) ? (float)(-1.0 / sqrt(s)) : 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < L; i += (long)gridDim.x * blockDim.x) {
        float v = r[(long)b * r_bs + i] * inv;
        if (post) v *= post[i];
        out[(long)b * out_bs + i] = v;
    }
}

// STFT-domain guidance distances (get_rec_grads :105-115 -> utils/blind_bwe_utils.py:148-247): X = S(rec), R = S(y), both
// [B][frames][nbins] complex, w[nbins] the frequency weighting.  mode 0: D = ||w (X - R)||_2 (apply_norm_STFT_fweighted),
// 1: ||w|X| - w|R|||_2, 2: ||log10(w|X| + 1e-8) - log10(w|R| + 1e-8)||_2 (apply_norm_STFTmag_fweighted).
__device__ __forceinline__ float stft_dist_term(float2 x, float2 r, float w, int mode, float& mx) {
    if (mode == 0) {
        const float dr = w * x.x - w * r.x, di = w * x.y - w * r.y;
        mx = 0.f;
        return dr * dr + di * di;
    }
    mx = sqrtf(x.x * x.x + x.y * x.y);
    const float mr = sqrtf(r.x * r.x + r.y * r.y);
    const float d = mode == 1 ? mx * w - mr * w : log10f(mx * w + 1e-8f) - log10f(mr * w + 1e-8f);
    return d * d;
}
// grid (nblk, B): partial sums of the squared terms
__global__ __launch_bounds__(256) void stft_dist_partial_kernel(const float* __restrict__ X, const float* __restrict__ R,
                                                                const float* __restrict__ w, double* __restrict__ part,
                                                                int nblk, int nbins, long n, int mode) {
    __shared__ double sh[4];
    const int b = blockIdx.y;
    const float2* x2 = reinterpret_cast<const float2*>(X) + (long)b * n;
    const float2* r2 = reinterpret_cast<const float2*>(R) + (long)b * n;
    double acc = 0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float mx;
        acc += (double)stft_dist_term(x2[i], r2[i], w[i % nbins], mode, mx);
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part[(long)b * nblk + blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}
// G = dD/dX (real, imag); shared: one D over the whole batch (the reference's coupling), else one per batch item
__global__ __launch_bounds__(256) void stft_dist_grad_kernel(const float* __restrict__ X, const float* __restrict__ R,
                                                             const float* __restrict__ w, const double* __restrict__ part,
                                                             int nblk, float* __restrict__ G, int nbins, long n, int mode,
                                                             int shared, int B) {
    const int b = blockIdx.y;
    double s = 0;
    const int b0 = shared ? 0 : b, b1 = shared ? B : b + 1;
    for (int bb = b0; bb < b1; ++bb)
        for (int i = 0; i < nblk; ++i) s += part[(long)bb * nblk + i];
    const float invD = s > 0 ? (float)(1.0 / sqrt(s)) : 0.f;
    const float2* x2 = reinterpret_cast<const float2*>(X) + (long)b * n;
    const float2* r2 = reinterpret_cast<const float2*>(R) + (long)b * n;
    float2* g2 = reinterpret_cast<float2*>(G) + (long)b * n;
    const float iln10 = 0.43429448190325176f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float2 x = x2[i], r = r2[i];
        const float ww = w[i % nbins];
        float2 g;
        if (mode == 0) {
            g.x = ww * (ww * x.x - ww * r.x) * invD;
            g.y = ww * (ww * x.y - ww * r.y) * invD;
        } else {
            const float mx = sqrtf(x.x * x.x + x.y * x.y), mr = sqrtf(r.x * r.x + r.y * r.y);
            float c;                                   // dD/d|X|
            if (mode == 1) c = (mx * ww - mr * ww) * invD * ww;
            else c = (log10f(mx * ww + 1e-8f) - log10f(mr * ww + 1e-8f)) * invD * iln10 / (mx * ww + 1e-8f) * ww;
            const float im = mx > 0.f ? 1.f / mx : 0.f;
            g.x = c * x.x * im;
            g.y = c * x.y * im;
        }
        g2[i] = g;
    }
}

// grid (ceil(nbins/256), Bout): threads over bins, loop over frames (and batch if shared)
__global__ __launch_bounds__(256) void mag_stats_kernel(const float* __restrict__ sx, const float* __restrict__ sy,
                                                        double* __restrict__ stats, int B, int nbins, int frames,
                                                        int shared) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nbins) return;
    const int bo = blockIdx.y;
    const int b0 = shared ? 0 : bo, b1 = shared ? B : bo + 1;
    double sxx = 0, sxy = 0, syy = 0;
    for (int b = b0; b < b1; ++b) {
        const float2* X = reinterpret_cast<const float2*>(sx) + (long)b * frames * nbins;
        const float2* Y = reinterpret_cast<const float2*>(sy) + (long)b * frames * nbins;
        for (int t = 0; t < frames; ++t) {
            const float2 xv = X[(long)t * nbins + k], yv = Y[(long)t * nbins + k];
            const float mx = sqrtf(xv.x * xv.x + xv.y * xv.y);
            const float my = sqrtf(yv.x * yv.x + yv.y * yv.y);
            sxx += (double)mx * mx;
            sxy += (double)mx * my;
            syy += (double)my * my;
        }
    }
    stats[((long)bo * 3 + 0) * nbins + k] = sxx;
    stats[((long)bo * 3 + 1) * nbins + k] = sxy;
    stats[((long)bo * 3 + 2) * nbins + k] = syy;
}

constexpr int KMAX = 8;

struct Filt {
    float fc[KMAX], A[KMAX], anchor[KMAX];
    int kstar[KMAX];
};

// first bin whose float32 frequency is >= fc (exact float32 comparisons like `f >= fc` in the reference)
__device__ int first_bin_ge(float fc, float df, int nbins) {
    int k = (int)floorf(fc / df);
    if (k < 0) k = 0;
    if (k > nbins - 1) k = nbins - 1;
    while (k > 0 && (float)(k - 1) * df >= fc) --k;
    while (k < nbins && (float)k * df < fc) ++k;
    return k;                                       // == nbins when no bin qualifies
}

__device__ __forceinline__ float seg_val(float f, float fc, float A) {
    // 10 ** (A * log2(f / fc) / 20) in float32, same operation order as the reference
    return powf(10.f, A * log2f(f / fc) / 20.f);
}

// serial anchor chain (K <= 8): anchor_i = value written by segment i-1 at the first bin >= fc_i
__device__ void build_filter(Filt& F, int K, float df, int nbins) {
    F.anchor[0] = 1.f;
    F.kstar[0] = first_bin_ge(F.fc[0], df, nbins);
    for (int i = 1; i < K; ++i) {
        const int ks = first_bin_ge(F.fc[i], df, nbins);
        F.kstar[i] = ks;
        const float fk = (float)(ks < nbins ? ks : nbins - 1) * df;
        // the bin may sit below fc_{i-1}'s first bin only if fc is unsorted; then H there is still the older value
        int j = i - 1;
        while (j > 0 && fk < F.fc[j]) --j;
        float v;
        if (fk < F.fc[j]) v = 1.f;
        else v = seg_val(fk, F.fc[j], F.A[j]) * F.anchor[j];
        F.anchor[i] = v;
    }
}

__device__ __forceinline__ int seg_of(const Filt& F, int K, float f) {
    int s = -1;
    for (int i = 0; i < K; ++i)
        if (f >= F.fc[i]) s = i;                    // later breakpoints overwrite earlier ones
    return s;
}

__global__ __launch_bounds__(256) void design_filter_kernel(const float* __restrict__ params, float* __restrict__ H,
                                                            int K, int nbins, float df) {
    __shared__ Filt F;
    const int p = blockIdx.x;
    if (threadIdx.x == 0) {
        for (int i = 0; i < K; ++i) {
            F.fc[i] = params[((long)p * 2 + 0) * K + i];
            F.A[i] = params[((long)p * 2 + 1) * K + i];
        }
        build_filter(F, K, df, nbins);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < nbins; k += blockDim.x) {
        const float f = (float)k * df;
        const int s = seg_of(F, K, f);
        H[(long)p * nbins + k] = (s < 0) ? 1.f : seg_val(f, F.fc[s], F.A[s]) * F.anchor[s];
    }
}

__device__ __forceinline__ float weight_sq(int k, int nbins, int kind) {
    const float fr = (float)k / (float)(nbins - 1);          // torch.linspace(0,1,nbins)[k]
    float w;
    switch (kind) {
        case 1: w = sqrtf(fr); break;
        case 2: w = fr; break;
        case 3: w = log2f(1.f + fr); break;
        default: w = 1.f;
    }
    return w * w;
}

// one block per clip
// lossgrad != NULL: no descent - the loss and its gradient at `params` are written ([P][1 + 2K]: loss, d/dfc_j, d/dA_j) and the
// parameters are left alone (BlindSampler.compute_sweep, testing/blind_bwe_sampler.py:598-616); stats_ps: stride between the
// statistics of consecutive parameter sets, in doubles (0: one set of statistics for all of them)
__global__ __launch_bounds__(256) void filter_fit_kernel(const double* __restrict__ stats, float* __restrict__ params,
                                                         int* __restrict__ n_iter, int K, int nbins, float df,
                                                         babe_fit_cfg cfg, long stats_ps, float* __restrict__ lossgrad) {
    __shared__ Filt F;
    __shared__ double red[4][2 * KMAX + 1];
    __shared__ float prev[2 * KMAX];
    __shared__ int done;
    const int p = blockIdx.x;
    const double* Sxx = stats + (long)p * stats_ps;
    const double* Sxy = Sxx + nbins;
    const double* Syy = Sxx + 2 * nbins;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) {
        for (int i = 0; i < K; ++i) {
            F.fc[i] = params[((long)p * 2 + 0) * K + i];
            F.A[i] = params[((long)p * 2 + 1) * K + i];
        }
        done = 0;
    }
    int it = 0;
    const double cln = 0.11512925464970228;   // ln(10)/20
    for (; it < cfg.max_iter; ++it) {
        __syncthreads();
        if (threadIdx.x == 0) build_filter(F, K, df, nbins);
        __syncthreads();
        double E[KMAX], Ep[KMAX], loss2 = 0;
#pragma unroll
        for (int i = 0; i < KMAX; ++i) E[i] = Ep[i] = 0;
        for (int k = threadIdx.x; k < nbins; k += blockDim.x) {
            const float f = (float)k * df;
            const int s = seg_of(F, K, f);
            const double w2 = weight_sq(k, nbins, cfg.weighting);
            double Hk = 1.0, lg = 0.0;
            if (s >= 0) {
                Hk = (double)(seg_val(f, F.fc[s], F.A[s]) * F.anchor[s]);
                lg = (double)log2f(f / F.fc[s]);
            }
            loss2 += w2 * (Hk * Hk * Sxx[k] - 2.0 * Hk * Sxy[k] + Syy[k]);
            if (s >= 0) {
                const double e = w2 * (Hk * Sxx[k] - Sxy[k]) * Hk * cln;
#pragma unroll
                for (int i = 0; i < KMAX; ++i)
                    if (i == s) {
                        E[i] += e;
                        Ep[i] += e * lg;
                    }
            }
        }
        loss2 = wave_sum(loss2);
#pragma unroll
        for (int i = 0; i < KMAX; ++i) {
            E[i] = wave_sum(E[i]);
            Ep[i] = wave_sum(Ep[i]);
        }
        if (lane == 0) {
            red[wave][0] = loss2;
            for (int i = 0; i < KMAX; ++i) {
                red[wave][1 + i] = E[i];
                red[wave][1 + KMAX + i] = Ep[i];
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            double l2 = 0, Et[KMAX], Ept[KMAX];
            for (int i = 0; i < KMAX; ++i) Et[i] = Ept[i] = 0;
            for (int w = 0; w < 4; ++w) {
                l2 += red[w][0];
                for (int i = 0; i < KMAX; ++i) {
                    Et[i] += red[w][1 + i];
                    Ept[i] += red[w][1 + KMAX + i];
                }
            }
            const double loss = sqrt(l2 > 0 ? l2 : 0);
            const double il = loss > 0 ? 1.0 / loss : 0.0;
            // suffix sums of E
            double suf[KMAX + 1];
            suf[K] = 0;
            for (int i = K - 1; i >= 0; --i) suf[i] = suf[i + 1] + Et[i];
            float nfc[KMAX], nA[KMAX];
            for (int j = 0; j < K; ++j) {
                double Lj = 0;
                if (j + 1 < K) {
                    const int ks = F.kstar[j + 1] < nbins ? F.kstar[j + 1] : nbins - 1;
                    Lj = (double)log2f(((float)ks * df) / F.fc[j]);
                }
                const double gA = (Ept[j] + Lj * suf[j + 1]) * il;
                const double gfc = -(double)F.A[j] / ((double)F.fc[j] * 0.6931471805599453) * suf[j] * il;
                nfc[j] = F.fc[j] - cfg.mu_fc * (float)gfc;
                nA[j] = F.A[j] - cfg.mu_A * (float)gA;
                if (lossgrad) {
                    lossgrad[(long)p * (1 + 2 * K) + 1 + j] = (float)gfc;
                    lossgrad[(long)p * (1 + 2 * K) + 1 + K + j] = (float)gA;
                }
            }
            if (lossgrad) {
                lossgrad[(long)p * (1 + 2 * K)] = (float)loss;
                done = 2;                                      // (evaluation only: parameters untouched)
            }
            if (cfg.clamp_fc) {
                nfc[0] = fminf(fmaxf(nfc[0], cfg.fcmin), cfg.fcmax);
                for (int j = 1; j < K; ++j) nfc[j] = fminf(fmaxf(nfc[j], nfc[j - 1] + 1.f), cfg.fcmax);
            }
            if (cfg.clamp_A) {
                nA[0] = fminf(fmaxf(nA[0], cfg.Amin), cfg.only_negative_A ? -1.f : cfg.Amax);
                for (int j = 1; j < K; ++j)
                    nA[j] = fminf(fmaxf(nA[j], cfg.Amin), cfg.only_negative_A ? nA[j - 1] : cfg.Amax);
            }
            if (it > 0) {
                float dfc = 0, dA = 0;
                for (int j = 0; j < K; ++j) {
                    dfc += fabsf(nfc[j] - prev[j]);
                    dA += fabsf(nA[j] - prev[KMAX + j]);
                }
                if (dfc / K < cfg.tol_fc && dA / K < cfg.tol_A) done = 1;
            }
            for (int j = 0; j < K && done != 2; ++j) {
                F.fc[j] = nfc[j];
                F.A[j] = nA[j];
                prev[j] = nfc[j];
                prev[KMAX + j] = nA[j];
            }
        }
        __syncthreads();
        if (done) {
            ++it;
            break;
        }
    }
    if (threadIdx.x == 0 && !lossgrad) {
        for (int i = 0; i < K; ++i) {
            params[((long)p * 2 + 0) * K + i] = F.fc[i];
            params[((long)p * 2 + 1) * K + i] = F.A[i];
        }
        if (n_iter) n_iter[p] = it;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Fast variant of the fit (round 4): same projected gradient descent, same statistics, same update rule and early exit, built
// for LATENCY - the kernel is one workgroup on every evaluation's critical path (138 calls per benchmark step).
// What the first kernel paid per iteration (17 us): three workgroup barriers around two single-thread sections (the anchor
// chain with K powf / log2f / divisions; the K-parameter update), 8 bins per thread with powf + 2 log2f + 2 divisions each,
// and the statistics re-read from L2 every iteration.  Here:
//  * 256 threads = ONE wave per SIMD (every wave executes the serial sections redundantly, and two waves of a SIMD would
//    share its one vector issue slot: 512 threads measured 594 us per 100 iterations, 6 us each), at most 10 bins per thread,
//    whose three statistics, weight and log2(f) live in REGISTERS for all iterations;
//  * H_k = exp2(A_s * (log2 f_k - log2 fc_s) * log2(10)/20) * anchor_s on the hardware's v_exp_f32 / v_log_f32 (1 ulp; the
//    libm powf / log2f of the first kernel differ from the reference's CPU libm by as much - the reference's own H carries
//    ~1e-6 relative rounding noise at 50 dB of attenuation - so this changes the noise realisation, not its size; the filter
//    that is APPLIED is still designed by design_filter_kernel with the reference's operation order and exact masks);
//  * no single-thread section: the anchor chain and the parameter update are computed redundantly by every lane from
//    wave-uniform data; ONE barrier per iteration (per-wave partial sums in a double-buffered LDS array, every wave adds
//    the eight partials in the same order, so all waves hold bit-identical parameters).
// Segment masks use the exact float32 comparisons `f >= fc` of the reference, as before.
constexpr int FIT_NT = 256, FIT_NB = 10;                // threads, bins per thread (nbins <= 2560: NFFT <= 4096 + 1024)

// first_bin_ge with a multiplication for the initial guess (the two correction loops make the result exact whatever the guess)
__device__ __forceinline__ int first_bin_ge_mul(float fc, float df, float inv_df, int nbins) {
    int k = (int)floorf(fc * inv_df);
    k = k < 0 ? 0 : (k > nbins - 1 ? nbins - 1 : k);
    while (k > 0 && (float)(k - 1) * df >= fc) --k;
    while (k < nbins && (float)k * df < fc) ++k;
    return k;
}
__device__ __forceinline__ float fast_log2(float x) { return __builtin_amdgcn_logf(x); }
// Sum over the 64 lanes of a wave without LDS round trips: four DPP row shifts give lane 15 of every 16-lane row its row's
// total (inclusive scan), v_readlane fetches the four row totals, which are added in a fixed order: the result is wave-uniform
// (and lives in scalar registers).  ds_bpermute butterflies (wave_sum) cost six LDS latencies per value.
__device__ __forceinline__ double dpp_row_shr_add(double v, int sh) {
    int lo = __double2loint(v), hi = __double2hiint(v), lo2, hi2;
    switch (sh) {
        case 1: lo2 = __builtin_amdgcn_update_dpp(0, lo, 0x111, 0xf, 0xf, true); hi2 = __builtin_amdgcn_update_dpp(0, hi, 0x111, 0xf, 0xf, true); break;
        case 2: lo2 = __builtin_amdgcn_update_dpp(0, lo, 0x112, 0xf, 0xf, true); hi2 = __builtin_amdgcn_update_dpp(0, hi, 0x112, 0xf, 0xf, true); break;
        case 4: lo2 = __builtin_amdgcn_update_dpp(0, lo, 0x114, 0xf, 0xf, true); hi2 = __builtin_amdgcn_update_dpp(0, hi, 0x114, 0xf, 0xf, true); break;
        default: lo2 = __builtin_amdgcn_update_dpp(0, lo, 0x118, 0xf, 0xf, true); hi2 = __builtin_amdgcn_update_dpp(0, hi, 0x118, 0xf, 0xf, true); break;
    }
    return v + __hiloint2double(hi2, lo2);            // (bound_ctrl: lanes shifted in from outside the row read 0)
}
__device__ __forceinline__ double wave_sum_uniform(double v) {
    v = dpp_row_shr_add(v, 1);
    v = dpp_row_shr_add(v, 2);
    v = dpp_row_shr_add(v, 4);
    v = dpp_row_shr_add(v, 8);
    const int lo = __double2loint(v), hi = __double2hiint(v);
    double t[4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
        t[r] = __hiloint2double(__builtin_amdgcn_readlane(hi, 16 * r + 15), __builtin_amdgcn_readlane(lo, 16 * r + 15));
    return (t[0] + t[1]) + (t[2] + t[3]);
}
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

template <int K>                                        // (compile-time K: the K-loops are straight-line code, no predicates)
__global__ __launch_bounds__(FIT_NT) void filter_fit_fast_kernel(const double* __restrict__ stats, float* __restrict__ params,
                                                                  int* __restrict__ n_iter, int nbins, float df,
                                                                  babe_fit_cfg cfg) {
    constexpr int KMAX = K;                              // (shadows the file-wide bound inside this kernel)
    constexpr int NW = FIT_NT / 64, NV = 2 * KMAX + 1;
    __shared__ double red[2][NW][NV];
    const int p = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const double* Sxx = stats + ((long)p * 3 + 0) * nbins;
    const double* Sxy = stats + ((long)p * 3 + 1) * nbins;
    const double* Syy = stats + ((long)p * 3 + 2) * nbins;
    constexpr float C20 = 0.16609640474436813f;          // log2(10) / 20
    const float inv_df = 1.f / df;
    const double cln = 0.11512925464970228;              // ln(10) / 20
    // per-thread bins k = tid + 512 m: statistics pre-multiplied by the weight, frequency and its log2
    double sxx[FIT_NB], sxy[FIT_NB], syy[FIT_NB];
    float fk[FIT_NB], l2f[FIT_NB];
#pragma unroll
    for (int m = 0; m < FIT_NB; ++m) {
        const int k = tid + FIT_NT * m;
        const bool ok = k < nbins;
        const double w2 = ok ? (double)weight_sq(k, nbins, cfg.weighting) : 0.0;
        sxx[m] = ok ? w2 * Sxx[k] : 0.0;
        sxy[m] = ok ? w2 * Sxy[k] : 0.0;
        syy[m] = ok ? w2 * Syy[k] : 0.0;
        fk[m] = (float)k * df;
        l2f[m] = fast_log2(fk[m]);                       // (k = 0: -inf, never used: f = 0 lies below every breakpoint)
    }
    float fc[KMAX], A[KMAX], anchor[KMAX], l2fc[KMAX], prev_fc[KMAX], prev_A[KMAX];
    int kstar[KMAX];
#pragma unroll
    for (int i = 0; i < KMAX; ++i) {
        fc[i] = i < K ? params[((long)p * 2 + 0) * K + i] : 3.0e38f;      // (unused breakpoints: above every bin)
        A[i] = i < K ? params[((long)p * 2 + 1) * K + i] : 0.f;
        prev_fc[i] = prev_A[i] = 0.f;
        anchor[i] = 1.f;
        l2fc[i] = 0.f;
        kstar[i] = nbins;
    }
    int it = 0;
    for (; it < cfg.max_iter; ++it) {
        // ---- anchors (build_filter): every lane, wave-uniform data
#pragma unroll
        for (int i = 0; i < KMAX; ++i)
            if (i < K) {
                l2fc[i] = fast_log2(fc[i]);
                kstar[i] = first_bin_ge_mul(fc[i], df, inv_df, nbins);
            }
        anchor[0] = 1.f;
#pragma unroll
        for (int i = 1; i < KMAX; ++i)
            if (i < K) {
                const float fq = (float)(kstar[i] < nbins ? kstar[i] : nbins - 1) * df;
                // segment that wrote H at that bin before segment i did: the last j < i with fq >= fc[j] searched downwards as
                // build_filter does (j = i - 1, stepping down while fq < fc[j] and j > 0)
                float fcj = fc[0], Aj = A[0], aj = anchor[0], lj = l2fc[0];
                bool found = false;
#pragma unroll
                for (int j = KMAX - 1; j >= 1; --j)
                    if (j < i && !found && !(fq < fc[j])) {
                        fcj = fc[j]; Aj = A[j]; aj = anchor[j]; lj = l2fc[j];
                        found = true;
                    }
                anchor[i] = (fq < fcj) ? 1.f : fast_exp2(Aj * (fast_log2(fq) - lj) * C20) * aj;
            }
        // ---- per-bin pass.  H_k = 2^(P_s l2f_k + Q_s) with P_s = c A_s, Q_s = log2(anchor_s) - P_s log2(fc_s): two selected
        // values per bin; Ep = sum e (l2f_k - log2 fc_s) is accumulated as sum e l2f_k and corrected by log2(fc_s) E_s after the
        // workgroup sums (the subtraction moves out of the loop; same value up to rounding of the double sums)
        float Pq[KMAX], Qq[KMAX];
#pragma unroll
        for (int i = 0; i < KMAX; ++i) {
            Pq[i] = C20 * A[i];
            Qq[i] = fast_log2(anchor[i]) - Pq[i] * l2fc[i];
        }
        double E[KMAX], Ep[KMAX], loss2 = 0;
#pragma unroll
        for (int i = 0; i < KMAX; ++i) E[i] = Ep[i] = 0;
#pragma unroll
        for (int m = 0; m < FIT_NB; ++m) {
            const float f = fk[m];
            int sidx = -1;
            float Ps = 0.f, Qs = 0.f;
#pragma unroll
            for (int i = 0; i < KMAX; ++i)
                if (f >= fc[i]) {                        // later breakpoints overwrite earlier ones (seg_of)
                    sidx = i; Ps = Pq[i]; Qs = Qq[i];
                }
            const double Hk = sidx >= 0 ? (double)fast_exp2(Ps * l2f[m] + Qs) : 1.0;
            const double hx = Hk * sxx[m];
            loss2 += Hk * hx - 2.0 * Hk * sxy[m] + syy[m];
            const double e = sidx >= 0 ? (hx - sxy[m]) * Hk * cln : 0.0;
            const double el = sidx >= 0 ? e * (double)l2f[m] : 0.0;
#pragma unroll
            for (int i = 0; i < KMAX; ++i)
                if (i == sidx) {
                    E[i] += e;
                    Ep[i] += el;
                }
        }
        // ---- workgroup sums: DPP wave sums (uniform), per-wave partials through LDS, ONE barrier; then lane v < 2K + 1 of every
        // wave adds the eight partials of value v (same order in every wave) and v_readlane broadcasts the totals
        loss2 = wave_sum_uniform(loss2);
#pragma unroll
        for (int i = 0; i < KMAX; ++i) {
            E[i] = wave_sum_uniform(E[i]);
            Ep[i] = wave_sum_uniform(Ep[i]);
        }
        double (*rb)[NV] = red[it & 1];
        {
            double mine = loss2;                              // lane v keeps value v of this wave
#pragma unroll
            for (int i = 0; i < KMAX; ++i) {
                mine = lane == 1 + i ? E[i] : mine;
                mine = lane == 1 + KMAX + i ? Ep[i] : mine;
            }
            if (lane < NV) rb[wave][lane] = mine;
        }
        __syncthreads();
        double tot = 0;
        if (lane < NV) {
#pragma unroll
            for (int w = 0; w < NW; ++w) tot += rb[w][lane];
        }
        const int tlo = __double2loint(tot), thi = __double2hiint(tot);
        auto total = [&](int v) __attribute__((always_inline)) {
            return __hiloint2double(__builtin_amdgcn_readlane(thi, v), __builtin_amdgcn_readlane(tlo, v));
        };
        const double l2 = total(0);
        double Et[KMAX], Ept[KMAX];
#pragma unroll
        for (int i = 0; i < KMAX; ++i) {
            Et[i] = total(1 + i);
            Ept[i] = total(1 + KMAX + i) - (double)l2fc[i] * Et[i];
        }
        // ---- parameter update (fit_params :561-590), every lane
        // (1 / loss and A / (fc ln 2) in float: the products are rounded to float before they are used, :568-570 of the reference)
        const float lossf = sqrtf(l2 > 0 ? (float)l2 : 0.f);
        const double il = lossf > 0 ? (double)(1.f / lossf) : 0.0;
        double suf[KMAX + 1];
        suf[KMAX] = 0;
#pragma unroll
        for (int i = KMAX - 1; i >= 0; --i) suf[i] = suf[i + 1] + (i < K ? Et[i] : 0.0);
        float nfc[KMAX], nA[KMAX];
#pragma unroll
        for (int j = 0; j < KMAX; ++j) {
            nfc[j] = fc[j];
            nA[j] = A[j];
            if (j < K) {
                double Lj = 0;
                if (j + 1 < KMAX && j + 1 < K) {
                    const int ks = kstar[j + 1] < nbins ? kstar[j + 1] : nbins - 1;
                    Lj = (double)(fast_log2((float)ks * df) - l2fc[j]);
                }
                const double gA = (Ept[j] + Lj * suf[j + 1]) * il;
                const double gfc = (double)(-A[j] / (fc[j] * 0.6931471805599453f)) * suf[j] * il;
                nfc[j] = fc[j] - cfg.mu_fc * (float)gfc;
                nA[j] = A[j] - cfg.mu_A * (float)gA;
            }
        }
        if (cfg.clamp_fc) {
            nfc[0] = fminf(fmaxf(nfc[0], cfg.fcmin), cfg.fcmax);
#pragma unroll
            for (int j = 1; j < KMAX; ++j)
                if (j < K) nfc[j] = fminf(fmaxf(nfc[j], nfc[j - 1] + 1.f), cfg.fcmax);
        }
        if (cfg.clamp_A) {
            nA[0] = fminf(fmaxf(nA[0], cfg.Amin), cfg.only_negative_A ? -1.f : cfg.Amax);
#pragma unroll
            for (int j = 1; j < KMAX; ++j)
                if (j < K) nA[j] = fminf(fmaxf(nA[j], cfg.Amin), cfg.only_negative_A ? nA[j - 1] : cfg.Amax);
        }
        bool done = false;
        if (it > 0) {
            float dfc = 0, dA = 0;
#pragma unroll
            for (int j = 0; j < KMAX; ++j)
                if (j < K) {
                    dfc += fabsf(nfc[j] - prev_fc[j]);
                    dA += fabsf(nA[j] - prev_A[j]);
                }
            done = dfc / K < cfg.tol_fc && dA / K < cfg.tol_A;
        }
#pragma unroll
        for (int j = 0; j < KMAX; ++j)
            if (j < K) {
                fc[j] = nfc[j];
                A[j] = nA[j];
                prev_fc[j] = nfc[j];
                prev_A[j] = nA[j];
            }
        if (done) {                                      // (wave-uniform AND workgroup-uniform: identical data in every lane)
            ++it;
            break;
        }
    }
    if (tid == 0) {
#pragma unroll
        for (int i = 0; i < KMAX; ++i)
            if (i < K) {
                params[((long)p * 2 + 0) * K + i] = fc[i];
                params[((long)p * 2 + 1) * K + i] = A[i];
            }
        if (n_iter) n_iter[p] = it;
    }
}

int ilog2_exact(int n) {
    int l = 0;
    while ((1 << l) < n) ++l;
    return (1 << l) == n ? l : -1;
}

}  // namespace

extern "C" int babe_stft_fwd(const float* x, long x_bs, int L, const float* pre, float* spec, int B, int nfft,
                             int frames, const float* tw4096, void* stream) {
    const int lg = ilog2_exact(nfft);
    BABE_CHECK_ARG(x && spec && tw4096 && B > 0 && L > 0, "stft_fwd: bad arguments");
    BABE_CHECK_ARG(lg >= 8 && lg <= 12, "stft_fwd: nfft=%d unsupported (256..4096, power of two)", nfft);
    BABE_CHECK_ARG(frames == 1 + L / (nfft / 2), "stft_fwd: frames=%d inconsistent with L=%d", frames, L);
    BabeProfScope prof(BABE_SLOT_STFT_FWD, (double)B * (4.0 * L * (pre ? 2 : 1) + 8.0 * frames * (nfft / 2 + 1)), 0, 0, stream);
    hipLaunchKernelGGL(stft_fwd_kernel, dim3(frames, B), dim3(256), 0, (hipStream_t)stream, x, x_bs, L, pre, spec, lg,
                       frames, reinterpret_cast<const float2*>(tw4096));
    BABE_LAUNCH_CHECK();
    return BABE_OK;
}

extern "C" int babe_spec_filter_istft(const float* spec, const float* H, long H_bs, float* frames_out, int B, int nfft,
                                      int frames, const float* tw4096, void* stream) {
    const int lg = ilog2_exact(nfft);
    BABE_CHECK_ARG(spec && H && frames_out && tw4096 && B > 0 && frames > 0, "spec_filter_istft: bad arguments");
    BABE_CHECK_ARG(lg >= 8 && lg <= 12, "spec_filter_istft: nfft=%d unsupported", nfft);
    BabeProfScope prof(BABE_SLOT_ISTFT, (double)B * frames * (8.0 * (nfft / 2 + 1) + 4.0 * nfft), 0, 0, stream);
    hipLaunchKernelGGL(spec_filter_istft_kernel, dim3(frames, B), dim3(256), 0, (hipStream_t)stream, spec, H, H_bs,
                       frames_out, lg, frames, reinterpret_cast<const float2*>(tw4096));
    BABE_LAUNCH_CHECK();
    return BABE_OK;
}

extern "C" int babe_ola(const float* frames_in, const float* post, const float* y, long y_bs, float* out, long out_bs,
                        double* part, int nblk, int B, int L, int nfft, int frames, void* stream) {
    BABE_CHECK_ARG(frames_in && out && B > 0 && L > 0 && nblk > 0, "ola: bad arguments");
    BABE_CHECK_ARG(!y || part, "ola: residual mode needs a partial-sum buffer");
    BabeProfScope prof(BABE_SLOT_ISTFT, (double)B * (4.0 * frames * nfft + 4.0 * L * (y ? 2 : 1)), 0, 0, stream);
    hipLaunchKernelGGL(ola_kernel, dim3(nblk, B), dim3(256), 0, (hipStream_t)stream, frames_in, post, y, y_bs, out,
                       out_bs, y ? part : nullptr, nblk, L, nfft, frames);
    BABE_LAUNCH_CHECK();
    return BABE_OK;
}

extern "C" int babe_residual_seed(const float* r, long r_bs, const double* part, int nblk, const float* post,
                                  float* out, long out_bs, int B, int L, void* stream) {
    BABE_CHECK_ARG(r && part && out && B > 0 && L > 0, "residual_seed: bad arguments");
    BabeProfScope prof(BABE_SLOT_ISTFT, 8.0 * B * (double)L, 0, 0, stream);
    int bx = cdiv(L, 1024);
    hipLaunchKernelGGL(residual_seed_kernel, dim3(bx, B), dim3(256), 0, (hipStream_t)stream, r, r_bs, part, nblk, post,
                       out, out_bs, L);
    BABE_LAUNCH_CHECK();
    return BABE_OK;
}

extern "C" int babe_stft_dist_partial(const float* X, const float* R, const float* w, double* part, int nblk, int B,
                                      int nbins, int frames, int mode, void* stream) {
    BABE_CHECK_ARG(X && R && w && part && nblk > 0 && B > 0 && nbins > 1 && frames > 0 && mode >= 0 && mode <= 2,
                   "stft_dist_partial: bad arguments");
    BabeProfScope prof(BABE_SLOT_SAMPLER, 16.0 * B * (double)nbins * frames, 0, 0, stream);
    hipLaunchKernelGGL(stft_dist_partial_kernel, dim3(nblk, B), dim3(256), 0, (hipStream_t)stream, X, R, w, part, nblk, nbins,
                       (long)nbins * frames, mode);
    BABE_LAUNCH_CHECK();
    return BABE_OK;
}

extern "C" int babe_stft_dist_grad(const float* X, const float* R, const float* w, const double* part, int nblk, float* G,
                                   int B, int nbins, int frames, int mode, int shared, void* stream) {
    BABE_CHECK_ARG(X && R && w && part && G && nblk > 0 && B > 0 && nbins > 1 && frames > 0 && mode >= 0 && mode <= 2,
                   "stft_dist_grad: bad arguments");
    BabeProfScope prof(BABE_SLOT_SAMPLER, 24.0 * B * (double)nbins * frames, 0, 0, stream);
    const long n = (long)nbins * frames;
    hipLaunchKernelGGL(stft_dist_grad_kernel, dim3(cdiv(n, 1024), B), dim3(256), 0, (hipStream_t)stream, X, R, w, part, nblk,
                       G, nbins, n, mode, shared, B);
    BABE_LAUNCH_CHECK();
    return BABE_OK;
}

extern "C" int babe_stft_mag_stats(const float* specX, const float* specY, double* stats, int B, int nbins, int frames,
                                   int shared, void* stream) {
    BABE_CHECK_ARG(specX && specY && stats && B > 0 && nbins > 1 && frames > 0, "stft_mag_stats: bad arguments");
    BabeProfScope prof(BABE_SLOT_MAG_STATS, 16.0 * B * (double)nbins * frames, 0, 0, stream);
    hipLaunchKernelGGL(mag_stats_kernel, dim3(cdiv(nbins, 256), shared ? 1 : B), dim3(256), 0, (hipStream_t)stream,
                       specX, specY, stats, B, nbins, frames, shared);
    BABE_LAUNCH_CHECK();
    return BABE_OK;
}

extern "C" int babe_design_filter(const float* params, float* H, int P, int K, int nbins, float fs, int nfft,
                                  void* stream) {
    BABE_CHECK_ARG(params && H && P > 0 && K > 0 && K <= KMAX && nbins > 1, "design_filter: bad arguments (K<=8)");
    BabeProfScope prof(BABE_SLOT_FILTER_FIT, 4.0 * P * (double)nbins, 0, 0, stream);
    const float df = fs / (float)nfft;
    hipLaunchKernelGGL(design_filter_kernel, dim3(P), dim3(256), 0, (hipStream_t)stream, params, H, K, nbins, df);
    BABE_LAUNCH_CHECK();
    return BABE_OK;
}

extern "C" int babe_filter_fit(const double* stats, float* params, int* n_iter, int P, int K, int nbins, float fs,
                               int nfft, const babe_fit_cfg* cfg, void* stream) {
    BABE_CHECK_ARG(stats && params && cfg && P > 0 && K > 0 && K <= KMAX && nbins > 1, "filter_fit: bad arguments");
    BabeProfScope prof(BABE_SLOT_FILTER_FIT, 24.0 * P * (double)nbins, 0, 0, stream);
    const float df = fs / (float)nfft;
    BABE_CHECK_ARG(cfg->kernel == 0 || cfg->kernel == 1, "filter_fit: cfg.kernel %d (0 fast, 1 reference-order)", cfg->kernel);
    if (nbins <= FIT_NT * FIT_NB && cfg->kernel == 0) {
#define FIT_CASE(k)                                                                                                      \
    case k:                                                                                                              \
        hipLaunchKernelGGL(filter_fit_fast_kernel<k>, dim3(P), dim3(FIT_NT), 0, (hipStream_t)stream, stats, params, n_iter, \
                           nbins, df, *cfg);                                                                             \
        break;
        switch (K) {
            FIT_CASE(1) FIT_CASE(2) FIT_CASE(3) FIT_CASE(4) FIT_CASE(5) FIT_CASE(6) FIT_CASE(7) FIT_CASE(8)
        }
#undef FIT_CASE
    } else
        hipLaunchKernelGGL(filter_fit_kernel, dim3(P), dim3(256), 0, (hipStream_t)stream, stats, params, n_iter, K, nbins,
                           df, *cfg, 3L * nbins, (float*)nullptr);
    BABE_LAUNCH_CHECK();
    return BABE_OK;
}

extern "C" int babe_filter_loss_grad(const double* stats, long stats_pstride, const float* params, float* lossgrad, int P, int K,
                                     int nbins, float fs, int nfft, const babe_fit_cfg* cfg, void* stream) {
    BABE_CHECK_ARG(stats && params && lossgrad && cfg && P > 0 && K > 0 && K <= KMAX && nbins > 1 && stats_pstride >= 0,
                   "filter_loss_grad: bad arguments");
    BabeProfScope prof(BABE_SLOT_FILTER_FIT, 24.0 * P * (double)nbins, 0, 0, stream);
    babe_fit_cfg c = *cfg;
    c.max_iter = 1;
    hipLaunchKernelGGL(filter_fit_kernel, dim3(P), dim3(256), 0, (hipStream_t)stream, stats, const_cast<float*>(params),
                       (int*)nullptr, K, nbins, fs / (float)nfft, c, stats_pstride, lossgrad);
    BABE_LAUNCH_CHECK();
    return BABE_OK;
}
