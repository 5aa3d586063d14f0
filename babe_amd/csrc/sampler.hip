// Element-wise steps of the guided EDM sampler (testing/blind_bwe_sampler.py:503-516, :125-135, :701-761;
// diff_params/edm.py:144-159).  Trivially HBM-bound: float4 streaming, double partial sums.
#include "common.h"
#include "../../include/babe_hip.h"
#include "prof.h"

namespace {

__global__ __launch_bounds__(256) void lincomb3_kernel(float* __restrict__ out, float a, const float* __restrict__ x,
                                                       float b, const float* __restrict__ y, float c,
                                                       const float* __restrict__ z, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float v = a * x[i];
        if (y) v += b * y[i];
        if (z) v += c * z[i];
        out[i] = v;
    }
}

__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ g, long g_bs,
                                                            double* __restrict__ part, int nblk, long n) {
    __shared__ double sh[4];
    const int b = blockIdx.y;
    const float* p = g + (long)b * g_bs;
    double acc = 0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const double v = p[i];
        acc += v * v;
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part[(long)b * nblk + blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}

// Observation-noise regularisation (posterior_sampling.SNR_observations, testing/blind_bwe_sampler.py:80-86 and :542-548):
//     y += sqrt(var(y, -1) / snr) * noise        per clip, var = the unbiased sample variance, IN PLACE (the reference mutates the
// observations at every call, so the noise accumulates over the sampling loop).  One workgroup of 1024 threads per clip: mean, then
// the sum of squared deviations (both in double, two passes over the 1.4 MB clip out of L2), then the update.  An optional
// regulariser off the benchmark's path (conf/tester/blind_bwe_2.yaml sets it): built for exactness, not for bandwidth.
__global__ __launch_bounds__(1024) void add_obs_noise_kernel(float* __restrict__ y, long y_bs, const float* __restrict__ noise,
                                                             long n_bs, float snr, long n) {
    __shared__ double sh[16];
    __shared__ double res;
    float* p = y + (long)blockIdx.x * y_bs;
    const float* q = noise + (long)blockIdx.x * n_bs;
    auto block_sum = [&](double v) {
        v = wave_sum(v);
        __syncthreads();                                   // (sh / res of the previous call are no longer read)
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
        __syncthreads();
        if (threadIdx.x == 0) {
            double t = 0;
            for (int w = 0; w < 16; ++w) t += sh[w];
            res = t;
        }
        __syncthreads();
        return res;
    };
    double acc = 0;
    for (long i = threadIdx.x; i < n; i += 1024) acc += (double)p[i];
    const double mean = block_sum(acc) / (double)n;
    acc = 0;
    for (long i = threadIdx.x; i < n; i += 1024) {
        const double d = (double)p[i] - mean;
        acc += d * d;
    }
    const float var = (float)(block_sum(acc) / (double)(n - 1));
    const float sigma = sqrtf(var / snr);
    for (long i = threadIdx.x; i < n; i += 1024) p[i] = p[i] + sigma * q[i];
}

// Alternative guidance distances of get_rec_grads (testing/blind_bwe_sampler.py:99-103).  r = y - rec is what the
// filter / overlap-add kernels hand over; rec = y - r.
// cosine: partial sums (rec.rec, rec.y, y.y) per block
__global__ __launch_bounds__(256) void cos_partial_kernel(const float* __restrict__ r, long r_bs,
                                                          const float* __restrict__ y, long y_bs,
                                                          double* __restrict__ part, int nblk, long n) {
    __shared__ double sh[4][3];
    const int b = blockIdx.y;
    const float* pr = r + (long)b * r_bs;
    const float* py = y + (long)b * y_bs;
    double a0 = 0, a1 = 0, a2 = 0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const double yv = py[i], rc = yv - (double)pr[i];
        a0 += rc * rc;
        a1 += rc * yv;
        a2 += yv * yv;
    }
    a0 = wave_sum(a0);
    a1 = wave_sum(a1);
    a2 = wave_sum(a2);
    if ((threadIdx.x & 63) == 0) {
        sh[threadIdx.x >> 6][0] = a0;
        sh[threadIdx.x >> 6][1] = a1;
        sh[threadIdx.x >> 6][2] = a2;
    }
    __syncthreads();
    if (threadIdx.x < 3)
        part[((long)b * nblk + blockIdx.x) * 3 + threadIdx.x] =
            sh[0][threadIdx.x] + sh[1][threadIdx.x] + sh[2][threadIdx.x] + sh[3][threadIdx.x];
}

// seed = d(distance)/d(rec) [* post]:  mode 1 smooth-L1 (sum reduction): -clamp(r / beta, -1, 1);
// mode 2 cosine, distance = clamp(1 - cos(rec, y), min 0): -(y / (|rec||y|) - cos * rec / |rec|^2) where 1 - cos >= 0
__global__ __launch_bounds__(256) void alt_seed_kernel(const float* __restrict__ r, long r_bs,
                                                       const float* __restrict__ y, long y_bs,
                                                       const double* __restrict__ part, int nblk,
                                                       const float* __restrict__ post, float* __restrict__ out,
                                                       long out_bs, int L, int mode, float beta) {
    const int b = blockIdx.y;
    float ca = 0.f, cb = 0.f;
    if (mode == 2) {
        double s0 = 0, s1 = 0, s2 = 0;
        for (int i = 0; i < nblk; ++i) {
            s0 += part[((long)b * nblk + i) * 3];
            s1 += part[((long)b * nblk + i) * 3 + 1];
            s2 += part[((long)b * nblk + i) * 3 + 2];
        }
        const double den = sqrt(s0) * sqrt(s2);
        if (den > 0) {
            const double c = s1 / den;
            if (1.0 - c >= 0.0) {
                ca = (float)(1.0 / den);
                cb = (float)(c / s0);
            }
        }
    }
    const float ib = 1.f / beta;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < L; i += (long)gridDim.x * blockDim.x) {
        const float rv = r[(long)b * r_bs + i];
        float v;
        if (mode == 1) {
            v = -fminf(fmaxf(rv * ib, -1.f), 1.f);
        } else if (mode == 3) {
            v = rv;                                    // r already IS d(distance)/d(rec) (STFT-domain distances): only * post
        } else {
            const float yv = y[(long)b * y_bs + i];
            v = -(yv * ca - (yv - rv) * cb);
        }
        if (post) v *= post[i];
        out[(long)b * out_bs + i] = v;
    }
}

__global__ __launch_bounds__(256) void score_direction_kernel(const float* __restrict__ xden,
                                                              const float* __restrict__ xhat,
                                                              const float* __restrict__ g,
                                                              const double* __restrict__ part, int nblk,
                                                              float* __restrict__ d, float t, float xi,
                                                              float sqrt_len, int shared_norm, int mode, int B,
                                                              long n) {
    const int b = blockIdx.y;
    double s2 = 0;
    if (shared_norm) {
        for (int i = 0; i < B * nblk; ++i) s2 += part[i];
    } else {
        for (int i = 0; i < nblk; ++i) s2 += part[(long)b * nblk + i];
    }
    const float normguide = (float)sqrt(s2) / sqrt_len;
    const float s = (mode == 0) ? xi / (normguide + 1e-6f) / t : xi / (normguide * t + 1e-6f);
    const float it2 = 1.f / (t * t);
    const long base = (long)b * n;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float score = (xden[base + i] - xhat[base + i]) * it2 - s * g[base + i];
        d[base + i] = -t * score;
    }
}

// grid (ceil(L/1024), B): each block produces 1024 outputs from 1024+ntaps-1 inputs staged in LDS
__global__ __launch_bounds__(256) void fir_same_kernel(const float* __restrict__ x, long x_bs,
                                                       const float* __restrict__ taps, int ntaps,
                                                       float* __restrict__ out, long out_bs, int L, int adjoint) {
    extern __shared__ float sh[];
    float* st = sh;                 // taps
    float* sx = sh + ntaps;         // input window
    const int b = blockIdx.y;
    const int n0 = blockIdx.x * 1024;
    const int padl = (ntaps - 1) / 2;
    // forward: out[n] = sum_k taps[k] x[n+k-padl]; adjoint: out[m] = sum_k taps[k] g[m-k+padl]
    const int lo = adjoint ? n0 - (ntaps - 1) + padl : n0 - padl;
    const int nin = 1024 + ntaps - 1;
    for (int i = threadIdx.x; i < ntaps; i += blockDim.x) st[i] = taps[i];
    for (int i = threadIdx.x; i < nin; i += blockDim.x) {
        const int s = lo + i;
        sx[i] = (s >= 0 && s < L) ? x[(long)b * x_bs + s] : 0.f;
    }
    __syncthreads();
    for (int j = threadIdx.x; j < 1024; j += blockDim.x) {
        const int n = n0 + j;
        if (n >= L) break;
        float acc = 0.f;
        if (!adjoint) {
            for (int k = 0; k < ntaps; ++k) acc += st[k] * sx[j + k];
        } else {
            for (int k = 0; k < ntaps; ++k) acc += st[k] * sx[j + (ntaps - 1) - k];
        }
        out[(long)b * out_bs + n] = acc;
    }
}

__global__ __launch_bounds__(256) void mask_blend_kernel(float* __restrict__ out, const float* __restrict__ mask,
                                                         long mask_bs, const float* __restrict__ a,
                                                         const float* __restrict__ b_, long n) {
    const int b = blockIdx.y;
    const float* m = mask + (long)b * mask_bs;
    const long base = (long)b * n;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float mv = m[i];
        float v = 0.f;
        if (a) v += mv * a[base + i];
        if (b_) v += (1.f - mv) * b_[base + i];
        out[base + i] = v;
    }
}

}  // namespace

extern "C" int babe_mask_blend(float* out, const float* mask, long mask_bs, const float* a, const float* b_, int B,
                               long n, void* stream) {
    BABE_CHECK_ARG(out && mask && (a || b_) && B > 0 && n > 0, "mask_blend: bad arguments");
    BabeProfScope prof(BABE_SLOT_SAMPLER, 4.0 * B * (double)n * (2 + (a ? 1 : 0) + (b_ ? 1 : 0)), 0, 0, stream);
    int bx = cdiv(n, 1024);
    if (bx > 1024) bx = 1024;
    hipLaunchKernelGGL(mask_blend_kernel, dim3(bx, B), dim3(256), 0, (hipStream_t)stream, out, mask, mask_bs, a, b_, n);
    BABE_LAUNCH_CHECK();
    return BABE_OK;
}

extern "C" int babe_fir_same(const float* x, long x_bs, const float* taps, int ntaps, float* out, long out_bs, int B,
                             int L, int adjoint, void* stream) {
    BABE_CHECK_ARG(x && taps && out && B > 0 && L > 0 && ntaps > 0 && ntaps <= 4096, "fir_same: bad arguments");
    BabeProfScope prof(BABE_SLOT_SAMPLER, 8.0 * B * (double)L, 2.0 * B * (double)L * ntaps, 0, stream);
    const size_t lds = (size_t)(ntaps + 1024 + ntaps - 1) * sizeof(float);
    hipLaunchKernelGGL(fir_same_kernel, dim3(cdiv(L, 1024), B), dim3(256), lds, (hipStream_t)stream, x, x_bs, taps,
                       ntaps, out, out_bs, L, adjoint);
    BABE_LAUNCH_CHECK();
    return BABE_OK;
}

extern "C" int babe_add_obs_noise(float* y, long y_bs, const float* noise, long noise_bs, float snr, int B, long n, void* stream) {
    BABE_CHECK_ARG(y && noise && B > 0 && n > 1 && snr > 0.f, "add_obs_noise: bad arguments");
    hipLaunchKernelGGL(add_obs_noise_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, y, y_bs, noise, noise_bs, snr, n);
    BABE_LAUNCH_CHECK();
    return BABE_OK;
}

extern "C" int babe_lincomb3(float* out, float a, const float* x, float b, const float* y, float c, const float* z,
                             long n, void* stream) {
    BABE_CHECK_ARG(out && x && n > 0, "lincomb3: bad arguments");
    BabeProfScope prof(BABE_SLOT_SAMPLER, 4.0 * (double)n * (2 + (y ? 1 : 0) + (z ? 1 : 0)), 0, 0, stream);
    int bx = cdiv(n, 1024);
    if (bx > 2048) bx = 2048;
    hipLaunchKernelGGL(lincomb3_kernel, dim3(bx), dim3(256), 0, (hipStream_t)stream, out, a, x, b, y, c, z, n);
    BABE_LAUNCH_CHECK();
    return BABE_OK;
}

extern "C" int babe_sumsq_partial(const float* g, long g_bs, double* part, int nblk, int B, long n, void* stream) {
    BABE_CHECK_ARG(g && part && nblk > 0 && B > 0 && n > 0, "sumsq_partial: bad arguments");
    BabeProfScope prof(BABE_SLOT_SAMPLER, 4.0 * B * (double)n, 0, 0, stream);
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3(nblk, B), dim3(256), 0, (hipStream_t)stream, g, g_bs, part, nblk, n);
    BABE_LAUNCH_CHECK();
    return BABE_OK;
}

extern "C" int babe_cos_partial(const float* r, long r_bs, const float* y, long y_bs, double* part, int nblk, int B,
                                long n, void* stream) {
    BABE_CHECK_ARG(r && y && part && nblk > 0 && B > 0 && n > 0, "cos_partial: bad arguments");
    BabeProfScope prof(BABE_SLOT_SAMPLER, 8.0 * B * (double)n, 0, 0, stream);
    hipLaunchKernelGGL(cos_partial_kernel, dim3(nblk, B), dim3(256), 0, (hipStream_t)stream, r, r_bs, y, y_bs, part, nblk, n);
    BABE_LAUNCH_CHECK();
    return BABE_OK;
}

extern "C" int babe_residual_seed_alt(const float* r, long r_bs, const float* y, long y_bs, const double* part, int nblk,
                                      const float* post, float* out, long out_bs, int B, int L, int mode, float beta,
                                      void* stream) {
    BABE_CHECK_ARG(r && out && B > 0 && L > 0 && mode >= 1 && mode <= 3, "residual_seed_alt: bad arguments");
    BABE_CHECK_ARG(mode != 1 || beta > 0.f, "residual_seed_alt: mode 1 needs beta > 0");
    BABE_CHECK_ARG(mode != 2 || (y && part && nblk > 0), "residual_seed_alt: mode 2 needs y and the cos_partial sums");
    BabeProfScope prof(BABE_SLOT_SAMPLER, 12.0 * B * (double)L, 0, 0, stream);
    int bx = cdiv(L, 1024);
    hipLaunchKernelGGL(alt_seed_kernel, dim3(bx, B), dim3(256), 0, (hipStream_t)stream, r, r_bs, y, y_bs, part, nblk, post,
                       out, out_bs, L, mode, beta);
    BABE_LAUNCH_CHECK();
    return BABE_OK;
}

extern "C" int babe_score_direction(const float* xden, const float* xhat, const float* g, const double* part, int nblk,
                                    float* d, float t, float xi, float audio_len, int shared_norm, int mode, int B,
                                    long n, void* stream) {
    BABE_CHECK_ARG(xden && xhat && g && part && d && B > 0 && n > 0 && t > 0, "score_direction: bad arguments");
    BabeProfScope prof(BABE_SLOT_SAMPLER, 16.0 * B * (double)n, 0, 0, stream);
    int bx = cdiv(n, 1024);
    if (bx > 1024) bx = 1024;
    hipLaunchKernelGGL(score_direction_kernel, dim3(bx, B), dim3(256), 0, (hipStream_t)stream, xden, xhat, g, part,
                       nblk, d, t, xi, sqrtf(audio_len), shared_norm, mode, B, n);
    BABE_LAUNCH_CHECK();
    return BABE_OK;
}
