// Sample-rate conversion of the file-level flows: torchaudio.functional.resample as published (Hann-windowed sinc,
// lowpass_filter_width 6, rolloff 0.99, polyphase by gcd), call sites /root/reference/testing/blind_bwe_tester.py:410,744,930 and
// /root/reference/testing/denoise_and_bwe_tester.py:282-289 (recording -> 22.05 kHz denoiser -> 16 kHz model, config #5).
// torchaudio pads the waveform by (width, width + orig) zeros and runs F.conv1d(x[:, None], kernel[new][1][2 width + orig],
// stride = orig): output sample m = i * new + j is sum_k kernel[j][k] * xpad[i * orig + k].  The published kernel is dense but
// the Hann window is exactly 0 beyond |t| = lowpass_filter_width, i.e. outside ~2 * width / ... taps per phase: the host passes
// the first and one-past-last non-zero tap of every phase (babe_amd/resample.py) and only those are multiplied - exact zeros
// change no sum.  HBM-bound and tiny (a 30 s file once per flow): one thread per output sample, taps through the scalar cache.
#include "common.h"
#include "../../include/babe_hip.h"
#include "prof.h"

namespace {

__global__ __launch_bounds__(256) void resample_sinc_kernel(const float* __restrict__ x, long x_bs, float* __restrict__ out,
                                                            long out_bs, long L_in, long L_out, const float* __restrict__ kern,
                                                            const int* __restrict__ krange, int orig, int new_, int width,
                                                            int taps) {
    const int b = blockIdx.y;
    const float* xb = x + (long)b * x_bs;
    float* ob = out + (long)b * out_bs;
    for (long m = (long)blockIdx.x * blockDim.x + threadIdx.x; m < L_out; m += (long)gridDim.x * blockDim.x) {
        const long i = m / new_;
        const int j = (int)(m - i * new_);
        const int k0 = krange[2 * j], k1 = krange[2 * j + 1];
        const float* kj = kern + (long)j * taps;
        const long p0 = i * orig - width;                   // input index of tap 0
        float acc = 0.f;
        for (int k = k0; k < k1; ++k) {
            const long p = p0 + k;
            const float v = (p >= 0 && p < L_in) ? xb[p] : 0.f;
            acc = fmaf(kj[k], v, acc);
        }
        ob[m] = acc;
    }
}

}  // namespace

extern "C" int babe_resample_sinc(const float* x, long x_bs, float* out, long out_bs, int B, long L_in, long L_out,
                                  const float* kernel, const int* krange, int orig, int new_, int width, void* stream) {
    BABE_CHECK_ARG(x && out && kernel && krange && B > 0 && L_in > 0 && L_out > 0, "resample_sinc: bad arguments");
    BABE_CHECK_ARG(orig > 0 && new_ > 0 && width > 0, "resample_sinc: orig=%d new=%d width=%d", orig, new_, width);
    // every output sample must lie inside the frames the padded convolution produces: floor(L_in / orig) + 1 frames of new_
    BABE_CHECK_ARG(L_out <= (L_in / orig + 1) * (long)new_, "resample_sinc: L_out %ld beyond the %ld samples the transform yields",
                   L_out, (L_in / orig + 1) * (long)new_);
    const int taps = 2 * width + orig;
    BabeProfScope prof(BABE_SLOT_SAMPLER, 4.0 * B * (double)(L_in + L_out), 0, 0, stream);
    long bx = (L_out + 255) / 256;
    if (bx > 4096) bx = 4096;
    hipLaunchKernelGGL(resample_sinc_kernel, dim3((unsigned)bx, B), dim3(256), 0, (hipStream_t)stream, x, x_bs, out, out_bs, L_in,
                       L_out, kernel, krange, orig, new_, width, taps);
    BABE_LAUNCH_CHECK();
    return BABE_OK;
}
