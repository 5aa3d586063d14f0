// Small helpers: library info / errors, strided axpby (cat / slice / residual merges), Linear, RFF.
#include "common.h"
#include "../../include/babe_hip.h"
#include "prof.h"
#include <cstdarg>
#include <vector>

static thread_local char g_err[512] = "";

void babe_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* babe_last_error(void) { return g_err; }
extern "C" const char* babe_version(void) { return "babe_hip 0.1 (gfx950)"; }

// ---- measurement hook (prof.h) ---------------------------------------------------------------------------------------
namespace {
struct ProfRec { hipEvent_t e0, e1; int slot; void* stream; double flops; };
struct Prof {
    bool on = false;
    int conv_slot_override = -1;          // >= 0: conv launches are tallied there (the DFT stages of the CQT)
    std::vector<ProfRec> rec;
    size_t used = 0;
    double bytes[BABE_NSLOTS] = {}, flops[BABE_NSLOTS] = {}, exec[BABE_NSLOTS] = {};
    long dispatch[BABE_NSLOTS] = {};      // always-on launch counters
    bool open = false;
} g_prof;
const char* const kSlotNames[BABE_NSLOTS] = {
    "conv53_wino4", "conv53_wino2", "conv53_direct", "conv11", "conv_bf16", "dft_stage", "gn_stats", "scale_gelu",
    "gn_bwd_partial", "gn_bwd_apply", "resample", "axpby", "film", "cqt_band_analysis", "cqt_band_synthesis",
    "cqt_gather", "stft_fwd", "istft", "mag_stats", "filter_fit", "sampler", "denoiser", "conv53_fewco",
    "conv_bf16p", "conv53_wino45", "conv53_wino85"};
}  // namespace

extern "C" void babe_prof_begin(int slot, double bytes, double flops, double exec_flops, void* stream) {
    if (slot <= BABE_SLOT_CONV_BF16 && g_prof.conv_slot_override >= 0) slot = g_prof.conv_slot_override;
    g_prof.dispatch[slot]++;
    if (!g_prof.on) return;
    if (g_prof.used == g_prof.rec.size()) {
        ProfRec r;
        (void)hipEventCreate(&r.e0);
        (void)hipEventCreate(&r.e1);
        g_prof.rec.push_back(r);
    }
    ProfRec& r = g_prof.rec[g_prof.used];
    r.slot = slot;
    r.stream = stream;
    r.flops = flops;
    (void)hipEventRecord(r.e0, (hipStream_t)stream);
    g_prof.bytes[slot] += bytes;
    g_prof.flops[slot] += flops;
    g_prof.exec[slot] += exec_flops;
    g_prof.open = true;
}
extern "C" void babe_prof_end(void* stream) {
    if (!g_prof.open) return;
    (void)hipEventRecord(g_prof.rec[g_prof.used++].e1, (hipStream_t)stream);
    g_prof.open = false;
}
extern "C" int babe_prof_nslots(void) { return BABE_NSLOTS; }
extern "C" const char* babe_prof_slot_name(int slot) { return (slot >= 0 && slot < BABE_NSLOTS) ? kSlotNames[slot] : ""; }
/* on: start tallying (a fresh tally unless one is pending); off: stop, what was recorded stays until babe_prof_read() */
extern "C" int babe_prof_enable(int on) {
    if (on < 0) return g_prof.on ? 1 : 0;           // query (the host's graph capture asks the library, not a shadow flag)
    g_prof.on = on != 0;
    g_prof.open = false;
    return BABE_OK;
}
extern "C" int babe_prof_conv_slot(int slot) {
    g_prof.conv_slot_override = (slot >= 0 && slot < BABE_NSLOTS) ? slot : -1;
    return BABE_OK;
}
/* ms/bytes/flops/exec_flops/launches: arrays of babe_prof_nslots() entries; waits for the recorded events; resets. */
extern "C" int babe_prof_read(double* ms, double* bytes, double* flops, double* exec_flops, long* launches) {
    for (int i = 0; i < BABE_NSLOTS; ++i) {
        if (ms) ms[i] = 0;
        if (launches) launches[i] = 0;
    }
    for (size_t i = 0; i < g_prof.used; ++i) {
        ProfRec& r = g_prof.rec[i];
        if (hipEventSynchronize(r.e1) != hipSuccess) {
            babe_set_error("prof_read: hipEventSynchronize failed");
            return BABE_ERR_HIP;
        }
        float t = 0;
        (void)hipEventElapsedTime(&t, r.e0, r.e1);
        if (ms) ms[r.slot] += t;
        if (launches) launches[r.slot]++;
    }
    for (int i = 0; i < BABE_NSLOTS; ++i) {
        if (bytes) bytes[i] = g_prof.bytes[i];
        if (flops) flops[i] = g_prof.flops[i];
        if (exec_flops) exec_flops[i] = g_prof.exec[i];
        g_prof.bytes[i] = g_prof.flops[i] = g_prof.exec[i] = 0;
    }
    g_prof.used = 0;
    return BABE_OK;
}
/* Timeline of the pending records (call BEFORE babe_prof_read, which resets): for record i, t0_ms[i] / t1_ms[i] = GPU time of
 * its two events relative to the first record's start event (events of different streams share the device clock), slot[i],
 * lane[i] = index of its stream in order of first appearance, flops[i] = the algorithmic flops it was tallied with.  An
 * interval is [the stream reached the launch, the kernel finished]; lanes that run concurrently are NOT serialised (a
 * rocprofv3 kernel trace serialises them).  Returns the number of records written (<= cap), or a negative error. */
extern "C" long babe_prof_timeline(double* t0_ms, double* t1_ms, int* slot, int* lane, double* flops, long cap) {
    std::vector<void*> streams;
    long n = 0;
    for (size_t i = 0; i < g_prof.used && n < cap; ++i, ++n) {
        ProfRec& r = g_prof.rec[i];
        if (hipEventSynchronize(r.e1) != hipSuccess) {
            babe_set_error("prof_timeline: hipEventSynchronize failed");
            return BABE_ERR_HIP;
        }
        float a = 0, b = 0;
        (void)hipEventElapsedTime(&a, g_prof.rec[0].e0, r.e0);
        (void)hipEventElapsedTime(&b, g_prof.rec[0].e0, r.e1);
        size_t k = 0;
        while (k < streams.size() && streams[k] != r.stream) ++k;
        if (k == streams.size()) streams.push_back(r.stream);
        if (t0_ms) t0_ms[n] = a;
        if (t1_ms) t1_ms[n] = b;
        if (slot) slot[n] = r.slot;
        if (lane) lane[n] = (int)k;
        if (flops) flops[n] = r.flops;
    }
    return n;
}
extern "C" long babe_prof_pending(void) { return (long)g_prof.used; }
/* always-on launch counters per slot (which kernel a conv call really dispatched to); reset != 0 clears them */
extern "C" int babe_prof_dispatch_counts(long* counts, int reset) {
    for (int i = 0; i < BABE_NSLOTS; ++i) {
        if (counts) counts[i] = g_prof.dispatch[i];
        if (reset) g_prof.dispatch[i] = 0;
    }
    return BABE_OK;
}

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

// out = alpha*x + beta*y over the same [B][C] planes (out may be a strided sub-view; x, y dense or strided): the residual merge
// (x + h)/sqrt2 at the end of a ResnetBlock without res_conv (networks/cqtdiff+.py:493) in ONE pass - it was axpby(z -> out) followed
// by axpby(x, out += ...), 20 bytes per element instead of 12.  Rounding as the two-pass form: fl(fl(alpha x) + fl(beta y)).
__global__ __launch_bounds__(256) void axpby2_4d_kernel(const float* __restrict__ xin, long x_bs, long x_cs,
                                                        const float* __restrict__ yin, long y_bs, long y_cs,
                                                        float* __restrict__ out, long out_bs, long out_cs, int C, long n,
                                                        float alpha, float beta) {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const int b = blockIdx.y / C, c = blockIdx.y % C;
    const float* x = xin + (long)b * x_bs + (long)c * x_cs;
    const float* y = yin + (long)b * y_bs + (long)c * y_cs;
    float* o = out + (long)b * out_bs + (long)c * out_cs;
    const long stride = (long)gridDim.x * blockDim.x * 4;
    for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
        const f32x4 xv = *reinterpret_cast<const f32x4*>(x + i), yv = *reinterpret_cast<const f32x4*>(y + i);
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
#pragma clang fp contract(off)                     // (HIP's __fmul_rn / __fadd_rn are plain operators: they would be fused)
            const float p = alpha * xv[e], q = beta * yv[e];
            v[e] = p + q;
        }
        *reinterpret_cast<f32x4*>(o + i) = v;
    }
}

// out[b][c][0 .. n) = value over [B][C] planes of n = F*T contiguous floats (zeroing the frequency sub-view a gradient is
// accumulated into: unet_engine.py's gR[:, :, :bpo, :].zero_()).  grid: (blocks, B*C)
__global__ __launch_bounds__(256) void fill4d_kernel(float* __restrict__ out, long out_bs, long out_cs, int C, long n, float value) {
    const int b = blockIdx.y / C, c = blockIdx.y % C;
    float* y = out + (long)b * out_bs + (long)c * out_cs;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) y[i] = value;
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
    BabeProfScope prof(BABE_SLOT_AXPBY, (beta != 0.f ? 12.0 : 8.0) * B * C * (double)n, 0, 0, stream);
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

extern "C" int babe_axpby2_4d(const float* x, long x_bs, long x_cs, const float* y, long y_bs, long y_cs, float* out, long out_bs,
                              long out_cs, int B, int C, int F, int T, float alpha, float beta, void* stream) {
    BABE_CHECK_ARG(x && y && out && B > 0 && C > 0 && F > 0 && T > 0, "axpby2_4d: bad arguments");
    BABE_CHECK_ARG((long)B * C <= 65535, "axpby2_4d: grid too large");
    const long n = (long)F * T;
    auto al16 = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
    BABE_CHECK_ARG(n % 4 == 0 && al16(x) && al16(y) && al16(out) && x_bs % 4 == 0 && x_cs % 4 == 0 && y_bs % 4 == 0 && y_cs % 4 == 0 &&
                       out_bs % 4 == 0 && out_cs % 4 == 0,
                   "axpby2_4d: planes must be 16-byte aligned with F*T %% 4 == 0 (use two babe_axpby4d calls otherwise)");
    BabeProfScope prof(BABE_SLOT_AXPBY, 12.0 * B * C * (double)n, 0, 0, stream);
    int bx = cdiv(n / 4, 256);
    if (bx > 64) bx = 64;
    hipLaunchKernelGGL(axpby2_4d_kernel, dim3(bx, B * C), dim3(256), 0, (hipStream_t)stream, x, x_bs, x_cs, y, y_bs, y_cs, out, out_bs,
                       out_cs, C, n, alpha, beta);
    BABE_LAUNCH_CHECK();
    return BABE_OK;
}

extern "C" int babe_fill4d(float* out, long out_bs, long out_cs, int B, int C, int F, int T, float value, void* stream) {
    BABE_CHECK_ARG(out && B > 0 && C > 0 && F > 0 && T > 0 && (long)B * C <= 65535, "fill4d: bad arguments");
    const long n = (long)F * T;
    int bx = cdiv(n, 1024);
    if (bx > 64) bx = 64;
    hipLaunchKernelGGL(fill4d_kernel, dim3(bx, B * C), dim3(256), 0, (hipStream_t)stream, out, out_bs, out_cs, C, n, value);
    BABE_LAUNCH_CHECK();
    return BABE_OK;
}

extern "C" int babe_linear(const float* x, const float* W, const float* bias, float* out, int B, int K, int J,
                           int relu, void* stream) {
    BABE_CHECK_ARG(x && W && out && B > 0 && K > 0 && J > 0, "linear: bad arguments");
    BabeProfScope prof(BABE_SLOT_FILM, 4.0 * ((double)J * K + (double)B * (K + J)), 2.0 * B * J * K, 0, stream);
    hipLaunchKernelGGL(linear_kernel, dim3(cdiv(J, 4)), dim3(256), 0, (hipStream_t)stream, x, W, bias, out, B, K, J,
                       relu);
    BABE_LAUNCH_CHECK();
    return BABE_OK;
}

extern "C" int babe_rff(const float* cnoise, const float* freq, float* out, int B, int R, void* stream) {
    BABE_CHECK_ARG(cnoise && freq && out && B > 0 && R > 0, "rff: bad arguments");
    BabeProfScope prof(BABE_SLOT_FILM, 4.0 * B * (1 + 3 * R), 0, 0, stream);
    hipLaunchKernelGGL(rff_kernel, dim3(cdiv(B * R, 64)), dim3(64), 0, (hipStream_t)stream, cnoise, freq, out, B, R);
    BABE_LAUNCH_CHECK();
    return BABE_OK;
}
