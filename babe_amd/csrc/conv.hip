// Frequency-dilated Conv2d "same" (and 1x1 convs) as an fp32 MFMA implicit GEMM for gfx950.
//
// Replaces F.conv2d in /root/reference/networks/cqtdiff+.py:79-88 (ResnetBlock.H :433-436,
// proj_in/proj_out/res_conv :412-415, pyr_down_proj :676) and, with flipped/transposed packed
// weights, autograd's convolution_backward w.r.t. the input.
//
//   D[co][pos] = sum_k W[co][k] * X[k][pos],   k = (kh, kw, ci)
//
// MFMA v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain): A operand = packed weights (rows = co),
// B operand = activations (cols = 32 consecutive time steps) so that accumulator columns are
// consecutive positions and the epilogue stores are 128-B coalesced per half-wave.
// Workgroup = 4 waves; every wave owns WP*32 positions x NT*32 output channels; the block stages
// one (kh, 8-input-channel) slab of activations [KC][PR][PT+2] and weights [KW][KC][BN] in LDS
// per step.  Zero "same" padding is applied at staging time.
#include "common.h"
#include "../../include/babe_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

struct ConvGeom {
    int CinP, CoutP, pt_log2, pr_log2, tiles_t;
};

template <int NT, int WP, int KC>
__global__ __launch_bounds__(256) void conv_mfma_kernel(babe_conv_args a, ConvGeom g) {
    constexpr int BN = NT * 32;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int PT = 1 << g.pt_log2;
    const int PR = 1 << g.pr_log2;
    const int XROW = PT + 2;
    const int XCH = PR * XROW;
    float* Xs = smem;
    float* Ws = smem + ((KC * XCH + 3) & ~3);

    const int tile_t = blockIdx.x % g.tiles_t;
    const int tile_f = blockIdx.x / g.tiles_t;
    const int t0 = tile_t << g.pt_log2;
    const int f0 = tile_f << g.pr_log2;
    const int co0 = blockIdx.y * BN;
    const int b = blockIdx.z;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int h = lane >> 5;
    const int l31 = lane & 31;

    f32x16 acc[NT][WP];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < WP; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    int boff[WP];
#pragma unroll
    for (int wp = 0; wp < WP; ++wp) {
        const int p = (wave * WP + wp) * 32 + l31;
        boff[wp] = (p >> g.pt_log2) * XROW + (p & (PT - 1)) + h * XCH;
    }
    const int aoff = h * BN + l31;

    const int padt = a.KW >> 1;
    const int khc = a.KH >> 1;
    // staging geometry: lpr lanes per activation row
    const int lpr_log2 = g.pt_log2 < 6 ? g.pt_log2 : 6;
    const int lpr = 1 << lpr_log2;
    const int rows_per_iter = 256 >> lpr_log2;
    const int nrows = KC << g.pr_log2;
    const int srow0 = tid >> lpr_log2;
    const int scol = tid & (lpr - 1);
    const int cin_split = a.in2 ? a.cin_split : a.Cin;

    for (int kh = 0; kh < a.KH; ++kh) {
        const int foff = (kh - khc) * a.dil;
        // skip taps that fall entirely outside the frequency range for this tile
        if (f0 + foff + PR <= 0 || f0 + foff >= a.F) continue;
        for (int ci0 = 0; ci0 < g.CinP; ci0 += KC) {
            __syncthreads();
            // ---- stage activations
            for (int row = srow0; row < nrows; row += rows_per_iter) {
                const int ci = ci0 + (row >> g.pr_log2);
                const int r = row & (PR - 1);
                const int f = f0 + r + foff;
                const bool rowok = (ci < a.Cin) && (f >= 0) && (f < a.F);
                const float* src = nullptr;
                float sc = 1.f;
                if (rowok) {
                    src = (ci < cin_split) ? a.in + (long)b * a.in_bs + (long)ci * a.in_cs
                                           : a.in2 + (long)b * a.in2_bs + (long)(ci - cin_split) * a.in2_cs;
                    src += (long)f * a.T;
                    if (a.in_scale) sc = a.in_scale[b * a.Cin + ci];
                }
                float* dst = Xs + row * XROW;
                for (int tt = scol; tt < XROW; tt += lpr) {
                    const int t = t0 + tt - padt;
                    float v = 0.f;
                    if (rowok && t >= 0 && t < a.T) v = src[t] * sc;
                    dst[tt] = v;
                }
            }
            // ---- stage weights: KW*KC rows of BN contiguous floats
            {
                constexpr int V4 = BN / 4;
                const int nv = a.KW * KC * V4;
                for (int idx = tid; idx < nv; idx += 256) {
                    const int row = idx / V4;           // kw*KC + ci_l
                    const int c4 = idx - row * V4;
                    const int kw = row / KC;
                    const int ci_l = row - kw * KC;
                    const float4 v = *reinterpret_cast<const float4*>(
                        a.w_packed + ((long)((kh * a.KW + kw) * g.CinP + ci0 + ci_l)) * g.CoutP + co0 + c4 * 4);
                    *reinterpret_cast<float4*>(Ws + row * BN + c4 * 4) = v;
                }
            }
            __syncthreads();
            // ---- MFMA
            for (int kw = 0; kw < a.KW; ++kw) {
#pragma unroll
                for (int q = 0; q < KC / 2; ++q) {
                    float av[NT], bv[WP];
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) av[nt] = Ws[(kw * KC + 2 * q) * BN + aoff + nt * 32];
#pragma unroll
                    for (int wp = 0; wp < WP; ++wp) bv[wp] = Xs[2 * q * XCH + boff[wp] + kw];
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int wp = 0; wp < WP; ++wp)
                            acc[nt][wp] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[nt], bv[wp], acc[nt][wp], 0, 0, 0);
                }
            }
        }
    }

    // ---- epilogue
#pragma unroll
    for (int wp = 0; wp < WP; ++wp) {
        const int p = (wave * WP + wp) * 32 + l31;
        const int f = f0 + (p >> g.pt_log2);
        const int t = t0 + (p & (PT - 1));
        if (f >= a.F || t >= a.T) continue;
        const long sp = (long)f * a.T + t;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + nt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (co < a.Cout) {
                    float v = acc[nt][wp][r] * a.alpha;
                    if (a.oscale) v *= a.oscale[b * a.Cout + co];
                    if (a.res) v += a.rbeta * a.res[(long)b * a.res_bs + (long)co * a.res_cs + sp];
                    a.out[(long)b * a.out_bs + (long)co * a.out_cs + sp] = v;
                }
            }
        }
    }
}

__global__ void pack_weights_kernel(const float* __restrict__ w, float* __restrict__ dst, int Cout, int Cin, int KH,
                                    int KW, int tf, int CinP, int CoutP, long total) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int co = (int)(i % CoutP);
    long r = i / CoutP;
    const int ci = (int)(r % CinP);
    r /= CinP;
    const int kw = (int)(r % KW);
    const int kh = (int)(r / KW);
    float v = 0.f;
    if (!tf) {
        if (co < Cout && ci < Cin) v = w[(((long)co * Cin + ci) * KH + kh) * KW + kw];
    } else {
        // packed "Cout" = reference Cin, packed "Cin" = reference Cout
        if (co < Cin && ci < Cout) v = w[(((long)ci * Cin + co) * KH + (KH - 1 - kh)) * KW + (KW - 1 - kw)];
    }
    dst[i] = v;
}

inline int ilog2_floor(int v) {
    int l = 0;
    while ((1 << (l + 1)) <= v) ++l;
    return l;
}
inline int ilog2_ceil(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}

template <int NT, int WP>
int launch_conv(const babe_conv_args& a, ConvGeom g, hipStream_t s) {
    constexpr int KC = 8;
    constexpr int NPOS = 128 * WP;
    const int npos_log2 = ilog2_floor(NPOS);
    g.pt_log2 = ilog2_ceil(a.T);
    if (g.pt_log2 > npos_log2) g.pt_log2 = npos_log2;
    if (g.pt_log2 < 4) g.pt_log2 = 4;
    g.pr_log2 = npos_log2 - g.pt_log2;
    const int PT = 1 << g.pt_log2, PR = 1 << g.pr_log2;
    g.tiles_t = cdiv(a.T, PT);
    const int tiles_f = cdiv(a.F, PR);
    dim3 grid(g.tiles_t * tiles_f, g.CoutP / (NT * 32), a.B);
    const size_t lds = ((size_t)((KC * PR * (PT + 2) + 3) & ~3) + (size_t)3 * KC * NT * 32) * sizeof(float);
    hipLaunchKernelGGL((conv_mfma_kernel<NT, WP, KC>), grid, dim3(256), lds, s, a, g);
    return 0;
}

}  // namespace

// ---- optional per-launch timing of the conv kernel (bench.py roofline): HIP events on the launch stream
#include <vector>
namespace {
struct ConvProf {
    bool on = false;
    bool paused = false;
    std::vector<hipEvent_t> ev;      // pairs (start, stop)
    size_t used = 0;
    double flops = 0;
} g_prof;
}  // namespace

extern "C" int babe_conv_prof_enable(int on) {
    g_prof.on = on != 0;
    g_prof.used = 0;
    g_prof.flops = 0;
    return BABE_OK;
}

extern "C" int babe_conv_prof_pause(int paused) {
    g_prof.paused = paused != 0;
    return BABE_OK;
}

extern "C" int babe_conv_prof_read(double* ms_total, double* flops_total, long* launches) {
    double ms = 0;
    for (size_t i = 0; i + 1 < g_prof.used; i += 2) {
        if (hipEventSynchronize(g_prof.ev[i + 1]) != hipSuccess) {
            babe_set_error("conv_prof_read: hipEventSynchronize failed");
            return BABE_ERR_HIP;
        }
        float t = 0;
        hipEventElapsedTime(&t, g_prof.ev[i], g_prof.ev[i + 1]);
        ms += t;
    }
    if (ms_total) *ms_total = ms;
    if (flops_total) *flops_total = g_prof.flops;
    if (launches) *launches = (long)(g_prof.used / 2);
    g_prof.used = 0;
    g_prof.flops = 0;
    return BABE_OK;
}

static hipEvent_t prof_event() {
    if (g_prof.used == g_prof.ev.size()) {
        hipEvent_t e;
        hipEventCreate(&e);
        g_prof.ev.push_back(e);
    }
    return g_prof.ev[g_prof.used++];
}

extern "C" long babe_conv_packed_size(int Cout, int Cin, int KH, int KW, int transpose_flip) {
    const int co = transpose_flip ? Cin : Cout;
    const int ci = transpose_flip ? Cout : Cin;
    return (long)KH * KW * ((ci + 7) / 8 * 8) * ((co + 31) / 32 * 32);
}

extern "C" int babe_conv_pack_weights(const float* w, float* dst, int Cout, int Cin, int KH, int KW,
                                      int transpose_flip, void* stream) {
    BABE_CHECK_ARG(w && dst && Cout > 0 && Cin > 0 && KH > 0 && KW > 0, "conv_pack_weights: bad arguments");
    const int co = transpose_flip ? Cin : Cout;
    const int ci = transpose_flip ? Cout : Cin;
    const int CinP = (ci + 7) / 8 * 8, CoutP = (co + 31) / 32 * 32;
    const long total = (long)KH * KW * CinP * CoutP;
    hipLaunchKernelGGL(pack_weights_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, w, dst, Cout,
                       Cin, KH, KW, transpose_flip, CinP, CoutP, total);
    BABE_LAUNCH_CHECK();
    return BABE_OK;
}

extern "C" int babe_conv2d(const babe_conv_args* ap, void* stream) {
    BABE_CHECK_ARG(ap, "conv2d: null args");
    const babe_conv_args& a = *ap;
    BABE_CHECK_ARG(a.in && a.w_packed && a.out, "conv2d: null pointer");
    BABE_CHECK_ARG(a.B > 0 && a.Cin > 0 && a.Cout > 0 && a.F > 0 && a.T > 0, "conv2d: bad shape");
    BABE_CHECK_ARG((a.KH == 5 || a.KH == 1) && (a.KW == 3 || a.KW == 1) && a.dil >= 1,
                   "conv2d: kernel %dx%d unsupported (need 5x3 or 1x1)", a.KH, a.KW);
    BABE_CHECK_ARG(!a.in2 || (a.cin_split > 0 && a.cin_split < a.Cin), "conv2d: bad cin_split");
    ConvGeom g;
    g.CinP = (a.Cin + 7) / 8 * 8;
    g.CoutP = (a.Cout + 31) / 32 * 32;
    const int n32 = g.CoutP / 32;
    int NT = 1;
    for (int c = 4; c >= 1; --c)
        if (n32 % c == 0) { NT = c; break; }
    // positions per block: 256 (WP=2) unless that leaves the chip under-filled
    const long npos = (long)a.F * a.T;
    const long blocks256 = ((npos + 255) / 256) * (n32 / NT) * a.B;
    const bool wp2 = blocks256 >= 512;
    hipStream_t s = (hipStream_t)stream;
    const bool prof = g_prof.on && !g_prof.paused;
    if (prof) {
        hipEventRecord(prof_event(), s);
        g_prof.flops += 2.0 * a.B * (double)a.Cout * a.Cin * a.KH * a.KW * (double)a.F * a.T;
    }
    struct ProfStop {
        hipStream_t s;
        bool on;
        ~ProfStop() { if (on) hipEventRecord(prof_event(), s); }
    } prof_stop{s, prof};
    if (wp2) {
        switch (NT) {
            case 4: launch_conv<4, 2>(a, g, s); break;
            case 3: launch_conv<3, 2>(a, g, s); break;
            case 2: launch_conv<2, 2>(a, g, s); break;
            default: launch_conv<1, 2>(a, g, s); break;
        }
    } else {
        switch (NT) {
            case 4: launch_conv<4, 1>(a, g, s); break;
            case 3: launch_conv<3, 1>(a, g, s); break;
            case 2: launch_conv<2, 1>(a, g, s); break;
            default: launch_conv<1, 1>(a, g, s); break;
        }
    }
    BABE_LAUNCH_CHECK();
    return BABE_OK;
}
