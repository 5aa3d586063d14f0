// Winograd F(4,3) along TIME for the frequency-dilated (5,3) Conv2d, fp32 MFMA.
//
// Same idea as conv_wino.hip (the 3 time taps of networks/cqtdiff+.py:433-436 are dense) with the larger tile: 4 outputs
// y[4j..4j+3] from the 6 inputs d = x[4j-1 .. 4j+4] with 6 multiplies per (kh, ci) - 1.5 per output instead of 2 for
// F(2,3) and 3 for the direct kernel.  Interpolation points 0, +-1, +-2, inf:
//   U = B^T d:  U0 = 4d0 - 5d2 + d4        V = G w:  V0 = w0/4
//               U1 = -4d1 - 4d2 + d3 + d4            V1 = -(w0 + w1 + w2)/6
//               U2 =  4d1 - 4d2 - d3 + d4            V2 = -(w0 - w1 + w2)/6
//               U3 = -2d1 - d2 + 2d3 + d4            V3 = w0/24 + w1/12 + w2/6
//               U4 =  2d1 - d2 - 2d3 + d4            V4 = w0/24 - w1/12 + w2/6
//               U5 =  4d1 - 5d3 + d5                 V5 = w2
//   M_p = sum over (kh, ci) of V_p U_p  (6 independent GEMMs on v_mfma_f32_32x32x2_f32)
//   y0 = M0+M1+M2+M3+M4   y1 = M1-M2+2M3-2M4   y2 = M1+M2+4M3+4M4   y3 = M1-M2+8M3-8M4+M5
// fp32 throughout; the larger transform constants cost about 2.5x the rounding error of F(2,3) (1e-6 relative on a
// 1280-term sum against 4e-7; direct 1.5e-7), far inside the 1e-3 RMS bound of the north star.
//
// Waves split the 6 phases 3 + 3 ("phase triples", like the phase pairs of conv_wino_pp_kernel): a workgroup has
// 2 x WR x WC waves; wave (tr, wr, wc) accumulates phases 3tr..3tr+2 of NTW row tiles over 32 units (= 128 time steps).
// LDS images XQ[triple][8][units][4], WQ[triple][8][BN][4] (the 4th float is padding): one 16-byte read per operand
// and K-step.  The weights need no transform, so they travel global -> LDS by LDS-DMA (global_load_lds_dwordx4: no
// registers, no ds_write); only the activations pass through registers for the input transform.  The output transform
// exchanges three partial sums per output through LDS, one row tile at a time.
// Measured (MI355X, 256 channels, F=384, T=128, B=2): F(2,3) 161 -> 183 TFLOP/s algorithmic on the same box; with all
// staging removed the loop runs at 234, i.e. the activation staging is what is left to hide.
#include "common.h"
#include "../../include/babe_hip.h"
#include "prof.h"
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

struct Wino4Geom {
    int CinP, CoutP, pt_log2, pr_log2, tiles_t;
};

// ABL: compile-time ablation bits for profiling builds (1 no global loads, 2 no LDS stores, 4 no barrier)
template <int NTW, int WR, int WC, int ABL = 0>
__global__ __launch_bounds__(128 * WR * WC, (WR * WC <= 2 ? 2 : 1)) void conv_wino4_kernel(babe_conv_args a, Wino4Geom g,
                                                                                        const float* __restrict__ wq) {
    constexpr int NTH = 128 * WR * WC;
    constexpr int KC = 8;
    constexpr int BN = WR * NTW * 32;
    constexpr int NUNIT = WC * 32;                      // units (4 outputs each) per tile
    constexpr int NXQ = KC * NUNIT;                     // input quads per slab (one per unit)
    constexpr int XJ = (NXQ + NTH - 1) / NTH;
    constexpr int NW4 = 2 * KC * BN;                    // weight float4 per slab
    constexpr int WJ = (NW4 + NTH - 1) / NTH;
    constexpr int XSZ = 2 * KC * NUNIT;                 // float4 units
    constexpr int BUF = XSZ + NW4;
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    f32x4* smem = reinterpret_cast<f32x4*>(smem_f);

    const int PT = 1 << g.pt_log2;
    const int PR = 1 << g.pr_log2;
    const int upr_log2 = g.pt_log2 - 2;                 // units per row
    const int tile_t = blockIdx.x % g.tiles_t;
    const int tile_f = blockIdx.x / g.tiles_t;
    const int t0 = tile_t << g.pt_log2;
    const int f0 = tile_f << g.pr_log2;
    const int co0 = blockIdx.y * BN;
    const int b = blockIdx.z;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tr = wave / (WR * WC);
    const int wr = (wave / WC) % WR, wc = wave % WC;
    const int h = lane >> 5;
    const int l31 = lane & 31;
    const int khc = a.KH >> 1;
    const int cin_split = a.in2 ? a.cin_split : a.Cin;
    const float* isc = a.in_scale ? a.in_scale : a.in;
    const bool has_isc = a.in_scale != nullptr;

    f32x16 acc[NTW][3];
#pragma unroll
    for (int i = 0; i < NTW; ++i)
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][p][r] = 0.f;

    int xci[XJ], xrow[XJ], xt[XJ], xlds[XJ];
#pragma unroll
    for (int v = 0; v < XJ; ++v) {
        const int idx = tid + v * NTH;
        const int i4 = idx & ((PT >> 2) - 1);
        xrow[v] = (idx >> upr_log2) & (PR - 1);
        xci[v] = idx >> (upr_log2 + g.pr_log2);         // >= KC for the idle tail threads
        xt[v] = t0 + 4 * i4;
        xlds[v] = xci[v] * NUNIT + (xrow[v] << upr_log2) + i4;
    }
    // weights need no transform: they go global -> LDS directly (LDS-DMA, no registers, no ds_write).  One wave
    // instruction fills 64 consecutive float4 of the slab [triple][8][BN] (= slab index idx); the source is per lane.
    static_assert(NW4 % NTH == 0, "weight slab must be a whole number of wave instructions per wave");
    int wsrc[WJ];
#pragma unroll
    for (int jj = 0; jj < WJ; ++jj) {
        const int idx = tid + jj * NTH;
        const int wt = idx / (KC * BN);
        const int rem = idx - wt * (KC * BN);
        const int ci_l = rem / BN;
        const int co_l = rem - ci_l * BN;
        wsrc[jj] = (ci_l * 2 + wt) * g.CoutP + co0 + co_l;       // float4 units relative to the slab row
    }
    f32x4 xv[XJ];
    float xl[XJ], xrr[XJ], xsc[XJ];
    bool xok[XJ], xlok[XJ], xrok[XJ];

    ChanSrc chan_ptr;
    chan_ptr.init(a.in, a.in_bs, a.in_cs, a.in2, a.in2_bs, a.in2_cs, cin_split, b);
    auto kh_valid = [&](int kh) {
        const int foff = (kh - khc) * a.dil;
        return !(f0 + foff + PR <= 0 || f0 + foff >= a.F);
    };
    auto load_chunk = [&](int kh, int ci0, f32x4* buf) {
        const int foff = (kh - khc) * a.dil;
#pragma unroll
        for (int v = 0; v < XJ; ++v) {
            const int cir = ci0 + xci[v];
            const int f = f0 + xrow[v] + foff;
            const bool ok = xci[v] < KC && cir < a.Cin && f >= 0 && f < a.F && xt[v] < a.T;
            const int ci = cir < a.Cin ? cir : a.Cin - 1;
            const float* src = chan_ptr(ci);
            const long off = ok ? (long)f * a.T + xt[v] : 1;
            xv[v] = *reinterpret_cast<const f32x4*>(src + (ok ? off : 0));
            const bool lok = ok && xt[v] > 0;
            const bool rok = ok && xt[v] + 4 < a.T;
            xl[v] = src[lok ? off - 1 : 0];          // raw; masked when the slab is written
            xrr[v] = src[rok ? off + 4 : 0];
            xlok[v] = lok;
            xrok[v] = rok;
            xsc[v] = has_isc ? isc[b * a.Cin + ci] : 1.f;
            xok[v] = ok;
        }
        const f32x4* wrow = reinterpret_cast<const f32x4*>(wq) + (long)(kh * g.CinP + ci0) * 2 * g.CoutP;
#pragma unroll
        for (int jj = 0; jj < WJ; ++jj)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wrow + wsrc[jj]),
                                             (__attribute__((address_space(3))) void*)(buf + XSZ + jj * NTH + wave * 64), 16,
                                             0, 0);
    };
    auto store_chunk = [&](f32x4* buf) {
#pragma unroll
        for (int v = 0; v < XJ; ++v) {
            if (xci[v] < KC) {
                f32x4 q0 = {0.f, 0.f, 0.f, 0.f}, q1 = {0.f, 0.f, 0.f, 0.f};
                if (xok[v]) {
                    const float s = xsc[v];
                    const float d0 = xlok[v] ? xl[v] * s : 0.f, d1 = xv[v][0] * s, d2 = xv[v][1] * s, d3 = xv[v][2] * s,
                                d4 = xv[v][3] * s, d5 = xrok[v] ? xrr[v] * s : 0.f;
                    const float e = d4 - 4.f * d2, o = d3 - 4.f * d1;        // shared by U1/U2
                    const float e2 = d4 - d2, o2 = 2.f * (d3 - d1);          // shared by U3/U4
                    q0 = f32x4{4.f * d0 - 5.f * d2 + d4, e + o, e - o, 0.f};
                    q1 = f32x4{e2 + o2, e2 - o2, 4.f * d1 - 5.f * d3 + d5, 0.f};
                }
                buf[xlds[v]] = q0;
                buf[KC * NUNIT + xlds[v]] = q1;
            }
        }
    };

    const int boff = (tr * KC + h) * NUNIT + wc * 32 + l31;
    const int aoff = XSZ + (tr * KC + h) * BN + wr * (NTW * 32) + l31;

    int kh = 0;
    while (!kh_valid(kh)) ++kh;
    int ci0 = 0;
    load_chunk(kh, ci0, smem);
    store_chunk(smem);
    __syncthreads();
    int cur = 0;
    while (true) {
        int nkh = kh, nci = ci0 + KC;
        if (nci >= g.CinP) {
            nci = 0;
            ++nkh;
            while (nkh < a.KH && !kh_valid(nkh)) ++nkh;
        }
        const bool has_next = nkh < a.KH;
        if (has_next && (!(ABL & 1) || ci0 == 0)) load_chunk(nkh, nci, smem + (cur ^ 1) * BUF);
        const f32x4* Xs = smem + cur * BUF;
        f32x4 av[2][NTW], bv[2];
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) av[0][nt] = Xs[aoff + nt * 32];
        bv[0] = Xs[boff];
#pragma unroll
        for (int st = 0; st < KC / 2; ++st) {
            const int c = st & 1;
            if (st + 1 < KC / 2) {
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) av[c ^ 1][nt] = Xs[aoff + 2 * (st + 1) * BN + nt * 32];
                bv[c ^ 1] = Xs[boff + 2 * (st + 1) * NUNIT];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
                for (int p = 0; p < 3; ++p)
                    acc[nt][p] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c][nt][p], bv[c][p], acc[nt][p], 0, 0, 0);
        }
        if (has_next && !(ABL & 2)) store_chunk(smem + (cur ^ 1) * BUF);
        if (!(ABL & 4)) __syncthreads();
        if (!has_next) break;
        kh = nkh;
        ci0 = nci;
        cur ^= 1;
    }

    // ---- output transform.  With a = (M0+M1+M2, M1-M2, M1+M2) from the triple-0 wave and b = (M3+M4, 2(M3-M4), M5)
    // from the triple-1 wave:  y0 = a0+b0, y1 = a1+b1, y2 = a2+4 b0, y3 = a1+4 b1+M5.  Row tile nt is finished by the
    // triple-(nt & 1) wave; the other wave of the pair passes its three sums through LDS (one row tile per round).
    const int q = wc * 32 + l31;
    const int f = f0 + (q >> upr_log2);
    const int t = t0 + 4 * (q & ((1 << upr_log2) - 1));
    const bool pv = f < a.F && t < a.T;
    const long sp = pv ? (long)f * a.T + t : 0;
    const bool has_os = a.oscale != nullptr, has_res = a.res != nullptr;
    float* ex = smem_f + (wr * WC + wc) * 3072 + lane;
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
        if (nt) __syncthreads();                           // the previous round's reads are done
        if (tr != (nt & 1)) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float m0 = acc[nt][0][r], m1 = acc[nt][1][r], m2 = acc[nt][2][r];
                if (tr == 0) {
                    ex[r * 64] = m0 + m1 + m2;
                    ex[1024 + r * 64] = m1 - m2;
                    ex[2048 + r * 64] = m1 + m2;
                } else {
                    ex[r * 64] = m0 + m1;
                    ex[1024 + r * 64] = 2.f * (m0 - m1);
                    ex[2048 + r * 64] = m2;
                }
            }
        }
        __syncthreads();
        if (tr == (nt & 1)) {
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                const int co_q = co0 + wr * (NTW * 32) + nt * 32 + 8 * qd + 4 * h;
                int cc[4];
                float os[4];
                f32x4 rr[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    cc[k] = co_q + k < a.Cout ? co_q + k : a.Cout - 1;
                    os[k] = has_os ? a.oscale[b * a.Cout + cc[k]] : 1.f;
                    rr[k] = has_res ? *reinterpret_cast<const f32x4*>(a.res + (long)b * a.res_bs + (long)cc[k] * a.res_cs + sp)
                                    : f32x4{0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int r = 4 * qd + k;
                    const float e0 = ex[r * 64], e1 = ex[1024 + r * 64], e2 = ex[2048 + r * 64];
                    const float m0 = acc[nt][0][r], m1 = acc[nt][1][r], m2 = acc[nt][2][r];
                    float a0, a1, a2, b0, b1, m5;
                    if (nt & 1) {              // own triple 1; received a
                        a0 = e0; a1 = e1; a2 = e2;
                        b0 = m0 + m1; b1 = 2.f * (m0 - m1); m5 = m2;
                    } else {                   // own triple 0; received b
                        a0 = m0 + m1 + m2; a1 = m1 - m2; a2 = m1 + m2;
                        b0 = e0; b1 = e1; m5 = e2;
                    }
                    f32x4 y = {a0 + b0, a1 + b1, a2 + 4.f * b0, a1 + 4.f * b1 + m5};
                    const float sc = a.alpha * os[k];
                    y = y * sc + a.rbeta * rr[k];
                    if (pv && co_q + k < a.Cout)
                        *reinterpret_cast<f32x4*>(a.out + (long)b * a.out_bs + (long)(co_q + k) * a.out_cs + sp) = y;
                }
            }
        }
    }
}

// dst [KH][CinP][2 triples][CoutP][4] (4th float of every triple is 0)
__global__ void pack_wino4_kernel(const float* __restrict__ w, float* __restrict__ dst, int Cout, int Cin, int KH, int tf,
                                  int CinP, int CoutP, long total) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int co = (int)(i % CoutP);
    long r = i / CoutP;
    const int ci = (int)(r % CinP);
    const int kh = (int)(r / CinP);
    double w0 = 0, w1 = 0, w2 = 0;
    if (!tf) {
        if (co < Cout && ci < Cin) {
            const float* p = w + (((long)co * Cin + ci) * KH + kh) * 3;
            w0 = p[0]; w1 = p[1]; w2 = p[2];
        }
    } else {
        if (co < Cin && ci < Cout) {      // packed "Cout" = reference Cin; taps flipped in both axes
            const float* p = w + (((long)ci * Cin + co) * KH + (KH - 1 - kh)) * 3;
            w0 = p[2]; w1 = p[1]; w2 = p[0];
        }
    }
    f32x4* d = reinterpret_cast<f32x4*>(dst) + ((long)(kh * CinP + ci) * 2) * CoutP + co;
    d[0] = f32x4{(float)(w0 / 4), (float)(-(w0 + w1 + w2) / 6), (float)(-(w0 - w1 + w2) / 6), 0.f};
    d[CoutP] = f32x4{(float)(w0 / 24 + w1 / 12 + w2 / 6), (float)(w0 / 24 - w1 / 12 + w2 / 6), (float)w2, 0.f};
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

template <int NTW, int WR, int WC, int ABL = 0>
void launch4(const babe_conv_args& a, Wino4Geom g, const float* wq, hipStream_t s) {
    constexpr int NPOS = 128 * WC;
    const int npos_log2 = ilog2_floor(NPOS);
    g.pt_log2 = ilog2_ceil(a.T);
    if (g.pt_log2 > npos_log2) g.pt_log2 = npos_log2;
    if (g.pt_log2 < 4) g.pt_log2 = 4;
    g.pr_log2 = npos_log2 - g.pt_log2;
    const int PT = 1 << g.pt_log2, PR = 1 << g.pr_log2;
    g.tiles_t = cdiv(a.T, PT);
    const int tiles_f = cdiv(a.F, PR);
    constexpr int BN = WR * NTW * 32;
    dim3 grid(g.tiles_t * tiles_f, g.CoutP / BN, a.B);
    size_t lds = 2 * (size_t)(2 * 8 * (WC * 32) + 2 * 8 * BN) * 16;
    const size_t ex = (size_t)WR * WC * 3072 * 4;
    if (ex > lds) lds = ex;
    hipLaunchKernelGGL((conv_wino4_kernel<NTW, WR, WC, ABL>), grid, dim3(128 * WR * WC), lds, s, a, g, wq);
}

}  // namespace


extern "C" long babe_conv_packed_size_wino4(int Cout, int Cin, int KH, int transpose_flip) {
    const int co = transpose_flip ? Cin : Cout;
    const int ci = transpose_flip ? Cout : Cin;
    return (long)KH * ((ci + 7) / 8 * 8) * ((co + 31) / 32 * 32) * 8;
}

extern "C" int babe_conv_pack_weights_wino4(const float* w, float* dst, int Cout, int Cin, int KH, int KW,
                                            int transpose_flip, void* stream) {
    BABE_CHECK_ARG(w && dst && Cout > 0 && Cin > 0 && KH > 0 && KW == 3, "conv_pack_weights_wino4: needs KW == 3");
    const int co = transpose_flip ? Cin : Cout;
    const int ci = transpose_flip ? Cout : Cin;
    const int CinP = (ci + 7) / 8 * 8, CoutP = (co + 31) / 32 * 32;
    const long total = (long)KH * CinP * CoutP;
    hipLaunchKernelGGL(pack_wino4_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, w, dst, Cout, Cin, KH,
                       transpose_flip, CinP, CoutP, total);
    BABE_LAUNCH_CHECK();
    return BABE_OK;
}

int babe_conv2d_wino4p_supported(const babe_conv_args& a);                              // conv_wino4p.hip
int babe_conv2d_wino4p_launch(const babe_conv_args& a, const float* w_wino4, hipStream_t s);

/* returns 1 if the F(4,3) kernel can run this problem (the caller then passes the wino4-packed weights) */
extern "C" int babe_conv2d_wino4_supported(const babe_conv_args* ap) {
    if (!ap) return 0;
    const babe_conv_args& a = *ap;
    auto al16 = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
    if (a.KW != 3 || a.KH < 1 || a.T % 4 != 0 || a.T < 16) return 0;
    if (!al16(a.in) || a.in_bs % 4 || a.in_cs % 4) return 0;
    if (a.in2 && (!al16(a.in2) || a.in2_bs % 4 || a.in2_cs % 4)) return 0;
    if (!al16(a.out) || a.out_bs % 4 || a.out_cs % 4) return 0;
    if (a.res && (!al16(a.res) || a.res_bs % 4 || a.res_cs % 4)) return 0;
    const int n32 = (a.Cout + 31) / 32;
    return (n32 == 2 || n32 == 3 || n32 % 4 == 0) ? 1 : 0;
}

extern "C" int babe_conv2d_wino4(const babe_conv_args* ap, const float* w_wino4, void* stream) {
    BABE_CHECK_ARG(ap && w_wino4, "conv2d_wino4: null args");
    BABE_CHECK_ARG(babe_conv2d_wino4_supported(ap), "conv2d_wino4: unsupported problem (use babe_conv2d_wino / babe_conv2d)");
    const babe_conv_args& a = *ap;
    Wino4Geom g;
    g.CinP = (a.Cin + 7) / 8 * 8;
    g.CoutP = (a.Cout + 31) / 32 * 32;
    const int n32 = g.CoutP / 32;
    hipStream_t s = (hipStream_t)stream;
    const double flops = babe_conv_flops(a);     // F(4,3): 6 multiplies per 4 outputs instead of 12
    BabeProfScope prof(BABE_SLOT_CONV53_WINO4, babe_conv_bytes(a), flops, flops * 0.5, stream);
    if (babe_conv2d_wino4p_supported(a)) babe_conv2d_wino4p_launch(a, w_wino4, s);   // pipelined 128 co x 256 pos
    else if (n32 == 2) launch4<2, 1, 2>(a, g, w_wino4, s);       //  64 co x 256 positions, 4 waves
    else if (n32 == 3) launch4<3, 1, 2>(a, g, w_wino4, s);       //  96 co x 256 positions, 4 waves
    else launch4<2, 2, 2>(a, g, w_wino4, s);                     // 128 co x 256 positions, 8 waves
    BABE_LAUNCH_CHECK();
    return BABE_OK;
}
