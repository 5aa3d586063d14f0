// Winograd F(2,3) along TIME for the frequency-dilated (5,3) Conv2d, fp32 MFMA.
//
// The (5,3) kernel of the CQTDiff+ dilation layers (networks/cqtdiff+.py:433-436) is dense over 3 consecutive time
// taps (dilation 1 along T).  For a pair of outputs y[2j], y[2j+1] and the inputs d = x[2j-1 .. 2j+2]:
//     U = (d0-d2, d1+d2, d2-d1, d1-d3)            input transform  (adds only, done while staging)
//     V = (w0, (w0+w1+w2)/2, (w0-w1+w2)/2, w2)    filter transform (done once when the weights are packed)
//     M_p = sum over (kh, ci) of V_p * U_p         4 independent GEMMs -> v_mfma_f32_32x32x2_f32
//     y[2j] = M0+M1+M2 ,  y[2j+1] = M1-M2-M3      output transform (epilogue)
// i.e. 4 multiplies per 2 outputs and tap-set instead of 6: 2/3 of the MFMA work of the direct kernel for the same
// result up to fp32 rounding (the transforms use only +-1 and 1/2).  The frequency taps (kh, dilation 2^d) stay a
// direct sum.  Same argument block / epilogue semantics as conv.hip; used for KH=5, KW=3, even T with 16-byte aligned
// rows, and output-channel counts that tile by the variants below; everything else falls back to conv.hip.
//
// LDS images (double buffered, one barrier per (kh, 8-channel) slab): XW[8][pairs][4 phases], WW[8][BN][4 phases]:
// one 16-byte read gives a lane all 4 phases of its B (resp. A) operand.
#include "common.h"
#include "../../include/babe_hip.h"
#include "prof.h"
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

namespace {

struct WinoGeom {
    int CinP, CoutP, pt_log2, pr_log2, tiles_t;
};

// ABL: compile-time ablation bits for profiling builds (1 no global loads, 2 no LDS stores, 4 no barrier, 8 no MFMA)
template <int NTW, int WR, int WC, int ABL = 0>
__global__ __launch_bounds__(64 * WR * WC, 2) void conv_wino_kernel(babe_conv_args a, WinoGeom g,
                                                                 const float* __restrict__ wq) {
    constexpr int NTH = 64 * WR * WC;
    constexpr int KC = 8;
    constexpr int BN = WR * NTW * 32;
    constexpr int NPAIR = WC * 32;
    constexpr int NPOS = 2 * NPAIR;
    constexpr int NXQ = KC * NPOS / 4;                  // activation quads per slab
    constexpr int XJ = (NXQ + NTH - 1) / NTH;
    constexpr int NWU = KC * BN;                        // weight units (float4 of 4 phases) per slab
    constexpr int WJ = (NWU + NTH - 1) / NTH;
    constexpr int BUF = KC * NPAIR + NWU;               // float4 units per buffer
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    f32x4* smem = reinterpret_cast<f32x4*>(smem_f);

    const int PT = 1 << g.pt_log2;
    const int PR = 1 << g.pr_log2;
    const int ppr_log2 = g.pt_log2 - 1;                 // pairs per row
    const int tile_t = blockIdx.x % g.tiles_t;
    const int tile_f = blockIdx.x / g.tiles_t;
    const int t0 = tile_t << g.pt_log2;
    const int f0 = tile_f << g.pr_log2;
    const int co0 = blockIdx.y * BN;
    const int b = blockIdx.z;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WC, wc = wave % WC;
    const int h = lane >> 5;
    const int l31 = lane & 31;
    const int khc = a.KH >> 1;
    const int cin_split = a.in2 ? a.cin_split : a.Cin;
    const float* isc = a.in_scale ? a.in_scale : a.in;
    const bool has_isc = a.in_scale != nullptr;

    f32x16 acc[NTW][4];
#pragma unroll
    for (int i = 0; i < NTW; ++i)
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][p][r] = 0.f;

    // ---- per-thread staging slots
    int xci[XJ], xrow[XJ], xt[XJ], xlds[XJ];
#pragma unroll
    for (int v = 0; v < XJ; ++v) {
        const int idx = tid + v * NTH;
        const int i4 = idx & ((PT >> 2) - 1);
        xrow[v] = (idx >> (g.pt_log2 - 2)) & (PR - 1);
        xci[v] = idx >> (g.pt_log2 + g.pr_log2 - 2);     // >= KC for the idle tail threads
        xt[v] = t0 + 4 * i4;
        xlds[v] = xci[v] * NPAIR + (xrow[v] << ppr_log2) + 2 * i4;
    }
    f32x4 xv[XJ], wr4[WJ];
    float xl[XJ], xrr[XJ], xsc[XJ];
    bool xok[XJ], xlok[XJ], xrok[XJ];

    ChanSrc chan_ptr;
    chan_ptr.init(a.in, a.in_bs, a.in_cs, a.in2, a.in2_bs, a.in2_cs, cin_split, b);
    auto kh_valid = [&](int kh) {
        const int foff = (kh - khc) * a.dil;
        return !(f0 + foff + PR <= 0 || f0 + foff >= a.F);
    };
    auto load_chunk = [&](int kh, int ci0) {
        const int foff = (kh - khc) * a.dil;
#pragma unroll
        for (int v = 0; v < XJ; ++v) {
            const int cir = ci0 + xci[v];
            const int f = f0 + xrow[v] + foff;
            const bool ok = xci[v] < KC && cir < a.Cin && f >= 0 && f < a.F && xt[v] < a.T;
            const int ci = cir < a.Cin ? cir : a.Cin - 1;
            const float* src = chan_ptr(ci);
            const long off = ok ? (long)f * a.T + xt[v] : 1;          // >= 1 so that off-1 stays readable
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
#pragma unroll
        for (int jj = 0; jj < WJ; ++jj) {
            int idx = tid + jj * NTH;
            if (idx > NWU - 1) idx = NWU - 1;
            const int co_l = idx % BN;
            const int ci_l = idx / BN;
            wr4[jj] = *reinterpret_cast<const f32x4*>(
                wq + (((long)(kh * g.CinP + ci0 + ci_l)) * g.CoutP + co0 + co_l) * 4);
        }
    };
    auto store_chunk = [&](f32x4* buf) {
#pragma unroll
        for (int v = 0; v < XJ; ++v) {
            if (xci[v] < KC) {
                f32x4 u0 = {0.f, 0.f, 0.f, 0.f}, u1 = {0.f, 0.f, 0.f, 0.f};
                if (xok[v]) {
                    const float s = xsc[v];
                    const float dm = xlok[v] ? xl[v] * s : 0.f, d0 = xv[v][0] * s, d1 = xv[v][1] * s, d2 = xv[v][2] * s,
                                d3 = xv[v][3] * s, d4 = xrok[v] ? xrr[v] * s : 0.f;
                    // pair 2*i4   : d = (dm, d0, d1, d2);   pair 2*i4+1 : d = (d1, d2, d3, d4)
                    u0 = f32x4{dm - d1, d0 + d1, d1 - d0, d0 - d2};
                    u1 = f32x4{d1 - d3, d2 + d3, d3 - d2, d2 - d4};
                }
                buf[xlds[v]] = u0;
                buf[xlds[v] + 1] = u1;
            }
        }
#pragma unroll
        for (int jj = 0; jj < WJ; ++jj) {
            const int idx = tid + jj * NTH;
            if (idx < NWU) buf[KC * NPAIR + idx] = wr4[jj];
        }
    };

    const int boff = wc * 32 + l31 + h * NPAIR;                     // B operand unit (pair) for ci_l = h
    const int aoff = KC * NPAIR + wr * (NTW * 32) + l31 + h * BN;   // A operand unit for ci_l = h, nt = 0

    int kh = 0;
    while (!kh_valid(kh)) ++kh;
    int ci0 = 0;
    load_chunk(kh, ci0);
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
        if (has_next && (!(ABL & 1) || ci0 == 0)) load_chunk(nkh, nci);
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
                bv[c ^ 1] = Xs[boff + 2 * (st + 1) * NPAIR];
            }
            __builtin_amdgcn_sched_barrier(0);
            if (!(ABL & 8)) {
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
                for (int p = 0; p < 4; ++p)
                    acc[nt][p] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c][nt][p], bv[c][p], acc[nt][p], 0, 0, 0);
            } else {
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
                for (int p = 0; p < 4; ++p) acc[nt][p][0] += av[c][nt][p] * bv[c][p];
            }
        }
        if (has_next && !(ABL & 2)) store_chunk(smem + (cur ^ 1) * BUF);
        if (!(ABL & 4)) __syncthreads();
        if (!has_next) break;
        kh = nkh;
        ci0 = nci;
        cur ^= 1;
    }

    // ---- output transform + epilogue: out = alpha*y*oscale + rbeta*res, two adjacent time steps per lane
    const int q = wc * 32 + l31;
    const int f = f0 + (q >> ppr_log2);
    const int t = t0 + 2 * (q & ((1 << ppr_log2) - 1));
    const bool pv = f < a.F && t < a.T;
    const long sp = pv ? (long)f * a.T + t : 0;
    const bool has_os = a.oscale != nullptr, has_res = a.res != nullptr;
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
        int cc[16];
        float os[16];
        f32x2 rr[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + wr * (NTW * 32) + nt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            cc[r] = co < a.Cout ? co : a.Cout - 1;
        }
        if (has_os) {
#pragma unroll
            for (int r = 0; r < 16; ++r) os[r] = a.oscale[b * a.Cout + cc[r]];
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) os[r] = 1.f;
        }
        if (has_res) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                rr[r] = *reinterpret_cast<const f32x2*>(a.res + (long)b * a.res_bs + (long)cc[r] * a.res_cs + sp);
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) rr[r] = f32x2{0.f, 0.f};
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + wr * (NTW * 32) + nt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            const float m0 = acc[nt][0][r], m1 = acc[nt][1][r], m2 = acc[nt][2][r], m3 = acc[nt][3][r];
            f32x2 y;
            y[0] = (m0 + m1 + m2) * a.alpha * os[r] + a.rbeta * rr[r][0];
            y[1] = (m1 - m2 - m3) * a.alpha * os[r] + a.rbeta * rr[r][1];
            if (pv && co < a.Cout)
                *reinterpret_cast<f32x2*>(a.out + (long)b * a.out_bs + (long)co * a.out_cs + sp) = y;
        }
    }
}

// ---- phase-pair variant -------------------------------------------------------------------------------------------
// For channel counts whose 32-row tiles do not split evenly over the 4 SIMDs (96 = 3 tiles) the 4 Winograd phases are
// split instead: waves 0..WC-1 accumulate phases (0,1), waves WC..2WC-1 phases (2,3), each over ALL NTW row tiles of
// the block, and the output transform exchanges half of the partial sums through LDS once at the end.  LDS images
// XQ[pair][8][units][2], WQ[pair][8][BN][2]: one 8-byte read per operand and K-step, 16-byte staging stores.
template <int NTW, int WC>
__global__ __launch_bounds__(128 * WC, (NTW <= 3 ? 3 : 2)) void conv_wino_pp_kernel(babe_conv_args a, WinoGeom g,
                                                                const float* __restrict__ wq) {
    constexpr int NTH = 128 * WC;
    constexpr int KC = 8;
    constexpr int BN = NTW * 32;
    constexpr int NPAIR = WC * 32;
    constexpr int NPOS = 2 * NPAIR;
    constexpr int NXQ = KC * NPOS / 4;
    constexpr int XJ = (NXQ + NTH - 1) / NTH;
    constexpr int WROW4 = BN / 2;                        // float4 per (pair, ci) weight row
    constexpr int NWV = 2 * KC * WROW4;                  // weight float4 per slab
    constexpr int WJ = (NWV + NTH - 1) / NTH;
    constexpr int XSZ = 4 * KC * NPAIR;                  // floats
    constexpr int BUF = XSZ + 4 * KC * BN;               // floats per buffer
    extern __shared__ __attribute__((aligned(16))) float smem_f[];

    const int PT = 1 << g.pt_log2;
    const int PR = 1 << g.pr_log2;
    const int ppr_log2 = g.pt_log2 - 1;
    const int tile_t = blockIdx.x % g.tiles_t;
    const int tile_f = blockIdx.x / g.tiles_t;
    const int t0 = tile_t << g.pt_log2;
    const int f0 = tile_f << g.pr_log2;
    const int co0 = blockIdx.y * BN;
    const int b = blockIdx.z;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pp = wave / WC, wc = wave % WC;
    const int h = lane >> 5;
    const int l31 = lane & 31;
    const int khc = a.KH >> 1;
    const int cin_split = a.in2 ? a.cin_split : a.Cin;
    const float* isc = a.in_scale ? a.in_scale : a.in;
    const bool has_isc = a.in_scale != nullptr;

    f32x16 acc[NTW][2];
#pragma unroll
    for (int i = 0; i < NTW; ++i)
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][p][r] = 0.f;

    int xci[XJ], xrow[XJ], xt[XJ], xlds[XJ];
#pragma unroll
    for (int v = 0; v < XJ; ++v) {
        const int idx = tid + v * NTH;
        const int i4 = idx & ((PT >> 2) - 1);
        xrow[v] = (idx >> (g.pt_log2 - 2)) & (PR - 1);
        xci[v] = idx >> (g.pt_log2 + g.pr_log2 - 2);
        xt[v] = t0 + 4 * i4;
        xlds[v] = (xci[v] * NPAIR + (xrow[v] << ppr_log2) + 2 * i4) * 2;
    }
    int wsrc[WJ], wlds[WJ];
#pragma unroll
    for (int jj = 0; jj < WJ; ++jj) {
        int idx = tid + jj * NTH;
        if (idx > NWV - 1) idx = NWV - 1;
        const int wp = idx / (KC * WROW4);
        const int rem = idx - wp * (KC * WROW4);
        const int ci_l = rem / WROW4;
        const int c4 = rem - ci_l * WROW4;
        wsrc[jj] = (ci_l * 2 + wp) * (g.CoutP / 2) + co0 / 2 + c4;      // float4 units, relative to the slab row
        wlds[jj] = XSZ + ((wp * KC + ci_l) * BN) * 2 + c4 * 4;
    }
    f32x4 xv[XJ], wr4[WJ];
    float xl[XJ], xrr[XJ], xsc[XJ];
    bool xok[XJ], xlok[XJ], xrok[XJ];

    ChanSrc chan_ptr;
    chan_ptr.init(a.in, a.in_bs, a.in_cs, a.in2, a.in2_bs, a.in2_cs, cin_split, b);
    auto kh_valid = [&](int kh) {
        const int foff = (kh - khc) * a.dil;
        return !(f0 + foff + PR <= 0 || f0 + foff >= a.F);
    };
    auto load_chunk = [&](int kh, int ci0) {
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
            xl[v] = src[lok ? off - 1 : 0];
            xrr[v] = src[rok ? off + 4 : 0];
            xlok[v] = lok;
            xrok[v] = rok;
            xsc[v] = has_isc ? isc[b * a.Cin + ci] : 1.f;
            xok[v] = ok;
        }
        const f32x4* wrow = reinterpret_cast<const f32x4*>(wq) + (long)(kh * g.CinP + ci0) * g.CoutP;
#pragma unroll
        for (int jj = 0; jj < WJ; ++jj) wr4[jj] = wrow[wsrc[jj]];
    };
    auto store_chunk = [&](float* buf) {
#pragma unroll
        for (int v = 0; v < XJ; ++v) {
            if (xci[v] < KC) {
                f32x4 q0 = {0.f, 0.f, 0.f, 0.f}, q1 = {0.f, 0.f, 0.f, 0.f};
                if (xok[v]) {
                    const float s = xsc[v];
                    const float dm = xlok[v] ? xl[v] * s : 0.f, d0 = xv[v][0] * s, d1 = xv[v][1] * s,
                                d2 = xv[v][2] * s, d3 = xv[v][3] * s, d4 = xrok[v] ? xrr[v] * s : 0.f;
                    q0 = f32x4{dm - d1, d0 + d1, d1 - d3, d2 + d3};      // phases (0,1) of the two pairs
                    q1 = f32x4{d1 - d0, d0 - d2, d3 - d2, d2 - d4};      // phases (2,3)
                }
                *reinterpret_cast<f32x4*>(buf + xlds[v]) = q0;
                *reinterpret_cast<f32x4*>(buf + KC * NPAIR * 2 + xlds[v]) = q1;
            }
        }
#pragma unroll
        for (int jj = 0; jj < WJ; ++jj)
            if (tid + jj * NTH < NWV) *reinterpret_cast<f32x4*>(buf + wlds[jj]) = wr4[jj];
    };

    const int boff = ((pp * KC + h) * NPAIR + wc * 32 + l31) * 2;
    const int aoff = XSZ + ((pp * KC + h) * BN + l31) * 2;

    int kh = 0;
    while (!kh_valid(kh)) ++kh;
    int ci0 = 0;
    load_chunk(kh, ci0);
    store_chunk(smem_f);
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
        if (has_next) load_chunk(nkh, nci);
        const float* Xs = smem_f + cur * BUF;
        f32x2 av[2][NTW], bv[2];
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) av[0][nt] = *reinterpret_cast<const f32x2*>(Xs + aoff + nt * 64);
        bv[0] = *reinterpret_cast<const f32x2*>(Xs + boff);
#pragma unroll
        for (int st = 0; st < KC / 2; ++st) {
            const int c = st & 1;
            if (st + 1 < KC / 2) {
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt)
                    av[c ^ 1][nt] = *reinterpret_cast<const f32x2*>(Xs + aoff + 4 * (st + 1) * BN + nt * 64);
                bv[c ^ 1] = *reinterpret_cast<const f32x2*>(Xs + boff + 4 * (st + 1) * NPAIR);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
                for (int p = 0; p < 2; ++p)
                    acc[nt][p] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c][nt][p], bv[c][p], acc[nt][p], 0, 0, 0);
        }
        if (has_next) store_chunk(smem_f + (cur ^ 1) * BUF);
        __syncthreads();
        if (!has_next) break;
        kh = nkh;
        ci0 = nci;
        cur ^= 1;
    }

    // ---- exchange: tile nt is finished by the pair-(nt & 1) wave; the other wave passes (x, y) through LDS with
    //   nt even (owner has M0, M1): x = M2, y = M2 + M3        nt odd (owner has M2, M3): x = M0 + M1, y = M1
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
        if (pp != (nt & 1)) {
            float* e = smem_f + ((wc * NTW + nt) * 2) * 1024 + lane;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float m0 = acc[nt][0][r], m1 = acc[nt][1][r];
                e[r * 64] = (nt & 1) ? m0 + m1 : m0;
                e[1024 + r * 64] = (nt & 1) ? m1 : m0 + m1;
            }
        }
    }
    __syncthreads();

    const int q = wc * 32 + l31;
    const int f = f0 + (q >> ppr_log2);
    const int t = t0 + 2 * (q & ((1 << ppr_log2) - 1));
    const bool pv = f < a.F && t < a.T;
    const long sp = pv ? (long)f * a.T + t : 0;
    const bool has_os = a.oscale != nullptr, has_res = a.res != nullptr;
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
        if (pp != (nt & 1)) continue;
        const float* e = smem_f + ((wc * NTW + nt) * 2) * 1024 + lane;
        int cc[16];
        float os[16];
        f32x2 rr[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + nt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            cc[r] = co < a.Cout ? co : a.Cout - 1;
        }
        if (has_os) {
#pragma unroll
            for (int r = 0; r < 16; ++r) os[r] = a.oscale[b * a.Cout + cc[r]];
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) os[r] = 1.f;
        }
        if (has_res) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                rr[r] = *reinterpret_cast<const f32x2*>(a.res + (long)b * a.res_bs + (long)cc[r] * a.res_cs + sp);
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) rr[r] = f32x2{0.f, 0.f};
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + nt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            const float ex = e[r * 64], ey = e[1024 + r * 64];
            const float p0 = acc[nt][0][r], p1 = acc[nt][1][r];
            f32x2 y;
            if (nt & 1) {                     // own M2, M3;  ex = M0+M1, ey = M1
                y[0] = ex + p0;
                y[1] = ey - p0 - p1;
            } else {                          // own M0, M1;  ex = M2, ey = M2+M3
                y[0] = p0 + p1 + ex;
                y[1] = p1 - ey;
            }
            y[0] = y[0] * a.alpha * os[r] + a.rbeta * rr[r][0];
            y[1] = y[1] * a.alpha * os[r] + a.rbeta * rr[r][1];
            if (pv && co < a.Cout)
                *reinterpret_cast<f32x2*>(a.out + (long)b * a.out_bs + (long)co * a.out_cs + sp) = y;
        }
    }
}

/* which output-channel tilings use the phase-pair kernel (and its weight image) */
/* (measured: for 64 / 128 / 256 output channels the phase-pair split is 0-10 % slower than the 4-phase waves) */
inline bool wino_use_pp(int n32) { return n32 == 3; }

// dst [KH][CinP][CoutP][4], or for the phase-pair variant [KH][CinP][2 pairs][CoutP][2]

__global__ void pack_wino_kernel(const float* __restrict__ w, float* __restrict__ dst, int Cout, int Cin, int KH,
                                 int tf, int CinP, int CoutP, long total, int pp) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int co = (int)(i % CoutP);
    long r = i / CoutP;
    const int ci = (int)(r % CinP);
    const int kh = (int)(r / CinP);
    float w0 = 0.f, w1 = 0.f, w2 = 0.f;
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
    f32x4 v = {w0, 0.5f * (w0 + w1 + w2), 0.5f * (w0 - w1 + w2), w2};
    if (pp) {
        f32x2* d2 = reinterpret_cast<f32x2*>(dst) + ((long)(kh * CinP + ci) * 2) * CoutP + co;
        d2[0] = f32x2{v[0], v[1]};
        d2[CoutP] = f32x2{v[2], v[3]};
    } else {
        reinterpret_cast<f32x4*>(dst)[i] = v;
    }
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
void launch(const babe_conv_args& a, WinoGeom g, const float* wq, hipStream_t s) {
    constexpr int NPOS = 64 * WC;
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
    const size_t lds = 2 * (size_t)(8 * (WC * 32) + 8 * BN) * 16;
    hipLaunchKernelGGL((conv_wino_kernel<NTW, WR, WC, ABL>), grid, dim3(64 * WR * WC), lds, s, a, g, wq);
}

template <int NTW, int WC>
void launch_pp(const babe_conv_args& a, WinoGeom g, const float* wq, hipStream_t s) {
    constexpr int NPOS = 64 * WC;
    const int npos_log2 = ilog2_floor(NPOS);
    g.pt_log2 = ilog2_ceil(a.T);
    if (g.pt_log2 > npos_log2) g.pt_log2 = npos_log2;
    if (g.pt_log2 < 4) g.pt_log2 = 4;
    g.pr_log2 = npos_log2 - g.pt_log2;
    const int PT = 1 << g.pt_log2, PR = 1 << g.pr_log2;
    g.tiles_t = cdiv(a.T, PT);
    const int tiles_f = cdiv(a.F, PR);
    constexpr int BN = NTW * 32;
    dim3 grid(g.tiles_t * tiles_f, g.CoutP / BN, a.B);
    size_t lds = 2 * (size_t)(4 * 8 * (WC * 32 + BN)) * 4;
    const size_t ex = (size_t)WC * NTW * 2 * 1024 * 4;
    if (ex > lds) lds = ex;
    hipLaunchKernelGGL((conv_wino_pp_kernel<NTW, WC>), grid, dim3(128 * WC), lds, s, a, g, wq);
}

}  // namespace


extern "C" long babe_conv_packed_size_wino(int Cout, int Cin, int KH, int transpose_flip) {
    const int co = transpose_flip ? Cin : Cout;
    const int ci = transpose_flip ? Cout : Cin;
    return (long)KH * ((ci + 7) / 8 * 8) * ((co + 31) / 32 * 32) * 4;
}

extern "C" int babe_conv_pack_weights_wino(const float* w, float* dst, int Cout, int Cin, int KH, int KW,
                                           int transpose_flip, void* stream) {
    BABE_CHECK_ARG(w && dst && Cout > 0 && Cin > 0 && KH > 0 && KW == 3, "conv_pack_weights_wino: needs KW == 3");
    const int co = transpose_flip ? Cin : Cout;
    const int ci = transpose_flip ? Cout : Cin;
    const int CinP = (ci + 7) / 8 * 8, CoutP = (co + 31) / 32 * 32;
    const long total = (long)KH * CinP * CoutP;
    hipLaunchKernelGGL(pack_wino_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, w, dst, Cout, Cin,
                       KH, transpose_flip, CinP, CoutP, total, wino_use_pp(CoutP / 32) ? 1 : 0);
    BABE_LAUNCH_CHECK();
    return BABE_OK;
}

/* returns 1 if the Winograd kernel can run this problem (the caller then passes the wino-packed weights) */
extern "C" int babe_conv2d_wino_supported(const babe_conv_args* ap) {
    if (!ap) return 0;
    const babe_conv_args& a = *ap;
    auto al16 = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
    auto al8 = [](const void* p) { return ((uintptr_t)p & 7) == 0; };
    if (a.KW != 3 || a.KH < 1 || a.T % 4 != 0 || a.T < 16) return 0;
    if (!al16(a.in) || a.in_bs % 4 || a.in_cs % 4) return 0;
    if (a.in2 && (!al16(a.in2) || a.in2_bs % 4 || a.in2_cs % 4)) return 0;
    if (!al8(a.out) || a.out_bs % 2 || a.out_cs % 2) return 0;
    if (a.res && (!al8(a.res) || a.res_bs % 2 || a.res_cs % 2)) return 0;
    const int n32 = (a.Cout + 31) / 32;
    return (n32 == 1 || n32 == 3 || n32 % 2 == 0) ? 1 : 0;
}

extern "C" int babe_conv2d_wino(const babe_conv_args* ap, const float* w_wino, void* stream) {
    BABE_CHECK_ARG(ap && w_wino, "conv2d_wino: null args");
    BABE_CHECK_ARG(babe_conv2d_wino_supported(ap), "conv2d_wino: unsupported problem (use babe_conv2d)");
    const babe_conv_args& a = *ap;
    WinoGeom g;
    g.CinP = (a.Cin + 7) / 8 * 8;
    g.CoutP = (a.Cout + 31) / 32 * 32;
    const int n32 = g.CoutP / 32;
    hipStream_t s = (hipStream_t)stream;
    const double flops = babe_conv_flops(a);     // F(2,3): 4 multiplies per 2 outputs instead of 6
    BabeProfScope prof(BABE_SLOT_CONV53_WINO2, babe_conv_bytes(a), flops, flops * (2.0 / 3.0), stream);
    if (n32 == 1) launch<1, 1, 4>(a, g, w_wino, s);            //  32 co x 256 positions
    else if (wino_use_pp(n32)) launch_pp<3, 2>(a, g, w_wino, s);   //  96 co x 128 positions, phase-pair split
    else if (n32 == 2) launch<2, 1, 4>(a, g, w_wino, s);       //  64 co x 256 positions
    else launch<2, 2, 2>(a, g, w_wino, s);                     // 128 co x 128 positions
    BABE_LAUNCH_CHECK();
    return BABE_OK;
}
