// NESTED Winograd for the frequency-dilated (5,3) Conv2d (networks/cqtdiff+.py:79-88, 433-436), fp32 MFMA (round 3):
// F(2,5) along FREQUENCY nested with F(4,3) along TIME.  A unit = 2 output rows (f, f + dil: neighbours in their residue
// class mod dil) x 4 time steps; its 8 outputs come from a 6-row x 6-sample input patch with 36 multiplies per (ci, co)
// - 4.5 per output, against 7.5 for F(4,3) along time alone (conv_wino4p.hip) and 15 for the direct kernel.  Both
// transforms use the interpolation points 0, +-1, +-2, inf, so B^T is the same 6x6 matrix in both directions:
//   U = B^T D B   (6x6 patch D -> 36 phases)        V = G5 W G3^T   (5x3 taps -> 36 phases, in double at pack time)
//   M = sum_ci U (.) V                               Y = A2^T M A4   (2 rows x 4 steps)
//   A2^T = [1 1 1 1 1 0; 0 1 -1 2 -2 1]
// 36 accumulators per (co, unit) do not fit the register file at a useful tile size, so the 6 frequency phases are
// processed in THREE PASSES of two - (1,2), (3,4), (0,5) - over the input channels, 12 phase GEMMs per pass.  The
// partial sums of a finished pass are carried INSIDE the accumulators of the next one: with Q_r = sum over finished
// phases of A2[r][fp] M_fp (r = output row 0/1, still in the 6-phase time domain), the next pass (a, b) starts from
// [M_a; M_b] = C^-1 [Q_0; Q_1], C = [[A2[0][a], A2[0][b]], [A2[1][a], A2[1][b]]]:
//   after (1,2):  M_3 = 3/4 M_1 + 1/4 M_2,  M_4 = 1/4 M_1 + 3/4 M_2        after (3,4):  M_0 = M_3 + M_4,  M_5 = 2 M_3 - 2 M_4
// and after pass (0,5) the two output rows are simply A4^T M_0 and A4^T M_5 (time transform only, no LDS exchange).
//
// Tile: 64 output channels x 64 units (4 row pairs of one residue class x 16 time units = 8 rows x 64 steps), 8 waves;
// wave (cw, uw) owns the 16-channel tile cw and row pairs 2uw, 2uw+1: 2 x 12 accumulators of v_mfma_f32_16x16x4_f32
// (96 registers).  K-slab = (pass, 8 input channels) = 2 K-steps: 24 KB of transformed activations [ci][unit][12] and
// 24 KB of weights [ci][co][12] (one 16-byte LDS read hands a lane 4 phases of an operand; 48-byte strides are
// conflict-free for ds_read_b128 / ds_write_b128), 48 MFMAs per wave - the same cadence as conv_wino4p, whose pipeline
// this kernel keeps: THREE LDS buffers (slab j+2 staged while slab j is multiplied, first operands of slab j+1 read
// before the barrier), activations by raw buffer loads with the hardware range check as zero padding, weights by
// LDS-DMA.  Staging: thread = (ci, unit) loads its 6 rows x (16 bytes + 2 neighbours), forms the two frequency phases
// of the pass with wave-uniform coefficients (one code path for all passes), applies the time transform and writes 48
// bytes.  3 * Cin/8 slabs per tile instead of 5 * Cin/8: 0.6 of the matrix work of conv_wino4p per output.
// Rounding: 3.4x the F(4,3) kernel's (2-3e-6 relative against float64 at 256 channels; tests/test_gpu_ops.py).
#include "common.h"
#include "../../include/babe_hip.h"
#include "prof.h"
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

struct Wino45Geom {
    int CinP, CoutP, tiles_t, groups;
};

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
constexpr unsigned OOBH = 0xC0000000u;     // invalid offsets start here: +-(a source view < 1 GiB) stays >= 2^31

template <bool HAS_ISC>
__global__ __launch_bounds__(512, 1) void conv_wino45_kernel(babe_conv_args a, Wino45Geom g, const float* __restrict__ wq) {
#if __HIP_DEVICE_COMPILE__
    constexpr int NTH = 512, KC = 8, BN = 64, NU = 64;
    constexpr int XSZ = KC * NU * 3;                    // float4 per activation image (12 floats per (ci, unit))
    constexpr int WSZ = KC * BN * 3;
    constexpr int BUF = XSZ + WSZ;
    constexpr int WJ = WSZ / NTH;                       // 3 weight float4 per thread and slab
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    f32x4* smem = reinterpret_cast<f32x4*>(smem_f);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cw = wave & 3, uw = wave >> 2;
    const int l15 = lane & 15, lk = lane >> 4;
    const int b = blockIdx.z;
    const int co0 = blockIdx.y * BN;
    const int tile_t = blockIdx.x % g.tiles_t;
    const int rest = blockIdx.x / g.tiles_t;
    const int grp = rest % g.groups;
    const int cls = rest / g.groups;                    // residue class of the tile's rows (mod dil)
    const int t0 = tile_t * 64;
    const int split = a.in2 ? a.cin_split : a.Cin;
    const int nci = g.CinP / KC;
    const int nslab = 3 * nci;

    // ---- descriptors
    const float* p1 = a.in + (long)b * a.in_bs;
    const float* p2 = a.in2 ? a.in2 + (long)b * a.in2_bs : p1;
    const int cs1 = (int)a.in_cs, cs2 = a.in2 ? (int)a.in2_cs : (int)a.in_cs;
    const int nb1 = split * cs1 * 4, nb2 = (a.Cin - split) * cs2 * 4;
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)wq, 0, 3 * g.CinP * g.CoutP * 48, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsi = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(HAS_ISC ? a.in_scale + (long)b * a.Cin : a.in), 0, HAS_ISC ? a.Cin * 4 : 0, 0x00020000);

    // ---- per-thread staging constants: thread = (ci, unit), unit = rp * 16 + tu
    const int s_tu = tid & 15, s_rp = (tid >> 4) & 3, s_ci = tid >> 6;
    const int s_t = t0 + 4 * s_tu;
    const int s_fa = cls + 2 * (grp * 4 + s_rp) * a.dil;             // first output row of the pair
    unsigned er[6];                                                     // byte offset of (row r, t) or OOBH
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        const int fr = s_fa + (r - 2) * a.dil;
        const bool ok = fr >= 0 && fr < a.F && s_t < a.T;
        er[r] = ok ? (unsigned)((fr * a.T + s_t) * 4) : OOBH;
    }
    const unsigned leftbad = s_t > 0 ? 0u : OOBH;
    const unsigned rightbad = s_t + 4 < a.T ? 0u : OOBH;
    const int xcs1 = s_ci * cs1 * 4, xcs2 = s_ci * cs2 * 4;           // bytes
    const int xlds = (s_ci * NU + (tid & 63)) * 3;                     // float4 index of this thread's 12 floats
    int wvo[WJ];
#pragma unroll
    for (int jj = 0; jj < WJ; ++jj) {
        const int idx = tid + jj * NTH;
        const int ci_l = idx / (BN * 3);
        const int rem = idx - ci_l * (BN * 3);
        wvo[jj] = ci_l * g.CoutP * 48 + rem * 16;
    }

    f32x4 acc[2][12];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int p = 0; p < 12; ++p) acc[i][p] = f32x4{0.f, 0.f, 0.f, 0.f};

    // staging registers: raw loads of the slab that is next to be transformed
    f32x4 xv[6];
    float xl[6], xr[6], xsc = 1.f;

    // rows 0 and 5 of the patch are read by pass 2 (phases 0, 5) only: in the other passes their offsets are forced out of
    // range, so the loads return 0 without touching memory and the code stays one straight line
    auto issue_act = [&](int ps, int ci0) {
        const bool s2 = ci0 >= split;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(s2 ? p2 : p1), 0, s2 ? nb2 : nb1, 0x00020000);
        const unsigned sb = (unsigned)((s2 ? (ci0 - split) * cs2 : ci0 * cs1) * 4) + (unsigned)(s2 ? xcs2 : xcs1);
        const unsigned edge = ps == 2 ? 0u : OOBH;                    // scalar
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            const unsigned e = (er[r] + sb) | ((r == 0 || r == 5) ? edge : 0u);
            xv[r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, e, 0, 0));
            xl[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (e - 4u) | leftbad, 0, 0));
            xr[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (e + 16u) | rightbad, 0, 0));
        }
        if (HAS_ISC) xsc = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsi, (s_ci + ci0) * 4, 0, 0));
    };
    // frequency phases of pass ps from the 6 rows d0..d5 of one column:
    //   e = d4 - k2 d2,  o = k3 d3 - k1 d1,  Ea = x0 d0 + e + za o,  Eb = x5 d5 + ye e - o
    //   (1,2): k2 4, k1 4, k3 1, x0 0, za 1, x5 0, ye 1      (3,4): k2 1, k1 2, k3 2, x0 0, za 1, x5 0, ye 1
    //   (0,5): k2 5, k1 4, k3 5, x0 4, za 0, x5 1, ye 0      (Ea = 4 d0 - 5 d2 + d4,  Eb = 4 d1 - 5 d3 + d5)
    auto store_act = [&](int ps, f32x4* buf) {
        const float k2 = ps == 0 ? 4.f : (ps == 1 ? 1.f : 5.f);
        const float k1 = ps == 1 ? 2.f : 4.f;
        const float k3 = ps == 0 ? 1.f : (ps == 1 ? 2.f : 5.f);
        const float x0 = ps == 2 ? 4.f : 0.f, za = ps == 2 ? 0.f : 1.f;
        const float x5 = ps == 2 ? 1.f : 0.f, ye = ps == 2 ? 0.f : 1.f;
        float Ea[6], Eb[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            float d[6];
#pragma unroll
            for (int r = 0; r < 6; ++r) d[r] = j == 0 ? xl[r] : (j == 5 ? xr[r] : xv[r][j - 1]);
            const float e = d[4] - k2 * d[2];
            const float o = k3 * d[3] - k1 * d[1];
            Ea[j] = x0 * d[0] + (e + za * o);
            Eb[j] = x5 * d[5] + (ye * e - o);
        }
        if (HAS_ISC) {
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                Ea[j] *= xsc;
                Eb[j] *= xsc;
            }
        }
        // time transform (same B^T): U0 = 4E0-5E2+E4, U1/U2 = (E4-4E2) +- (E3-4E1), U3/U4 = (E4-E2) +- 2(E3-E1), U5 = 4E1-5E3+E5
        auto tt = [](const float (&E)[6], float (&U)[6]) {
            const float e = E[4] - 4.f * E[2], o = E[3] - 4.f * E[1];
            const float e2 = E[4] - E[2], o2 = 2.f * (E[3] - E[1]);
            U[0] = 4.f * E[0] - 5.f * E[2] + E[4];
            U[1] = e + o;
            U[2] = e - o;
            U[3] = e2 + o2;
            U[4] = e2 - o2;
            U[5] = 4.f * E[1] - 5.f * E[3] + E[5];
        };
        float Ua[6], Ub[6];
        tt(Ea, Ua);
        tt(Eb, Ub);
        buf[xlds] = f32x4{Ua[0], Ua[1], Ua[2], Ua[3]};
        buf[xlds + 1] = f32x4{Ua[4], Ua[5], Ub[0], Ub[1]};
        buf[xlds + 2] = f32x4{Ub[2], Ub[3], Ub[4], Ub[5]};
    };
    auto dma_w = [&](int ps, int ci0, f32x4* buf, int j0, int j1) {
        const int so = ((ps * g.CinP + ci0) * g.CoutP + co0) * 48;     // bytes, scalar
#pragma unroll
        for (int jj = j0; jj < j1; ++jj)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, LDS_PTR(buf + XSZ + jj * NTH + wave * 64), 16, wvo[jj], so, 0, 0);
    };
    auto advance = [&](int& ps, int& ci0) {             // next slab, clamped at the last one
        int nc = ci0 + KC, np = ps;
        if (nc >= g.CinP) {
            nc = 0;
            ++np;
        }
        if (np <= 2) {
            ps = np;
            ci0 = nc;
        }
    };

    // operand addresses (float4 units inside a buffer): phases 4pg..4pg+3 of K-step ks at + ks*4*64*3 + pg
    const int aoff = XSZ + (lk * BN + cw * 16 + l15) * 3;
    const int boff = (lk * NU + uw * 32 + l15) * 3;

    // ---- prologue: slabs 0 and 1 into buffers 0 and 1, loads of slab 2 in flight
    int pA = 0, cA = 0;                          // slab whose activations are in the staging registers
    issue_act(pA, cA);
    dma_w(pA, cA, smem, 0, WJ);
    store_act(pA, smem);
    int pW = pA, cW = cA;                        // slab whose weights are DMA'd next
    advance(pA, cA);
    advance(pW, cW);
    issue_act(pA, cA);
    dma_w(pW, cW, smem + BUF, 0, WJ);
    store_act(pA, smem + BUF);
    advance(pA, cA);
    advance(pW, cW);
    issue_act(pA, cA);                            // slab 2 (or a clamped copy of the last slab)
    int pS = pA;                                  // pass of the data held in the staging registers
    __syncthreads();

    f32x4 av[2], bv[2][2];
    av[0] = smem[aoff];
    bv[0][0] = smem[boff];
    bv[0][1] = smem[boff + 16 * 3];

    int rb = 0;                                   // ring slot of the slab being multiplied
    int pM = 0, cM = 0;                           // slab being multiplied (for the pass boundaries)
    for (int j = 0; j < nslab; ++j) {
        const int rn = rb == 2 ? 0 : rb + 1;      // slab j+1
        const int rw = rn == 2 ? 0 : rn + 1;      // slab j+2: staged during this slab
        const f32x4* Xs = smem + rb * BUF;
        const f32x4* Xn = smem + rn * BUF;
        f32x4* Xw = smem + rw * BUF;
#define MFMA_GRP(c, pg)                                                                                               \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                                    \
        acc[0][4 * (pg) + i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c][i], bv[c][0][i], acc[0][4 * (pg) + i], 0, 0, 0); \
        acc[1][4 * (pg) + i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c][i], bv[c][1][i], acc[1][4 * (pg) + i], 0, 0, 0); \
    }
#define READ_GRP(c, base, ks, pg)                                                   \
    av[c] = (base)[aoff + (ks) * 4 * BN * 3 + (pg)];                                \
    bv[c][0] = (base)[boff + (ks) * 4 * NU * 3 + (pg)];                             \
    bv[c][1] = (base)[boff + (ks) * 4 * NU * 3 + 16 * 3 + (pg)];
        // group (ks 0, pg 0): transform + write the staged activations of slab j+2, re-issue the staging loads (slab j+3),
        // first part of the weight DMA of slab j+2
        READ_GRP(1, Xs, 0, 1)
        store_act(pS, Xw);
        advance(pA, cA);
        issue_act(pA, cA);
        pS = pA;
        dma_w(pW, cW, Xw, 0, 2);
        MFMA_GRP(0, 0)
        __builtin_amdgcn_sched_barrier(0);
        READ_GRP(0, Xs, 0, 2)
        dma_w(pW, cW, Xw, 2, WJ);
        MFMA_GRP(1, 1)
        __builtin_amdgcn_sched_barrier(0);
        READ_GRP(1, Xs, 1, 0)
        MFMA_GRP(0, 2)
        __builtin_amdgcn_sched_barrier(0);
        READ_GRP(0, Xs, 1, 1)
        MFMA_GRP(1, 0)
        __builtin_amdgcn_sched_barrier(0);
        READ_GRP(1, Xs, 1, 2)
        MFMA_GRP(0, 1)
        __builtin_amdgcn_sched_barrier(0);
        // last group: first operands of slab j+1 (its buffer was completed by the PREVIOUS barrier)
        advance(pW, cW);
        READ_GRP(0, Xn, 0, 0)
        MFMA_GRP(1, 2)
        __syncthreads();                           // slab j+2 complete (DMA + ds_write), slab j's buffer free
        rb = rn;
        // pass boundary: carry the finished phases into the accumulators of the next pass (see the header)
        cM += KC;
        if (cM >= g.CinP) {
            cM = 0;
            if (pM == 0) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int p = 0; p < 6; ++p) {
                        const f32x4 m1 = acc[i][p], m2 = acc[i][6 + p];
                        acc[i][p] = 0.75f * m1 + 0.25f * m2;
                        acc[i][6 + p] = 0.25f * m1 + 0.75f * m2;
                    }
            } else if (pM == 1) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int p = 0; p < 6; ++p) {
                        const f32x4 m3 = acc[i][p], m4 = acc[i][6 + p];
                        acc[i][p] = m3 + m4;
                        acc[i][6 + p] = 2.f * (m3 - m4);
                    }
            }
            ++pM;
        }
    }
#undef MFMA_GRP
#undef READ_GRP

    // ---- output: rows fa (from M_0) and fa + dil (from M_5), time transform A4^T; lane = (unit l15, channels 4 lk .. 4 lk + 3)
    const bool has_os = a.oscale != nullptr, has_res = a.res != nullptr;
    const int t = t0 + 4 * l15;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int rp = uw * 2 + i;
        const int fa = cls + 2 * (grp * 4 + rp) * a.dil;
#pragma unroll
        for (int row = 0; row < 2; ++row) {
            const int f = fa + row * a.dil;
            const bool pv = f < a.F && t < a.T;
            const long sp = pv ? (long)f * a.T + t : 0;
            int cc[4];
            float os[4];
            f32x4 rr[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int co = co0 + cw * 16 + 4 * lk + k;
                cc[k] = co < a.Cout ? co : a.Cout - 1;
                os[k] = has_os ? a.oscale[b * a.Cout + cc[k]] : 1.f;
                rr[k] = has_res ? *reinterpret_cast<const f32x4*>(a.res + (long)b * a.res_bs + (long)cc[k] * a.res_cs + sp)
                                : f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int co = co0 + cw * 16 + 4 * lk + k;
                const float m0 = acc[i][6 * row + 0][k], m1 = acc[i][6 * row + 1][k], m2 = acc[i][6 * row + 2][k];
                const float m3 = acc[i][6 * row + 3][k], m4 = acc[i][6 * row + 4][k], m5 = acc[i][6 * row + 5][k];
                const float s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
                f32x4 y = {m0 + s12 + s34, d12 + 2.f * d34, s12 + 4.f * s34, d12 + 8.f * d34 + m5};
                const float sc = a.alpha * os[k];
                y = y * sc + a.rbeta * rr[k];
                if (pv && co < a.Cout)
                    *reinterpret_cast<f32x4*>(a.out + (long)b * a.out_bs + (long)co * a.out_cs + sp) = y;
            }
        }
    }
#endif
}

// dst [3 passes][CinP][CoutP][12]: pass ps holds frequency phases (1,2), (3,4), (0,5); entry 6*fpl + tp
__global__ void pack_wino45_kernel(const float* __restrict__ w, float* __restrict__ dst, int Cout, int Cin, int tf, int CinP,
                                   int CoutP, long total) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int co = (int)(i % CoutP);
    long r = i / CoutP;
    const int ci = (int)(r % CinP);
    const int ps = (int)(r / CinP);
    double wk[5][3];
#pragma unroll
    for (int kh = 0; kh < 5; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) wk[kh][kw] = 0;
    if (!tf) {
        if (co < Cout && ci < Cin) {
            const float* p = w + ((long)co * Cin + ci) * 15;
            for (int kh = 0; kh < 5; ++kh)
                for (int kw = 0; kw < 3; ++kw) wk[kh][kw] = p[kh * 3 + kw];
        }
    } else {
        if (co < Cin && ci < Cout) {      // packed "Cout" = reference Cin; taps flipped in both axes
            const float* p = w + ((long)ci * Cin + co) * 15;
            for (int kh = 0; kh < 5; ++kh)
                for (int kw = 0; kw < 3; ++kw) wk[kh][kw] = p[(4 - kh) * 3 + (2 - kw)];
        }
    }
    const double G5[6][5] = {{0.25, 0, 0, 0, 0},
                             {-1.0 / 6, -1.0 / 6, -1.0 / 6, -1.0 / 6, -1.0 / 6},
                             {-1.0 / 6, 1.0 / 6, -1.0 / 6, 1.0 / 6, -1.0 / 6},
                             {1.0 / 24, 1.0 / 12, 1.0 / 6, 1.0 / 3, 2.0 / 3},
                             {1.0 / 24, -1.0 / 12, 1.0 / 6, -1.0 / 3, 2.0 / 3},
                             {0, 0, 0, 0, 1}};
    const double G3[6][3] = {{0.25, 0, 0},          {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                             {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6},  {0, 0, 1}};
    const int fps[3][2] = {{1, 2}, {3, 4}, {0, 5}};
    float* d = dst + ((long)(ps * CinP + ci) * CoutP + co) * 12;
    for (int fl = 0; fl < 2; ++fl) {
        const int fp = fps[ps][fl];
        double fw[3];                                  // frequency transform of the 5 taps, per time tap
        for (int kw = 0; kw < 3; ++kw) {
            double s = 0;
            for (int kh = 0; kh < 5; ++kh) s += G5[fp][kh] * wk[kh][kw];
            fw[kw] = s;
        }
        for (int tp = 0; tp < 6; ++tp) d[6 * fl + tp] = (float)(G3[tp][0] * fw[0] + G3[tp][1] * fw[1] + G3[tp][2] * fw[2]);
    }
}

}  // namespace

extern "C" long babe_conv_packed_size_wino45(int Cout, int Cin, int transpose_flip) {
    const int co = transpose_flip ? Cin : Cout;
    const int ci = transpose_flip ? Cout : Cin;
    return 36L * ((ci + 7) / 8 * 8) * ((co + 63) / 64 * 64);
}

extern "C" int babe_conv_pack_weights_wino45(const float* w, float* dst, int Cout, int Cin, int KH, int KW,
                                             int transpose_flip, void* stream) {
    BABE_CHECK_ARG(w && dst && Cout > 0 && Cin > 0 && KH == 5 && KW == 3, "conv_pack_weights_wino45: needs a (5,3) kernel");
    const int co = transpose_flip ? Cin : Cout;
    const int ci = transpose_flip ? Cout : Cin;
    const int CinP = (ci + 7) / 8 * 8, CoutP = (co + 63) / 64 * 64;
    const long total = 3L * CinP * CoutP;
    hipLaunchKernelGGL(pack_wino45_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, w, dst, Cout, Cin,
                       transpose_flip, CinP, CoutP, total);
    BABE_LAUNCH_CHECK();
    return BABE_OK;
}

/* 1 if the nested-Winograd kernel can run this problem (the caller then passes the wino45-packed weights) */
extern "C" int babe_conv2d_wino45_supported(const babe_conv_args* ap) {
    if (!ap) return 0;
    static const char* ov = getenv("BABE_CONV_WINO45");
    if (ov && ov[0] == '0') return 0;
    const babe_conv_args& a = *ap;
    auto al16 = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
    if (a.KH != 5 || a.KW != 3 || a.T % 4 != 0 || a.T < 64 || a.dil < 1) return 0;   // (tiles are 64 time steps wide)
    if (a.Cin < 8 || a.Cout < 33) return 0;                  // (few-channel convs: conv_fewco / direct kernels)
    if (!al16(a.in) || a.in_bs % 4 || a.in_cs % 4) return 0;
    if (a.in2 && (!al16(a.in2) || a.in2_bs % 4 || a.in2_cs % 4)) return 0;
    if (!al16(a.out) || a.out_bs % 4 || a.out_cs % 4) return 0;
    if (a.res && (!al16(a.res) || a.res_bs % 4 || a.res_cs % 4)) return 0;
    if (a.in2 && (a.cin_split % 8 != 0)) return 0;           // a slab never straddles the two sources
    const long lim = 0x3fffffffL / 4;                        // source views below 1 GiB per batch item (OOBH arithmetic)
    const int split = a.in2 ? a.cin_split : a.Cin;
    if ((long)split * a.in_cs >= lim) return 0;
    if (a.in2 && (long)(a.Cin - split) * a.in2_cs >= lim) return 0;
    if ((long)a.F * a.T >= lim) return 0;
    if (36L * ((a.Cin + 7) / 8 * 8) * ((a.Cout + 63) / 64 * 64) * 4 >= 0x7fffffffL) return 0;
    return 1;
}

extern "C" int babe_conv2d_wino45(const babe_conv_args* ap, const float* w_wino45, void* stream) {
    BABE_CHECK_ARG(ap && w_wino45, "conv2d_wino45: null args");
    BABE_CHECK_ARG(babe_conv2d_wino45_supported(ap), "conv2d_wino45: unsupported problem (use babe_conv2d_wino4 / babe_conv2d)");
    const babe_conv_args& a = *ap;
    Wino45Geom g;
    g.CinP = (a.Cin + 7) / 8 * 8;
    g.CoutP = (a.Cout + 63) / 64 * 64;
    g.tiles_t = cdiv(a.T, 64);
    const int n = cdiv(a.F, a.dil);                  // rows per residue class (at most)
    g.groups = cdiv(cdiv(n, 2), 4);                  // 4 row pairs per tile
    hipStream_t s = (hipStream_t)stream;
    const double flops = babe_conv_flops(a);         // 36 multiplies per 8 outputs instead of 120: 0.3 of the direct count
    BabeProfScope prof(BABE_SLOT_CONV53_WINO45, babe_conv_bytes(a), flops, flops * 0.3, stream);
    dim3 grid(g.tiles_t * g.groups * a.dil, g.CoutP / 64, a.B);
    const size_t lds = 3 * (size_t)(8 * 64 * 3 + 8 * 64 * 3) * 16;           // 144 KB
    static std::atomic<unsigned long long> attr_done{0};
    if (babe_lds_optin(attr_done, {reinterpret_cast<const void*>(&conv_wino45_kernel<true>),
                                   reinterpret_cast<const void*>(&conv_wino45_kernel<false>)}, (int)lds) == hipSuccess) {
        if (a.in_scale) hipLaunchKernelGGL((conv_wino45_kernel<true>), grid, dim3(512), lds, s, a, g, w_wino45);
        else hipLaunchKernelGGL((conv_wino45_kernel<false>), grid, dim3(512), lds, s, a, g, w_wino45);
    }
    BABE_LAUNCH_CHECK();
    return BABE_OK;
}
