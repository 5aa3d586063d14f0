// Winograd F(4,3)-along-time (5,3) conv, PIPELINED main loop (round 2).  Same arithmetic, tiling, LDS images, packed
// weights and epilogue as conv_wino4.hip (read its header first); what changes is how a K-slab (8 input channels of one
// frequency tap) travels, because PMC on the round-1 kernel showed the matrix pipe busy only 60 % of the time with the
// rest split between waves parked at s_waitcnt / s_barrier (25 % of wave cycles) and non-MFMA issue (18 %) that never
// overlapped the MFMAs (profiles/r02_pmc_wino4_v8.txt: MFMA + parked + issue add up to the slab time):
//
//   * THREE LDS buffers.  Slab j+2 is staged while slab j is multiplied, so the buffer of slab j+1 is complete one
//     barrier early and the first operands of slab j+1 are read BEFORE the barrier that ends slab j: no wave leaves a
//     barrier into an LDS read burst (8 waves x 6 KB) any more, the MFMAs of the next slab start at once.
//   * activations through raw BUFFER loads: zero padding (rows outside [0,F), columns outside [0,T), channels >= Cin)
//     is the hardware range check (out-of-range offset -> 0), so the staging code has no masks and no branches; the
//     per-slab address is two scalar adds.  Loads for slab j+3 are issued right after the registers they land in have
//     been transformed and written (slab j+2): one register set, a full slab of latency cover.
//   * weights by LDS-DMA through a buffer descriptor: per-thread offsets are loop constants, the slab is the scalar offset.
//   * one basic block per slab (tail slabs are clamped re-stagings of the last slab, never read), with the staging work
//     distributed behind the four K-steps so that it issues in the shadow of the MFMAs.
//
// Requirements beyond conv_wino4.hip's: cin_split % 8 == 0 (a slab never straddles the two sources) and every source
// view below 2 GiB per batch item (32-bit buffer offsets); anything else runs on the round-1 kernel.
#include "common.h"
#include "../../include/babe_hip.h"
#include "prof.h"
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

namespace {

struct Wino4pGeom {
    int CinP, CoutP, pt_log2, pr_log2, tiles_t;
};

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
constexpr unsigned OOB = 0x80000000u;      // beyond every descriptor's num_records: the load returns 0, touches nothing

// XSOA: the transformed activations are stored phase-planar, [triple][ci][phase][unit] floats (12 bytes per unit and
// triple instead of a padded float4), and a B operand is three ds_read_b32: 24 KB instead of 32 KB per slab for 128 units,
// which is what lets the 96-channel tile (96 co x 512 positions) keep three buffers inside 160 KB of LDS.
template <int NTW, int WR, int WC, bool HAS_ISC, bool XSOA = false>
__global__ __launch_bounds__(128 * WR * WC, 1) void conv_wino4p_kernel(babe_conv_args a, Wino4pGeom g,
                                                                       const float* __restrict__ wq) {
#if __HIP_DEVICE_COMPILE__      // the buffer-descriptor builtins exist in the device pass only; the host pass needs just the stub
    constexpr int NTH = 128 * WR * WC;
    constexpr int KC = 8;
    constexpr int BN = WR * NTW * 32;
    constexpr int NUNIT = WC * 32;                      // units (4 outputs each) per tile
    constexpr int NXQ = KC * NUNIT;                     // input quads per slab (one per unit and channel)
    static_assert(NXQ % NTH == 0, "every thread stages the same number of quads");
    constexpr int XJ = NXQ / NTH;
    constexpr int NW4 = 2 * KC * BN;                    // weight float4 per slab
    static_assert(NW4 % NTH == 0, "weight slab = whole wave instructions");
    constexpr int WJ = NW4 / NTH;
    constexpr int XSZ = XSOA ? 2 * KC * 3 * NUNIT / 4 : 2 * KC * NUNIT;      // float4 units
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
    const int split = a.in2 ? a.cin_split : a.Cin;

    // ---- frequency taps that touch this tile's rows: a contiguous range [kh_lo, kh_hi]
    int kh_lo = 0, kh_hi = a.KH - 1;
    while ((f0 + (kh_lo - khc) * a.dil + PR <= 0 || f0 + (kh_lo - khc) * a.dil >= a.F) && kh_lo < kh_hi) ++kh_lo;
    while ((f0 + (kh_hi - khc) * a.dil + PR <= 0 || f0 + (kh_hi - khc) * a.dil >= a.F) && kh_hi > kh_lo) --kh_hi;
    const int nci = g.CinP / KC;
    const int nslab = (kh_hi - kh_lo + 1) * nci;

    // ---- descriptors (wave-uniform by construction: kernargs and block indices only)
    const float* p1 = a.in + (long)b * a.in_bs;
    const float* p2 = a.in2 ? a.in2 + (long)b * a.in2_bs : p1;
    const int cs1 = (int)a.in_cs, cs2 = a.in2 ? (int)a.in2_cs : (int)a.in_cs;
    const int nb1 = split * cs1 * 4, nb2 = (a.Cin - split) * cs2 * 4;
    const __amdgpu_buffer_rsrc_t rsw =
        __builtin_amdgcn_make_buffer_rsrc((void*)wq, 0, a.KH * g.CinP * 2 * g.CoutP * 16, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsi = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(HAS_ISC ? a.in_scale + (long)b * a.Cin : a.in), 0, HAS_ISC ? a.Cin * 4 : 0, 0x00020000);

    // ---- per-thread staging constants
    // Validity is carried in the TOP BIT of the byte offset (any offset >= 2^31 is out of range for every descriptor
    // here), built with integer arithmetic only: selects on the load address make hipcc branch around the loads.
    int xspat[XJ], xfrow[XJ], xcs1[XJ], xcs2[XJ], xci4[XJ], xlds[XJ];
    unsigned xcolbad[XJ], xleftbad[XJ], xrightbad[XJ];
#pragma unroll
    for (int v = 0; v < XJ; ++v) {
        const int idx = tid + v * NTH;
        const int i4 = idx & ((PT >> 2) - 1);
        const int row = (idx >> upr_log2) & (PR - 1);
        const int ci = idx >> (upr_log2 + g.pr_log2);               // 0..KC-1
        const int t = t0 + 4 * i4;
        xspat[v] = (f0 + row) * a.T + t;
        xfrow[v] = f0 + row;
        xcs1[v] = ci * cs1;
        xcs2[v] = ci * cs2;
        xci4[v] = ci * 4;
        xcolbad[v] = t < a.T ? 0u : OOB;
        xleftbad[v] = (t < a.T && t > 0) ? 0u : OOB;
        xrightbad[v] = t + 4 < a.T ? 0u : OOB;
        xlds[v] = XSOA ? ci * 3 * NUNIT + (row << upr_log2) + i4 : ci * NUNIT + (row << upr_log2) + i4;
    }
    int wvo[WJ];                       // byte offset of this thread's float4 inside a weight slab, and its LDS slot
#pragma unroll
    for (int jj = 0; jj < WJ; ++jj) {
        const int idx = tid + jj * NTH;
        const int wt = idx / (KC * BN);
        const int rem = idx - wt * (KC * BN);
        const int ci_l = rem / BN;
        const int co_l = rem - ci_l * BN;
        wvo[jj] = ((ci_l * 2 + wt) * g.CoutP + co0 + co_l) * 16;
    }

    f32x16 acc[NTW][3];
#pragma unroll
    for (int i = 0; i < NTW; ++i)
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][p][r] = 0.f;

    // staging registers (ONE set): raw loads of the slab that is next to be transformed
    f32x4 xv[XJ];
    float xl[XJ], xrr[XJ], xsc[XJ];

    // slab cursor -> scalars.  (kh, ci0) of slab s, clamped to the last slab.
    auto issue_act = [&](int kh, int ci0) {
        const int foff = (kh - khc) * a.dil;
        const bool s2 = ci0 >= split;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(s2 ? p2 : p1), 0, s2 ? nb2 : nb1, 0x00020000);
        const int sbase = (s2 ? (ci0 - split) * cs2 : ci0 * cs1) + foff * a.T;          // elements, scalar
#pragma unroll
        for (int v = 0; v < XJ; ++v) {
            const int fr = xfrow[v] + foff;
            const unsigned rowbad = (unsigned)(fr | (a.F - 1 - fr)) & OOB;               // fr < 0 or fr >= F
            const unsigned e = (unsigned)((xspat[v] + (s2 ? xcs2[v] : xcs1[v]) + sbase) * 4) | rowbad;
            xv[v] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, e | xcolbad[v], 0, 0));
            xl[v] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (e - 4u) | rowbad | xleftbad[v], 0, 0));
            xrr[v] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (e + 16u) | rowbad | xrightbad[v], 0, 0));
            // (offset entirely in the VGPR operand: only that one is range-checked against Cin)
            if (HAS_ISC) xsc[v] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsi, xci4[v] + ci0 * 4, 0, 0));
        }
    };
    auto store_act = [&](f32x4* buf) {
#pragma unroll
        for (int v = 0; v < XJ; ++v) {
            float d0 = xl[v], d1 = xv[v][0], d2 = xv[v][1], d3 = xv[v][2], d4 = xv[v][3], d5 = xrr[v];
            if (HAS_ISC) {
                const float s = xsc[v];
                d0 *= s; d1 *= s; d2 *= s; d3 *= s; d4 *= s; d5 *= s;
            }
            const float e = d4 - 4.f * d2, o = d3 - 4.f * d1;        // shared by U1/U2
            const float e2 = d4 - d2, o2 = 2.f * (d3 - d1);          // shared by U3/U4
            if (XSOA) {
                float* xf = reinterpret_cast<float*>(buf) + xlds[v];
                xf[0] = 4.f * d0 - 5.f * d2 + d4;
                xf[NUNIT] = e + o;
                xf[2 * NUNIT] = e - o;
                xf[KC * 3 * NUNIT] = e2 + o2;
                xf[KC * 3 * NUNIT + NUNIT] = e2 - o2;
                xf[KC * 3 * NUNIT + 2 * NUNIT] = 4.f * d1 - 5.f * d3 + d5;
            } else {
                buf[xlds[v]] = f32x4{4.f * d0 - 5.f * d2 + d4, e + o, e - o, 0.f};
                buf[KC * NUNIT + xlds[v]] = f32x4{e2 + o2, e2 - o2, 4.f * d1 - 5.f * d3 + d5, 0.f};
            }
        }
    };
    auto dma_w = [&](int kh, int ci0, f32x4* buf, int j0, int j1) {
        const int so = (kh * g.CinP + ci0) * 2 * g.CoutP * 16;       // bytes, scalar
#pragma unroll
        for (int jj = j0; jj < j1; ++jj)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, LDS_PTR(buf + XSZ + jj * NTH + wave * 64), 16, wvo[jj], so, 0, 0);
    };
    auto advance = [&](int& kh, int& ci0) {             // next slab, clamped at the last one
        int nc = ci0 + KC, nk = kh;
        if (nc >= g.CinP) {
            nc = 0;
            ++nk;
        }
        if (nk <= kh_hi) {
            kh = nk;
            ci0 = nc;
        }
    };

    const int boff = XSOA ? (tr * KC + h) * 3 * NUNIT + wc * 32 + l31 : (tr * KC + h) * NUNIT + wc * 32 + l31;
    const int aoff = XSZ + (tr * KC + h) * BN + wr * (NTW * 32) + l31;

    // ---- prologue: slabs 0 and 1 into buffers 0 and 1, loads of slab 2 in flight
    int kA = kh_lo, cA = 0;                      // cursor of the slab whose activations are in the staging registers
    issue_act(kA, cA);
    dma_w(kA, cA, smem, 0, WJ);
    store_act(smem);
    int kW = kA, cW = cA;                        // cursor of the slab whose weights are DMA'd next
    advance(kA, cA);
    advance(kW, cW);
    issue_act(kA, cA);
    dma_w(kW, cW, smem + BUF, 0, WJ);
    store_act(smem + BUF);
    advance(kA, cA);
    advance(kW, cW);
    issue_act(kA, cA);                            // slab 2 (or a clamped copy of the last slab)
    __syncthreads();

    f32x4 av[2][NTW], bv[2];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) av[0][nt] = smem[aoff + nt * 32];
    if (XSOA) {
        const float* xf = reinterpret_cast<const float*>(smem) + boff;
        bv[0] = f32x4{xf[0], xf[NUNIT], xf[2 * NUNIT], 0.f};
    } else {
        bv[0] = smem[boff];
    }

    int rb = 0;                                   // ring slot of the slab being multiplied
    for (int j = 0; j < nslab; ++j) {
        const int rn = rb == 2 ? 0 : rb + 1;      // slab j+1
        const int rw = rn == 2 ? 0 : rn + 1;      // slab j+2: staged during this slab
        const f32x4* Xs = smem + rb * BUF;
        const f32x4* Xn = smem + rn * BUF;
        f32x4* Xw = smem + rw * BUF;
#define MFMA_STEP(c)                                                                                        \
    _Pragma("unroll") for (int nt = 0; nt < NTW; ++nt) _Pragma("unroll") for (int p = 0; p < 3; ++p)        \
        acc[nt][p] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c][nt][p], bv[c][p], acc[nt][p], 0, 0, 0);
#define READ_STEP(c, base, st)                                                                              \
    _Pragma("unroll") for (int nt = 0; nt < NTW; ++nt) av[c][nt] = (base)[aoff + 2 * (st) * BN + nt * 32];  \
    if (XSOA) {                                                                                             \
        const float* xf_ = reinterpret_cast<const float*>(base) + boff + 2 * (st) * 3 * NUNIT;              \
        bv[c] = f32x4{xf_[0], xf_[NUNIT], xf_[2 * NUNIT], 0.f};                                             \
    } else {                                                                                                \
        bv[c] = (base)[boff + 2 * (st) * NUNIT];                                                            \
    }
        // K-step 0: transform + write the staged activations of slab j+2, re-issue the staging loads (slab j+3), first
        // half of the weight DMA of slab j+2 (issued early: it has to land before the barrier at the end of this slab)
        READ_STEP(1, Xs, 1)
        store_act(Xw);
        advance(kA, cA);
        issue_act(kA, cA);
        dma_w(kW, cW, Xw, 0, WJ / 2);
        MFMA_STEP(0)
        __builtin_amdgcn_sched_barrier(0);
        // K-step 1: second half of the weight DMA
        READ_STEP(0, Xs, 2)
        dma_w(kW, cW, Xw, WJ / 2, WJ);
        MFMA_STEP(1)
        __builtin_amdgcn_sched_barrier(0);
        // K-step 2
        READ_STEP(1, Xs, 3)
        MFMA_STEP(0)
        __builtin_amdgcn_sched_barrier(0);
        // K-step 3: first operands of slab j+1 (its buffer was completed by the PREVIOUS barrier)
        advance(kW, cW);
        READ_STEP(0, Xn, 0)
        MFMA_STEP(1)
        __syncthreads();                           // slab j+2 complete (DMA + ds_write), slab j's buffer free
        rb = rn;
    }
#undef MFMA_STEP
#undef READ_STEP

    // ---- output transform (identical to conv_wino4_kernel).  With a = (M0+M1+M2, M1-M2, M1+M2) from the triple-0 wave
    // and b = (M3+M4, 2(M3-M4), M5) from the triple-1 wave:  y0 = a0+b0, y1 = a1+b1, y2 = a2+4 b0, y3 = a1+4 b1+M5.
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
#endif
}

inline int ilog2_floor_p(int v) {
    int l = 0;
    while ((1 << (l + 1)) <= v) ++l;
    return l;
}
inline int ilog2_ceil_p(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}

template <int NTW, int WR, int WC, bool XSOA = false>
void launch4p(const babe_conv_args& a, Wino4pGeom g, const float* wq, hipStream_t s) {
    constexpr int NPOS = 128 * WC;
    const int npos_log2 = ilog2_floor_p(NPOS);
    g.pt_log2 = ilog2_ceil_p(a.T);
    if (g.pt_log2 > npos_log2) g.pt_log2 = npos_log2;
    if (g.pt_log2 < 4) g.pt_log2 = 4;
    g.pr_log2 = npos_log2 - g.pt_log2;
    const int PT = 1 << g.pt_log2, PR = 1 << g.pr_log2;
    g.tiles_t = cdiv(a.T, PT);
    const int tiles_f = cdiv(a.F, PR);
    constexpr int BN = WR * NTW * 32;
    dim3 grid(g.tiles_t * tiles_f, g.CoutP / BN, a.B);
    size_t lds = 3 * (size_t)((XSOA ? 2 * 8 * 3 * (WC * 32) / 4 : 2 * 8 * (WC * 32)) + 2 * 8 * BN) * 16;
    const size_t ex = (size_t)WR * WC * 3072 * 4;
    if (ex > lds) lds = ex;
    static std::atomic<unsigned long long> attr_done{0};          // 144 KB of dynamic LDS: above the 64 KB default cap
    if (babe_lds_optin(attr_done, {reinterpret_cast<const void*>(&conv_wino4p_kernel<NTW, WR, WC, true, XSOA>),
                                   reinterpret_cast<const void*>(&conv_wino4p_kernel<NTW, WR, WC, false, XSOA>)},
                       (int)lds) != hipSuccess)
        return;                                                   // (the caller's BABE_LAUNCH_CHECK reports it)
    if (a.in_scale)
        hipLaunchKernelGGL((conv_wino4p_kernel<NTW, WR, WC, true, XSOA>), grid, dim3(128 * WR * WC), lds, s, a, g, wq);
    else
        hipLaunchKernelGGL((conv_wino4p_kernel<NTW, WR, WC, false, XSOA>), grid, dim3(128 * WR * WC), lds, s, a, g, wq);
}

}  // namespace

/* 1 if the pipelined kernel takes this problem (the caller has already checked babe_conv2d_wino4_supported) */
int babe_conv2d_wino4p_supported(const babe_conv_args& a) {
    static const char* ov = getenv("BABE_CONV_WINO4P");
    if (ov && ov[0] == '0') return 0;
    const int n32 = (a.Cout + 31) / 32;
    if (n32 % 4 != 0 && n32 != 2 && n32 != 3) return 0;   // 8-wave workgroups: 128 co x 256 pos, 64 / 96 co x 512 pos
    if (a.in2 && (a.cin_split % 8 != 0)) return 0;
    const long lim = 0x7fffffffL / 4;
    const int split = a.in2 ? a.cin_split : a.Cin;
    if ((long)split * a.in_cs >= lim) return 0;
    if (a.in2 && (long)(a.Cin - split) * a.in2_cs >= lim) return 0;
    if ((long)a.KH * ((a.Cin + 7) / 8 * 8) * 2 * ((a.Cout + 31) / 32 * 32) * 16 >= 0x7fffffffL) return 0;
    return 1;
}

int babe_conv2d_wino4p_launch(const babe_conv_args& a, const float* w_wino4, hipStream_t s) {
    Wino4pGeom g;
    g.CinP = (a.Cin + 7) / 8 * 8;
    g.CoutP = (a.Cout + 31) / 32 * 32;
    if (g.CoutP == 64) launch4p<2, 1, 4>(a, g, w_wino4, s);             //  64 co x 512 positions, 8 waves
    else if (g.CoutP == 96) launch4p<3, 1, 4, true>(a, g, w_wino4, s);  //  96 co x 512 positions, phase-planar activations
    else launch4p<2, 2, 2>(a, g, w_wino4, s);                           // 128 co x 256 positions, 8 waves
    return 0;
}
