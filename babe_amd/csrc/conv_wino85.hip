// Nested Winograd F(4,5) along FREQUENCY x F(4,3) along TIME for the
// frequency-dilated (5,3) Conv2d (networks/cqtdiff+.py:79-88, 433-436), fp32 MFMA.  A unit = 4 output rows of one residue class
// (f, f + d, f + 2d, f + 3d) x 4 time steps from an 8-row x 6-sample patch: 8 x 6 = 48 products per (ci, co) per 16 outputs =
// 3.0 per output (conv_wino45.hip: 4.5).  Interpolation points 0, +-1, +-2, +-1/2, inf along frequency (the classic 8-point set),
// 0, +-1, +-2, inf along time (as conv_wino45.hip).  Paper estimate and rounding study: profiles/r05_wino_f45_estimate.txt.
//   frequency input transform B^T (rows of the 8-row patch d0..d7):
//     p0  (0)    = -d0 + 21/4 (d2 - d4) + d6                     p7 (inf) = -d1 + 21/4 (d3 - d5) + d7
//     p1,2 (+-1) = (d2 - 17/4 d4 + d6) +- (d1 - 17/4 d3 + d5)
//     p3,4 (+-2) = (1/4 d2 - 5/4 d4 + d6) +- (1/2 d1 - 5/2 d3 + 2 d5)
//     p5,6 (+-1/2) = (4 d2 - 5 d4 + d6) +- (2 d1 - 5/2 d3 + 1/2 d5)
//   output A^T (4 rows x 8 phases): [1 1 1 1 1 1 1 0; 0 1 -1 2 -2 1/2 -1/2 0; 0 1 1 4 4 1/4 1/4 0; 0 1 -1 8 -8 1/8 -1/8 1]
// 48 accumulators per (co, unit) do not fit: TWO PASSES of four frequency phases, A = (p1, p2, p3, p4) - rows 1..6 only - and
// B = (p5, p6, p0, p7); the partial sums of pass A are carried inside the accumulators of pass B: with C_X = A^T[:, X],
// M'_B = C_B^-1 C_A M_A:   p5' = 3 m1 + m2 + 10 m3 + 6 m4,  p6' = m1 + 3 m2 + 6 m3 + 10 m4,  p0' = -3 (m1 + m2) - 15 (m3 + m4),
//                          p7' = 3/4 (m1 - m2) + 15/2 (m3 - m4).
// Tile: 128 output channels x 16 units (one row quad x 64 steps), 8 waves; wave w owns channel tile w: 4 x 6 = 24 accumulators of
// v_mfma_f32_16x16x4_f32 (96 registers) = ONE (tile, segment) item per wave.  One barrier per 16-channel super-slab; a super-slab is 8
// half-slots (4 ci x 16 co x 12 of the pass's 24 phases) = 24 MFMA groups of 4.  The WEIGHT operand never touches LDS: the 96
// accumulators leave room for a ring of 12 operand registers sets, loaded straight from global memory (L2-resident packed image, one
// coalesced 1 KB load per wave and group, 12 groups ahead); LDS holds only the transformed activations (2 x 24 KB).  The first build
// of this kernel staged the weights through a wave-private LDS-DMA ring like conv_wino45x_kernel: its ablations (tools/f45_ablate.py,
// profiles/r05_f45_ablate.txt) charged 20 % of the time to the DMA and 18 % to the operand reads - LDS bandwidth was the limit.
// Transform (128-channel kernel): thread = (ci, unit) in waves 0-3 (wave w and w + 4 share a SIMD): they load the rows once and compute
// both phase pairs of the pass, coefficients wave-uniform; waves 4-7 only multiply (W85_HALFLOAD; the first form - every wave one
// pair from its own copy of the rows - is the 0 setting).
// 96- and 64-channel tiles: conv_wino85s_kernel below - waves 0 .. NW-1 only multiply, waves NW .. 7 only load and transform, which
// balances the four SIMDs where six (four) multiplying waves alone cannot.  All three produce the same sums in the same order for a
// given (co, output): a conv run as 64-channel tiles equals the same conv run as one 128-channel tile bit for bit.
// Round 6: the 128-channel tile runs by default as conv_wino85s_kernel<., 8, 12> - 8 multiplying + 4 transform waves, three per SIMD
// at 159 registers (babe_conv2d_wino85_set_waves / BABE_W85_12W=0 for the 8-wave form below): +3 % on the whole job.
// Measured (profiles/r05_f45_check.txt, MI355X, us per launch F45 / F(2,5)xF(4,3)): 128 ch 164 / 205, 256 ch 273 / 350, 96 ch
// 212 / 268, 64 ch 109 / 145; whole job 2.136 -> 2.40 audio-sec/s.
#include "common.h"
#include "../../include/babe_hip.h"
#include "prof.h"
#include "gelu.h"
#include <atomic>
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifndef W85_HALFLOAD
#define W85_HALFLOAD 1 // in the 128-channel kernel waves 0-3 load the rows and transform BOTH phase pairs of their (ci, unit), waves 4-7
#endif                 // only multiply (0: every wave loads and transforms one pair - twice the row loads; -0.6 % on the whole job)
#ifndef W85_RD
#define W85_RD 12      // weight register ring: groups in flight (a multiple of 3 that divides 24)
#endif
#ifndef W85_TFIRST
#define W85_TFIRST 1    // specialised-wave kernels: the transform waves are the OLDEST waves of the workgroup (0: the youngest, round 5 - 6a)
#endif
// W85_FLAGS (round 6, the default): no barrier inside the main loop of the specialised-wave kernels.  THREE X buffers; every transform
// wave publishes the number of super-slabs it has finished (prog[t]), every multiplying wave the number it has consumed (cons[w]), in
// LDS words of their own (no atomics); a consumer polls the words of the other side (one ds_read_b128 per side and super-slab when
// nothing is late).  A wave then waits only for data it needs, not for the slowest wave of the workgroup: the first multiplying wave
// of a SIMD, which the oldest-first issue order lets through a super-slab in half its time, goes on into the next one and fills the
// matrix pipe while its partner waits for weights.  Same arithmetic in the same order: bit-identical to the barrier form
// (-DW85_FLAGS=0, which the s_memtime probe needs).  +1.0 ... 1.25 % on the job, profiles/r06_f45_flags_ab.txt.
#ifndef W85_FLAGS
#define W85_FLAGS 1
#endif
#ifndef W85_NXB
#define W85_NXB 3       // X buffers of the flag-synchronised form (3 or 4)
#endif
#ifndef W85_SLEEP
#define W85_SLEEP 1     // s_sleep argument inside its polling loops (0: none)
#endif
#ifndef W85_TPRIO
#define W85_TPRIO 0
#endif
// timing probe (ablation bit 16384, tools/f45_barrier_probe.py): cycles a wave spends at the per-super-slab barrier and in total,
// written by lane 0 of the first multiplying and the first transform wave to stat_part[tile * 4 ..] (stat_mode 99)
#define W85_PROBE (W85_ABL & 16384)
#if W85_PROBE && W85_FLAGS
#error "the barrier probe instruments the barrier form: build with -DW85_FLAGS=0"
#endif
#if W85_PROBE
#define W85_BARRIER(acc_)                                              \
    {                                                                  \
        const unsigned long long tb_ = __builtin_amdgcn_s_memtime();   \
        __builtin_amdgcn_s_barrier();                                  \
        acc_ += __builtin_amdgcn_s_memtime() - tb_;                    \
    }
#else
#define W85_BARRIER(acc_) __builtin_amdgcn_s_barrier();
#endif
#ifndef W85_ABL
#define W85_ABL 0      // timing ablations only (tools/f45_ablate.py): 1 no transform arithmetic, 2 no row loads, 4 no weight loads,
#endif                 // 8 no X reads, 16 no MFMA, 32 rows loaded by waves 0-3 only, 64 / 128 row loads that always hit L2 / L1, 512 no epilogue, 1024 no pass carry - results are wrong

namespace {

struct Wino85Geom {
    int CinP, CoutP, tiles_t, nquads;      // nquads: row quads per residue class
    int xcd, per_xcd, total, ncb;          // XCD-contiguous tile order: tiles per XCD, tiles per batch item, channel blocks
};

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
constexpr unsigned OOBH = 0xC0000000u;

// ---- arithmetic shared by the two kernels, written with explicit fused multiply-adds and contraction off, so that the 128-channel
// kernel and the specialised-wave kernel round identically (tests/test_gpu_ops.py::test_f45_tile_widths_are_bit_identical)
__device__ __forceinline__ float w85_dpp_shr1(float old, float src) {
    asm("s_nop 1\n\tv_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(old) : "v"(src));
    return old;
}
__device__ __forceinline__ float w85_dpp_shl1(float old, float src) {
    asm("s_nop 1\n\tv_mov_b32_dpp %0, %1 row_shl:1 row_mask:0xf bank_mask:0xf" : "+v"(old) : "v"(src));
    return old;
}
// time transform B^T of F(4,3) (points 0, +-1, +-2, inf) of six samples
__device__ __forceinline__ void w85_tt(const float (&E)[6], float (&U)[6]) {
#pragma clang fp contract(off)
    const float e = __builtin_fmaf(-4.f, E[2], E[4]), o = __builtin_fmaf(-4.f, E[1], E[3]);
    const float e2 = E[4] - E[2], o2 = E[3] - E[1];
    U[0] = __builtin_fmaf(4.f, E[0], __builtin_fmaf(-5.f, E[2], E[4]));
    U[1] = e + o;
    U[2] = e - o;
    U[3] = __builtin_fmaf(2.f, o2, e2);
    U[4] = __builtin_fmaf(-2.f, o2, e2);
    U[5] = __builtin_fmaf(4.f, E[1], __builtin_fmaf(-5.f, E[3], E[5]));
}
// one thread's (ci, unit): patch rows xv[0..7] (4 samples each), halo samples xh[0..7] (left neighbour's last / right neighbour's
// first sample, by DPP inside the 16-lane row of units) -> the 12 transformed values of phase pair sel = 2 pass + half:
//   X = c0 d0 + c2 d2 + c4 d4 + d6,  Y = c1 d1 + c3 d3 + c5 d5 + c7 d7,  (Ea, Eb) = (X + Y, X - Y)   [sel 3: (X, Y)]
//   sel 0 p1/p2: c2 1     c4 -17/4  c1 1    c3 -17/4  c5 1
//   sel 1 p3/p4: c2 1/4   c4 -5/4   c1 1/2  c3 -5/2   c5 2
//   sel 2 p5/p6: c2 4     c4 -5     c1 2    c3 -5/2   c5 1/2
//   sel 3 p0/p7: c0 -1 c2 21/4 c4 -21/4   c1 -1 c3 21/4 c5 -21/4 c7 1
// (coefficients selected as integers so that they stay in scalar registers: a float ?: chain became a tree of branches)
template <bool HAS_ISC>
__device__ __forceinline__ void w85_transform(const f32x4 (&xv)[8], const float (&xh)[8], int sel, float xsc, f32x4& o0, f32x4& o1,
                                              f32x4& o2) {
#pragma clang fp contract(off)
    auto pick = [&](unsigned v0, unsigned v1, unsigned v2, unsigned v3) __attribute__((always_inline)) {
        return __builtin_bit_cast(float, sel == 0 ? v0 : (sel == 1 ? v1 : (sel == 2 ? v2 : v3)));
    };
    const float c2 = pick(0x3f800000u, 0x3e800000u, 0x40800000u, 0x40a80000u);        // 1.0 0.25 4.0 5.25
    const float c4 = pick(0xc0880000u, 0xbfa00000u, 0xc0a00000u, 0xc0a80000u);        // -4.25 -1.25 -5.0 -5.25
    const float c1 = pick(0x3f800000u, 0x3f000000u, 0x40000000u, 0xbf800000u);        // 1.0 0.5 2.0 -1.0
    const float c3 = pick(0xc0880000u, 0xc0200000u, 0xc0200000u, 0x40a80000u);        // -4.25 -2.5 -2.5 5.25
    const float c5 = pick(0x3f800000u, 0x40000000u, 0x3f000000u, 0xc0a80000u);        // 1.0 2.0 0.5 -5.25
    const bool special = sel == 3;                      // (wave-uniform: the only user of rows 0 and 7)
    float Ea[6], Eb[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        float d[8];
#pragma unroll
        for (int r = 1; r < 7; ++r) d[r] = j == 0 ? w85_dpp_shr1(xh[r], xv[r][3]) : (j == 5 ? w85_dpp_shl1(xh[r], xv[r][0]) : xv[r][j - 1]);
        float X = __builtin_fmaf(c2, d[2], __builtin_fmaf(c4, d[4], d[6]));
        float Y = __builtin_fmaf(c1, d[1], __builtin_fmaf(c3, d[3], c5 * d[5]));
        if (special) {
            d[0] = j == 0 ? w85_dpp_shr1(xh[0], xv[0][3]) : (j == 5 ? w85_dpp_shl1(xh[0], xv[0][0]) : xv[0][j - 1]);
            d[7] = j == 0 ? w85_dpp_shr1(xh[7], xv[7][3]) : (j == 5 ? w85_dpp_shl1(xh[7], xv[7][0]) : xv[7][j - 1]);
            Ea[j] = X - d[0];
            Eb[j] = Y + d[7];
        } else {
            Ea[j] = X + Y;
            Eb[j] = X - Y;
        }
    }
    if (HAS_ISC) {
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            Ea[j] *= xsc;
            Eb[j] *= xsc;
        }
    }
    float Ua[6], Ub[6];
    w85_tt(Ea, Ua);
    w85_tt(Eb, Ub);
    o0 = f32x4{Ua[0], Ua[1], Ua[2], Ua[3]};
    o1 = f32x4{Ua[4], Ua[5], Ub[0], Ub[1]};
    o2 = f32x4{Ub[2], Ub[3], Ub[4], Ub[5]};
}
// pass boundary: carry the finished phases (p1, p2, p3, p4) into the accumulators of (p5, p6, p0, p7)
__device__ __forceinline__ void w85_carry(f32x4 (&acc)[2][12]) {
#pragma clang fp contract(off)
    asm volatile("s_nop 15\n\ts_nop 15");
#pragma unroll
    for (int p = 0; p < 6; ++p)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float m1 = acc[0][p][e], m2 = acc[0][6 + p][e], m3 = acc[1][p][e], m4 = acc[1][6 + p][e];
            const float s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
            acc[0][p][e] = __builtin_fmaf(2.f, s12, d12) + __builtin_fmaf(8.f, s34, 2.f * d34);          // 3 m1 + m2 + 10 m3 + 6 m4
            acc[0][6 + p][e] = __builtin_fmaf(2.f, s12, -d12) + __builtin_fmaf(8.f, s34, -2.f * d34);    // m1 + 3 m2 + 6 m3 + 10 m4
            acc[1][p][e] = __builtin_fmaf(-3.f, s12, -15.f * s34);
            acc[1][6 + p][e] = __builtin_fmaf(0.75f, d12, 7.5f * d34);
        }
}
// output: rows r = 0..3 from (M5, M6, M0, M7) = acc[0][0..5], acc[0][6..11], acc[1][0..5], acc[1][6..11]; cot0 = the wave's first channel
// tile / ntiles: this workgroup's index in the launch's list of (row quad, time tile) tiles of a batch item and the list's length -
// the slot of the optional fused reduction (babe_conv_args::stat_mode, include/babe_hip.h)
__device__ __forceinline__ void w85_epilogue(const babe_conv_args& a, const f32x4 (&acc)[2][12], int b, int cot0, int fa, int t0, int lk,
                                             int l15, int tile, int ntiles) {
#pragma clang fp contract(off)
    const bool has_os = a.oscale != nullptr, has_res = a.res != nullptr;
    const int t = t0 + 4 * l15;
    const int smode = a.stat_mode;                      // (kernel-uniform)
    double ssum = 0.0, ssq = 0.0;                        // sums over this lane's 4 channels x 4 rows x 4 steps (mode 1: y, y^2; mode 2: ssum)
    // Every load of the epilogue is unconditional (an out-of-range point reads the channel's first float4 and is masked at the
    // store) and issued a row ahead of its use: with the loads under `if (pv)` each of the 16 (row, channel) steps waited out
    // its own memory latency (s_waitcnt vmcnt(0) per step in the ISA).
    // Offsets inside one batch element are 32-bit (babe_conv2d_wino85_supported bounds Cout * out_cs and Cout * res_cs).
    float os[4] = {1.f, 1.f, 1.f, 1.f}, scx[4] = {0.f, 0.f, 0.f, 0.f};
    const int co0 = cot0 + 4 * lk;
    if (has_os) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) os[kk] = a.oscale[b * a.Cout + co0 + kk];
    }
    if (smode == 2) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) scx[kk] = a.stat_scale[b * a.Cout + co0 + kk];
    }
    float* const outb = a.out + (long)b * a.out_bs;
    const int ocs = (int)a.out_cs;
    // res and stat_x are never both given (the launcher rejects it): one staging array, two rows in flight
    const float* const lsrc = smode == 2 ? a.stat_x + (long)b * a.out_bs : (has_res ? a.res + (long)b * a.res_bs : nullptr);
    const int lcs = smode == 2 ? ocs : (int)a.res_cs;
    const bool has_ld = lsrc != nullptr;
    f32x4 ld[4][4];
    bool pvr[4];
    int spr[4];
#pragma unroll
    for (int row = 0; row < 4; ++row) {
        const int f = fa + row * a.dil;
        pvr[row] = f < a.F && t < a.T;
        spr[row] = pvr[row] ? f * a.T + t : 0;
    }
#pragma unroll
    for (int row = 0; row < 4; ++row)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) ld[row][kk] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (has_ld) {
#pragma unroll
        for (int row = 0; row < 2; ++row)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) ld[row][kk] = *reinterpret_cast<const f32x4*>(lsrc + ((co0 + kk) * lcs + spr[row]));
    }
#pragma unroll
    for (int row = 0; row < 4; ++row) {
        const bool pv = pvr[row];
        const int sp = spr[row];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            float m[6];
            if (W85_ABL & 4096) {                       // (no output transform: raw accumulators stored)
                const f32x4 y = {acc[0][row][kk], acc[0][6 + row][kk], acc[1][row][kk], acc[1][6 + row][kk]};
                if (pv) *reinterpret_cast<f32x4*>(outb + ((co0 + kk) * ocs + sp)) = y;
                continue;
            }
#pragma unroll
            for (int tp = 0; tp < 6; ++tp) {
                const float M5 = acc[0][tp][kk], M6 = acc[0][6 + tp][kk], M0 = acc[1][tp][kk], M7 = acc[1][6 + tp][kk];
                m[tp] = row == 0 ? (M5 + M6) + M0 : (row == 1 ? 0.5f * (M5 - M6) : (row == 2 ? 0.25f * (M5 + M6) : __builtin_fmaf(0.125f, M5 - M6, M7)));
            }
            const float s12 = m[1] + m[2], d12 = m[1] - m[2], s34 = m[3] + m[4], d34 = m[3] - m[4];
            const float y0 = m[0] + s12 + s34, y1 = __builtin_fmaf(2.f, d34, d12), y2 = __builtin_fmaf(4.f, s34, s12),
                        y3 = __builtin_fmaf(8.f, d34, d12) + m[5];
            const float sc = a.alpha * os[kk];
            const f32x4 r4 = smode == 2 ? f32x4{0.f, 0.f, 0.f, 0.f} : ld[row][kk];
            f32x4 y;
            y[0] = __builtin_fmaf(y0, sc, a.rbeta * r4[0]);
            y[1] = __builtin_fmaf(y1, sc, a.rbeta * r4[1]);
            y[2] = __builtin_fmaf(y2, sc, a.rbeta * r4[2]);
            y[3] = __builtin_fmaf(y3, sc, a.rbeta * r4[3]);
            if (W85_ABL & 2048) {                       // (no stores: the arithmetic stays)
                if (y[0] == 12345.f && y[1] == 5.f) *reinterpret_cast<f32x4*>(a.out) = y;
            } else if (pv) *reinterpret_cast<f32x4*>(outb + ((co0 + kk) * ocs + sp)) = y;
            if (smode == 1 && pv) {
                // babe_gn_partial's sums of the output this conv writes (csrc/norm.hip: sum and sum of squares in double)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const double v = (double)y[e];
                    ssum += v;
                    ssq += v * v;
                }
            }
            if (smode == 2 && pv) {
                // babe_gn_bwd_partial's term for these four outputs (csrc/norm.hip: dv = da * gelu'(x * sc) in float, dv * x summed
                // in double, times sc per channel): y IS the da this conv writes
                const f32x4 x4 = ld[row][kk];
                double cs = 0.0;
#pragma unroll
                for (int e = 0; e < 4; ++e) cs += (double)(y[e] * babe_gelu::gelu_grad_f(x4[e] * scx[kk])) * (double)x4[e];
                ssum += (double)scx[kk] * cs;
            }
        }
        if (row < 2 && has_ld) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) ld[row + 2][kk] = *reinterpret_cast<const f32x4*>(lsrc + ((co0 + kk) * lcs + spr[row + 2]));
        }
    }
    if (smode == 1 || smode == 2) {
        // The 16 lanes of a row (same lk) hold the same channel quad, the wave 16 consecutive channels: a slot = the largest run of
        // channels (4, 8 or 16) that divides the group size, summed in the wave in a fixed order and written exactly once.
        const int sg = (a.stat_cg & 15) == 0 ? 16 : ((a.stat_cg & 7) == 0 ? 8 : 4);
#pragma unroll
        for (int o = 8; o >= 1; o >>= 1) {
            ssum += __shfl_xor(ssum, o, 16);
            if (smode == 1) ssq += __shfl_xor(ssq, o, 16);
        }
        if (sg >= 8) {
            ssum += __shfl_xor(ssum, 16, 64);
            if (smode == 1) ssq += __shfl_xor(ssq, 16, 64);
        }
        if (sg == 16) {
            ssum += __shfl_xor(ssum, 32, 64);
            if (smode == 1) ssq += __shfl_xor(ssq, 32, 64);
        }
        if (l15 == 0 && ((4 * lk) & (sg - 1)) == 0) {
            const int spg = a.stat_cg / sg, cu = (cot0 + 4 * lk) / sg;      // slots per (tile, group); this slot's channel run
            const int g = cu / spg, q = cu - g * spg;
            const long S = (long)ntiles * spg;
            const long idx = ((long)b * (a.Cout / a.stat_cg) + g) * S + (long)tile * spg + q;
            if (smode == 1) {
                a.stat_part[2 * idx] = ssum;
                a.stat_part[2 * idx + 1] = ssq;
            } else
                a.stat_part[idx] = ssum;
        }
    }
}

template <bool HAS_ISC>
__global__ __launch_bounds__(512, 1) void conv_wino85_kernel(babe_conv_args a, Wino85Geom g, const float* __restrict__ wq) {
#if __HIP_DEVICE_COMPILE__
    constexpr int KS = 16, KQ = 4, NU = 16, BN = 128;
    constexpr int XSZ = KS * NU * 6;                    // float4 per activation super-slab (16 ci x 16 units x 24 floats)
    constexpr int RD = W85_RD;                          // weight register ring: groups in flight (a multiple of 3 that divides 24)
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    f32x4* smem = reinterpret_cast<f32x4*>(smem_f);
    f32x4* const Xb = smem;                             // X[2]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lk = lane >> 4;
    const int b = blockIdx.z;
    // XCD-aware tile order: workgroups go to the 8 XCDs round-robin by linear id, each XCD has its own L2.  The tiles that read the
    // same input rows - the channel blocks of a tile, then the neighbouring row quads of a residue class (4 of their 8 rows are
    // shared) - are handed to ONE XCD as a contiguous chunk of the list [time tile][class][quad][channel block].
    int bx = blockIdx.x, by = blockIdx.y;
    if (g.xcd) {
        const int lin = (bx & 7) * g.per_xcd + (bx >> 3);
        if ((bx >> 3) >= g.per_xcd || lin >= g.total) return;
        by = lin % g.ncb;
        bx = lin / g.ncb;
    }
    const int co0 = by * BN;
    const int nQ = a.dil * g.nquads;
    const int tile_t = g.xcd ? bx / nQ : bx % g.tiles_t;
    const int Q = g.xcd ? bx % nQ : bx / g.tiles_t;
    const int t0 = tile_t * 64;
    const int cls = Q / g.nquads;
    const int fa = Q < a.dil * g.nquads ? cls + 4 * (Q - cls * g.nquads) * a.dil : a.F + 8 * a.dil;   // first output row
    const int NS = 2 * (g.CinP / KS);                   // super-slabs

    const float* p1 = a.in + (long)b * a.in_bs;
    const int cs1 = (int)a.in_cs;
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)p1, 0, a.Cin * cs1 * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)wq, 0, 4 * g.CinP * g.CoutP * 48, 0x00020000);

    // ---- staging constants: thread = (ci = 4 (wave & 3) + (lane >> 4), unit = lane & 15); half = wave >> 2
    const int half = wave >> 2, wq4 = wave & 3;
    const int s_tu = lane & 15, s_ch = lane >> 4;
    const int s_t = t0 + 4 * s_tu;
    const unsigned chb = (unsigned)(s_ch * cs1 * 4);
    unsigned er[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int fr = fa + (r - 2) * a.dil;
        const bool ok = fr >= 0 && fr < a.F && s_t < a.T;
        er[r] = ok ? (unsigned)((fr * a.T + s_t) * 4) + chb : OOBH;
        if (W85_ABL & 128) er[r] = (unsigned)((r * 64 + 4 * s_tu) * 4) + chb;       // (always the same 8 KB: L1 hits)
    }
    unsigned ehalo = OOBH;
    {
        const int hr = lane >> 3, hc = (lane >> 1) & 3, hs = lane & 1;       // 8 rows x 4 channels x 2 sides = 64 lanes
        const int fr = fa + (hr - 2) * a.dil;
        const int th = t0 + (hs ? 64 : -1);
        if (fr >= 0 && fr < a.F && th >= 0 && th < a.T) ehalo = (unsigned)((fr * a.T + th) * 4 + hc * cs1 * 4);
    }
    const int hsrc = (2 * s_ch + (s_tu == 15 ? 1 : 0)) * 4;               // bpermute byte index of row 0; row r adds 32
    const int xlds = ((wq4 * 4 + s_ch) * NU + s_tu) * 6 + half * 3;         // float4 index of this thread's 12 floats
    // weights: global -> registers, one coalesced 1 KB load per group ([lk][l15] float4 of the packed image
    // [pass][ci quad][half][co tile][pg][64 lanes][4]); the ring aw[RD] keeps RD groups in flight (the compiler counts vmcnt)
    const int NT = g.CoutP >> 4;
    const int wstep = NT * 3072;                        // next half-slot of this wave's channel tile
    const int sWend = 4 * g.CinP * g.CoutP * 48;
    int sW = ((co0 >> 4) + wave) * 3072;
    const unsigned wvo = (unsigned)(lane * 16);
    auto w_next = [&]() __attribute__((always_inline)) {
        if (W85_ABL & 256) return;                      // (always the first half-slot: L1 hits)
        sW += wstep;
        sW = sW < sWend ? sW : sWend;
    };
#define Y_FENCE __builtin_amdgcn_sched_barrier(0);

    f32x4 xv[8];
    xv[0] = xv[7] = f32x4{0.f, 0.f, 0.f, 0.f};
    float xhl = 0.f, xsc = 1.f;
    float xh[8];
    int pS = 0;
    // the patch rows of super-slab (ps, ci0): rows 1..6 always, rows 0 and 7 in pass B only (a wave-uniform branch; pass A's phases
    // +-1, +-2 do not touch them).  (hipcc counts the vmcnt of every wait from the loads it sees: nothing here is hand-counted.)
    auto issue_rows = [&](int ps, int ci0, int r0, int r1) __attribute__((always_inline)) {
        const int so = (W85_ABL & (64 | 128)) ? 0 : (ci0 + 4 * wq4) * cs1 * 4;       // (64: always the first 4 channels: L2 hits)
        if ((W85_ABL & 2) || ((W85_ABL & 32) && half) || (W85_HALFLOAD && half)) return;
#pragma unroll
        for (int r = 1; r < 7; ++r) {
            if (r < r0 || r >= r1) continue;
            xv[r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs1, er[r], so, 0));
        }
        if (ps == 1) {
            if (r0 == 0) xv[0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs1, er[0], so, 0));
            if (r1 == 8) xv[7] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs1, er[7], so, 0));
        }
    };
    auto issue_halo = [&](int ci0) __attribute__((always_inline)) {
        const int so = (ci0 + 4 * wq4) * cs1 * 4;
        if ((W85_ABL & 2) || ((W85_ABL & 32) && half) || (W85_HALFLOAD && half)) return;
        xhl = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs1, ehalo, so, 0));
    };
    auto issue_isc = [&](int ci0) __attribute__((always_inline)) {
        if (W85_HALFLOAD && half) return;
        if (HAS_ISC) xsc = a.in_scale[(long)b * a.Cin + ci0 + 4 * wq4 + s_ch];
    };
    auto halo_permute = [&]() __attribute__((always_inline)) {
        if (W85_HALFLOAD && half) return;
#pragma unroll
        for (int r = 0; r < 8; ++r)
            asm volatile("ds_bpermute_b32 %0, %1, %2 offset:%3" : "=v"(xh[r]) : "v"(hsrc), "v"(xhl), "n"(32 * r));
    };
    auto store_act = [&](f32x4* buf) __attribute__((always_inline)) {
        if (W85_ABL & 1) {
            buf[xlds] = xv[1] + xv[0];
            buf[xlds + 1] = xv[2] + xv[7];
            buf[xlds + 2] = xv[3] + f32x4{xh[1], xh[2], xh[0], xh[7]};
            return;
        }
        if (W85_HALFLOAD) {                                  // waves 0-3: both phase pairs of their (ci, unit) from one set of rows
            if (half) return;
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xh[0]), "+v"(xh[1]), "+v"(xh[2]), "+v"(xh[3]), "+v"(xh[4]), "+v"(xh[5]), "+v"(xh[6]), "+v"(xh[7]));
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                f32x4 o0, o1, o2;
                w85_transform<HAS_ISC>(xv, xh, pS * 2 + hf, xsc, o0, o1, o2);
                buf[xlds + 3 * hf] = o0;
                buf[xlds + 3 * hf + 1] = o1;
                buf[xlds + 3 * hf + 2] = o2;
            }
            return;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xh[0]), "+v"(xh[1]), "+v"(xh[2]), "+v"(xh[3]), "+v"(xh[4]), "+v"(xh[5]), "+v"(xh[6]), "+v"(xh[7]));
        f32x4 o0, o1, o2;
        w85_transform<HAS_ISC>(xv, xh, pS * 2 + half, xsc, o0, o1, o2);      // (the selector is wave-uniform)
        buf[xlds] = o0;
        buf[xlds + 1] = o1;
        buf[xlds + 2] = o2;
    };
    auto advance = [&](int& ps, int& ci0) __attribute__((always_inline)) {    // next super-slab, clamped at the last one
        int nc = ci0 + KS, np = ps;
        if (nc >= g.CinP) {
            nc = 0;
            ++np;
        }
        if (np <= 1) {
            ps = np;
            ci0 = nc;
        }
    };

    // B operand address (float4 units) in X ([ci][unit][6 float4]); the A operand comes straight from the weight ring in registers
    const int boff = (lk * NU + l15) * 6;

    // ---- prologue: super-slab 0 transformed into X[0], rows of super-slab 1 in flight, ring entries 0-2 loaded
    int pA = 0, cA = 0;
    issue_rows(pA, cA, 0, 8);
    issue_halo(cA);
    issue_isc(cA);
    pS = pA;
    f32x4 aw[RD];
#pragma unroll
    for (int i = 0; i < RD; ++i) {
        aw[i] = (W85_ABL & 4) ? f32x4{1.f, 2.f, 3.f, 4.f} : __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsw, wvo, sW + (i % 3) * 1024, 0));
        if (i % 3 == 2) w_next();
    }
    Y_FENCE
    halo_permute();
    store_act(Xb);
    Y_FENCE
    advance(pA, cA);
    issue_rows(pA, cA, 0, 8);
    issue_halo(cA);
    issue_isc(cA);
    pS = pA;
    Y_FENCE
    asm volatile("s_waitcnt lgkmcnt(0)");
    __builtin_amdgcn_s_barrier();
    Y_FENCE

    f32x4 acc[2][12];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int p = 0; p < 12; ++p) acc[i][p] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 bv[2];
    bv[0] = bv[1] = f32x4{1.f, 2.f, 3.f, 4.f};
    // group G = 3 hs + pg of a super-slab (hs = half-slot 0..7 = (ci quad hs >> 1, phase half hs & 1), pg = 4 of its 12 phases):
    //   [row / halo loads] [X read of group G + 1 into the other operand set] [transform at G = 0] [4 MFMAs of group G]
    //   [weight load of group G + RD into the ring entry the MFMAs just read]
#if W85_ABL & 8
#define Y_READ(c, Xp, hs, pg) asm volatile("" : "+v"(bv[c]));
#else
#define Y_READ(c, Xp, hs, pg) bv[c] = (Xp)[boff + ((hs) >> 1) * KQ * NU * 6 + ((hs) & 1) * 3 + (pg)];
#endif
#ifdef W85_PRIO
#define Y_PRIO(n) __builtin_amdgcn_s_setprio(n);
#else
#define Y_PRIO(n)
#endif
#define Y_MFMA(c, GN)                                                  \
    if (!(W85_ABL & 16))                                               \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                      \
        acc[((GN) / 3) & 1][4 * ((GN) % 3) + i] =                      \
            __builtin_amdgcn_mfma_f32_16x16x4f32(aw[(GN) % RD][i], bv[c][i], acc[((GN) / 3) & 1][4 * ((GN) % 3) + i], 0, 0, 0);
#define Y_WLOAD(GN)                                                                                                          \
    if (!(W85_ABL & 4)) aw[(GN) % RD] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsw, wvo, sW + ((GN) % 3) * 1024, 0)); \
    if constexpr ((GN) % 3 == 2) w_next();
#define Y_G(GN)                                                                                                 \
    if constexpr ((GN) == 1) {                                                                                  \
        issue_rows(pA, cA, 4, 8);                                                                               \
        Y_FENCE                                                                                                 \
    }                                                                                                           \
    if constexpr ((GN) == 2) {                                                                                  \
        issue_halo(cA);                                                                                         \
        Y_FENCE                                                                                                 \
    }                                                                                                           \
    if constexpr ((GN) < 23) {                                                                                  \
        Y_READ(((GN) + 1) & 1, Xs, ((GN) + 1) / 3, ((GN) + 1) % 3)                                              \
        Y_FENCE                                                                                                 \
    }                                                                                                           \
    if constexpr ((GN) == 0) {                                                                                  \
        store_act(Xw);                                                                                          \
        advance(pA, cA);                                                                                        \
        issue_isc(cA);                                                                                          \
        pS = pA;                                                                                                \
        Y_FENCE                                                                                                 \
        issue_rows(pA, cA, 0, 4);                                                                               \
        Y_FENCE                                                                                                 \
    }                                                                                                           \
    if constexpr ((GN) < 23) {                                                                                  \
        Y_PRIO(1)                                                                                               \
        Y_MFMA((GN) & 1, GN)                                                                                    \
        Y_PRIO(0)                                                                                               \
        Y_FENCE                                                                                                 \
        Y_WLOAD(GN)                                                                                             \
        Y_FENCE                                                                                                 \
    }

    Y_READ(0, Xb, 0, 0)
    halo_permute();                                       // (super-slab 1's halo: waits for its load)
    int cM = 0, pM = 0;
    for (int S = 0; S < NS; ++S) {
        const f32x4* Xs = Xb + (S & 1) * XSZ;
        f32x4* Xw = Xb + ((S + 1) & 1) * XSZ;
        Y_G(0) Y_G(1) Y_G(2) Y_G(3) Y_G(4) Y_G(5) Y_G(6) Y_G(7) Y_G(8) Y_G(9) Y_G(10) Y_G(11)
        Y_G(12) Y_G(13) Y_G(14) Y_G(15) Y_G(16) Y_G(17) Y_G(18) Y_G(19) Y_G(20) Y_G(21) Y_G(22)
        // G23: barrier (X[(S + 1) & 1] complete, nobody reads X[S & 1] any more: the operands of this group are in registers), first
        // read of the next super-slab, 4 MFMAs
        Y_G(23)
        asm volatile("s_waitcnt lgkmcnt(0)");
        __builtin_amdgcn_s_barrier();
        Y_FENCE
        Y_READ(0, Xw, 0, 0)
        halo_permute();
        Y_FENCE
        Y_MFMA(1, 23)
        Y_FENCE
        Y_WLOAD(23)
        Y_FENCE
        cM += KS;
        if (cM >= g.CinP) {
            cM = 0;
            if (pM == 0 && !(W85_ABL & 1024)) {
                w85_carry(acc);
            }
            ++pM;
        }
    }
#undef Y_G
#undef Y_MFMA
#undef Y_READ
#undef Y_WLOAD
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)");

    if (W85_ABL & 512) {                                 // (no output transform: one store keeps the accumulators alive)
        f32x4 sacc = acc[0][0];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int p = 0; p < 12; ++p) sacc += acc[i][p];
        if (sacc[0] == 12345.f) *reinterpret_cast<f32x4*>(a.out) = sacc;
        return;
    }
    w85_epilogue(a, acc, b, co0 + wave * 16, fa, t0, lk, l15, g.xcd ? bx : Q * g.tiles_t + tile_t, g.tiles_t * nQ);
#endif
}

// ---- 96- and 64-channel tiles: SPECIALISED waves.  With NW = 6 (4) channel tiles the eight waves do not split evenly over the four
// SIMDs as multipliers (wave w and w + 4 share SIMD w & 3): instead waves 0 .. NW-1 ONLY multiply (weights ring, X reads, MFMAs,
// epilogue) and waves NW .. 7 ONLY load and transform - all 16 ci x 16 units x 4 phases of a super-slab, 4 / (8 - NW) channel
// quads per wave, both phase pairs per thread from ONE set of row loads.  NW = 6: SIMDs 0 and 1 host two multiplying waves, SIMDs 2
// and 3 one multiplying wave plus a transform wave carrying four wave-shares of transform: about the same time.  NW = 4: every SIMD
// hosts one of each.  The multiplying waves' memory queue holds weights only, the transform waves' rows only (two register sets: the
// rows of super-slab S + 2 are in flight while S + 1 is transformed).  One barrier per super-slab, as in the 128-channel kernel.
// NWV = 12 (round 6, BABE_W85_12W=1): the 128-channel tile as 8 multiplying + 4 transform waves, three waves per SIMD at <= 168
// registers - the multiplying waves never carry the transform in their own instruction stream.
template <bool HAS_ISC, int NW, int NWV = 8>
__global__ __launch_bounds__(64 * NWV, 1) void conv_wino85s_kernel(babe_conv_args a, Wino85Geom g, const float* __restrict__ wq) {
#if __HIP_DEVICE_COMPILE__
    constexpr int KS = 16, KQ = 4, NU = 16, BN = 16 * NW, RD = W85_RD;
    constexpr int TW = NWV - NW, PPL = 4 / TW;          // transform waves; channel quads per transform wave
    static_assert((NWV == 8 && (NW == 6 || NW == 4)) || (NWV == 12 && NW == 8), "tile width");
    constexpr int XSZ = KS * NU * 6;
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    f32x4* const Xb = reinterpret_cast<f32x4*>(smem_f);  // X[2]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    // The hardware issues oldest wave first: with the transform waves as the workgroup's LAST waves their instructions got what the
    // two multiplying waves of their SIMD left over, most of a super-slab's transform ran alone after the MFMAs, and the multiplying
    // waves stood at the barrier for 53 % (first wave of a SIMD) / 20 % (second) of the main loop (s_memtime probe, ablation bit 16384,
    // profiles/r06_f45_barrier_probe.txt).  The FIRST TW hardware waves transform; `wave` stays the logical index the rest of the
    // kernel uses: 0 .. NW-1 multiply, NW .. NWV-1 transform.
    const int wave_hw = __builtin_amdgcn_readfirstlane(tid >> 6);
    // (W85_TFIRST == 2, experiment, 12-wave form only: the transform wave between the two multiplying waves of its SIMD in age)
    const int wave = (W85_TFIRST == 2 && NWV == 12) ? (wave_hw < 4 ? wave_hw : (wave_hw < 8 ? NW + wave_hw - 4 : wave_hw - 4))
                     : W85_TFIRST                   ? (wave_hw < NWV - NW ? NW + wave_hw : wave_hw - (NWV - NW))
                                                    : wave_hw;
    const int l15 = lane & 15, lk = lane >> 4;
    const int b = blockIdx.z;
    int bx = blockIdx.x, by = blockIdx.y;
    if (g.xcd) {                                        // XCD-contiguous tile order (see conv_wino85_kernel)
        const int lin = (bx & 7) * g.per_xcd + (bx >> 3);
        if ((bx >> 3) >= g.per_xcd || lin >= g.total) return;
        by = lin % g.ncb;
        bx = lin / g.ncb;
    }
    const int co0 = by * BN;
    const int nQ = a.dil * g.nquads;
    const int tile_t = g.xcd ? bx / nQ : bx % g.tiles_t;
    const int Q = g.xcd ? bx % nQ : bx / g.tiles_t;
    const int t0 = tile_t * 64;
    const int cls = Q / g.nquads;
    const int fa = Q < a.dil * g.nquads ? cls + 4 * (Q - cls * g.nquads) * a.dil : a.F + 8 * a.dil;
    const int NS = 2 * (g.CinP / KS);                   // super-slabs (even)
#if W85_FLAGS
    // flags behind X[W85_NXB]: prog[0..3] (transform waves; unused words stay at "infinity"), cons[0..7] (multiplying waves)
    unsigned* const flg = reinterpret_cast<unsigned*>(Xb + W85_NXB * XSZ);
    const unsigned flg_a = (unsigned)(unsigned long long)flg;     // LDS byte address of the flags (low word of the flat address)
    if (tid < 12) flg[tid] = (tid < 4 ? tid < TW : tid - 4 < NW) ? 0u : 0x7fffffffu;
    __syncthreads();
    // min over four flag words at byte address `ad`
    auto flag_min4 = [&](unsigned ad) __attribute__((always_inline)) {
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        u32x4 v;
        asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(ad) : "memory");
        const unsigned m01 = v[0] < v[1] ? v[0] : v[1], m23 = v[2] < v[3] ? v[2] : v[3];
        return __builtin_amdgcn_readfirstlane(m01 < m23 ? m01 : m23);
    };
    auto flag_set = [&](unsigned ad, unsigned val) __attribute__((always_inline)) {
        asm volatile("s_waitcnt lgkmcnt(0)\n\tds_write_b32 %0, %1" ::"v"(ad), "v"(val) : "memory");
    };
#endif

    if (wave >= NW) {
        // ================= transform waves =================
#if W85_TPRIO            // (experiment: the transform waves above the multiplying waves of their SIMD)
        __builtin_amdgcn_s_setprio(W85_TPRIO);
#endif
        unsigned long long pb_wait = 0;
        const unsigned long long pb_t0 = W85_PROBE ? __builtin_amdgcn_s_memtime() : 0ull;
        const float* p1 = a.in + (long)b * a.in_bs;
        const int cs1 = (int)a.in_cs;
        const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)p1, 0, a.Cin * cs1 * 4, 0x00020000);
        const int vw0 = (wave - NW) * PPL;              // first channel quad of this wave
        const int s_tu = lane & 15, s_ch = lane >> 4;
        const int s_t = t0 + 4 * s_tu;
        const unsigned chb = (unsigned)(s_ch * cs1 * 4);
        unsigned er[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int fr = fa + (r - 2) * a.dil;
            const bool ok = fr >= 0 && fr < a.F && s_t < a.T;
            er[r] = ok ? (unsigned)((fr * a.T + s_t) * 4) + chb : OOBH;
        }
        unsigned ehalo = OOBH;
        {
            const int hr = lane >> 3, hc = (lane >> 1) & 3, hs = lane & 1;
            const int fr = fa + (hr - 2) * a.dil;
            const int th = t0 + (hs ? 64 : -1);
            if (fr >= 0 && fr < a.F && th >= 0 && th < a.T) ehalo = (unsigned)((fr * a.T + th) * 4 + hc * cs1 * 4);
        }
        const int hsrc = (2 * s_ch + (s_tu == 15 ? 1 : 0)) * 4;
        f32x4 xv[2][PPL][8];
        float xhl[2][PPL], xsc[2][PPL];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int p = 0; p < PPL; ++p) {
                xv[i][p][0] = xv[i][p][7] = f32x4{0.f, 0.f, 0.f, 0.f};
                xsc[i][p] = 1.f;
            }
        // the rows of super-slab (ps, ci0) into register set `st` (rows 0 and 7 in pass B only), its halo and input scale
        auto issue = [&](int st, int ps, int ci0) __attribute__((always_inline)) {
            if (W85_ABL & 2) return;
#pragma unroll
            for (int p = 0; p < PPL; ++p) {
                const int so = (ci0 + 4 * (vw0 + p)) * cs1 * 4;
#pragma unroll
                for (int r = 1; r < 7; ++r) xv[st][p][r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs1, er[r], so, 0));
                if (ps == 1) {
                    xv[st][p][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs1, er[0], so, 0));
                    xv[st][p][7] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs1, er[7], so, 0));
                }
                xhl[st][p] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs1, ehalo, so, 0));
                if (HAS_ISC) xsc[st][p] = a.in_scale[(long)b * a.Cin + ci0 + 4 * (vw0 + p) + s_ch];
            }
        };
        // register set `st` (rows of a pass-`ps` super-slab) -> X buffer `buf`: both phase pairs of the pass for this thread's
        // (ci, unit) pairs; the arithmetic of conv_wino85_kernel's store_act, term for term
        unsigned long long pb_vm = 0;                      // (probe: time waiting for the rows of the set about to be transformed)
        auto transform = [&](int st, int ps, f32x4* buf) __attribute__((always_inline)) {
#if W85_PROBE
            {
                const unsigned long long tv_ = __builtin_amdgcn_s_memtime();
                asm volatile("s_waitcnt vmcnt(9)" ::: "memory");     // (at most 9 loads per channel quad are younger than this set's)
                pb_vm += __builtin_amdgcn_s_memtime() - tv_;
            }
#endif
#pragma unroll
            for (int p = 0; p < PPL; ++p) {
                float xh[8];
#pragma unroll
                for (int r = 0; r < 8; ++r)
                    asm volatile("ds_bpermute_b32 %0, %1, %2 offset:%3" : "=v"(xh[r]) : "v"(hsrc), "v"(xhl[st][p]), "n"(32 * r));
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xh[0]), "+v"(xh[1]), "+v"(xh[2]), "+v"(xh[3]), "+v"(xh[4]), "+v"(xh[5]), "+v"(xh[6]), "+v"(xh[7]));
                const int xl = (((vw0 + p) * 4 + s_ch) * NU + s_tu) * 6;
                if (W85_ABL & 1) {
                    buf[xl] = xv[st][p][1] + xv[st][p][0];
                    buf[xl + 3] = xv[st][p][2] + xv[st][p][7] + f32x4{xh[1], xh[2], xh[0], xh[7]};
                    continue;
                }
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    f32x4 o0, o1, o2;
                    w85_transform<HAS_ISC>(xv[st][p], xh, ps * 2 + hf, xsc[st][p], o0, o1, o2);
                    buf[xl + hf * 3] = o0;
                    buf[xl + hf * 3 + 1] = o1;
                    buf[xl + hf * 3 + 2] = o2;
                }
            }
        };
        auto advance = [&](int& ps, int& ci0) __attribute__((always_inline)) {    // next super-slab, clamped at the last one
            int nc = ci0 + KS, np = ps;
            if (nc >= g.CinP) {
                nc = 0;
                ++np;
            }
            if (np <= 1) {
                ps = np;
                ci0 = nc;
            }
        };
        // super-slab s lives in register set s & 1; (pT, cT) = the slab transformed next, (pL, cL) = the slab loaded next
        int pT = 0, cT = 0, pL = 0, cL = 0;
#if W85_FLAGS
        issue(0, pL, cL);
        advance(pL, cL);
        issue(1, pL, cL);
        advance(pL, cL);
        int kb = 0;                                        // k mod NXB: the X buffer of slab k
        const unsigned my_prog = flg_a + (unsigned)(wave - NW) * 4;
        auto slab = [&](int st, int k) __attribute__((always_inline)) {
            // buffer k mod NXB last held slab k - NXB: every multiplying wave must have consumed it (cons >= k - NXB + 1)
            if (k >= W85_NXB) {
                while (true) {
                    const unsigned c0 = flag_min4(flg_a + 16), c1 = flag_min4(flg_a + 32);
                    if ((int)(c0 < c1 ? c0 : c1) >= k - W85_NXB + 1) break;
                    if (W85_SLEEP) __builtin_amdgcn_s_sleep(W85_SLEEP);
                }
            }
            transform(st, pT, Xb + kb * XSZ);
            advance(pT, cT);
            flag_set(my_prog, (unsigned)(k + 1));          // (behind this wave's LDS writes: s_waitcnt lgkmcnt(0) first)
            issue(st, pL, cL);                             // the set just transformed is free: rows of slab k + 2 (clamped at the end)
            advance(pL, cL);
            kb = kb == W85_NXB - 1 ? 0 : kb + 1;
        };
        for (int S = 0; S < NS; S += 2) {
            slab(0, S);
            slab(1, S + 1);
        }
        return;
#endif
        issue(0, pL, cL);
        advance(pL, cL);
        issue(1, pL, cL);
        advance(pL, cL);
        transform(0, pT, Xb);                              // super-slab 0 -> X[0]
        advance(pT, cT);
        asm volatile("s_waitcnt lgkmcnt(0)");
        W85_BARRIER(pb_wait)
        for (int S = 0; S < NS; S += 2) {
            // iteration S (even): set 0 is free (slab S was transformed last time): rows of slab S + 2; transform slab S + 1 (set 1)
            issue(0, pL, cL);
            advance(pL, cL);
            transform(1, pT, Xb + XSZ);
            advance(pT, cT);
            asm volatile("s_waitcnt lgkmcnt(0)");
            W85_BARRIER(pb_wait)
            // iteration S + 1: rows of slab S + 3 into set 1; transform slab S + 2 (set 0) -> X[0]
            issue(1, pL, cL);
            advance(pL, cL);
            transform(0, pT, Xb);
            advance(pT, cT);
            asm volatile("s_waitcnt lgkmcnt(0)");
            W85_BARRIER(pb_wait)
        }
        if (W85_PROBE && a.stat_mode == 99 && wave == NW && lane == 0) {
            double* dbg = a.stat_part + 4 * ((long)b * g.total + (g.xcd ? blockIdx.x : blockIdx.x + gridDim.x * blockIdx.y));
            dbg[2] = (double)pb_wait + 1e-9 * (double)pb_vm;     // (the row wait rides in the fraction: 1e-9 x ticks)
            dbg[3] = (double)(__builtin_amdgcn_s_memtime() - pb_t0);
        }
        return;
    }

    // ================= multiplying waves =================
    // (a static s_setprio 1 / 3 for these waves over the transform wave of their SIMD: no effect, profiles/r06_12wave_ab.txt)
    unsigned long long pb_wait = 0;
    const unsigned long long pb_t0 = W85_PROBE ? __builtin_amdgcn_s_memtime() : 0ull;
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)wq, 0, 4 * g.CinP * g.CoutP * 48, 0x00020000);
    const int NT = g.CoutP >> 4;
    const int wstep = NT * 3072;
    const int sWend = 4 * g.CinP * g.CoutP * 48;
    int sW = ((co0 >> 4) + wave) * 3072;
    const unsigned wvo = (unsigned)(lane * 16);
    auto w_next = [&]() __attribute__((always_inline)) {
        if (W85_ABL & 256) return;                      // (timing ablation: always the first half-slot - L1 hits)
        sW += wstep;
        sW = sW < sWend ? sW : sWend;
    };
    const int boff = (lk * NU + l15) * 6;
    f32x4 aw[RD];
#pragma unroll
    for (int i = 0; i < RD; ++i) {
        aw[i] = (W85_ABL & 4) ? f32x4{1.f, 2.f, 3.f, 4.f} : __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsw, wvo, sW + (i % 3) * 1024, 0));
        if (i % 3 == 2) w_next();
    }
    f32x4 acc[2][12];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int p = 0; p < 12; ++p) acc[i][p] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 bv[2];
    bv[0] = bv[1] = f32x4{1.f, 2.f, 3.f, 4.f};
#if W85_FLAGS
    const unsigned my_cons = flg_a + 16 + (unsigned)wave * 4;
    auto wait_prog = [&](int need) __attribute__((always_inline)) {
        while ((int)flag_min4(flg_a) < need)
            if (W85_SLEEP) __builtin_amdgcn_s_sleep(W85_SLEEP);
    };
    wait_prog(1);                                          // X[0] is complete
    int kb = 0;
#else
    W85_BARRIER(pb_wait)                          // X[0] is complete
#endif
    Y_FENCE
#if W85_ABL & 8
#define Z_READ(c, Xp, hs, pg) asm volatile("" : "+v"(bv[c]));
#else
#define Z_READ(c, Xp, hs, pg) bv[c] = (Xp)[boff + ((hs) >> 1) * KQ * NU * 6 + ((hs) & 1) * 3 + (pg)];
#endif
#define Z_MFMA(c, GN)                                                  \
    if (!(W85_ABL & 16))                                               \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                      \
        acc[((GN) / 3) & 1][4 * ((GN) % 3) + i] =                      \
            __builtin_amdgcn_mfma_f32_16x16x4f32(aw[(GN) % RD][i], bv[c][i], acc[((GN) / 3) & 1][4 * ((GN) % 3) + i], 0, 0, 0);
#define Z_WLOAD(GN)                                                                                                          \
    if (!(W85_ABL & 4)) aw[(GN) % RD] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsw, wvo, sW + ((GN) % 3) * 1024, 0)); \
    if constexpr ((GN) % 3 == 2) w_next();
#if W85_ABL & 8192        // (timing probe: waves 4, 5 of the 96-channel tile do half their MFMAs - the balance a K-split would give)
#define Z_SKIP(GN) (NW == 6 && wave >= 4 && (GN) >= 12)
#else
#define Z_SKIP(GN) false
#endif
#define Z_G(GN)                                                        \
    Z_READ(((GN) + 1) & 1, Xs, ((GN) + 1) / 3, ((GN) + 1) % 3)         \
    Y_FENCE                                                            \
    if (!Z_SKIP(GN)) Z_MFMA((GN) & 1, GN)                              \
    Y_FENCE                                                            \
    Z_WLOAD(GN)                                                        \
    Y_FENCE
    Z_READ(0, Xb, 0, 0)
    int cM = 0, pM = 0;
    for (int S = 0; S < NS; ++S) {
#if W85_FLAGS
        const int kbn = kb == W85_NXB - 1 ? 0 : kb + 1;
        const f32x4* Xs = Xb + kb * XSZ;
        const f32x4* Xw = Xb + kbn * XSZ;
        kb = kbn;
#else
        const f32x4* Xs = Xb + (S & 1) * XSZ;
        const f32x4* Xw = Xb + ((S + 1) & 1) * XSZ;
#endif
        Z_G(0) Z_G(1) Z_G(2) Z_G(3) Z_G(4) Z_G(5) Z_G(6) Z_G(7) Z_G(8) Z_G(9) Z_G(10) Z_G(11)
        Z_G(12) Z_G(13) Z_G(14) Z_G(15) Z_G(16) Z_G(17) Z_G(18) Z_G(19) Z_G(20) Z_G(21) Z_G(22)
        // G23: its operands are in registers; barrier (X[(S + 1) & 1] complete, nobody reads X[S & 1] any more)
#if W85_FLAGS
        flag_set(my_cons, (unsigned)(S + 1));              // (behind this wave's reads of slab S: s_waitcnt lgkmcnt(0) first)
        if (S + 1 < NS) wait_prog(S + 2);                  // slab S + 1 is complete
#else
        asm volatile("s_waitcnt lgkmcnt(0)");
        W85_BARRIER(pb_wait)
#endif
        Y_FENCE
        Z_READ(0, Xw, 0, 0)
        Y_FENCE
        Z_MFMA(1, 23)
        Y_FENCE
        Z_WLOAD(23)
        Y_FENCE
        cM += KS;
        if (cM >= g.CinP) {
            cM = 0;
            if (pM == 0) w85_carry(acc);
            ++pM;
        }
    }
#undef Z_G
#undef Z_MFMA
#undef Z_READ
#undef Z_WLOAD
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)");
    if (W85_PROBE && a.stat_mode == 99 && lane == 0) {
        double* dbg = a.stat_part + 4 * ((long)b * g.total + (g.xcd ? blockIdx.x : blockIdx.x + gridDim.x * blockIdx.y));
        if (wave == 0) {
            dbg[0] = (double)pb_wait;
            dbg[1] = (double)(__builtin_amdgcn_s_memtime() - pb_t0);
        }
        // every multiplying wave's barrier share, behind the per-tile records (tools/f45_barrier_probe.py: offset 4 * (tiles + 64))
        a.stat_part[4 * ((long)g.total + 64) + 8 * (g.xcd ? blockIdx.x : blockIdx.x + gridDim.x * blockIdx.y) + wave] =
            (double)pb_wait / (double)(__builtin_amdgcn_s_memtime() - pb_t0);
    }

    w85_epilogue(a, acc, b, co0 + wave * 16, fa, t0, lk, l15, g.xcd ? bx : Q * g.tiles_t + tile_t, g.tiles_t * nQ);
#endif
}

// dst [2 passes][CinP / 4][2 halves][CoutP / 16][3 phase groups][4 ci][16 co][4]: one group's A operand of one wave = 1 KB, lane
// (ci, co) its float4; entry 6 * fl + tp of the 12 phases of a half; phase pairs (A,0) = (p1,p2), (A,1) = (p3,p4), (B,0) = (p5,p6),
// (B,1) = (p0,p7)
__global__ void pack_wino85_kernel(const float* __restrict__ w, float* __restrict__ dst, int Cout, int Cin, int tf, int CinP,
                                   int CoutP, long total) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int co = (int)(i % CoutP);
    long r = i / CoutP;
    const int c4 = (int)(r % 4);
    r /= 4;
    const int hf = (int)(r % 2);
    r /= 2;
    const int cq = (int)(r % (CinP / 4));
    const int ps = (int)(r / (CinP / 4));
    const int ci = cq * 4 + c4;
    double wk[5][3];
    for (int kh = 0; kh < 5; ++kh)
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
    const double G8[8][5] = {{-1, 0, 0, 0, 0},
                             {-2.0 / 9, -2.0 / 9, -2.0 / 9, -2.0 / 9, -2.0 / 9},
                             {-2.0 / 9, 2.0 / 9, -2.0 / 9, 2.0 / 9, -2.0 / 9},
                             {1.0 / 90, 1.0 / 45, 2.0 / 45, 4.0 / 45, 8.0 / 45},
                             {1.0 / 90, -1.0 / 45, 2.0 / 45, -4.0 / 45, 8.0 / 45},
                             {32.0 / 45, 16.0 / 45, 8.0 / 45, 4.0 / 45, 2.0 / 45},
                             {32.0 / 45, -16.0 / 45, 8.0 / 45, -4.0 / 45, 2.0 / 45},
                             {0, 0, 0, 0, 1}};
    const double G3[6][3] = {{0.25, 0, 0},          {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                             {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6},  {0, 0, 1}};
    const int fps[2][2][2] = {{{1, 2}, {3, 4}}, {{5, 6}, {0, 7}}};
    float d[12];
    for (int fl = 0; fl < 2; ++fl) {
        const int fp = fps[ps][hf][fl];
        double fw[3];
        for (int kw = 0; kw < 3; ++kw) {
            double s = 0;
            for (int kh = 0; kh < 5; ++kh) s += G8[fp][kh] * wk[kh][kw];
            fw[kw] = s;
        }
        for (int tp = 0; tp < 6; ++tp) d[6 * fl + tp] = (float)(G3[tp][0] * fw[0] + G3[tp][1] * fw[1] + G3[tp][2] * fw[2]);
    }
    const long hslot = ((long)ps * (CinP / 4) + cq) * 2 + hf;
    for (int pg = 0; pg < 3; ++pg) {
        float* o = dst + ((((hslot * (CoutP / 16) + (co >> 4)) * 3 + pg) * 4 + c4) * 16 + (co & 15)) * 4;
        for (int e = 0; e < 4; ++e) o[e] = d[4 * pg + e];
    }
}

}  // namespace

extern "C" long babe_conv_packed_size_wino85(int Cout, int Cin, int transpose_flip) {
    const int co = transpose_flip ? Cin : Cout;
    const int ci = transpose_flip ? Cout : Cin;
    return 48L * ((ci + 15) / 16 * 16) * ((co + 15) / 16 * 16);
}

extern "C" int babe_conv_pack_weights_wino85(const float* w, float* dst, int Cout, int Cin, int KH, int KW, int transpose_flip,
                                             void* stream) {
    BABE_CHECK_ARG(w && dst && Cout > 0 && Cin > 0 && KH == 5 && KW == 3, "conv_pack_weights_wino85: needs a (5,3) kernel");
    const int co = transpose_flip ? Cin : Cout;
    const int ci = transpose_flip ? Cout : Cin;
    const int CinP = (ci + 15) / 16 * 16, CoutP = (co + 15) / 16 * 16;
    const long total = 4L * CinP * CoutP;
    hipLaunchKernelGGL(pack_wino85_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, w, dst, Cout, Cin,
                       transpose_flip, CinP, CoutP, total);
    BABE_LAUNCH_CHECK();
    return BABE_OK;
}

/* 1 if the F(4,5) x F(4,3) kernels can run this problem (the contract include/babe_hip.h states): Cout a multiple of 128
 * (conv_wino85_kernel), 96 or 64 (conv_wino85s_kernel<., 6> / <., 4>: specialised multiplying / transform waves), Cin a multiple
 * of 16, T a multiple of 4 and at least 64, one source, 16-byte aligned views. */
extern "C" int babe_conv2d_wino85_supported(const babe_conv_args* ap) {
    if (!ap) return 0;
    const babe_conv_args& a = *ap;
    auto al16 = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
    if (a.KH != 5 || a.KW != 3 || a.T % 4 != 0 || a.T < 64 || a.dil < 1) return 0;
    if (a.Cin < 16 || a.Cin % 16 != 0 || (a.Cout % 128 != 0 && a.Cout % 96 != 0 && a.Cout % 64 != 0)) return 0;
    if (!al16(a.in) || a.in_bs % 4 || a.in_cs % 4 || a.in2) return 0;
    if (!al16(a.out) || a.out_bs % 4 || a.out_cs % 4) return 0;
    if (a.res && (!al16(a.res) || a.res_bs % 4 || a.res_cs % 4)) return 0;
    const long lim = 0x3fffffffL / 4;
    if ((long)a.Cin * a.in_cs >= lim || (long)a.F * a.T >= lim || (long)a.Cout * a.out_cs >= lim ||
        (a.res && (long)a.Cout * a.res_cs >= lim))
        return 0;
    if (48L * a.Cin * a.Cout * 4 >= 0x7fffffffL) return 0;
    return 1;
}

extern "C" int babe_conv2d_wino85_stat_slots(const babe_conv_args* ap) {
    if (!ap || ap->stat_cg < 4 || ap->stat_cg % 4 != 0 || ap->dil < 1 || ap->F < 1 || ap->T < 1) return 0;
    const babe_conv_args& a = *ap;
    const int sg = a.stat_cg % 16 == 0 ? 16 : (a.stat_cg % 8 == 0 ? 8 : 4);
    return cdiv(a.T, 64) * a.dil * cdiv(cdiv(a.F, a.dil), 4) * (a.stat_cg / sg);
}

/* fraction of the row-quad x time slots of a launch that hold real outputs */
static double wino85_fill(const babe_conv_args& a) {
    const long n = (a.F + a.dil - 1) / a.dil, nq = (n + 3) / 4;
    return ((double)a.F / (4.0 * nq * a.dil)) * ((double)a.T / (64.0 * ((a.T + 63) / 64)));
}

extern "C" int babe_conv2d_wino85_preferred(const babe_conv_args* ap) {
    if (!babe_conv2d_wino85_supported(ap)) return 0;
    static const double min_fill = [] { const char* e = getenv("BABE_W85_FILL"); return e ? atof(e) : 0.80; }();
    return wino85_fill(*ap) >= min_fill ? 1 : 0;
}

/* Form of the 128-channel tile: 12 = 8 multiplying + 4 transform waves (conv_wino85s_kernel<., 8, 12>, round 6, the default:
 * +3 % on the whole job, profiles/r06_12wave_ab.txt), 8 = conv_wino85_kernel (waves 0-3 transform and multiply).  Same sums in
 * the same order: bit-identical outputs.  BABE_W85_12W=0 selects 8 for the process; babe_conv2d_wino85_set_waves at run time. */
static std::atomic<int>& w85_waves128() {
    static std::atomic<int> v{[] { const char* e = getenv("BABE_W85_12W"); return (e && atoi(e) == 0) ? 8 : 12; }()};
    return v;
}
extern "C" int babe_conv2d_wino85_set_waves(int waves) {
    BABE_CHECK_ARG(waves == 8 || waves == 12, "conv2d_wino85_set_waves: 8 or 12 (got %d)", waves);
    w85_waves128().store(waves, std::memory_order_relaxed);
    return BABE_OK;
}

extern "C" int babe_conv2d_wino85(const babe_conv_args* ap, const float* w_wino85, void* stream) {
    BABE_CHECK_ARG(ap && w_wino85, "conv2d_wino85: null args");
    BABE_CHECK_ARG(babe_conv2d_wino85_supported(ap), "conv2d_wino85: unsupported problem");
    const babe_conv_args& a = *ap;
    if (a.stat_mode) {
        BABE_CHECK_ARG((a.stat_mode == 1 || a.stat_mode == 2 || (W85_PROBE && a.stat_mode == 99)) && a.stat_part && a.stat_cg >= 4 && a.stat_cg % 4 == 0 && a.Cout % a.stat_cg == 0,
                       "conv2d_wino85: fused reduction: mode %d, group size %d (Cout %d)", a.stat_mode, a.stat_cg, a.Cout);
        BABE_CHECK_ARG(a.stat_mode != 2 || (a.stat_x && a.stat_scale && ((uintptr_t)a.stat_x & 15) == 0 && a.res == nullptr),
                       "conv2d_wino85: fused reduction, mode 2: stat_x (16-byte aligned) and stat_scale are needed, res is not taken");
    }
    Wino85Geom g;
    g.CinP = a.Cin;
    g.CoutP = a.Cout;
    g.tiles_t = cdiv(a.T, 64);
    g.nquads = cdiv(cdiv(a.F, a.dil), 4);
    const double flops = babe_conv_flops(a);         // 48 multiplies per 16 outputs instead of 240: 0.2 of the direct count
    BabeProfScope prof(BABE_SLOT_CONV53_WINO85, babe_conv_bytes(a), flops, flops * 0.2, stream);
#if W85_FLAGS
    const size_t lds = (size_t)(W85_NXB * 16 * 16 * 6) * 16 + 64;             // X[NXB] + the progress words: 72 KB at 3
    {
        static std::atomic<unsigned long long> attr_done{0};
        if (babe_lds_optin(attr_done, {reinterpret_cast<const void*>(&conv_wino85s_kernel<true, 8, 12>), reinterpret_cast<const void*>(&conv_wino85s_kernel<false, 8, 12>),
                                       reinterpret_cast<const void*>(&conv_wino85s_kernel<true, 6>), reinterpret_cast<const void*>(&conv_wino85s_kernel<false, 6>),
                                       reinterpret_cast<const void*>(&conv_wino85s_kernel<true, 4>), reinterpret_cast<const void*>(&conv_wino85s_kernel<false, 4>)},
                           128 * 1024) != hipSuccess) {
            babe_set_error("conv2d_wino85: cannot opt in to %zu bytes of LDS", lds);
            return BABE_ERR_HIP;
        }
    }
#else
    const size_t lds = (size_t)(2 * 16 * 16 * 6) * 16;                        // X[2]: 48 KB
#endif
    static const int xcd_order = [] { const char* e = getenv("BABE_W85_XCD"); return e ? atoi(e) : 1; }();
    const int bn = a.Cout % 128 == 0 ? 128 : (a.Cout % 96 == 0 ? 96 : 64);
    g.xcd = xcd_order;
    g.ncb = a.Cout / bn;
    g.total = g.tiles_t * a.dil * g.nquads * g.ncb;
    g.per_xcd = (g.total + 7) / 8;
    dim3 grid(g.tiles_t * a.dil * g.nquads, g.ncb, a.B);
    if (g.xcd) grid = dim3(8 * g.per_xcd, 1, a.B);
    const hipStream_t st = (hipStream_t)stream;
    const bool isc = a.in_scale != nullptr;
    if (bn == 128 && w85_waves128().load(std::memory_order_relaxed) == 12) {
        if (isc) hipLaunchKernelGGL((conv_wino85s_kernel<true, 8, 12>), grid, dim3(768), lds, st, a, g, w_wino85);
        else hipLaunchKernelGGL((conv_wino85s_kernel<false, 8, 12>), grid, dim3(768), lds, st, a, g, w_wino85);
    } else if (bn == 128) {
        if (isc) hipLaunchKernelGGL((conv_wino85_kernel<true>), grid, dim3(512), lds, st, a, g, w_wino85);
        else hipLaunchKernelGGL((conv_wino85_kernel<false>), grid, dim3(512), lds, st, a, g, w_wino85);
    } else if (bn == 96) {
        if (isc) hipLaunchKernelGGL((conv_wino85s_kernel<true, 6>), grid, dim3(512), lds, st, a, g, w_wino85);
        else hipLaunchKernelGGL((conv_wino85s_kernel<false, 6>), grid, dim3(512), lds, st, a, g, w_wino85);
    } else {
        if (isc) hipLaunchKernelGGL((conv_wino85s_kernel<true, 4>), grid, dim3(512), lds, st, a, g, w_wino85);
        else hipLaunchKernelGGL((conv_wino85s_kernel<false, 4>), grid, dim3(512), lds, st, a, g, w_wino85);
    }
    BABE_LAUNCH_CHECK();
    return BABE_OK;
}
