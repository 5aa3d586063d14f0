// Constant-Q (NSGT, "oct") analysis / synthesis kernels.  Definition: oracle/nsgt.py header and
// babe_amd/cqt_plan.py; call sites in the reference: networks/cqtdiff+.py:743,841,
// testing/blind_bwe_sampler.py:156 (cqt_nsgt_pytorch.CQT_nsgt.fwd/.bwd/.apply_hpf_DC).
// HBM/latency-bound: a workgroup owns 4096 complex points of one octave (4096/T bands) in LDS.
#include "common.h"
#include "../../include/babe_hip.h"
#include "prof.h"

namespace {

// ---- per-band FFTs: Stockham radix-16 passes with register butterflies --------------------------------------------------
// A workgroup of 256 threads owns 4096 complex points of ONE octave: 4096/T bands of T points each (64 bands when
// T <= 64), so every workgroup of the launch has the same amount of work whatever the octave - the round-1 kernel gave a
// 512-thread workgroup to every band, and 192 of the 448 bands used an eighth of it.  A band of T = R0 * TB points
// (R0 = min(16, T)) is transformed by TB threads in ceil(log16 T) passes: a thread holds the R0 inputs of its butterflies
// in registers (stride T/R), multiplies by the inter-pass twiddles (exp(-2 pi i q/4096): four table entries per butterfly
// from global memory, the other eleven as products), runs the radix-R DFT in registers (R = 16 as 4 x 4, 8 as 4 x 2) and
// scatters the results autosorted through LDS to the next pass, so input and output are both in natural order and nothing
// is bit-reversed.  One LDS buffer between passes: write - barrier - read - barrier.
// Element i of the workgroup's image lives at i + (i >> 4) (the first pass writes with stride 16: without the pad all lanes
// of a wave hit two banks).
#ifndef ABL
#define ABL 0     // timing ablations of band_fft_kernel<0> (tools/ab/abl_build.sh cqt <bits>; results are WRONG with any bit set):
#endif            // 1 wide (16-byte) loads of the same bytes, 2 wide stores, 4 no window loads, 8 no FFT (copy only),
                  // 16 half the LDS image (indices wrap: more workgroups per CU), 32 register cap for 6 waves per SIMD,
                  // 128 non-temporal spectrum loads (slower: 41 -> 50 us)
#if ABL & 16
#define CQ_AT(i) (((i) + ((i) >> 4)) & 2047)
#else
#define CQ_AT(i) ((i) + ((i) >> 4))
#endif
constexpr int CQ_PTS = 4096;                           // points per workgroup
constexpr int CQ_LDS = (ABL & 16) ? 2048 : CQ_PTS + (CQ_PTS >> 4);         // float2 elements

// Complex values are 2-vectors (register pairs): hipcc turns the arithmetic below into packed fp32 instructions
// (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32) plus some moves of the halves.  (Round 3 tried the same operations as inline
// asm with op_sel / neg modifiers instead of the moves: 921 -> 571 vector instructions per 4096-point band and NO change in
// the kernel time - the kernel is bound by its memory traffic, not by the vector ALU; dropped.)
typedef float f2 __attribute__((ext_vector_type(2)));
// a + (S i) b
template <int S>
__device__ __forceinline__ f2 cq_add_i(f2 a, f2 b) {
    return S > 0 ? f2{a.x - b.y, a.y + b.x} : f2{a.x + b.y, a.y - b.x};
}
// x * w (CONJ: x * conj(w))
template <bool CONJ>
__device__ __forceinline__ f2 cq_mul(f2 x, f2 w) {
    return CONJ ? f2{x.x * w.x + x.y * w.y, x.y * w.x - x.x * w.y} : f2{x.x * w.x - x.y * w.y, x.x * w.y + x.y * w.x};
}

// 4-point DFT, sign S (exp(S 2 pi i nk/4)), in place, natural order
template <int S>
__device__ __forceinline__ void cq_fft4(f2& x0, f2& x1, f2& x2, f2& x3) {
    const f2 a0 = x0 + x2, a1 = x0 - x2, a2 = x1 + x3, d = x1 - x3;
    x0 = a0 + a2;
    x2 = a0 - a2;
    x1 = cq_add_i<S>(a1, d);
    x3 = cq_add_i<-S>(a1, d);
}
// R-point DFT of v[0..R-1] in registers, R in {2, 4, 8, 16}.  R = 4 * R2: v[R2*k1 + k2] receives X[k1 + 4*k2]; the
// caller un-permutes with cq_perm<R>(r) when it stores.
template <int R, int S>
__device__ __forceinline__ void cq_fft(f2* v) {
    if constexpr (R == 2) {
        const f2 a = v[0], b = v[1];
        v[0] = a + b;
        v[1] = a - b;
    } else if constexpr (R == 4) {
        cq_fft4<S>(v[0], v[1], v[2], v[3]);
    } else {
        constexpr int R2 = R / 4;
#pragma unroll
        for (int n2 = 0; n2 < R2; ++n2) cq_fft4<S>(v[n2], v[R2 + n2], v[2 * R2 + n2], v[3 * R2 + n2]);
        // twiddles W_R^(n2*k1), W_R = exp(S 2 pi i / R): the table holds exp(+2 pi i m/16), S < 0 multiplies by the conjugate
        constexpr float C16[10] = {1.f, 0.92387953251128674f, 0.70710678118654752f, 0.38268343236508977f, 0.f,
                                   -0.38268343236508977f, -0.70710678118654752f, -0.92387953251128674f, -1.f,
                                   -0.92387953251128674f};
        constexpr float S16[10] = {0.f, 0.38268343236508977f, 0.70710678118654752f, 0.92387953251128674f, 1.f,
                                   0.92387953251128674f, 0.70710678118654752f, 0.38268343236508977f, 0.f,
                                   -0.38268343236508977f};
#pragma unroll
        for (int k1 = 1; k1 < 4; ++k1)
#pragma unroll
            for (int n2 = 1; n2 < R2; ++n2) {
                constexpr int unit = 16 / R;                    // index into the 16th-root table
                const int m = n2 * k1 * unit;                   // <= 9
                if (m == 4) v[R2 * k1 + n2] = cq_add_i<S>(f2{0.f, 0.f}, v[R2 * k1 + n2]);       // (S i) x
                else v[R2 * k1 + n2] = cq_mul<(S < 0)>(v[R2 * k1 + n2], f2{C16[m], S16[m]});
            }
#pragma unroll
        for (int k1 = 0; k1 < 4; ++k1) cq_fft<R2, S>(v + R2 * k1);
    }
}
template <int R>
__device__ __forceinline__ constexpr int cq_perm(int r) {       // register r of cq_fft<R> holds output bin cq_perm(r)
    if constexpr (R <= 4) return r;
    else {
        constexpr int R2 = R / 4;
        const int k1 = r / R2, k2 = r % R2;
        return k1 + 4 * (R2 <= 4 ? k2 : cq_perm<R2>(k2));
    }
}

// ---- pass plan of a T = 2^LT point band: radices 16,16,16 / 16,16,8 / 16,16,4 / 16,8,4 / 16,16 / 16,8 / 16,4 / 16,2 / 16 / 8 / 4
__host__ __device__ constexpr int cq_npass(int LT) { return LT >= 9 ? 3 : (LT >= 5 ? 2 : 1); }
__host__ __device__ constexpr int cq_radix(int LT, int p) {
    if (LT <= 3) return 1 << LT;
    if (p == 0) return 16;
    if (LT >= 10) return p == 1 ? 16 : (1 << (LT - 8));
    if (LT == 9) return p == 1 ? 8 : 4;
    return 1 << (LT - 4);                       // LT 5..8, p == 1
}
__host__ __device__ constexpr int cq_log2(int v) { return v <= 1 ? 0 : 1 + cq_log2(v >> 1); }
__host__ __device__ constexpr int cq_nsl(int LT, int p) { return p == 0 ? 0 : cq_nsl(LT, p - 1) + cq_log2(cq_radix(LT, p - 1)); }

// table entry exp(S' 2 pi i q / 4096), q in [0, 2048); the table (global memory, 16 KB: L1 / L2 resident) holds
// exp(-2 pi i q / 4096) for q < 2048
// (callers: q = k t with k < 4096 / (Ns R) * Ns and t in {1, 2, 4, 8} <= R / 2: always below 2048, no sign fold)
__device__ __forceinline__ f2 cq_tw(const f2* __restrict__ tw, int q) { return tw[q]; }

// Stockham pass P of a band, data in REGISTERS between the butterflies and LDS only between passes:
//   butterfly c of thread u: j = u + c*TB; inputs t = 0..R-1 are elements j + t*N/R of the previous stage, multiplied by
//   W^(k t), k = j mod Ns (Ns = product of the earlier radices); outputs r land at d + cq_perm(r)*Ns,
//   d = (j / Ns) * Ns * R + k (autosort: natural order in and out).
// v[] holds the R0 = min(16, T) values of the thread: as [CNT][R] for a pass of radix R (CNT = R0 / R butterflies).
template <int LT, int P>
struct CqPass {
    static constexpr int N = 1 << LT;
    static constexpr int R0 = N < 16 ? N : 16;
    static constexpr int TB = N / R0;
    static constexpr int R = cq_radix(LT, P);
    static constexpr int LR = cq_log2(R);
    static constexpr int CNT = R0 / R;
    static constexpr int NR = N / R;
    static constexpr int NSL = cq_nsl(LT, P);
    static constexpr int Ns = 1 << NSL;
    // element index (inside the band) of output r of butterfly c
    static __device__ __forceinline__ int out_index(int u, int c, int r) {
        const int j = u + c * TB;
        return ((j >> NSL) << (NSL + LR)) + (j & (Ns - 1)) + cq_perm<R>(r) * Ns;
    }
    // (LDS addresses: one padded base address per butterfly + compile-time offsets.  With i = i0 + 16 q the padded index is
    // CQ_AT(i0) + 17 q: every stride that occurs - N/R between the inputs of a butterfly, Ns between its outputs - is a
    // multiple of 16, or (first pass, Ns = 1) the 16 outputs are one aligned run.)
    static __device__ __forceinline__ void scatter(const f2* v, f2* a, int base, int u) {
#pragma unroll
        for (int c = 0; c < CNT; ++c) {
            const int j = u + c * TB;
            f2* ap = a + CQ_AT(base + ((j >> NSL) << (NSL + LR)) + (j & (Ns - 1)));
#pragma unroll
            for (int r = 0; r < R; ++r) {
                static_assert(Ns == 1 || Ns % 16 == 0, "pass plan");
                constexpr int unit = Ns == 1 ? 1 : Ns + Ns / 16;
                ap[cq_perm<R>(r) * unit] = v[c * R + r];
            }
        }
    }
    // number of table entries a butterfly of this pass loads: W^(k t) for t = 1, 2, 4, 8 below R
    static constexpr int NTW = R >= 16 ? 4 : (R >= 8 ? 3 : (R >= 4 ? 2 : 1));
    // The table entries of this pass depend on the thread only (k = j mod Ns): they are loaded BEFORE the first butterflies of
    // the transform (round 5) - issued right after the band's input loads, so that their global-memory latency runs behind those
    // and pass 0 instead of being paid after every inter-pass barrier (hipcc does not move a load across __syncthreads()).
    static __device__ __forceinline__ void load_tw(f2 (&pre)[CNT * NTW], const f2* __restrict__ tw, int u) {
#pragma unroll
        for (int c = 0; c < CNT; ++c) {
            const int j = u + c * TB;
            constexpr int stride = 4096 / (Ns * R);
            const int q1 = (j & (Ns - 1)) * stride;               // < 4096 / R
#pragma unroll
            for (int e = 0; e < NTW; ++e) pre[c * NTW + e] = cq_tw(tw, q1 << e);
        }
    }
    // LDS -> registers, twiddles, butterflies (P >= 1)
    template <int S>
    static __device__ __forceinline__ void gather_bfly(f2* v, const f2* a, const f2 (&pre)[CNT * NTW], int base, int u) {
#pragma unroll
        for (int c = 0; c < CNT; ++c) {
            const int j = u + c * TB;
            f2* x = v + c * R;
            static_assert(NR % 16 == 0, "pass plan");
            const f2* ap = a + CQ_AT(base + j);
#pragma unroll
            for (int t = 0; t < R; ++t) x[t] = ap[t * (NR + NR / 16)];
            // W^(k t): t = 1, 2, 4, 8 from the table, the rest as products of two of those (<= 3 roundings)
            // (table = exp(-2 pi i q/4096): S < 0 multiplies by it, S > 0 by its conjugate)
            f2 w[R];
            w[1] = pre[c * NTW];
            if constexpr (R >= 4) w[2] = pre[c * NTW + 1];
            if constexpr (R >= 8) w[4] = pre[c * NTW + 2];
            if constexpr (R >= 16) w[8] = pre[c * NTW + 3];
#pragma unroll
            for (int t = 3; t < R; ++t) {
                const int hb = t >= 8 ? 8 : (t >= 4 ? 4 : 2);      // highest power of two in t
                if (t != hb) w[t] = cq_mul<false>(w[hb], w[t - hb]);
            }
#pragma unroll
            for (int t = 1; t < R; ++t) x[t] = cq_mul<(S > 0)>(x[t], w[t]);
            cq_fft<R, S>(x);
        }
    }
};

// The whole transform of the thread's band: `v` enters with the inputs of pass 0 (v[t] = element u + t*TB) and leaves with
// the outputs of the last pass (CqPass<LT, NP-1>::out_index tells which).  Every thread of the workgroup must call
// (barriers); `active` guards the LDS traffic of threads without a band.
template <int LT, int S>
__device__ __forceinline__ void cq_band_fft_regs(f2* v, f2* a, const f2* __restrict__ tw, int base, int u, bool active) {
    constexpr int NP = cq_npass(LT);
    using P0 = CqPass<LT, 0>;
    using P1 = CqPass<LT, (NP >= 2 ? 1 : 0)>;
    using P2 = CqPass<LT, (NP >= 3 ? 2 : 0)>;
    f2 tw1[P1::CNT * P1::NTW], tw2[P2::CNT * P2::NTW];
    if constexpr (NP >= 2) P1::load_tw(tw1, tw, u);
    if constexpr (NP >= 3) P2::load_tw(tw2, tw, u);
    cq_fft<P0::R, S>(v);
    if constexpr (NP >= 2) {
        if (active) P0::scatter(v, a, base, u);
        __syncthreads();
        if (active) P1::template gather_bfly<S>(v, a, tw1, base, u);
        if constexpr (NP >= 3) {
            __syncthreads();
            if (active) P1::scatter(v, a, base, u);
            __syncthreads();
            if (active) P2::template gather_bfly<S>(v, a, tw2, base, u);
        }
    }
}

// Kaiser window g(m) / T of a band of M samples, m = -(M/2) .. M - M/2 - 1 (include/babe_hip.h, babe_cqt_bands::kpoly):
// Horner over a = 1 - (2 m / M)^2 with the coefficients in scalar registers
struct CqKaiser { float c[12]; int deg; float tom; };     // coefficients in (scalar) registers, 2 / M of the thread's band
template <int LT>
__device__ __forceinline__ float cq_kaiser(const CqKaiser& kz, int m) {
    const float x = (float)m * kz.tom;
    const float a = fmaxf(fmaf(-x, x, 1.f), 0.f);
    float p;                                           // (entries above deg are zero; the branch is uniform)
    if (kz.deg <= 6) {
        p = kz.c[6];
#pragma unroll
        for (int j = 5; j >= 0; --j) p = fmaf(p, a, kz.c[j]);
    } else {
        p = kz.c[11];
#pragma unroll
        for (int j = 10; j >= 0; --j) p = fmaf(p, a, kz.c[j]);
    }
    return p * (1.f / (float)(1 << LT));
}

// ANALYSIS (MODE 0): x[pos] = spec[(c + m) mod L] * win[m], m = pos for pos < M - M/2, pos - T for pos >= T - M/2, 0 between;
// IFFT_T (unnormalised; win carries 1/T); coefficients out planar.
// SYNTHESIS (MODE 1): x = coefficients; FFT_T; bs[woff + mi] = X[(mi - M/2) mod T] * win[mi].
// Global memory -> registers -> (LDS between passes only) -> registers -> global memory: the inputs of the first pass are
// loaded in butterfly order (element u + t*T/16 of the band: consecutive threads = consecutive samples) and the outputs of
// the last pass are stored from registers (runs of Ns >= 16 consecutive samples per instruction).  Round 2 staged the band
// through LDS before the first and after the last pass and kept the twiddle table in LDS: 256 KB of LDS traffic + 100 KB of
// twiddle reads + a 16 KB table copy per 4096 points - at B >= 32 the kernel was bound by LDS, not HBM.  Now 128 KB.
template <int LT, int MODE, bool AN, bool NT>
__device__ __forceinline__ void cq_run(const babe_cqt_bands& bd, f2* a, int k0, int nb, int oc, int bin0, int b,
                                       const float* __restrict__ spec, float* __restrict__ bs, long bs_stride,
                                       const float* __restrict__ win) {
    constexpr int T = 1 << LT;
    constexpr int R0 = T < 16 ? T : 16;
    constexpr int TB = T / R0;
    constexpr int NP = cq_npass(LT);
    constexpr unsigned OOB = 0x80000000u;                  // beyond every buffer below: loads return 0, stores are dropped
    using PL = CqPass<LT, NP - 1>;
    const int tid = threadIdx.x;
    const int s = tid / TB, u = tid % TB;
    const bool active = s < nb;
    const int k = k0 + (active ? s : 0);
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    const i32x4 br = reinterpret_cast<const i32x4*>(bd.band_rec)[k];          // {c, M, woff, 0}
    const int M = br[1], wo = br[2], half = M >> 1;
    const f2* tw = reinterpret_cast<const f2*>(bd.tw4096);
    constexpr bool analytic = AN;                         // Kaiser window evaluated here instead of read
    CqKaiser kz;
    kz.c[0] = bd.kpoly[0]; kz.c[1] = bd.kpoly[1]; kz.c[2] = bd.kpoly[2]; kz.c[3] = bd.kpoly[3]; kz.c[4] = bd.kpoly[4]; kz.c[5] = bd.kpoly[5];
    kz.c[6] = bd.kpoly[6]; kz.c[7] = bd.kpoly[7]; kz.c[8] = bd.kpoly[8]; kz.c[9] = bd.kpoly[9]; kz.c[10] = bd.kpoly[10]; kz.c[11] = bd.kpoly[11];
    kz.deg = bd.kdeg;
    kz.tom = 2.f / (float)M;
    // All global traffic goes through buffer descriptors: 32-bit per-thread offsets, the compile-time part of an address in
    // the instruction's scalar offset, the range check as the "outside the window" zero - the kernel is bound by the
    // vector ALU (round 3 PMC: 1100 vector instructions per thread and 16 points, a third of them address arithmetic).
    // coefficients of the workgroup's octave, clip b: [2][binsoct][T]; bands of a workgroup are consecutive bins
    const unsigned plane = (unsigned)bd.binsoct * T * 4;                                   // bytes of the real plane
    const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(bd.coef[oc] + (long)b * 2 * bd.binsoct * T, 0, 2 * plane, 0x00020000);
    const unsigned cfo = active ? (unsigned)((bin0 + s) * T) * 4 : OOB;                    // this thread's band
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)win, 0, (unsigned)bd.sum_M * 4, 0x00020000);
    f2 v[R0];
    if (MODE == 0) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(spec + (long)b * 2 * bd.KX), 0, 2 * (unsigned)bd.KX * 4, 0x00020000);
        const int c = br[0];
        const unsigned im = (unsigned)bd.KX * 4;
        float vr[R0], vi[R0], vw[R0];
        // the band's window covers spectrum bins c - half .. c + M - half - 1; almost every band lies inside 0 .. L/2
        if ((ABL & 1) && R0 == 16) {
            typedef float f32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
            for (int t4 = 0; t4 < 4; ++t4) {
                const unsigned so = active ? (unsigned)(c + 4 * u + t4 * 4 * TB) * 4 : OOB;
                const unsigned wv = active ? (unsigned)(wo + 4 * u + t4 * 4 * TB) * 4 : OOB;
                const f32x4 a0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, so, 0, 0));
                const f32x4 a1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, so, im, 0));
                const f32x4 a2 = (ABL & 4) ? f32x4{1.f, 1.f, 1.f, 1.f} : __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, wv, 0, 0));
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    vr[4 * t4 + i] = a0[i];
                    vi[4 * t4 + i] = a1[i];
                    vw[4 * t4 + i] = a2[i];
                }
            }
        } else if (c - half >= 0 && c + M - half - 1 <= bd.L / 2) {
#pragma unroll
            for (int t = 0; t < R0; ++t) {
                const int pos = u + t * TB;
                const bool hi = pos >= T - half;
                const bool in = active && (pos < M - half || hi);
                const int m = hi ? pos - T : pos;
                const unsigned so = in ? (unsigned)(c + m) * 4 : OOB;
                const unsigned wv = in ? (unsigned)(wo + half + m) * 4 : OOB;
                vr[t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, so, 0, (ABL & 128) ? 2 : 0));
                vi[t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, so, im, (ABL & 128) ? 2 : 0));
                if (analytic) vw[t] = cq_kaiser<LT>(kz, m);       // (outside the window the spectrum loads return 0)
                else vw[t] = (ABL & 4) ? 1.f : __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rw, wv, 0, 0));
            }
        } else {                                          // bands that wrap around bin 0 or reach past L/2 (mirror: conjugate)
#pragma unroll
            for (int t = 0; t < R0; ++t) {
                const int pos = u + t * TB;
                const int m = pos >= T - half ? pos - T : pos;
                const bool in = active && (pos < M - half || pos >= T - half);
                int n = c + m;
                n = n < 0 ? n + bd.L : n;
                n = n >= bd.L ? n - bd.L : n;
                const bool mir = n > bd.L / 2;
                const int nn = mir ? bd.L - n : n;
                const unsigned so = in ? (unsigned)nn * 4 : OOB;
                const unsigned wv = in ? (unsigned)(wo + half + m) * 4 : OOB;
                vr[t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, so, 0, 0));
                vi[t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, so, im, 0));
                if (analytic) vw[t] = cq_kaiser<LT>(kz, m);
                else vw[t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rw, wv, 0, 0));
                if (mir) vi[t] = -vi[t];
            }
        }
#pragma unroll
        for (int t = 0; t < R0; ++t) v[t] = f2{vr[t], vi[t]} * vw[t];
    } else {
        float vr[R0], vi[R0];
#pragma unroll
        for (int t = 0; t < R0; ++t) {
            vr[t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rc, cfo + u * 4, t * TB * 4, 0));
            vi[t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rc, cfo + u * 4, plane + t * TB * 4, 0));
        }
#pragma unroll
        for (int t = 0; t < R0; ++t) v[t] = f2{vr[t], vi[t]};
    }
    if (!(ABL & 8) || MODE != 0) cq_band_fft_regs<LT, (MODE == 0 ? +1 : -1)>(v, a, tw, s << LT, u, active);
    if (MODE == 0 && (ABL & 2) && R0 == 16) {
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        typedef int i32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4) {
            const unsigned o0 = cfo + (unsigned)(4 * u + t4 * 4 * TB) * 4;
            const f32x4 yr = {v[4 * t4].x, v[4 * t4 + 1].x, v[4 * t4 + 2].x, v[4 * t4 + 3].x};
            const f32x4 yi = {v[4 * t4].y, v[4 * t4 + 1].y, v[4 * t4 + 2].y, v[4 * t4 + 3].y};
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, yr), rc, o0, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, yi), rc, o0, plane, 0);
        }
    } else if (MODE == 0) {
#pragma unroll
        for (int c = 0; c < PL::CNT; ++c) {
            const unsigned o0 = cfo + (unsigned)PL::out_index(u, c, 0) * 4;                // (cq_perm(0) = 0)
#pragma unroll
            for (int r = 0; r < PL::R; ++r) {
                const int d = cq_perm<PL::R>(r) * PL::Ns * 4;
                // (floats first: hipcc 7.2 miscompiles __builtin_bit_cast of an ext_vector ELEMENT - it reads element 0)
                const float yr = v[c * PL::R + r].x, yi = v[c * PL::R + r].y;
                // (NT: non-temporal stores from 8 clips on - the coefficients of a large batch are read much later, by the UNet's
                // first convolutions, and need not displace the spectrum in L2: -3.4 % at 32 clips, profiles/r06_cqt_bench.txt)
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(yr), rc, o0, d, NT ? 2 : 0);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(yi), rc, o0, plane + d, NT ? 2 : 0);
            }
        }
    } else {
        // band spectra: [B][sum_M] complex; one descriptor over the whole tensor would pass 4 GB at large B: per clip
        typedef int i32x2 __attribute__((ext_vector_type(2)));
        const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(bs + (long)b * bs_stride * 2, 0, (unsigned)bd.sum_M * 8, 0x00020000);
#pragma unroll
        for (int c = 0; c < PL::CNT; ++c)
#pragma unroll
            for (int r = 0; r < PL::R; ++r) {
                const int mi = (PL::out_index(u, c, r) + half) & (T - 1);
                const bool ok = active && mi < M;
                const float w = analytic ? cq_kaiser<LT>(kz, mi - half)
                                         : __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rw, ok ? (unsigned)(wo + mi) * 4 : OOB, 0, 0));
                const f2 o = v[c * PL::R + r] * w;
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(i32x2, o), ro, ok ? (unsigned)(wo + mi) * 8 : OOB, 0, 0);
            }
    }
}

template <int MODE, bool AN, bool NT>
__global__ __launch_bounds__(256)
#if ABL & 32
__attribute__((amdgpu_waves_per_eu(6, 6)))
#endif
void band_fft_kernel(babe_cqt_bands bd, const float* __restrict__ spec,
                                                       float* __restrict__ bs, long bs_stride,
                                                       const float* __restrict__ win) {
    __shared__ f2 a[CQ_LDS];
    const int wg = blockIdx.x, b = blockIdx.y;
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    const i32x4 wr = reinterpret_cast<const i32x4*>(bd.wg_rec)[wg];           // {first band, bands, log2 T, oct | binoct << 8}
    const int k0 = wr[0], nb = wr[1], lt = wr[2], oc = wr[3] & 255, bin0 = wr[3] >> 8;
    if (nb > 64 || (nb << lt) > 4096 || oc >= 8) __builtin_trap();      // a table that contradicts its own summary fields: fail loudly
    switch (lt) {
#define CQ_CASE(L_) case L_: cq_run<L_, MODE, AN, NT>(bd, a, k0, nb, oc, bin0, b, spec, bs, bs_stride, win); break;
#ifdef CQ_ONLY                    // (instruction counting: one band length per build)
        CQ_CASE(CQ_ONLY)
#else
        CQ_CASE(12) CQ_CASE(11) CQ_CASE(10) CQ_CASE(9) CQ_CASE(8) CQ_CASE(7) CQ_CASE(6) CQ_CASE(5) CQ_CASE(4) CQ_CASE(3) CQ_CASE(2)
#endif
#undef CQ_CASE
        default: __builtin_trap();
    }
}

// overlap-add in frequency as a gather (deterministic: fixed summation order, no atomics).  A thread owns bin n of GB
// consecutive clips; the sources of a bin (the same for every clip) come as ONE 16-byte record {s0, s1, s2, count} built from
// the CSR on the host when no bin has more than three (NSGT windows overlap pairwise; a third source appears where rounded
// window lengths or the mirrored bands reach one bin further) - round 2 walked rowptr -> src -> data, three dependent
// loads deep, and the kernel took as long as the band FFTs.  Sources beyond the count get an out-of-range offset and load 0.
constexpr int GB = 8;
__global__ __launch_bounds__(256) void gather_rec_kernel(const float* __restrict__ bs, long bs_stride, unsigned bs_bytes,
                                                         const int* __restrict__ rec, float* __restrict__ spec, int KX,
                                                         int L, float scale, const float* __restrict__ mul, int B) {
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const int b0 = blockIdx.y * GB;
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= KX) return;
    float re[GB], im[GB];
#pragma unroll
    for (int i = 0; i < GB; ++i) re[i] = im[i] = 0.f;
    if (n <= L / 2) {
        const i32x4 r = reinterpret_cast<const i32x4*>(rec)[n];
        unsigned off[3];
        float sg[3];
#pragma unroll
        for (int e = 0; e < 3; ++e) {
            off[e] = e < r[3] ? (unsigned)(r[e] & 0x7fffffff) * 8u : 0x80000000u;
            sg[e] = r[e] < 0 ? -1.f : 1.f;
        }
        float sc = scale;
        if (mul) sc *= mul[n];
        f32x2 v[GB][3];
#pragma unroll
        for (int i = 0; i < GB; ++i) {
            // (clips past the batch: a zero-length descriptor, every load returns 0)
            const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)(bs + (long)(b0 + i < B ? b0 + i : 0) * bs_stride * 2), 0,
                                                                                b0 + i < B ? bs_bytes : 0u, 0x00020000);
#pragma unroll
            for (int e = 0; e < 3; ++e) v[i][e] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rb, off[e], 0, 0));
        }
#pragma unroll
        for (int i = 0; i < GB; ++i) {
#pragma unroll
            for (int e = 0; e < 3; ++e) {
                const float vx = v[i][e].x, vy = v[i][e].y;
                re[i] += vx;
                im[i] += sg[e] * vy;
            }
            re[i] *= sc;
            im[i] *= sc;
        }
    }
#pragma unroll
    for (int i = 0; i < GB; ++i)
        if (b0 + i < B) {
            spec[(long)(b0 + i) * 2 * KX + n] = re[i];
            spec[(long)(b0 + i) * 2 * KX + KX + n] = im[i];
        }
}

// general form (any number of sources per bin): walks the CSR
__global__ __launch_bounds__(256) void gather_kernel(const float* __restrict__ bs, long bs_stride,
                                                     const int* __restrict__ rowptr, const int* __restrict__ src,
                                                     float* __restrict__ spec, int KX, int L, float scale,
                                                     const float* __restrict__ mul) {
    const int b = blockIdx.y;
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= KX) return;
    float re = 0.f, im = 0.f;
    if (n <= L / 2) {
        const float2* p = reinterpret_cast<const float2*>(bs) + (long)b * bs_stride;
        const int e0 = rowptr[n], e1 = rowptr[n + 1];
        for (int e = e0; e < e1; ++e) {
            const int s = src[e];
            const float2 v = p[s & 0x7fffffff];
            re += v.x;
            im += (s < 0) ? -v.y : v.y;
        }
        float sc = scale;
        if (mul) sc *= mul[n];
        re *= sc;
        im *= sc;
    }
    spec[(long)b * 2 * KX + n] = re;
    spec[(long)b * 2 * KX + KX + n] = im;
}

__global__ __launch_bounds__(256) void spec_scale_kernel(const float* __restrict__ s1, const float* __restrict__ s2,
                                                         float* __restrict__ out, const float* __restrict__ mul,
                                                         int KX, int L, float sc1, float sc2) {
    const int b = blockIdx.y;
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= KX) return;
    float re = 0.f, im = 0.f;
    if (n <= L / 2) {
        const float m = mul ? mul[n] : 1.f;
        const long o = (long)b * 2 * KX + n;
        re = s1[o] * m * sc1;
        im = s1[o + KX] * m * sc1;
        if (s2) {
            re += s2[o] * m * sc2;
            im += s2[o + KX] * m * sc2;
        }
    }
    out[(long)b * 2 * KX + n] = re;
    out[(long)b * 2 * KX + KX + n] = im;
}

// 32x32 tiled transpose with complex twiddle.  grid (ceil(N2/32), ceil(N1/32), B)
__global__ __launch_bounds__(256) void twiddle_transpose_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                                const float2* __restrict__ tw, int N1, int N2,
                                                                int adjoint) {
    __shared__ float tr[32][33], ti[32][33];
    const int b = blockIdx.z;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;       // 32 x 8
    if (!adjoint) {
        // in [2*N1][N2] -> out [2*N2][N1]
        const float* ire = in + (long)b * 2 * N1 * N2;
        const float* iim = ire + (long)N1 * N2;
        float* ore = out + (long)b * 2 * N2 * N1;
        float* oim = ore + (long)N2 * N1;
        const int n2 = blockIdx.x * 32 + tx;
        for (int r = ty; r < 32; r += 8) {
            const int k1 = blockIdx.y * 32 + r;
            float vr = 0.f, vi = 0.f;
            if (k1 < N1 && n2 < N2) {
                const float ar = ire[(long)k1 * N2 + n2], ai = iim[(long)k1 * N2 + n2];
                const float2 w = tw[(long)k1 * N2 + n2];
                vr = ar * w.x - ai * w.y;
                vi = ar * w.y + ai * w.x;
            }
            tr[r][tx] = vr;
            ti[r][tx] = vi;
        }
        __syncthreads();
        const int k1 = blockIdx.y * 32 + tx;
        for (int r = ty; r < 32; r += 8) {
            const int n2o = blockIdx.x * 32 + r;
            if (k1 < N1 && n2o < N2) {
                ore[(long)n2o * N1 + k1] = tr[tx][r];
                oim[(long)n2o * N1 + k1] = ti[tx][r];
            }
        }
    } else {
        // in [2*N2][N1] -> out [2*N1][N2], multiply by conj(tw[k1][n2])
        const float* ire = in + (long)b * 2 * N2 * N1;
        const float* iim = ire + (long)N2 * N1;
        float* ore = out + (long)b * 2 * N1 * N2;
        float* oim = ore + (long)N1 * N2;
        const int k1 = blockIdx.y * 32 + tx;
        for (int r = ty; r < 32; r += 8) {
            const int n2 = blockIdx.x * 32 + r;
            float vr = 0.f, vi = 0.f;
            if (k1 < N1 && n2 < N2) {
                vr = ire[(long)n2 * N1 + k1];
                vi = iim[(long)n2 * N1 + k1];
            }
            tr[r][tx] = vr;
            ti[r][tx] = vi;
        }
        __syncthreads();
        const int n2 = blockIdx.x * 32 + tx;
        for (int r = ty; r < 32; r += 8) {
            const int k1o = blockIdx.y * 32 + r;
            if (k1o < N1 && n2 < N2) {
                const float ar = tr[tx][r], ai = ti[tx][r];
                const float2 w = tw[(long)k1o * N2 + n2];
                ore[(long)k1o * N2 + n2] = ar * w.x + ai * w.y;
                oim[(long)k1o * N2 + n2] = ai * w.x - ar * w.y;
            }
        }
    }
}

}  // namespace

extern "C" int babe_fft_twiddle_transpose(const float* in, float* out, const float* tw, int B, int N1, int N2,
                                          int adjoint, void* stream) {
    BABE_CHECK_ARG(in && out && tw && B > 0 && N1 > 0 && N2 > 0, "fft_twiddle_transpose: bad arguments");
    BabeProfScope prof(BABE_SLOT_CQT_GATHER, 24.0 * B * (double)N1 * N2, 0, 0, stream);
    hipLaunchKernelGGL(twiddle_transpose_kernel, dim3(cdiv(N2, 32), cdiv(N1, 32), B), dim3(256), 0,
                       (hipStream_t)stream, in, out, reinterpret_cast<const float2*>(tw), N1, N2, adjoint);
    BABE_LAUNCH_CHECK();
    return BABE_OK;
}

static int check_bands(const babe_cqt_bands* bd) {
    BABE_CHECK_ARG(bd && bd->nbands > 0 && bd->c && bd->M && bd->woff && bd->log2T && bd->oct && bd->binoct &&
                       bd->tw4096 && bd->nocts <= 8 && bd->wg_first && bd->wg_count && bd->nwg > 0 && bd->wg_rec && bd->band_rec,
                   "cqt: bad band table");
    // what the band-FFT kernel relies on: <= 64 bands per workgroup (thread = band slot x butterfly), T in 4..4096 (its dispatch)
    BABE_CHECK_ARG(bd->max_wg_count >= 1 && bd->max_wg_count <= 64, "cqt: wg_count must be in 1..64 (got %d)", bd->max_wg_count);
    BABE_CHECK_ARG(bd->min_log2T >= 2 && bd->max_log2T <= 12 && bd->min_log2T <= bd->max_log2T,
                   "cqt: band lengths must be 2^2..2^12 (got 2^%d..2^%d)", bd->min_log2T, bd->max_log2T);
    BABE_CHECK_ARG(bd->binsoct >= 1 && bd->nwg <= bd->nbands, "cqt: workgroup table inconsistent with the band table");
    return 0;
}

extern "C" int babe_cqt_band_analysis(const babe_cqt_bands* bd, const float* spec, const float* win, int B,
                                      void* stream) {
    if (check_bands(bd)) return BABE_ERR_ARG;
    BABE_CHECK_ARG(spec && B > 0, "cqt_band_analysis: bad arguments");
    BABE_CHECK_ARG(win || (bd->kdeg > 0 && bd->kdeg <= 11), "cqt_band_analysis: no window table and no analytic window (kdeg = %d)", bd->kdeg);
    BabeProfScope prof(BABE_SLOT_CQT_ANALYSIS, (double)B * (8.0 * (bd->L / 2 + 1) + 8.0 * bd->sum_T + (win ? 4.0 : 0.0) * bd->sum_M), 5.0 * B * bd->sum_TlogT, 0, stream);
    const dim3 grid(bd->nwg, B);
    const hipStream_t st = (hipStream_t)stream;
    if (win && B >= 8) hipLaunchKernelGGL((band_fft_kernel<0, false, true>), grid, dim3(256), 0, st, *bd, spec, (float*)nullptr, 0L, win);
    else if (win) hipLaunchKernelGGL((band_fft_kernel<0, false, false>), grid, dim3(256), 0, st, *bd, spec, (float*)nullptr, 0L, win);
    else if (B >= 8) hipLaunchKernelGGL((band_fft_kernel<0, true, true>), grid, dim3(256), 0, st, *bd, spec, (float*)nullptr, 0L, win);
    else hipLaunchKernelGGL((band_fft_kernel<0, true, false>), grid, dim3(256), 0, st, *bd, spec, (float*)nullptr, 0L, win);
    BABE_LAUNCH_CHECK();
    return BABE_OK;
}

extern "C" int babe_cqt_band_synthesis(const babe_cqt_bands* bd, float* bs, const float* win, long bs_stride, int B,
                                       void* stream) {
    if (check_bands(bd)) return BABE_ERR_ARG;
    BABE_CHECK_ARG(bs && B > 0, "cqt_band_synthesis: bad arguments");
    BABE_CHECK_ARG(win || (bd->kdeg > 0 && bd->kdeg <= 11), "cqt_band_synthesis: no window table and no analytic window (kdeg = %d)", bd->kdeg);
    BabeProfScope prof(BABE_SLOT_CQT_SYNTHESIS, (double)B * (8.0 * bd->sum_T + (win ? 12.0 : 8.0) * bd->sum_M), 5.0 * B * bd->sum_TlogT, 0, stream);
    if (win) hipLaunchKernelGGL((band_fft_kernel<1, false, false>), dim3(bd->nwg, B), dim3(256), 0, (hipStream_t)stream, *bd, (const float*)nullptr, bs, bs_stride, win);
    else hipLaunchKernelGGL((band_fft_kernel<1, true, false>), dim3(bd->nwg, B), dim3(256), 0, (hipStream_t)stream, *bd, (const float*)nullptr, bs, bs_stride, win);
    BABE_LAUNCH_CHECK();
    return BABE_OK;
}

extern "C" int babe_cqt_gather(const float* bs, long bs_stride, const int* rowptr, const int* src, const int* rec,
                               float* spec, int KX, int L, float scale, const float* mul, int B, void* stream) {
    BABE_CHECK_ARG(bs && rowptr && src && spec && KX > L / 2 && B > 0, "cqt_gather: bad arguments");
    BabeProfScope prof(BABE_SLOT_CQT_GATHER, (double)B * 8.0 * (bs_stride + KX), 0, 0, stream);
    if (rec && bs_stride * 8 < 0x80000000L)
        hipLaunchKernelGGL(gather_rec_kernel, dim3(cdiv(KX, 256), cdiv(B, GB)), dim3(256), 0, (hipStream_t)stream, bs,
                           bs_stride, (unsigned)(bs_stride * 8), rec, spec, KX, L, scale, mul, B);
    else
        hipLaunchKernelGGL(gather_kernel, dim3(cdiv(KX, 256), B), dim3(256), 0, (hipStream_t)stream, bs, bs_stride, rowptr,
                           src, spec, KX, L, scale, mul);
    BABE_LAUNCH_CHECK();
    return BABE_OK;
}

extern "C" int babe_spec_scale(const float* s1, const float* s2, float* out, const float* mul, int KX, int L,
                               float sc1, float sc2, int B, void* stream) {
    BABE_CHECK_ARG(s1 && out && KX > L / 2 && B > 0, "spec_scale: bad arguments");
    BabeProfScope prof(BABE_SLOT_CQT_GATHER, (double)B * 8.0 * KX * (s2 ? 3 : 2), 0, 0, stream);
    hipLaunchKernelGGL(spec_scale_kernel, dim3(cdiv(KX, 256), B), dim3(256), 0, (hipStream_t)stream, s1, s2, out, mul,
                       KX, L, sc1, sc2);
    BABE_LAUNCH_CHECK();
    return BABE_OK;
}
