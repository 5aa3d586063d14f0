// Pieces shared by the direct conv kernels (conv.hip, conv11p.hip): the fused epilogue and the A-operand fragment read.
#pragma once
#include "common.h"
#include "../../include/babe_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

// Epilogue shared by all conv kernels: out = alpha*acc*oscale[b,co] + rbeta*res.  The 16 loads of a 32x32 tile are
// issued back-to-back inside ONE wave-uniform branch per operand: a per-element "if (ptr) load" makes hipcc branch
// around every load and wait vmcnt(0) each time (measured: the epilogue then serialises 128 load latencies).
template <int NT, int WP, bool HAS_OS, bool HAS_RES>
__device__ __forceinline__ void conv_epilogue_impl(const babe_conv_args& a, f32x16 (&acc)[NT][WP], int b, int co0,
                                                   int f0, int t0, int pt_log2, int wave, int l31, int h) {
    const int PT = 1 << pt_log2;
#pragma unroll
    for (int wp = 0; wp < WP; ++wp) {
        const int p = (wave * WP + wp) * 32 + l31;
        const int f = f0 + (p >> pt_log2);
        const int t = t0 + (p & (PT - 1));
        const bool pv = f < a.F && t < a.T;
        const long sp = pv ? (long)f * a.T + t : 0;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            float os[16], rr[16];
            int cc[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + nt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                cc[r] = co < a.Cout ? co : a.Cout - 1;
            }
            if constexpr (HAS_OS) {
#pragma unroll
                for (int r = 0; r < 16; ++r) os[r] = a.oscale[b * a.Cout + cc[r]];
            }
            if constexpr (HAS_RES) {
#pragma unroll
                for (int r = 0; r < 16; ++r) rr[r] = a.res[(long)b * a.res_bs + (long)cc[r] * a.res_cs + sp];
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + nt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                float v = acc[nt][wp][r] * a.alpha;
                if constexpr (HAS_OS) v *= os[r];
                if constexpr (HAS_RES) v += a.rbeta * rr[r];
                if (pv && co < a.Cout) a.out[(long)b * a.out_bs + (long)co * a.out_cs + sp] = v;
            }
        }
    }
}

template <int NT, int WP>
__device__ __forceinline__ void conv_epilogue(const babe_conv_args& a, f32x16 (&acc)[NT][WP], int b, int co0, int f0,
                                              int t0, int pt_log2, int wave, int l31, int h) {
    // four straight-line specialisations behind wave-uniform branches
    if (a.oscale) {
        if (a.res) conv_epilogue_impl<NT, WP, true, true>(a, acc, b, co0, f0, t0, pt_log2, wave, l31, h);
        else conv_epilogue_impl<NT, WP, true, false>(a, acc, b, co0, f0, t0, pt_log2, wave, l31, h);
    } else {
        if (a.res) conv_epilogue_impl<NT, WP, false, true>(a, acc, b, co0, f0, t0, pt_log2, wave, l31, h);
        else conv_epilogue_impl<NT, WP, false, false>(a, acc, b, co0, f0, t0, pt_log2, wave, l31, h);
    }
}

template <int N> struct AVec;
template <> struct AVec<1> { static __device__ __forceinline__ void ld(const float* p, float* v) { v[0] = p[0]; } };
template <> struct AVec<2> {
    static __device__ __forceinline__ void ld(const float* p, float* v) {
        const float2 t = *reinterpret_cast<const float2*>(p);
        v[0] = t.x; v[1] = t.y;
    }
};
template <> struct AVec<3> {
    static __device__ __forceinline__ void ld(const float* p, float* v) { v[0] = p[0]; v[1] = p[1]; v[2] = p[2]; }
};
template <> struct AVec<4> {
    static __device__ __forceinline__ void ld(const float* p, float* v) {
        const float4 t = *reinterpret_cast<const float4*>(p);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    }
};

}  // namespace
