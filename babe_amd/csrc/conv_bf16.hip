// bf16-MFMA variant of the dilated conv (same tiling, epilogue and C-ABI argument block as conv.hip).
//
// Activations stay fp32 in HBM; they are converted to bf16 while being staged into LDS, weights are packed to
// bf16 once.  v_mfma_f32_32x32x16_bf16 accumulates in fp32.  Two precisions:
//   SX = 1  "bf16"    : one product per k-block                      (configs #3-#5 of BASELINE.json)
//   SX = 2  "bf16x3"  : x = x_hi + x_lo, w = w_hi + w_lo (both bf16), products hi*hi + hi*lo + lo*hi:
//                       16-bit-mantissa multiplies, fp32 accumulate, at 3/16 of the fp32-MFMA cost.
// LDS images are arrays of 16-byte units (8 consecutive input channels of one position / one output channel),
// which is exactly one MFMA operand fragment: A = weights [k-group h][co], B = activations [k-group h][pos].
#include "common.h"
#include "../../include/babe_hip.h"
#include "prof.h"
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

namespace {

__device__ __forceinline__ float sel_scale(bool has, float loaded) { return has ? loaded : 1.f; }

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

struct ConvGeomB {
    int GP, CoutP, pt_log2, pr_log2, tiles_t;
    long split_stride;      // elements (shorts) between the hi and lo weight images
};

template <int NT, int WP, int KW, int SX>
__global__ __launch_bounds__(256, 2) void conv_bf16_kernel(babe_conv_args a, ConvGeomB g,
                                                           const unsigned short* __restrict__ wq) {
    constexpr int KC = 16, G = 2;
    constexpr int BN = NT * 32;
    constexpr int NPOS = 128 * WP;
    constexpr int TG = 256 / NPOS;
    constexpr int CPT = KC / TG;            // staged channels per thread (8 or 16)
    constexpr int GPT = CPT / 8;            // 16-byte units per thread and position
    constexpr int NWU = SX * KW * G * BN;   // weight units per chunk
    constexpr int WJ = (NWU + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16x8* smem = reinterpret_cast<bf16x8*>(smem_raw);
    const int PT = 1 << g.pt_log2;
    const int PR = 1 << g.pr_log2;
    const int XROW = PT + 2;
    const int XCH = PR * XROW;
    const int XBUF = SX * G * XCH;          // units
    const int BUF = XBUF + NWU;

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
    const int aoff = XBUF + h * BN + l31;

    constexpr int padt = KW >> 1;
    const int khc = a.KH >> 1;
    const int cin_split = a.in2 ? a.cin_split : a.Cin;
    const float* isc = a.in_scale ? a.in_scale : a.in;     // always-readable address: the load below is unconditional
    const bool has_isc = a.in_scale != nullptr;
    const int CinP = g.GP * 8;

    const int pg = tid & (NPOS - 1);
    const int cgrp = (TG == 1) ? 0 : __builtin_amdgcn_readfirstlane(tid / NPOS) * CPT;
    int se[2], sr[2], srt[2];
    bool sv[2], stv[2];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
        const int e = pg + s2 * NPOS;
        se[s2] = e;
        sv[s2] = e < XCH;
        const int r = e / XROW;
        const int tt = e - r * XROW;
        const int t = t0 + tt - padt;
        sr[s2] = r;
        stv[s2] = sv[s2] && t >= 0 && t < a.T;
        srt[s2] = r * a.T + t;
    }
    float xr[2][CPT];
    u32x4 wr[WJ];
    float scj[CPT];
    bool cok[CPT];
    bool okm[2];

    ChanSrc chan_ptr;
    chan_ptr.init(a.in, a.in_bs, a.in_cs, a.in2, a.in2_bs, a.in2_cs, cin_split, b);
    auto kh_valid = [&](int kh) {
        const int foff = (kh - khc) * a.dil;
        return !(f0 + foff + PR <= 0 || f0 + foff >= a.F);
    };
    auto load_chunk = [&](int kh, int ci0) {
        const int foff = (kh - khc) * a.dil;
        const float* srcj[CPT];
#pragma unroll
        for (int j = 0; j < CPT; ++j) {
            const int cir = ci0 + cgrp + j;
            cok[j] = cir < a.Cin;
            const int ci = cok[j] ? cir : a.Cin - 1;
            srcj[j] = chan_ptr(ci);
            scj[j] = sel_scale(has_isc, isc[b * a.Cin + ci]);
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const int f = f0 + sr[s2] + foff;
            const bool ok = stv[s2] && f >= 0 && f < a.F;
            okm[s2] = ok;
            const long off = ok ? (long)(f0 + foff) * a.T + srt[s2] : 0;
#pragma unroll
            for (int j = 0; j < CPT; ++j) xr[s2][j] = srcj[j][off];
        }
#pragma unroll
        for (int jj = 0; jj < WJ; ++jj) {
            int idx = tid + jj * 256;
            if (idx > NWU - 1) idx = NWU - 1;
            const int co = idx % BN;
            int r = idx / BN;                  // (s*KW + kw)*G + gl
            const int gl = r % G;
            r /= G;
            const int kw = r % KW;
            const int s = r / KW;
            const long src = (long)s * g.split_stride +
                             ((((long)(kh * KW + kw) * g.GP + (ci0 >> 3) + gl) * g.CoutP) + co0 + co) * 8;
            wr[jj] = *reinterpret_cast<const u32x4*>(wq + src);
        }
    };
    auto store_chunk = [&](bf16x8* buf) {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
            if (sv[s2]) {
#pragma unroll
                for (int gq = 0; gq < GPT; ++gq) {
                    bf16x8 hi, lo;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int jj = gq * 8 + j;
                        const float v = (okm[s2] && cok[jj]) ? xr[s2][jj] * scj[jj] : 0.f;
                        hi[j] = (__bf16)v;
                        if constexpr (SX == 2) lo[j] = (__bf16)(v - (float)hi[j]);
                    }
                    const int gl = (cgrp >> 3) + gq;
                    buf[gl * XCH + se[s2]] = hi;
                    if constexpr (SX == 2) buf[(G + gl) * XCH + se[s2]] = lo;
                }
            }
#pragma unroll
        for (int jj = 0; jj < WJ; ++jj) {
            const int idx = tid + jj * 256;
            if (idx < NWU) *reinterpret_cast<u32x4*>(buf + XBUF + idx) = wr[jj];
        }
    };

    int kh = 0;
    while (!kh_valid(kh)) ++kh;
    int ci0 = 0;
    load_chunk(kh, ci0);
    store_chunk(smem);
    __syncthreads();
    int cur = 0;
    while (true) {
        int nkh = kh, nci = ci0 + KC;
        if (nci >= CinP) {
            nci = 0;
            ++nkh;
            while (nkh < a.KH && !kh_valid(nkh)) ++nkh;
        }
        const bool has_next = nkh < a.KH;
        if (has_next) load_chunk(nkh, nci);
        const bf16x8* Xs = smem + cur * BUF;
#pragma unroll
        for (int kw = 0; kw < KW; ++kw) {
            bf16x8 av[SX][NT], bv[SX][WP];
#pragma unroll
            for (int s = 0; s < SX; ++s) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) av[s][nt] = Xs[aoff + ((s * KW + kw) * G) * BN + nt * 32];
#pragma unroll
                for (int wp = 0; wp < WP; ++wp) bv[s][wp] = Xs[s * G * XCH + boff[wp] + kw];
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int wp = 0; wp < WP; ++wp) {
                    if constexpr (SX == 2) {      // small cross terms first, then the leading product
                        acc[nt][wp] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[1][nt], bv[0][wp], acc[nt][wp], 0, 0, 0);
                        acc[nt][wp] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[0][nt], bv[1][wp], acc[nt][wp], 0, 0, 0);
                    }
                    acc[nt][wp] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[0][nt], bv[0][wp], acc[nt][wp], 0, 0, 0);
                }
        }
        if (has_next) store_chunk(smem + (cur ^ 1) * BUF);
        __syncthreads();
        if (!has_next) break;
        kh = nkh;
        ci0 = nci;
        cur ^= 1;
    }

    conv_epilogue<NT, WP>(a, acc, b, co0, f0, t0, g.pt_log2, wave, l31, h);
}

// dst [s][kh][kw][g][coP][8]
__global__ void pack_weights_bf16_kernel(const float* __restrict__ w, unsigned short* __restrict__ dst, int Cout,
                                         int Cin, int KH, int KW, int tf, int GP, int CoutP, long per_split,
                                         int splits) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= per_split) return;
    const int j = (int)(i & 7);
    long r = i >> 3;
    const int co = (int)(r % CoutP);
    r /= CoutP;
    const int gq = (int)(r % GP);
    r /= GP;
    const int kw = (int)(r % KW);
    const int kh = (int)(r / KW);
    const int ci = gq * 8 + j;
    float v = 0.f;
    if (!tf) {
        if (co < Cout && ci < Cin) v = w[(((long)co * Cin + ci) * KH + kh) * KW + kw];
    } else {
        if (co < Cin && ci < Cout) v = w[(((long)ci * Cin + co) * KH + (KH - 1 - kh)) * KW + (KW - 1 - kw)];
    }
    const __bf16 hi = (__bf16)v;
    dst[i] = __builtin_bit_cast(unsigned short, hi);
    if (splits == 2) {
        const __bf16 lo = (__bf16)(v - (float)hi);
        dst[per_split + i] = __builtin_bit_cast(unsigned short, lo);
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
inline int pick_nt(int CoutP) {
    const int n32 = CoutP / 32;
    for (int c = 4; c >= 1; --c)
        if (n32 % c == 0) return c;
    return 1;
}

template <int NT, int WP, int KW, int SX>
void launch(const babe_conv_args& a, ConvGeomB g, const unsigned short* wq, hipStream_t s) {
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
    const size_t units = (size_t)SX * 2 * PR * (PT + 2) + (size_t)SX * KW * 2 * NT * 32;
    hipLaunchKernelGGL((conv_bf16_kernel<NT, WP, KW, SX>), grid, dim3(256), 2 * units * 16, s, a, g, wq);
}

}  // namespace

int babe_conv2d_bf16p_supported(const babe_conv_args& a);       // conv_bf16p.hip
int babe_conv2d_bf16p_launch(const babe_conv_args& a, const unsigned short* wq, hipStream_t s);

extern "C" long babe_conv_packed_size_bf16(int Cout, int Cin, int KH, int KW, int transpose_flip, int splits) {
    const int co = transpose_flip ? Cin : Cout;
    const int ci = transpose_flip ? Cout : Cin;
    return (long)splits * KH * KW * ((ci + 15) / 16 * 16) * ((co + 31) / 32 * 32);      // in bf16 elements
}

extern "C" int babe_conv_pack_weights_bf16(const float* w, void* dst, int Cout, int Cin, int KH, int KW,
                                           int transpose_flip, int splits, void* stream) {
    BABE_CHECK_ARG(w && dst && Cout > 0 && Cin > 0 && (splits == 1 || splits == 2), "conv_pack_weights_bf16: bad arguments");
    const int co = transpose_flip ? Cin : Cout;
    const int ci = transpose_flip ? Cout : Cin;
    const int GP = (ci + 15) / 16 * 2, CoutP = (co + 31) / 32 * 32;
    const long per_split = (long)KH * KW * GP * 8 * CoutP;
    hipLaunchKernelGGL(pack_weights_bf16_kernel, dim3(cdiv(per_split, 256)), dim3(256), 0, (hipStream_t)stream, w,
                       (unsigned short*)dst, Cout, Cin, KH, KW, transpose_flip, GP, CoutP, per_split, splits);
    BABE_LAUNCH_CHECK();
    return BABE_OK;
}


extern "C" int babe_conv2d_bf16(const babe_conv_args* ap, const void* w_bf16, int splits, void* stream) {
    BABE_CHECK_ARG(ap && w_bf16, "conv2d_bf16: null args");
    const babe_conv_args& a = *ap;
    BABE_CHECK_ARG(a.in && a.out, "conv2d_bf16: null pointer");
    BABE_CHECK_ARG(a.B > 0 && a.Cin > 0 && a.Cout > 0 && a.F > 0 && a.T > 0, "conv2d_bf16: bad shape");
    BABE_CHECK_ARG((a.KH == 5 || a.KH == 1) && (a.KW == 3 || a.KW == 1) && a.dil >= 1, "conv2d_bf16: kernel %dx%d unsupported", a.KH, a.KW);
    BABE_CHECK_ARG(splits == 1 || splits == 2, "conv2d_bf16: splits must be 1 (bf16) or 2 (bf16x3)");
    BABE_CHECK_ARG(!a.in2 || (a.cin_split > 0 && a.cin_split < a.Cin), "conv2d_bf16: bad cin_split");
    ConvGeomB g;
    g.GP = (a.Cin + 15) / 16 * 2;
    g.CoutP = (a.Cout + 31) / 32 * 32;
    g.split_stride = (long)a.KH * a.KW * g.GP * 8 * g.CoutP;
    const int NT = pick_nt(g.CoutP);
    const long npos = (long)a.F * a.T;
    const long blocks256 = ((npos + 255) / 256) * (g.CoutP / 32 / NT) * a.B;
    // plain bf16 is staging-bound: favour the larger tile (weights amortised twice); bf16x3 needs the LDS for 2 blocks/CU
    bool wp2 = (splits == 1) && blocks256 >= 1024;
    {
        static const char* ov = getenv("BABE_CONV_WP");
        if (ov && ov[0] == '1') wp2 = false;
        if (ov && ov[0] == '2') wp2 = true;
    }
    hipStream_t s = (hipStream_t)stream;
    const unsigned short* wq = (const unsigned short*)w_bf16;
    const double flops = babe_conv_flops(a);
    if (splits == 1 && babe_conv2d_bf16p_supported(a)) {       // round-2 pipelined kernel, same arguments and weights
        BabeProfScope prof(BABE_SLOT_CONV_BF16P, babe_conv_bytes(a), flops, flops, stream);
        babe_conv2d_bf16p_launch(a, wq, s);
        BABE_LAUNCH_CHECK();
        return BABE_OK;
    }
    BabeProfScope prof(BABE_SLOT_CONV_BF16, babe_conv_bytes(a), flops, flops * (splits == 2 ? 3 : 1), stream);
#define BC(NTv, WPv)                                                              \
    if (a.KW == 3) {                                                              \
        if (splits == 2) launch<NTv, WPv, 3, 2>(a, g, wq, s);                     \
        else launch<NTv, WPv, 3, 1>(a, g, wq, s);                                 \
    } else {                                                                      \
        if (splits == 2) launch<NTv, WPv, 1, 2>(a, g, wq, s);                     \
        else launch<NTv, WPv, 1, 1>(a, g, wq, s);                                 \
    }
    if (wp2) {
        switch (NT) {
            case 4: BC(4, 2) break;
            case 3: BC(3, 2) break;
            case 2: BC(2, 2) break;
            default: BC(1, 2) break;
        }
    } else {
        switch (NT) {
            case 4: BC(4, 1) break;
            case 3: BC(3, 1) break;
            case 2: BC(2, 1) break;
            default: BC(1, 1) break;
        }
    }
#undef BC
    BABE_LAUNCH_CHECK();
    return BABE_OK;
}
