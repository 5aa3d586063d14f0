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
#include "prof.h"
#include "conv_common.h"
#include <cstdlib>

namespace {

__device__ __forceinline__ float sel_scale(bool has, float loaded) { return has ? loaded : 1.f; }

struct ConvGeom {
    int CinP, CoutP, pt_log2, pr_log2, tiles_t;
};

// Packed weights are permuted inside every BN-wide output-channel tile so that the NT values one lane feeds to
// its NT MFMAs are contiguous in LDS: channel co0 + nt*32 + l  is stored at  co0 + l*NT + nt.
// VEC = true: activations are staged as 16-byte loads / ds_write_b128 (needs T % 4 == 0 and 16-byte aligned rows);
// the LDS row then carries its left halo at index 3 so that the body starts 16-byte aligned.
template <int NT, int WP, int KC, int KW, bool VEC>
__global__ __launch_bounds__(256, (WP == 1 ? 4 : 2)) void conv_mfma_kernel(babe_conv_args a, ConvGeom g) {
    constexpr int BN = NT * 32;
    constexpr int NPOS = 128 * WP;          // output positions per block
    constexpr int TG = 256 / NPOS;          // thread groups that split the KC staged channels
    constexpr int CPT = KC / TG;            // staged channels per thread
    constexpr int V4 = BN / 4;
    constexpr int NWV = KW * KC * V4;       // float4s of weights per chunk
    constexpr int WJ = (NWV + 255) / 256;
    constexpr int NSTEP = KW * (KC / 2);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int PT = 1 << g.pt_log2;
    const int PR = 1 << g.pr_log2;
    constexpr int padt = KW >> 1;
    constexpr int XPAD = VEC ? 8 : 2;
    constexpr int HOFF = VEC ? (4 - padt) : 0;   // LDS index of time t0 - padt within a row
    const int XROW = PT + XPAD;
    const int XCH = PR * XROW;              // staged elements per channel
    const int XBUF = (KC * XCH + 3) & ~3;
    const int BUF = XBUF + KW * KC * BN;    // floats per LDS buffer (double buffered)

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
        boff[wp] = (p >> g.pt_log2) * XROW + (p & (PT - 1)) + HOFF + h * XCH;
    }
    const int aoff = XBUF + h * BN + l31 * NT;

    const int khc = a.KH >> 1;
    const int cin_split = a.in2 ? a.cin_split : a.Cin;
    const float* isc = a.in_scale ? a.in_scale : a.in;     // always-readable address: the load below is unconditional
    const bool has_isc = a.in_scale != nullptr;

    // ---- per-thread staging slots: two halo-extended positions x CPT channels
    const int pg = tid & (NPOS - 1);
    const int cgrp = (TG == 1) ? 0 : __builtin_amdgcn_readfirstlane(tid / NPOS) * CPT;   // wave-uniform
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
    f32x4 wr[WJ];          // native vector type: stays in registers (HIP's float4 struct array went to scratch)
    float scj[CPT];        // per-channel input scale of the chunk in flight (wave-uniform)
    bool cok[CPT];         // channel < Cin
    bool okm[2];           // position inside the tensor

    // ---- VEC staging slots: WP float4 body pieces + one halo scalar per thread
    constexpr int NV = WP;                  // float4 pieces per thread (KC*NPOS/4/256)
    f32x4 xv[NV];
    float xh = 0.f;
    float vsc[NV], hsc = 1.f;
    bool vok[NV], hok = false;
    int vlds[NV], vrow[NV], vci[NV], vt[NV];
    const int q_per_row_log2 = g.pt_log2 - 2;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int idx = tid + v * 256;
        const int q4 = idx & ((1 << q_per_row_log2) - 1);
        const int r = (idx >> q_per_row_log2) & (PR - 1);
        vci[v] = idx >> (g.pt_log2 + g.pr_log2 - 2);           // local channel 0..KC-1
        vrow[v] = r;
        vt[v] = t0 + 4 * q4;
        vlds[v] = vci[v] * XCH + r * XROW + 4 + 4 * q4;
    }
    // halo: thread -> (channel, row, side)
    const int hci = tid >> (g.pr_log2 + 1);
    const int hrow = (tid >> 1) & (PR - 1);
    const int hside = tid & 1;
    const bool hactive = hci < KC && padt > 0;
    const int ht = hside ? t0 + PT : t0 - 1;
    const int hlds = hci * XCH + hrow * XROW + (hside ? PT + 4 : 3);

    auto kh_valid = [&](int kh) {
        const int foff = (kh - khc) * a.dil;
        return !(f0 + foff + PR <= 0 || f0 + foff >= a.F);
    };
    ChanSrc chan_ptr;
    chan_ptr.init(a.in, a.in_bs, a.in_cs, a.in2, a.in2_bs, a.in2_cs, cin_split, b);
    auto load_x_vec = [&](int kh, int ci0) {
        const int foff = (kh - khc) * a.dil;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int cir = ci0 + vci[v];
            const int f = f0 + vrow[v] + foff;
            const bool ok = cir < a.Cin && f >= 0 && f < a.F && vt[v] < a.T;
            const int ci = cir < a.Cin ? cir : a.Cin - 1;
            const float* src = chan_ptr(ci);
            const long off = ok ? (long)f * a.T + vt[v] : 0;
            xv[v] = *reinterpret_cast<const f32x4*>(src + off);
            vsc[v] = sel_scale(has_isc, isc[b * a.Cin + ci]);
            vok[v] = ok;
        }
        if (hactive) {
            const int cir = ci0 + hci;
            const int f = f0 + hrow + foff;
            const bool ok = cir < a.Cin && f >= 0 && f < a.F && ht >= 0 && ht < a.T;
            const int ci = cir < a.Cin ? cir : a.Cin - 1;
            const float* src = chan_ptr(ci);
            xh = src[ok ? (long)f * a.T + ht : 0];
            hsc = sel_scale(has_isc, isc[b * a.Cin + ci]);
            hok = ok;
        }
    };
    auto store_x_vec = [&](float* buf) {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            f32x4 o = {0.f, 0.f, 0.f, 0.f};
            if (vok[v]) o = xv[v] * vsc[v];
            *reinterpret_cast<f32x4*>(buf + vlds[v]) = o;
        }
        if (hactive) buf[hlds] = hok ? xh * hsc : 0.f;
    };
    auto load_chunk = [&](int kh, int ci0) {
        const int foff = (kh - khc) * a.dil;
        if constexpr (VEC) {
            load_x_vec(kh, ci0);
        } else {
        // wave-uniform per-channel base pointers and scales (scalar registers / scalar loads)
        const float* srcj[CPT];
#pragma unroll
        for (int j = 0; j < CPT; ++j) {
            const int cir = ci0 + cgrp + j;
            cok[j] = cir < a.Cin;
            const int ci = cok[j] ? cir : a.Cin - 1;          // padded channels read a valid address
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
            for (int j = 0; j < CPT; ++j) xr[s2][j] = srcj[j][off];      // raw; masked + scaled at store time
        }
        }
#pragma unroll
        for (int jj = 0; jj < WJ; ++jj) {
            int idx = tid + jj * 256;
            if (idx > NWV - 1) idx = NWV - 1;
            const int row = idx / V4;               // kw*KC + ci_l
            const int c4 = idx - row * V4;
            const int kw = row / KC;
            const int ci_l = row - kw * KC;
            wr[jj] = *reinterpret_cast<const f32x4*>(
                a.w_packed + ((long)((kh * KW + kw) * g.CinP + ci0 + ci_l)) * g.CoutP + co0 + c4 * 4);
        }
    };
    auto store_chunk = [&](float* buf) {
        if constexpr (VEC) {
            store_x_vec(buf);
        } else {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
            if (sv[s2]) {
#pragma unroll
                for (int j = 0; j < CPT; ++j)
                    buf[(cgrp + j) * XCH + se[s2]] = (okm[s2] && cok[j]) ? xr[s2][j] * scj[j] : 0.f;
            }
        }
#pragma unroll
        for (int jj = 0; jj < WJ; ++jj) {
            const int idx = tid + jj * 256;
            if (idx < NWV) *reinterpret_cast<f32x4*>(buf + XBUF + idx * 4) = wr[jj];
        }
    };

    int kh = 0;
    while (!kh_valid(kh)) ++kh;                 // kh = KH/2 is always valid
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
        if (has_next) load_chunk(nkh, nci);      // global loads stay in flight under the MFMAs below
        const float* Xs = smem + cur * BUF;
        // operand registers are prefetched one k-pair ahead of the MFMAs that consume them
        float av[2][NT], bv[2][WP];
        AVec<NT>::ld(Xs + aoff, av[0]);
#pragma unroll
        for (int wp = 0; wp < WP; ++wp) bv[0][wp] = Xs[boff[wp]];
#pragma unroll
        for (int st = 0; st < NSTEP; ++st) {
            const int c = st & 1;
            if (st + 1 < NSTEP) {
                const int kw = (st + 1) / (KC / 2), q = (st + 1) % (KC / 2);
                AVec<NT>::ld(Xs + aoff + (kw * KC + 2 * q) * BN, av[c ^ 1]);
#pragma unroll
                for (int wp = 0; wp < WP; ++wp) bv[c ^ 1][wp] = Xs[2 * q * XCH + boff[wp] + kw];
            }
            // keep the LDS reads of the NEXT k-pair ahead of this k-pair's MFMAs (hipcc otherwise sinks them
            // behind the MFMAs into the same registers and waits lgkmcnt(0) with nothing to overlap)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int wp = 0; wp < WP; ++wp)
                    acc[nt][wp] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c][nt], bv[c][wp], acc[nt][wp], 0, 0, 0);
        }
        if (has_next) store_chunk(smem + (cur ^ 1) * BUF);
        __syncthreads();
        if (!has_next) break;
        kh = nkh;
        ci0 = nci;
        cur ^= 1;
    }

    // ---- epilogue
    conv_epilogue<NT, WP>(a, acc, b, co0, f0, t0, g.pt_log2, wave, l31, h);
}

inline int pick_nt(int CoutP) {
    const int n32 = CoutP / 32;
    for (int c = 4; c >= 1; --c)
        if (n32 % c == 0) return c;
    return 1;
}

__global__ void pack_weights_kernel(const float* __restrict__ w, float* __restrict__ dst, int Cout, int Cin, int KH,
                                    int KW, int tf, int CinP, int CoutP, long total, int NT) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    // stored position -> logical output channel (inverse of the in-tile permutation co0 + l*NT + nt)
    const int cs = (int)(i % CoutP);
    const int BN = NT * 32;
    const int loc = cs % BN;
    const int co = cs - loc + (loc % NT) * 32 + loc / NT;
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

template <int NT, int WP, int KW, bool VEC>
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
    size_t lds = 2 * ((size_t)((KC * PR * (PT + (VEC ? 8 : 2)) + 3) & ~3) + (size_t)KW * KC * NT * 32) * sizeof(float);
    hipLaunchKernelGGL((conv_mfma_kernel<NT, WP, KC, KW, VEC>), grid, dim3(256), lds, s, a, g);
    return 0;
}

}  // namespace

extern "C" long babe_conv_packed_size(int Cout, int Cin, int KH, int KW, int transpose_flip) {
    const int co = transpose_flip ? Cin : Cout;
    const int ci = transpose_flip ? Cout : Cin;
    return (long)KH * KW * ((ci + 7) / 8 * 8) * ((co + 31) / 32 * 32);
}

extern "C" int babe_conv_pack_weights(const float* w, float* dst, int Cout, int Cin, int KH, int KW,
                                      int transpose_flip, void* stream) {
    return babe_conv_pack_weights_nt(w, dst, Cout, Cin, KH, KW, transpose_flip, 0, stream);
}

extern "C" int babe_conv_pack_weights_nt(const float* w, float* dst, int Cout, int Cin, int KH, int KW,
                                         int transpose_flip, int nt, void* stream) {
    BABE_CHECK_ARG(w && dst && Cout > 0 && Cin > 0 && KH > 0 && KW > 0, "conv_pack_weights: bad arguments");
    const int co = transpose_flip ? Cin : Cout;
    const int ci = transpose_flip ? Cout : Cin;
    const int CinP = (ci + 7) / 8 * 8, CoutP = (co + 31) / 32 * 32;
    const long total = (long)KH * KW * CinP * CoutP;
    BABE_CHECK_ARG(nt == 0 || (nt >= 1 && nt <= 4 && (CoutP / 32) % nt == 0), "conv_pack_weights_nt: nt=%d does not divide %d row tiles", nt, CoutP / 32);
    hipLaunchKernelGGL(pack_weights_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, w, dst, Cout,
                       Cin, KH, KW, transpose_flip, CinP, CoutP, total, nt > 0 ? nt : pick_nt(CoutP));
    BABE_LAUNCH_CHECK();
    return BABE_OK;
}

int babe_conv11p_supported(const babe_conv_args& a, int nt);                 // conv11p.hip
int babe_conv11p_launch(const babe_conv_args& a, int nt, hipStream_t s);

extern "C" int babe_conv2d(const babe_conv_args* ap, void* stream) { return babe_conv2d_nt(ap, 0, stream); }

extern "C" int babe_conv2d_nt(const babe_conv_args* ap, int nt, void* stream) {
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
    const int NT = nt > 0 ? nt : pick_nt(g.CoutP);
    BABE_CHECK_ARG(NT >= 1 && NT <= 4 && n32 % NT == 0, "conv2d_nt: nt=%d does not divide %d row tiles", NT, n32);
    // positions per block: measured on MI355X (tools/conv_shapes_bench.py) 128-position blocks (WP=1, 3-4 blocks/CU)
    // beat 256-position blocks (2/CU) on every wide layer because partial last rounds are cheaper; the
    // 64-channel layers (NT<=2) run equally fast with either, so they keep the larger tile.
    const long npos = (long)a.F * a.T;
    const long blocks256 = ((npos + 255) / 256) * (n32 / NT) * a.B;
    bool wp2 = (NT <= 2) && blocks256 >= 1024;
    {   // debug override: BABE_CONV_WP=1|2 forces the positions-per-wave variant
        static const char* ov = getenv("BABE_CONV_WP");
        if (ov && ov[0] == '1') wp2 = false;
        if (ov && ov[0] == '2') wp2 = true;
    }
    hipStream_t s = (hipStream_t)stream;
    const double flops = babe_conv_flops(a);
    BabeProfScope prof(a.KH > 1 ? BABE_SLOT_CONV53_DIRECT : BABE_SLOT_CONV11, babe_conv_bytes(a), flops, flops, stream);
    if (babe_conv11p_supported(a, NT)) {                    // pipelined (1,1) kernel (conv11p.hip)
        babe_conv11p_launch(a, NT, s);
        BABE_LAUNCH_CHECK();
        return BABE_OK;
    }
    // vector staging needs 16-byte aligned rows in every source tensor
    auto al16 = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
    bool vec = (a.T % 4 == 0) && al16(a.in) && (a.in_bs % 4 == 0) && (a.in_cs % 4 == 0) &&
               (!a.in2 || (al16(a.in2) && a.in2_bs % 4 == 0 && a.in2_cs % 4 == 0));
    {
        static const char* ov = getenv("BABE_CONV_VEC");
        if (ov && ov[0] == '0') vec = false;
    }
#define BABE_CONV_CASE(NTv, WPv)                                          \
    if (a.KW == 3) {                                                      \
        if (vec) launch_conv<NTv, WPv, 3, true>(a, g, s);                 \
        else launch_conv<NTv, WPv, 3, false>(a, g, s);                    \
    } else {                                                              \
        if (vec) launch_conv<NTv, WPv, 1, true>(a, g, s);                 \
        else launch_conv<NTv, WPv, 1, false>(a, g, s);                    \
    }
    if (wp2) {
        switch (NT) {
            case 4: BABE_CONV_CASE(4, 2) break;
            case 3: BABE_CONV_CASE(3, 2) break;
            case 2: BABE_CONV_CASE(2, 2) break;
            default: BABE_CONV_CASE(1, 2) break;
        }
    } else {
        switch (NT) {
            case 4: BABE_CONV_CASE(4, 1) break;
            case 3: BABE_CONV_CASE(3, 1) break;
            case 2: BABE_CONV_CASE(2, 1) break;
            default: BABE_CONV_CASE(1, 1) break;
        }
    }
#undef BABE_CONV_CASE
    BABE_LAUNCH_CHECK();
    return BABE_OK;
}
