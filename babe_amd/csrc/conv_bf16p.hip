// bf16-MFMA (5,3) dilated conv, PIPELINED (round 2): the kernel behind precision='bf16' for the configurations that the
// reference runs under bf16 autocast (BASELINE.json configs[2]-[4]).  Same contract, packed weights and epilogue as
// conv_bf16.hip (direct convolution, v_mfma_f32_32x32x16_bf16, fp32 accumulate, activations fp32 in HBM and rounded to
// bf16 on their way into LDS); what changes is how operands travel, because at bf16 rates the round-1 kernel was bound
// by its staging code (one dword load per thread, position and channel):
//
//   * 512 threads, one workgroup per CU, tile = 128 (or 64) output channels x 512 positions, K-slab = 32 input channels
//     of one frequency tap: 48 (24) MFMAs per wave between barriers, 0.75 (1.0) ds_read_b128 per MFMA.
//   * activations through raw BUFFER loads, 16 bytes = 4 time positions of one channel per lane; zero padding (rows,
//     columns, channels beyond Cin) is the hardware range check.  A thread owns 8 channels x 4 positions, i.e. four
//     complete 16-byte MFMA operand units; the two halo columns of a row are eight dword loads on a few lanes, predicated
//     by the same range check (idle lanes load nothing and write a dummy LDS slot: no branches in the slab body).
//   * LDS image [channel group][unit] with unit u stored at u ^ ((u >> 4) & 3): a thread's four units are 64 bytes apart
//     from its neighbour's, which would be a 4-way bank conflict on the ds_write_b128; the XOR makes the writes of 16
//     consecutive lanes hit 16 distinct 16-byte bank groups and keeps the reads (32 consecutive units) conflict-free.
//   * weights by LDS-DMA through a buffer descriptor (per-thread offsets are loop constants, the slab a scalar offset).
//   * two LDS buffers; slab j+1 is converted and written, and the raw loads of slab j+2 are issued, in the shadow of
//     the MFMAs of slab j.
//
// Requirements beyond conv_bf16.hip's: KW == 3, T % 4 == 0, Cin % 8 == 0, cin_split % 32 == 0, every source view below
// 2 GiB per batch item, Cout > 32.  Anything else (and bf16x3) runs on the round-1 kernel.
#include "conv_common.h"
#include "prof.h"
#include <cstdlib>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

namespace {

struct Bf16pGeom {
    int GP, CoutP, pt_log2, pr_log2, tiles_t;
};

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
constexpr unsigned OOB = 0x80000000u;      // beyond every descriptor's num_records: the load returns 0, touches nothing

__device__ __forceinline__ int swz(int u) { return u ^ ((u >> 4) & 3); }

template <int NT, int NP, int WR, int WC, bool HAS_ISC>
__global__ __launch_bounds__(512, 1) void conv_bf16p_kernel(babe_conv_args a, Bf16pGeom g,
                                                            const unsigned short* __restrict__ wq) {
#if __HIP_DEVICE_COMPILE__      // buffer-descriptor builtins exist in the device pass only
    static_assert(WR * WC == 8, "eight waves");
    constexpr int NTH = 512, KC = 32, G = 4;
    constexpr int BN = WR * NT * 32;
    constexpr int NPOS = WC * NP * 32;
    static_assert(NPOS == 512, "one staging quad per thread and channel group");
    constexpr int XCHP = NPOS + NPOS / 8;             // units per channel group: PR rows of PT+2, PR <= NPOS/16
    constexpr int XSZ = G * XCHP;
    constexpr int NWU = 3 * G * BN;                   // weight units per slab
    constexpr int WJ = (NWU + NTH - 1) / NTH;
    constexpr int BUF = XSZ + WJ * NTH + 4;           // + the dummy slot idle halo lanes write
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16x8* smem = reinterpret_cast<bf16x8*>(smem_raw);

    const int PT = 1 << g.pt_log2;
    const int PR = 1 << g.pr_log2;
    const int XROW = PT + 2;
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
    const int split = a.in2 ? a.cin_split : a.Cin;

    int kh_lo = 0, kh_hi = a.KH - 1;
    while ((f0 + (kh_lo - khc) * a.dil + PR <= 0 || f0 + (kh_lo - khc) * a.dil >= a.F) && kh_lo < kh_hi) ++kh_lo;
    while ((f0 + (kh_hi - khc) * a.dil + PR <= 0 || f0 + (kh_hi - khc) * a.dil >= a.F) && kh_hi > kh_lo) --kh_hi;
    const int CinP = (a.Cin + KC - 1) / KC * KC;
    const int nslab = (kh_hi - kh_lo + 1) * (CinP / KC);

    // ---- descriptors (wave-uniform: kernargs and block indices only)
    const float* p1 = a.in + (long)b * a.in_bs;
    const float* p2 = a.in2 ? a.in2 + (long)b * a.in2_bs : p1;
    const int cs1 = (int)a.in_cs, cs2 = a.in2 ? (int)a.in2_cs : (int)a.in_cs;
    const int nb1 = split * cs1 * 4, nb2 = (a.Cin - split) * cs2 * 4;
    const __amdgpu_buffer_rsrc_t rsw =
        __builtin_amdgcn_make_buffer_rsrc((void*)wq, 0, a.KH * 3 * g.GP * g.CoutP * 16, 0x00020000);

    // ---- staging constants.  Channel group = tid / 128 (wave-uniform), inside it 128 quads of 4 time positions.
    const int gw = __builtin_amdgcn_readfirstlane(tid >> 7);
    const int lt = tid & 127;
    const int i4 = lt & ((PT >> 2) - 1);
    const int srow = lt >> (g.pt_log2 - 2);
    const int st = t0 + 4 * i4;
    const int xspat4 = ((f0 + srow) * a.T + st) * 4;
    const unsigned xcolbad = st < a.T ? 0u : OOB;
    const int u0 = srow * XROW + 1 + 4 * i4;
    // halo: lanes lt < 2*PR of every group own (row, side) = (lt >> 1, lt & 1)
    const bool hl = lt < 2 * PR;
    const int hrow = lt >> 1;
    const int ht = (lt & 1) ? t0 + PT : t0 - 1;
    const int hspat4 = ((f0 + hrow) * a.T + ht) * 4;
    const unsigned hbad = (hl && ht >= 0 && ht < a.T) ? 0u : OOB;
    const int hslot = hl ? gw * XCHP + swz(hrow * XROW + ((lt & 1) ? PT + 1 : 0)) : XSZ + WJ * NTH;
    int wvo[WJ];
#pragma unroll
    for (int jj = 0; jj < WJ; ++jj) {
        const int idx = tid + jj * NTH;
        const int kw = idx / (G * BN);
        const int rem = idx - kw * (G * BN);
        const int gl = rem / BN;
        const int co_l = rem - gl * BN;
        wvo[jj] = (int)((unsigned)((((kw * g.GP + gl) * g.CoutP) + co0 + co_l) * 16) | (idx < NWU ? 0u : OOB));
    }

    f32x16 acc[NT][NP];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][p][r] = 0.f;

    // staging registers (ONE set): raw loads of the slab that is next to be converted
    f32x4 xv[8];
    float xh[8], xsc[8];

#ifndef ABL
#define ABL 0
#endif
    auto issue_act = [&](int kh, int ci0) {
        if (ABL == 1) return;
        const int foff = (kh - khc) * a.dil;
        const bool s2 = ci0 >= split;
        const int nch = s2 ? a.Cin - split : split;
        const int cl = (s2 ? ci0 - split : ci0) + 8 * gw;               // first channel of this group in its source
        const int cs = s2 ? cs2 : cs1;
        const __amdgpu_buffer_rsrc_t rs =
            __builtin_amdgcn_make_buffer_rsrc((void*)(s2 ? p2 : p1), 0, s2 ? nb2 : nb1, 0x00020000);
        const unsigned gbad = cl < nch ? 0u : OOB;                         // scalar: a group beyond Cin is all zeros
        const int sb4 = cl * cs * 4;                                       // scalar byte offset of channel j = 0 (soffset is unsigned)
        const int fo4 = foff * a.T * 4;                                    // the tap's row shift goes into the lane offset
        const int fr = f0 + srow + foff, fh = f0 + hrow + foff;
        const unsigned e = (unsigned)(xspat4 + fo4) | ((unsigned)(fr | (a.F - 1 - fr)) & OOB) | xcolbad | gbad;
        const unsigned eh = (unsigned)(hspat4 + fo4) | ((unsigned)(fh | (a.F - 1 - fh)) & OOB) | hbad | gbad;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            xv[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, e, gbad ? 0 : sb4 + j * cs * 4, 0));
#pragma unroll
        for (int j = 0; j < 8; ++j)
            xh[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, eh, gbad ? 0 : sb4 + j * cs * 4, 0));
        if (HAS_ISC) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                int ci = ci0 + 8 * gw + j;
                ci = ci < a.Cin ? ci : a.Cin - 1;
                xsc[j] = a.in_scale[(long)b * a.Cin + ci];                 // scalar loads (wave-uniform address)
            }
        }
    };
    auto pack8 = [&](const float (&v)[8]) {
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (__bf16)(HAS_ISC ? v[j] * xsc[j] : v[j]);
        return o;
    };
    auto store_main = [&](bf16x8* buf, int k0, int k1) {
#pragma unroll
        for (int k = k0; k < k1; ++k) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = xv[j][k];
            buf[gw * XCHP + swz(u0 + k)] = pack8(v);
        }
    };
    auto store_halo = [&](bf16x8* buf) { buf[hslot] = pack8(xh); };
    auto dma_w = [&](int kh, int ci0, bf16x8* buf) {
        const int so = ((kh * 3) * g.GP + (ci0 >> 3)) * g.CoutP * 16;      // bytes, scalar
#pragma unroll
        for (int jj = 0; jj < WJ; ++jj)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, LDS_PTR(buf + XSZ + jj * NTH + wave * 64), 16, wvo[jj], so, 0, 0);
    };
    auto advance = [&](int& kh, int& ci0) {             // next slab, clamped at the last one
        int nc = ci0 + KC, nk = kh;
        if (nc >= CinP) {
            nc = 0;
            ++nk;
        }
        if (nk <= kh_hi) {
            kh = nk;
            ci0 = nc;
        }
    };

    // operand addresses (units).  B: position p of tap kw is unit row*XROW + tt + kw of channel group 2*gp + h.
    int boff[NP][3];
#pragma unroll
    for (int np = 0; np < NP; ++np) {
        const int p = (wc * NP + np) * 32 + l31;
        const int u = (p >> g.pt_log2) * XROW + (p & (PT - 1));
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) boff[np][kw] = h * XCHP + swz(u + kw);
    }
    const int aoff = XSZ + h * BN + wr * (NT * 32) + l31;

    // ---- prologue: slab 0 into buffer 0, raw loads of slab 1 in flight
    int kA = kh_lo, cA = 0;
    issue_act(kA, cA);
    dma_w(kA, cA, smem);
    store_main(smem, 0, 4);
    store_halo(smem);
    advance(kA, cA);
    issue_act(kA, cA);
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");      // the weight DMA of slab 0 (older than the 16 raw loads of slab 1)
    __syncthreads();

    // MFMA schedule of a slab: six K-steps (channel-group pair gp = s / 3, time tap kw = s % 3), each in two halves of
    // NP/2 position tiles.  A operands are double-buffered per step, B operands per half-step: the reads of the next
    // half-step are issued before the MFMAs of the current one (NT*NP/2 MFMAs >= 128 cycles cover the LDS latency).
    constexpr int NH = NP / 2;
    bf16x8 av[2][NT], bv[2][NH];
    int cur = 0;
    for (int j = 0; j < nslab; ++j) {
        const bf16x8* Xs = smem + cur * BUF;
        bf16x8* Xw = smem + (cur ^ 1) * BUF;
#define READ_A(c, s) \
    _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) av[c][nt] = Xs[aoff + (((s) % 3) * G + 2 * ((s) / 3)) * BN + nt * 32];
#define READ_B(d, s, hf) \
    _Pragma("unroll") for (int q = 0; q < NH; ++q) bv[d][q] = Xs[boff[(hf) * NH + q][(s) % 3] + 2 * ((s) / 3) * XCHP];
#define MFMA_HALF(c, d, hf)                                                                                  \
    if (ABL != 2) _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) _Pragma("unroll") for (int q = 0; q < NH; ++q)         \
        acc[nt][(hf) * NH + q] =                                                                             \
            __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[c][nt], bv[d][q], acc[nt][(hf) * NH + q], 0, 0, 0);  \
    __builtin_amdgcn_sched_barrier(0);
        READ_A(0, 0)
        READ_B(0, 0, 0)
        dma_w(kA, cA, Xw);                         // weights of slab j+1: issued BEFORE this slab's raw loads (see the wait below)
        // step 0
        READ_B(1, 0, 1) READ_A(1, 1)
        store_main(Xw, 0, 1);
        MFMA_HALF(0, 0, 0)
        READ_B(0, 1, 0)
        store_main(Xw, 1, 2);
        MFMA_HALF(0, 1, 1)
        // step 1
        READ_B(1, 1, 1) READ_A(0, 2)
        store_main(Xw, 2, 3);
        MFMA_HALF(1, 0, 0)
        READ_B(0, 2, 0)
        store_main(Xw, 3, 4);
        store_halo(Xw);
        MFMA_HALF(1, 1, 1)
        // step 2: the staging registers are free again -> raw loads of slab j+2
        READ_B(1, 2, 1) READ_A(1, 3)
        advance(kA, cA);
        issue_act(kA, cA);
        MFMA_HALF(0, 0, 0)
        READ_B(0, 3, 0)
        MFMA_HALF(0, 1, 1)
        // step 3
        READ_B(1, 3, 1) READ_A(0, 4)
        MFMA_HALF(1, 0, 0)
        READ_B(0, 4, 0)
        MFMA_HALF(1, 1, 1)
        // step 4
        READ_B(1, 4, 1) READ_A(1, 5)
        MFMA_HALF(0, 0, 0)
        READ_B(0, 5, 0)
        MFMA_HALF(0, 1, 1)
        // step 5
        READ_B(1, 5, 1)
        MFMA_HALF(1, 0, 0)
        MFMA_HALF(1, 1, 1)
        // Vector-memory operations retire in order: WJ weight DMAs, then this slab's 16 raw loads.  At most 16
        // outstanding = this wave's share of the weight slab is in LDS; the barrier then publishes it to the other waves.
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        __syncthreads();
        cur ^= 1;
    }
#undef READ_A
#undef READ_B
#undef MFMA_HALF

    conv_epilogue<NT, NP>(a, acc, b, co0 + wr * (NT * 32), f0, t0, g.pt_log2, wc, l31, h);
#endif
}

inline int ilog2_ceil_b(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}

template <int NT, int NP, int WR, int WC>
void launch_bf16p(const babe_conv_args& a, Bf16pGeom g, const unsigned short* wq, hipStream_t s) {
    constexpr int BN = WR * NT * 32;
    g.pt_log2 = ilog2_ceil_b(a.T);
    if (g.pt_log2 > 9) g.pt_log2 = 9;
    if (g.pt_log2 < 4) g.pt_log2 = 4;
    g.pr_log2 = 9 - g.pt_log2;
    const int PT = 1 << g.pt_log2, PR = 1 << g.pr_log2;
    g.tiles_t = cdiv(a.T, PT);
    const int tiles_f = cdiv(a.F, PR);
    dim3 grid(g.tiles_t * tiles_f, cdiv(g.CoutP, BN), a.B);
    constexpr int WJ = (3 * 4 * BN + 511) / 512;
    const size_t lds = 2 * (size_t)(4 * 576 + WJ * 512 + 4) * 16;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_bf16p_kernel<NT, NP, WR, WC, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_bf16p_kernel<NT, NP, WR, WC, false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    if (a.in_scale)
        hipLaunchKernelGGL((conv_bf16p_kernel<NT, NP, WR, WC, true>), grid, dim3(512), lds, s, a, g, wq);
    else
        hipLaunchKernelGGL((conv_bf16p_kernel<NT, NP, WR, WC, false>), grid, dim3(512), lds, s, a, g, wq);
}

}  // namespace

/* 1 if the pipelined bf16 kernel takes this problem (splits == 1 only) */
int babe_conv2d_bf16p_supported(const babe_conv_args& a) {
    static const char* ov = getenv("BABE_CONV_BF16P");
    if (ov && ov[0] == '0') return 0;
    if (a.KW != 3 || (a.T & 3) || (a.Cin & 7) || a.Cout <= 32) return 0;
    if (a.in2 && (a.cin_split % 32 != 0)) return 0;
    const long lim = 0x7fffffffL / 4;
    const int split = a.in2 ? a.cin_split : a.Cin;
    if ((long)split * a.in_cs >= lim) return 0;
    if (a.in2 && (long)(a.Cin - split) * a.in2_cs >= lim) return 0;
    if ((long)a.KH * 3 * ((a.Cin + 15) / 16 * 2) * ((a.Cout + 31) / 32 * 32) * 16 >= 0x7fffffffL) return 0;
    return 1;
}

int babe_conv2d_bf16p_launch(const babe_conv_args& a, const unsigned short* wq, hipStream_t s) {
    Bf16pGeom g;
    g.GP = (a.Cin + 15) / 16 * 2;
    g.CoutP = (a.Cout + 31) / 32 * 32;
    if (g.CoutP == 64) launch_bf16p<2, 2, 1, 8>(a, g, wq, s);      //  64 co x 512 positions
    else launch_bf16p<2, 4, 2, 4>(a, g, wq, s);                    // 128 co x 512 positions
    return 0;
}
