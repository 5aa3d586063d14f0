// (1,1) Conv2d (proj_in / proj_out / res_conv and the k=(1,1) init / out ResnetBlocks of networks/cqtdiff+.py:412-415,
// 675, 690, 719; the same with transposed weights for the input-VJP) as a PIPELINED fp32-MFMA GEMM, round 2.
//
// These convs are 8.7 % of the benchmark's kernel time and sit between the two roofs: 2*Cin*Cout flops per position
// against 4*(Cin+Cout) bytes is 16-40 flop/B for the UNet's channel counts, i.e. 40-60 us of HBM time and about as much
// fp32-MFMA time per launch, where the generic direct kernel (8-channel slabs, one barrier each, register staging) took
// 80-240 us (2.2-2.8 TB/s algorithmic).  A (1,1) conv needs no halo, no taps and no transform, so BOTH operands travel
// global -> LDS by LDS-DMA through buffer descriptors (zero padding of ragged tiles / padded channels = the hardware range
// check) with no staging registers at all:
//   workgroup = 4 waves, 256 positions x BN = NT*32 output channels (the tile the weights were packed for);
//   K-slab = 16 input channels: X [16][256] floats + W [16][BN] floats, ring of THREE buffers, slab j+2 in flight while
//   slab j is multiplied, first operands of slab j+1 read before the barrier that ends slab j (as conv_wino4p.hip);
//   wave = 64 positions (2 MFMA column tiles) x all NT row tiles: per K-step NT floats of A (one 4/8/16-byte LDS read) and
//   2 of B feed 2*NT v_mfma_f32_32x32x2_f32; in_scale (the VJP's gate) multiplies the A fragment.
// Packed weights and epilogue are those of conv.hip (conv_common.h).  Requirements: KH = KW = 1, T % 4 == 0, 16-byte
// aligned views, cin_split % 16 == 0, views below 2 GiB; anything else stays on the generic kernel.
#include "common.h"
#include "../../include/babe_hip.h"
#include "prof.h"
#include "conv_common.h"
#include <cstdlib>

#ifndef C11_SYNC_AT_END
#define C11_SYNC_AT_END 0          // 1: the round-2 form (one __syncthreads() = vmcnt(0) + barrier at the end of every slab)
#endif

namespace {

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
constexpr unsigned OOB11 = 0x80000000u;

// LDS-DMA as inline asm.  With the builtin, hipcc's wait insertion makes every LDS read wait for ALL pending LDS-DMA (no alias scopes
// in LDS), i.e. vmcnt(0) in front of the first operand read after a slab's DMA was issued: the prefetch the ring exists for was
// waited for at once (rounds 2-3 ran like that; only the CU's second workgroup hid the latency).  The asm form is invisible to
// that pass; completion is the kernel's own counted s_waitcnt (see the K loop).  rs = {base lo, base hi, bytes, 0x00020000}.
typedef int c11_i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ c11_i32x4 c11_rsrc(const void* p, unsigned bytes) {
    const unsigned long a = (unsigned long)p;
    return c11_i32x4{(int)__builtin_amdgcn_readfirstlane((unsigned)a), (int)__builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xffffu),
                     (int)__builtin_amdgcn_readfirstlane(bytes), 0x00020000};
}
__device__ __forceinline__ void c11_dma16(c11_i32x4 rs, const float* lds_dst, unsigned voff) {
#if __HIP_DEVICE_COMPILE__
    const unsigned la = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)LDS_PTR(lds_dst));
    // (m0 is clobbered: without saying so hipcc may keep a live value there across the statement - movrel indexing, readlane,
    // a builtin LDS-DMA elsewhere in the kernel)
    asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(la), "v"(voff), "s"(rs) : "memory", "m0");
#endif
}

struct C11Geom {
    int CinP, CoutP, pt_log2, pr_log2, tiles_t;
};

// Epilogue through buffer descriptors: out = alpha*acc*oscale[b,co] + rbeta*res.  One per-lane byte offset per position
// (out of range when the position is padding) + one scalar term per output channel; a channel beyond Cout lands beyond the
// descriptor's size, so the hardware range check drops it - no per-element branches, no 64-bit address arithmetic, and
// none of the 132-1228 bytes per lane of scratch the pointer-based shared epilogue (conv_common.h) needed at 256 registers.
// With four K-slabs per tile (the 64-channel full-resolution layers) the epilogue is a quarter of the kernel.
template <int NT, int WP>
__device__ __forceinline__ void conv11p_epilogue(const babe_conv_args& a, f32x16 (&acc)[NT][WP], int b, int co0, int f0,
                                                 int t0, int pt_log2, int wave, int l31, int h) {
#if __HIP_DEVICE_COMPILE__
    const int PT = 1 << pt_log2;
    const bool has_os = a.oscale != nullptr, has_res = a.res != nullptr;
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)(a.out + (long)b * a.out_bs), 0, (unsigned)(a.Cout * a.out_cs * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rr_ = __builtin_amdgcn_make_buffer_rsrc((void*)(has_res ? a.res + (long)b * a.res_bs : a.out), 0,
                                                                         has_res ? (unsigned)(a.Cout * a.res_cs * 4) : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_ = __builtin_amdgcn_make_buffer_rsrc((void*)(has_os ? a.oscale + (long)b * a.Cout : a.out), 0,
                                                                         has_os ? (unsigned)(a.Cout * 4) : 0u, 0x00020000);
    const unsigned ocs = (unsigned)a.out_cs * 4u, rcs = (unsigned)a.res_cs * 4u;
    // per-position byte offsets of the WP column tiles (out of range when the position is padding)
    unsigned lo[WP], lr[WP];
#pragma unroll
    for (int wp = 0; wp < WP; ++wp) {
        const int p = (wave * WP + wp) * 32 + l31;
        const int f = f0 + (p >> pt_log2);
        const int t = t0 + (p & (PT - 1));
        const bool pv = f < a.F && t < a.T;
        const unsigned sp = (unsigned)(f * a.T + t) * 4u;
        lo[wp] = pv ? (unsigned)(co0 + 4 * h) * ocs + sp : 0x80000000u;
        lr[wp] = pv ? (unsigned)(co0 + 4 * h) * rcs + sp : 0x80000000u;
    }
    // The residual of group g + 1 (one (row tile, column tile) pair = 16 values per lane) is loaded while group g is scaled and
    // stored: two groups of loads in flight per wave instead of one (the res-carrying layers - every VJP - are bound by this
    // epilogue: 16 loads x 256 B per wave in flight is about 4.4 TB/s over the chip, which is where they sat).
    const float os_m = has_os ? a.alpha : 0.f, os_a = has_os ? 0.f : a.alpha;
    float rv[2][16];
    auto load_res = [&](int gidx, float (&dst)[16]) {
        const int nt = gidx / WP, wp = gidx % WP;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int cl = nt * 32 + (r & 3) + 8 * (r >> 2);
            dst[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rr_, lr[wp] + (unsigned)cl * rcs, 0, 0));
        }
    };
    if (has_res) load_res(0, rv[0]);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        // output scales of this lane's 16 channels (channel = co0 + nt*32 + (r & 3) + 8*(r >> 2) + 4*h), once per row tile
        float os[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int cl = nt * 32 + (r & 3) + 8 * (r >> 2);
            // (unconditional: without output scales the descriptor has size 0 and the load returns 0.  Under `has_os ? load : alpha`
            // the compiler put each of the 16 loads in its own branch with its own s_waitcnt vmcnt(0))
            const float sv = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_, (unsigned)(co0 + 4 * h + cl) * 4u, 0, 0));
            os[r] = __builtin_fmaf(sv, os_m, os_a);     // = sv * alpha with output scales, alpha without
        }
#pragma unroll
        for (int wp = 0; wp < WP; ++wp) {
            const int gidx = nt * WP + wp;
            if (has_res && gidx + 1 < NT * WP) load_res(gidx + 1, rv[(gidx + 1) & 1]);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int cl = nt * 32 + (r & 3) + 8 * (r >> 2);
                // (explicit rounding points: left to -ffp-contract the 128- and 256-position instantiations fused different
                // pairs of these three operations and gave results one ulp apart for the same input)
                float v = __fmul_rn(acc[nt][wp][r], os[r]);
                if (has_res) v = __builtin_fmaf(a.rbeta, rv[gidx & 1][r], v);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(v), ro, lo[wp] + (unsigned)cl * ocs, 0, 0);
            }
        }
    }
#endif
}

// NPW = 32-position MFMA column tiles per wave: 2 (256 positions per workgroup, 2 workgroups per CU) or 1 (128 positions,
// 3 per CU: finer tail quantisation for the launches with only a few hundred workgroups)
template <int NT, int NPW, bool HAS_ISC>
__global__ __launch_bounds__(256, NPW == 2 ? 2 : 3) void conv11p_kernel(babe_conv_args a, C11Geom g) {
#if __HIP_DEVICE_COMPILE__
    constexpr int KC = 16;
    constexpr int BN = NT * 32;
    constexpr int NPOS = 128 * NPW;
    constexpr int XF = KC * NPOS;                       // floats of X per slab
    constexpr int XJ = XF / 4 / 256;                    // DMA instructions per thread for X (4 or 2)
    constexpr int Q4 = NPOS / 4;                        // float4 per channel row
    constexpr int WF4 = KC * BN / 4;                    // float4 of W per slab
    constexpr int WJ = (WF4 + 255) / 256;
    constexpr int BUF = XF + WJ * 256 * 4;              // floats per ring slot
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int PT = 1 << g.pt_log2;
    const int tile_t = blockIdx.x % g.tiles_t;
    const int tile_f = blockIdx.x / g.tiles_t;
    const int t0 = tile_t << g.pt_log2;
    const int f0 = tile_f << g.pr_log2;
    const int co0 = blockIdx.y * BN;
    const int b = blockIdx.z;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5;
    const int l31 = lane & 31;
    const int split = a.in2 ? a.cin_split : a.Cin;

    const float* p1 = a.in + (long)b * a.in_bs;
    const float* p2 = a.in2 ? a.in2 + (long)b * a.in2_bs : p1;
    const int cs1 = (int)a.in_cs, cs2 = a.in2 ? (int)a.in2_cs : (int)a.in_cs;
    const int nb1 = split * cs1 * 4, nb2 = (a.Cin - split) * cs2 * 4;

    // per-thread DMA offsets: X piece v = float4 q of channel row ci_l;  W piece jj = float4 c4 of row ci_l
    int xoff1[XJ], xoff2[XJ];
    unsigned xbad[XJ];
#pragma unroll
    for (int v = 0; v < XJ; ++v) {
        const int idx = tid + v * 256;
        const int ci_l = idx / Q4;
        const int p = (idx % Q4) * 4;
        const int f = f0 + (p >> g.pt_log2), t = t0 + (p & (PT - 1));
        xbad[v] = (f < a.F && t < a.T) ? 0u : OOB11;
        xoff1[v] = (ci_l * cs1 + f * a.T + t) * 4;
        xoff2[v] = (ci_l * cs2 + f * a.T + t) * 4;
    }
    unsigned woff[WJ];
#pragma unroll
    for (int jj = 0; jj < WJ; ++jj) {
        const int idx = tid + jj * 256;
        const int row = idx / (BN / 4), c4 = idx - row * (BN / 4);
        woff[jj] = idx < WF4 ? (unsigned)((row * g.CoutP + co0 + c4 * 4) * 4) : OOB11;
    }
    const c11_i32x4 q1 = c11_rsrc(p1, (unsigned)nb1), q2 = c11_rsrc(p2, (unsigned)nb2), qw = c11_rsrc(a.w_packed, (unsigned)(g.CinP * g.CoutP * 4));
    auto dma_slab = [&](int ci0, float* buf) {
        const bool s2 = ci0 >= split;
        const c11_i32x4 rs = s2 ? q2 : q1;
        const int so = (s2 ? (ci0 - split) * cs2 : ci0 * cs1) * 4;
#pragma unroll
        for (int v = 0; v < XJ; ++v)
            c11_dma16(rs, buf + (v * 256 + wave * 64) * 4, (unsigned)((s2 ? xoff2[v] : xoff1[v]) + so) | xbad[v]);
#pragma unroll
        // (the whole offset is in the VGPR operand: the range check that zero-fills rows >= CinP covers only that one)
        for (int jj = 0; jj < WJ; ++jj)
            c11_dma16(qw, buf + XF + (jj * 256 + wave * 64) * 4, woff[jj] + (unsigned)(ci0 * g.CoutP * 4));
    };

    f32x16 acc[NT][NPW];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < NPW; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nslab = (g.CinP + KC - 1) / KC;
    const int boff = h * NPOS + wave * (32 * NPW) + l31;
    const int aoff = XF + h * BN + l31 * NT;
    const float* isp = HAS_ISC ? a.in_scale + (long)b * a.Cin : nullptr;

    // prologue: slabs 0 and 1 (a slab index beyond the last one is clamped: re-staged, never read)
    dma_slab(0, smem);
    dma_slab(nslab > 1 ? KC : 0, smem + BUF);
    float* const sc_lds = smem + 3 * BUF;                   // [nslab * KC] in_scale of this batch item (HAS_ISC)
    if (HAS_ISC)
        for (int c = tid; c < nslab * KC; c += 256) sc_lds[c] = isp[c < a.Cin ? c : a.Cin - 1];
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");     // (the asm DMAs are invisible to __syncthreads()' waits)

    // in_scale of this lane's k index for the 8 K-steps of a slab, loaded one slab ahead (a scalar load inside the K-step
    // would stall every step on its latency)
    float scur[KC / 2], snext[KC / 2];
    // (from an LDS copy of the batch item's Cin scales, made in the prologue: vector loads from global memory would sit in the
    // same queue as the slab DMA, and hipcc's wait for them would be a wait for the prefetch)
    auto load_scales = [&](int ci0, float* dst) {
#pragma unroll
        for (int st = 0; st < KC / 2; ++st) dst[st] = HAS_ISC ? sc_lds[ci0 + 2 * st + h] : 1.f;
    };
    if (HAS_ISC) load_scales(0, scur);

    float av[2][NT], bv[2][NPW];
    AVec<NT>::ld(smem + aoff, av[0]);
#pragma unroll
    for (int wp = 0; wp < NPW; ++wp) bv[0][wp] = smem[boff + 32 * wp];

    int rb = 0;
    for (int j = 0; j < nslab; ++j) {
        const int rn = rb == 2 ? 0 : rb + 1;
        const int rw = rn == 2 ? 0 : rn + 1;
        const float* Xs = smem + rb * BUF;
        const float* Xn = smem + rn * BUF;
        const int jw = j + 2 < nslab ? j + 2 : nslab - 1;
        // (the scale loads are OLDER than the slab's DMA in the vector-memory queue: waiting for them leaves the DMA in flight)
        if (HAS_ISC) load_scales((j + 1 < nslab ? j + 1 : j) * KC, snext);
        __builtin_amdgcn_sched_barrier(0);
        dma_slab(jw * KC, smem + rw * BUF);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int st = 0; st < KC / 2; ++st) {
            const int c = st & 1;
#if !C11_SYNC_AT_END
            if (st == KC / 2 - 1) {
                // COUNTED wait (round 4): the workgroup synchronises before the LAST K-step of slab j, and each wave waits only for
                // its part of slab j+1 (vmcnt(n) with n = the XJ + WJ DMA instructions of slab j+2, the youngest in the queue), so
                // slab j+2 stays in flight across the barrier: two slabs of prefetch distance instead of the one that
                // __syncthreads()' vmcnt(0) left.  Safe for the ring: the operands of this last step are already in registers
                // (read during step KC/2 - 2, complete at lgkmcnt(0)), so after the barrier nobody reads slab j's buffer again,
                // and the reads of slab j+1 below come after every wave's part of it has landed.
                asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(XJ + WJ) : "memory");
            }
#endif
            if (st + 1 < KC / 2) {
                AVec<NT>::ld(Xs + aoff + 2 * (st + 1) * BN, av[c ^ 1]);
#pragma unroll
                for (int wp = 0; wp < NPW; ++wp) bv[c ^ 1][wp] = Xs[boff + 2 * (st + 1) * NPOS + 32 * wp];
            } else {
                AVec<NT>::ld(Xn + aoff, av[c ^ 1]);          // first operands of slab j+1, before the barrier
#pragma unroll
                for (int wp = 0; wp < NPW; ++wp) bv[c ^ 1][wp] = Xn[boff + 32 * wp];
            }
            if (HAS_ISC) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) av[c][nt] *= scur[st];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int wp = 0; wp < NPW; ++wp)
                    acc[nt][wp] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c][nt], bv[c][wp], acc[nt][wp], 0, 0, 0);
        }
#if C11_SYNC_AT_END
        __syncthreads();                                   // slab j+2 landed (vmcnt(0)), slab j's buffer free
#endif
        if (HAS_ISC) {
#pragma unroll
            for (int st = 0; st < KC / 2; ++st) scur[st] = snext[st];
        }
        rb = rn;
    }
#if !C11_SYNC_AT_END
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the clamped re-stage of the last slab: nothing lands in LDS after the loop
#endif
    conv11p_epilogue<NT, NPW>(a, acc, b, co0, f0, t0, g.pt_log2, wave, l31, h);
#endif
}

template <int NT, int NPW>
void launch11(const babe_conv_args& a, C11Geom g, hipStream_t s) {
    constexpr int LOGP = NPW == 2 ? 8 : 7;
    g.pt_log2 = 0;
    while ((1 << g.pt_log2) < a.T && g.pt_log2 < LOGP) ++g.pt_log2;
    if (g.pt_log2 < 2) g.pt_log2 = 2;
    g.pr_log2 = LOGP - g.pt_log2;
    g.tiles_t = cdiv(a.T, 1 << g.pt_log2);
    const int tiles_f = cdiv(a.F, 1 << g.pr_log2);
    constexpr int BN = NT * 32;
    constexpr int WJ = (16 * BN / 4 + 255) / 256;
    const size_t lds = 3 * (size_t)(16 * 128 * NPW + WJ * 256 * 4) * 4 + (a.in_scale ? (size_t)((g.CinP + 15) / 16 * 16) * 4 : 0);
    dim3 grid(g.tiles_t * tiles_f, g.CoutP / BN, a.B);
    static std::atomic<unsigned long long> attr_done{0};
    if (babe_lds_optin(attr_done, {reinterpret_cast<const void*>(&conv11p_kernel<NT, NPW, true>),
                                   reinterpret_cast<const void*>(&conv11p_kernel<NT, NPW, false>)},
                       (int)(3 * (size_t)(16 * 128 * NPW + WJ * 256 * 4) * 4 + 2048 * 4)) != hipSuccess)     // (+ up to 2048 in_scale values)
        return;
    if (a.in_scale) hipLaunchKernelGGL((conv11p_kernel<NT, NPW, true>), grid, dim3(256), lds, s, a, g);
    else hipLaunchKernelGGL((conv11p_kernel<NT, NPW, false>), grid, dim3(256), lds, s, a, g);
}

}  // namespace

/* 1 if the pipelined (1,1) kernel takes this problem; nt = row tiles the weights were packed for */
int babe_conv11p_supported(const babe_conv_args& a, int nt) {
    static const char* ov = getenv("BABE_CONV11P");
    if (ov && ov[0] == '0') return 0;
    auto al16 = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
    if (a.KH != 1 || a.KW != 1 || a.T % 4 != 0 || nt < 1 || nt > 4) return 0;
    if (!al16(a.in) || a.in_bs % 4 || a.in_cs % 4) return 0;
    if (a.in2 && (!al16(a.in2) || a.in2_bs % 4 || a.in2_cs % 4 || a.cin_split % 16)) return 0;
    if (!al16(a.w_packed)) return 0;
    const long lim = 0x7fffffffL / 4;
    const int split = a.in2 ? a.cin_split : a.Cin;
    if ((long)split * a.in_cs >= lim || (a.in2 && (long)(a.Cin - split) * a.in2_cs >= lim)) return 0;
    if ((long)((a.Cin + 7) / 8 * 8) * ((a.Cout + 31) / 32 * 32) >= lim) return 0;
    // the epilogue addresses out / res of a batch item through buffer descriptors with 32-bit offsets
    const long coP = (a.Cout + 31) / 32 * 32;
    if (coP * a.out_cs >= lim || (a.res && coP * a.res_cs >= lim)) return 0;
    if (a.in_scale && a.Cin > 2032) return 0;                 // the LDS copy of the scales
    if ((long)a.F * a.T < 4096 && a.Cin < 256) return 0;     // tiny planes AND a short K loop: nothing to pipeline (the
                                                             // dense DFT stages, K ~ 2000 over a few hundred positions, qualify)
    return 1;
}

int babe_conv11p_launch(const babe_conv_args& a, int nt, hipStream_t s) {
    C11Geom g;
    g.CinP = (a.Cin + 7) / 8 * 8;
    g.CoutP = (a.Cout + 31) / 32 * 32;
    // Tile choice by tail quantisation: a launch costs about rounds x (resident workgroups per CU x positions per
    // workgroup); 128-position tiles run 3 per CU, 256-position tiles 2 per CU (measured: tools/conv_shapes_bench.py)
    const long cot = g.CoutP / (nt * 32);
    const long b128 = (((long)a.F * a.T + 127) / 128) * cot * a.B, b256 = (((long)a.F * a.T + 255) / 256) * cot * a.B;
    const long cost1 = ((b128 + 767) / 768) * 3, cost2 = ((b256 + 511) / 512) * 4;
    // Round 5: on the two-lane job the 128-position tiles win everywhere (2.441 / 2.440 vs 2.434 / 2.430 audio-sec/s with this cost
    // model, 2.424 / 2.422 with 256-position tiles everywhere; profiles/r05_f45_ablate.txt): the model prices a launch ALONE on the
    // GPU, and beside the other lane's kernels the smaller workgroups fill better.  BABE_CONV11P_NPW=2 / =a (the model) for A/B.
    static const char* ov = getenv("BABE_CONV11P_NPW");
    const bool big = ov ? (ov[0] == '2' || (ov[0] == 'a' && cost2 < cost1)) : false;
    switch (nt * 2 + (big ? 1 : 0)) {
        case 9: launch11<4, 2>(a, g, s); break;
        case 8: launch11<4, 1>(a, g, s); break;
        case 7: launch11<3, 2>(a, g, s); break;
        case 6: launch11<3, 1>(a, g, s); break;
        case 5: launch11<2, 2>(a, g, s); break;
        case 4: launch11<2, 1>(a, g, s); break;
        case 3: launch11<1, 2>(a, g, s); break;
        default: launch11<1, 1>(a, g, s); break;
    }
    return 0;
}
