// bf16-MFMA (5,3) dilated conv, PIPELINED (round 2): the kernel behind precision='bf16' for the configurations that the
// reference runs under bf16 autocast (BASELINE.json configs[2]-[4]).  Same contract, packed weights and arithmetic as
// conv_bf16.hip (direct convolution, v_mfma_f32_32x32x16_bf16, fp32 accumulate, activations fp32 in HBM and rounded to
// bf16 on their way into LDS); what changes is how operands travel, because at bf16 rates the round-1 kernel was bound
// by its staging code (one dword load per thread, position and channel) and by LDS reads (three shifted B operands
// per position tile, one per time tap):
//
//   * POSITION-INTERLEAVED tiles.  A wave owns 128 consecutive positions as FOUR 32-column MFMA tiles, tile q holding
//     positions 4l + q (l = lane & 31).  The B operand of tile q for time tap kw is then the unit set S(q + kw) =
//     {unit 4l + q + kw}: twelve (tile, tap) uses share SIX operand reads S(0..5) - half the LDS B traffic of
//     contiguous tiles - and the four tiles' accumulators of one output channel are four consecutive time steps in ONE
//     lane: the epilogue stores (and reads the residual as) 16-byte vectors.
//   * LDS image [channel group][row][plane u & 3][u >> 2] of 16-byte units (unit u = 8 channels of time step
//     t0 - 1 + u): every S(m) is a run of consecutive units (conflict-free ds_read_b128) and a staging thread's four
//     units go to four planes at one index (conflict-free ds_write_b128, no swizzle).
//   * activations through raw BUFFER loads, 16 bytes = 4 time steps of one channel per lane; zero padding (rows,
//     columns, channels beyond Cin) is the hardware range check.  A thread owns 8 channels x 4 time steps = four
//     complete units; the two halo columns of a row are eight dword loads on a few lanes, predicated by the same range
//     check (idle lanes load nothing and write a dummy LDS slot: no branches in the slab body).
//   * weights by LDS-DMA through a buffer descriptor (per-thread offsets are loop constants, the slab a scalar offset).
//   * two LDS buffers; slab j+1 is converted and written, and the raw loads of slab j+2 are issued, in the shadow of
//     the MFMAs of slab j.
//   * two shapes of one template <G = channel groups per slab>: G = 4 -> 512 threads, 128 co x 512 positions, 32-channel
//     slabs, one workgroup per CU; G = 2 -> 256 threads, 64 co x 512 positions, 16-channel slabs, two workgroups per CU
//     (the 64-channel layers are HBM-bound: a second workgroup covers the first one's prologue and epilogue).
//
// Kernels: (5,3) and (1,1) (one tap, no halo).  Requirements beyond conv_bf16.hip's: T % 4 == 0, Cin % 8 == 0, cin_split % 32 == 0, every source view below
// 2 GiB per batch item, Cout > 32, 16-byte aligned rows of out / res.  Anything else (and bf16x3) runs on the round-1 kernel.
#include "conv_common.h"
#include "prof.h"
#include <cstdlib>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

struct Bf16pGeom {
    int GP, CoutP, pt_log2, pr_log2, tiles_t, tiles_f, ncb, total;
};

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
constexpr unsigned OOB = 0x80000000u;      // beyond every descriptor's num_records: the load returns 0, touches nothing

#ifndef ABL
#define ABL 0
#endif

template <int G, int KW, bool HAS_ISC, bool UNITS = false>
__global__ __launch_bounds__(128 * G, (G == 4 ? 1 : 2)) void conv_bf16p_kernel(babe_conv_args a, Bf16pGeom g,
                                                                               const unsigned short* __restrict__ wq) {
#if __HIP_DEVICE_COMPILE__      // buffer-descriptor builtins exist in the device pass only
    constexpr int NT = 2, NP = 4, WC = 4, WR = G / 2;
    constexpr int NTH = 128 * G, KC = 8 * G, NGP = G / 2;
    constexpr int BN = WR * NT * 32;
    constexpr int XCHP = 640;                         // units per channel group: PR rows of PT + 4, PR <= 32
    constexpr int XSZ = G * XCHP;
    constexpr int NWU = KW * G * BN;                  // weight units per slab
    constexpr int WJ = (NWU + NTH - 1) / NTH;
    constexpr int BUF = XSZ + WJ * NTH + 4;           // + the dummy slot idle halo lanes write
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16x8* smem = reinterpret_cast<bf16x8*>(smem_raw);

    const int PT = 1 << g.pt_log2;
    const int PR = 1 << g.pr_log2;
    const int PI = (PT >> 2) + 1;                     // units per plane and row
    const int RU = 4 * PI;                            // units per row
    // UNITS: the activations arrive as bf16 units (babe_scale_gelu_units) and are copied by LDS-DMA, which writes
    // consecutive lanes to consecutive LDS slots: the channel groups are packed densely (PR*RU units each)
    const int xchp = UNITS ? PR * RU : XCHP;
    // XCD-aware tile order.  Workgroups go to the 8 XCDs round-robin by linear id; tiles that read the same input rows
    // (the two channel blocks of a tile, then the frequency-neighbours of a time column) are given to ONE XCD as a
    // contiguous chunk of the list [batch][time tile][row tile][channel block], so the five-tap re-reads hit that XCD's L2.
    const int per_xcd = (g.total + 7) >> 3;
    int lin = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= per_xcd || lin >= g.total) return;
    const int cb = lin % g.ncb;
    lin /= g.ncb;
    const int tile_f = lin % g.tiles_f;
    lin /= g.tiles_f;
    const int tile_t = lin % g.tiles_t;
    const int b = lin / g.tiles_t;
    const int t0 = tile_t << g.pt_log2;
    const int f0 = tile_f << g.pr_log2;
    const int co0 = cb * BN;
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
        __builtin_amdgcn_make_buffer_rsrc((void*)wq, 0, a.KH * KW * g.GP * g.CoutP * 16, 0x00020000);

    // ---- staging constants.  Channel group = tid / 128 (wave-uniform), inside it 128 quads of 4 time steps.
    const int gw = __builtin_amdgcn_readfirstlane(tid >> 7);
    const int lt = tid & 127;
    const int i4 = lt & ((PT >> 2) - 1);
    const int srow = lt >> (g.pt_log2 - 2);
    const int st = t0 + 4 * i4;
    const int xspat4 = ((f0 + srow) * a.T + st) * 4;
    const unsigned xcolbad = st < a.T ? 0u : OOB;
    // units 4*i4 + 1 .. 4*i4 + 4 of the row: planes 1, 2, 3 at index i4 and plane 0 at index i4 + 1
    const int xslot = gw * xchp + srow * RU + i4;
    // halo: lanes lt < 2*PR of every group own (row, side) = (lt >> 1, lt & 1): unit 0 (plane 0, index 0) or unit PT + 1
    // (plane 1, index PT / 4)
    const bool hl = lt < 2 * PR;
    const int hrow = lt >> 1;
    const int ht = (lt & 1) ? t0 + PT : t0 - 1;
    const int hspat4 = ((f0 + hrow) * a.T + ht) * 4;
    const unsigned hbad = (hl && ht >= 0 && ht < a.T) ? 0u : OOB;
    const int hslot = hl ? gw * xchp + hrow * RU + ((lt & 1) ? PI + (PT >> 2) : 0) : XSZ + WJ * NTH;
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

    // UNITS: X-image DMA tasks.  Slot s = tid + k*NTH of the dense image [group][row][plane][index] <- the unit at
    // [group][f][plane][t0/4 + index] of the tensor; entries beyond T/4 and slots beyond the image read as zeros.
    constexpr int XJ = XSZ / NTH;
    int xvo[UNITS ? XJ : 1], xrw[UNITS ? XJ : 1];
    const int PIg = (a.T >> 2) + 1;                    // entries per plane of a tensor row
    if (UNITS) {
#pragma unroll
        for (int k = 0; k < XJ; ++k) {
            const int sl = tid + k * NTH;
            const int gl = sl / xchp;
            const int r = sl - gl * xchp;
            const int row = r / RU;
            const int rr = r - row * RU;
            const int pl = rr / PI;
            const int ix = rr - pl * PI;
            const int e = (t0 >> 2) + ix;
            const bool ok = gl < G && e < PIg;
            xrw[k] = f0 + row;
            xvo[k] = (int)((unsigned)((gl * (int)a.in_cs + ((f0 + row) * 4 + pl) * PIg + e) * 16) | (ok ? 0u : OOB));
        }
    }
    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(UNITS ? reinterpret_cast<const char*>(a.in) + (long)b * a.in_bs * 16 : reinterpret_cast<const char*>(a.in)), 0,
        UNITS ? (a.Cin >> 3) * (int)a.in_cs * 16 : 0, 0x00020000);
    auto dma_x = [&](int kh, int ci0, bf16x8* buf) {
        const int foff = (kh - khc) * a.dil;
        const int so = ((ci0 >> 3) * (int)a.in_cs + foff * 4 * PIg) * 16;      // scalar; added to the LANE offset (range-checked)
#pragma unroll
        for (int k = 0; k < (UNITS ? XJ : 0); ++k) {
            const int fr = xrw[k] + foff;
            const unsigned e = (unsigned)(xvo[k] + so) | ((unsigned)(fr | (a.F - 1 - fr)) & OOB) | ((unsigned)xvo[k] & OOB);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, LDS_PTR(buf + k * NTH + wave * 64), 16, e, 0, 0, 0);
        }
    };

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

    auto issue_act = [&](int kh, int ci0) {
        if ((ABL & 1) || UNITS) return;
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
        if (KW == 3) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                xh[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, eh, gbad ? 0 : sb4 + j * cs * 4, 0));
        }
        if (HAS_ISC) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                int ci = ci0 + 8 * gw + j;
                ci = ci < a.Cin ? ci : a.Cin - 1;
                xsc[j] = a.in_scale[(long)b * a.Cin + ci];                 // wave-uniform address
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
        if ((ABL & 8) || UNITS) return;
#pragma unroll
        for (int k = k0; k < k1; ++k) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = xv[j][k];
            buf[xslot + (k == 3 ? 1 : (k + 1) * PI)] = pack8(v);           // unit 4*i4 + 1 + k
        }
    };
    auto store_halo = [&](bf16x8* buf) { if (KW == 3 && !(ABL & 8) && !UNITS) buf[hslot] = pack8(xh); };
    auto dma_w = [&](int kh, int ci0, bf16x8* buf) {
        if (ABL & 32) return;
        const int so = ((kh * KW) * g.GP + (ci0 >> 3)) * g.CoutP * 16;     // bytes, scalar
#pragma unroll
        for (int jj = 0; jj < WJ; ++jj)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, LDS_PTR(buf + XSZ + jj * NTH + wave * 64), 16, wvo[jj], so, 0, 0);
    };
    // Slab order: channel slab OUTER, frequency tap INNER.  The rows a tile reads for tap kh+1 are the rows its
    // frequency-neighbours read for tap kh: with the taps innermost those re-reads fall within a few slabs of each other,
    // while the live footprint (one channel slab of the XCD's rows) still fits the 4 MB L2.
    auto advance = [&](int& kh, int& ci0) {             // next slab, clamped at the last one
        int nk = kh + 1, nc = ci0;
        if (nk > kh_hi) {
            nk = kh_lo;
            nc += KC;
        }
        if (nc < CinP) {
            kh = nk;
            ci0 = nc;
        }
    };

    // ---- operand addresses (units).  This lane's four tiles hold positions p0 + q, p0 = wc*128 + 4*l31; S(m) is unit
    // tt0 + m of its row: plane m & 3, index tt0 / 4 + (m >> 2), of channel group 2*gp + h.
    const int p0 = wc * 128 + 4 * l31;
    const int prow = p0 >> g.pt_log2;
    const int tt0 = p0 & (PT - 1);
    int boff[6];
#pragma unroll
    for (int m = 0; m < 6; ++m) boff[m] = h * xchp + prow * RU + (m & 3) * PI + (tt0 >> 2) + (m >> 2);
    const int gpo = 2 * xchp;                        // second channel-group pair
    const int aoff = XSZ + h * BN + wr * (NT * 32) + l31;

    // ---- prologue: slab 0 into buffer 0, raw loads of slab 1 in flight
    int kA = kh_lo, cA = 0;
    if (UNITS) {
        dma_w(kA, cA, smem);
        dma_x(kA, cA, smem);
        advance(kA, cA);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        issue_act(kA, cA);
        dma_w(kA, cA, smem);
        store_main(smem, 0, 4);
        store_halo(smem);
        advance(kA, cA);
        issue_act(kA, cA);
        // the weight DMA of slab 0 is older than the raw loads of slab 1 (16 with the halo, 8 without)
        if (KW == 3) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
    __syncthreads();

    // MFMA schedule of a slab: NGP channel-group pairs x 3 time taps; tap kw of pair gp multiplies A(gp, kw) with
    // S(gp, kw .. kw + 3).  S(m + 4) and A(kw + 1) are read while tap kw is multiplied; the first operands of the next pair
    // are read during the last tap of the current one.
    bf16x8 av[2][NT], sv[2][6];
    // 96 output channels on the 128-channel tile: the upper row-tile of the wr = 1 waves is all padding - those waves skip
    // its MFMAs (wave-uniform; every SIMD hosts one wr = 0 and one wr = 1 wave, so the matrix pipes stay balanced)
    const bool both = co0 + wr * (NT * 32) + 32 < a.Cout;
    int cur = 0;
    for (int j = 0; j < nslab; ++j) {
        const bf16x8* Xs = smem + cur * BUF;
        bf16x8* Xw = smem + (cur ^ 1) * BUF;
#define READ_A(c, gp, kw) if (!(ABL & 4) || j == 0) \
    _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) av[c][nt] = Xs[aoff + ((kw) * G + 2 * (gp)) * BN + nt * 32];
#define READ_S(gp, m) if (!(ABL & 4) || j == 0) sv[(gp) & 1][m] = Xs[boff[m] + (gp) * gpo];
#define MFMA_TAP(c, gp, kw)                                                                                   \
    if (!(ABL & 2)) {                                                                                         \
        _Pragma("unroll") for (int q = 0; q < NP; ++q)                                                        \
            acc[0][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[c][0], sv[(gp) & 1][q + (kw)], acc[0][q], 0, 0, 0); \
        if (both) _Pragma("unroll") for (int q = 0; q < NP; ++q)                                              \
            acc[1][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[c][1], sv[(gp) & 1][q + (kw)], acc[1][q], 0, 0, 0); \
    }                                                                                                         \
    __builtin_amdgcn_sched_barrier(0);
        if (KW == 3) { READ_S(0, 0) }
        READ_S(0, 1) READ_A(0, 0, 0) READ_S(0, 2) READ_S(0, 3)
        if (KW == 1) { READ_S(0, 4) }              // one tap = the centre one: units tt + 1 .. tt + 4
        dma_w(kA, cA, Xw);                         // weights of slab j+1: issued BEFORE this slab's raw loads (see the wait below)
        if (UNITS) dma_x(kA, cA, Xw);              // ... and its activation units: both operands by DMA, no staging registers
        if constexpr (KW == 3) {
            // pair 0, tap 0
            READ_S(0, 4) READ_A(1, 0, 1)
            store_main(Xw, 0, 2);
            MFMA_TAP(0, 0, 0)
            // pair 0, tap 1
            READ_S(0, 5) READ_A(0, 0, 2)
            store_main(Xw, 2, 4);
            store_halo(Xw);
            MFMA_TAP(1, 0, 1)
            // pair 0, tap 2: the staging registers are free again -> raw loads of slab j+2
            if (NGP == 2) { READ_S(1, 0) READ_S(1, 1) READ_A(1, 1, 0) }
            advance(kA, cA);
            issue_act(kA, cA);
            MFMA_TAP(0, 0, 2)
            if (NGP == 2) {
                READ_S(1, 2) READ_S(1, 3)
                READ_S(1, 4) READ_A(0, 1, 1)
                MFMA_TAP(1, 1, 0)
                READ_S(1, 5) READ_A(1, 1, 2)
                MFMA_TAP(0, 1, 1)
                MFMA_TAP(1, 1, 2)
            }
        } else {
            // (1,1) kernel: one tap per pair, S(0..3) only; the slab is staging-bound, the MFMAs ride along
            if (NGP == 2) { READ_S(1, 1) READ_S(1, 2) READ_A(1, 1, 0) READ_S(1, 3) READ_S(1, 4) }
            store_main(Xw, 0, 4);
            advance(kA, cA);
            issue_act(kA, cA);
            MFMA_TAP(0, 0, 1)
            if (NGP == 2) { MFMA_TAP(1, 1, 1) }
        }
        // Vector-memory operations retire in order: WJ weight DMAs, then this slab's 16 raw loads.  At most 16
        // outstanding = this wave's share of the weight slab is in LDS; the barrier then publishes it to the other waves.
        if (UNITS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (KW == 3) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        if (!(ABL & 16)) __syncthreads();
        cur ^= 1;
    }
#undef READ_A
#undef READ_S
#undef MFMA_TAP

    // ---- epilogue: out = alpha*acc*oscale[b,co] + rbeta*res; a lane's four tiles are four consecutive time steps.
    // Round 6: through buffer descriptors like conv11p's - one per-lane byte offset per position (out of range when the position is
    // padding) + a scalar term per channel, a channel beyond Cout lands beyond the descriptor - so that NO load sits in a branch, and the
    // residual of group g + 1 (4 channels x 4 time steps) is in flight while group g is scaled and stored.  Before, every one of the
    // 4 NT groups issued its loads behind per-element `has_res ? load : 0` branches and waited vmcnt(0) before EVERY store: 8 - 16
    // exposed memory latencies per workgroup at one or two workgroups per CU.
    {
        typedef int i32x4 __attribute__((ext_vector_type(4)));
        const int f = f0 + prow;
        const int t = t0 + tt0;
        const bool pv = f < a.F && t < a.T;
        const bool has_os = a.oscale != nullptr, has_res = a.res != nullptr;
        const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)(a.out + (long)b * a.out_bs), 0, (unsigned)(a.Cout * a.out_cs * 4), 0x00020000);
        const __amdgpu_buffer_rsrc_t rr_ = __builtin_amdgcn_make_buffer_rsrc((void*)(has_res ? a.res + (long)b * a.res_bs : a.out), 0,
                                                                             has_res ? (unsigned)(a.Cout * a.res_cs * 4) : 0u, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_ = __builtin_amdgcn_make_buffer_rsrc((void*)(has_os ? a.oscale + (long)b * a.Cout : a.out), 0,
                                                                             has_os ? (unsigned)(a.Cout * 4) : 0u, 0x00020000);
        const unsigned sp = pv ? (unsigned)(f * a.T + t) * 4u : 0x80000000u;
        const unsigned ocs = (unsigned)a.out_cs * 4u, rcs = (unsigned)a.res_cs * 4u;
        const float os_m = has_os ? a.alpha : 0.f, os_a = has_os ? 0.f : a.alpha;      // scale = fma(oscale, os_m, os_a)
        constexpr int NG = NT * 4;
        auto chan = [&](int gi) { return co0 + wr * (NT * 32) + (gi >> 2) * 32 + 8 * (gi & 3) + 4 * h; };
        f32x4 rv[2][4];
        float sv[2][4];
        auto load_group = [&](int gi, f32x4 (&dr)[4], float (&ds)[4]) __attribute__((always_inline)) {
            const unsigned cq = (unsigned)chan(gi);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                ds[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_, (cq + k) * 4u, 0, 0));
                dr[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rr_, sp + (cq + k) * rcs, 0, 0));
            }
        };
        load_group(0, rv[0], sv[0]);
#pragma unroll
        for (int gi = 0; gi < NG; ++gi) {
            if (gi + 1 < NG) load_group(gi + 1, rv[(gi + 1) & 1], sv[(gi + 1) & 1]);
            const int nt = gi >> 2, qd = gi & 3;
            const unsigned cq = (unsigned)chan(gi);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int r = 4 * qd + k;
                const float sc = __builtin_fmaf(sv[gi & 1][k], os_m, os_a);
                const f32x4 rk = rv[gi & 1][k];
                f32x4 y;
                y[0] = __builtin_fmaf(acc[nt][0][r], sc, a.rbeta * rk[0]);
                y[1] = __builtin_fmaf(acc[nt][1][r], sc, a.rbeta * rk[1]);
                y[2] = __builtin_fmaf(acc[nt][2][r], sc, a.rbeta * rk[2]);
                y[3] = __builtin_fmaf(acc[nt][3][r], sc, a.rbeta * rk[3]);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, y), ro, sp + (cq + k) * ocs, 0, 0);
            }
        }
    }
#endif
}

inline int ilog2_ceil_b(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}

template <int G, int KW, bool UNITS = false>
void launch_bf16p(const babe_conv_args& a, Bf16pGeom g, const unsigned short* wq, hipStream_t s) {
    constexpr int BN = 32 * G, NTH = 128 * G;
    g.pt_log2 = ilog2_ceil_b(a.T);
    if (g.pt_log2 > 9) g.pt_log2 = 9;
    if (g.pt_log2 < 4) g.pt_log2 = 4;
    g.pr_log2 = 9 - g.pt_log2;
    const int PT = 1 << g.pt_log2, PR = 1 << g.pr_log2;
    g.tiles_t = cdiv(a.T, PT);
    g.tiles_f = cdiv(a.F, PR);
    g.ncb = cdiv(g.CoutP, BN);
    g.total = g.tiles_t * g.tiles_f * g.ncb * a.B;
    dim3 grid(8 * ((g.total + 7) / 8));
    constexpr int WJ = (KW * G * BN + NTH - 1) / NTH;
    const size_t lds = 2 * (size_t)(G * 640 + WJ * NTH + 4) * 16;
    static std::atomic<unsigned long long> attr_done{0};
    if (babe_lds_optin(attr_done, {reinterpret_cast<const void*>(&conv_bf16p_kernel<G, KW, true, UNITS>),
                                   reinterpret_cast<const void*>(&conv_bf16p_kernel<G, KW, false, UNITS>)}, (int)lds) != hipSuccess)
        return;
    if (a.in_scale) hipLaunchKernelGGL((conv_bf16p_kernel<G, KW, true, UNITS>), grid, dim3(NTH), lds, s, a, g, wq);
    else hipLaunchKernelGGL((conv_bf16p_kernel<G, KW, false, UNITS>), grid, dim3(NTH), lds, s, a, g, wq);
}

}  // namespace

/* 1 if the pipelined bf16 kernel takes this problem (splits == 1 only) */
int babe_conv2d_bf16p_supported(const babe_conv_args& a) {
    static const char* ov = getenv("BABE_CONV_BF16P");
    if (ov && ov[0] == '0') return 0;
    if (!((a.KW == 3) || (a.KW == 1 && a.KH == 1)) || (a.T & 3) || (a.Cin & 7) || a.Cout <= 32) return 0;
    if (a.in2 && (a.cin_split % 32 != 0)) return 0;
    auto al16 = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
    if (!al16(a.out) || (a.out_bs & 3) || (a.out_cs & 3)) return 0;                   // 16-byte epilogue vectors
    if (a.res && (!al16(a.res) || (a.res_bs & 3) || (a.res_cs & 3))) return 0;
    const long lim = 0x7fffffffL / 4;
    // (the epilogue addresses out / res of a batch item through buffer descriptors with 32-bit byte offsets)
    if ((long)((a.Cout + 31) / 32 * 32) * a.out_cs >= lim || (a.res && (long)((a.Cout + 31) / 32 * 32) * a.res_cs >= lim)) return 0;
    const int split = a.in2 ? a.cin_split : a.Cin;
    if ((long)split * a.in_cs >= lim) return 0;
    if (a.in2 && (long)(a.Cin - split) * a.in2_cs >= lim) return 0;
    if ((long)a.KH * a.KW * ((a.Cin + 15) / 16 * 2) * ((a.Cout + 31) / 32 * 32) * 16 >= 0x7fffffffL) return 0;
    return 1;
}

int babe_conv2d_bf16p_launch(const babe_conv_args& a, const unsigned short* wq, hipStream_t s) {
    Bf16pGeom g;
    g.GP = (a.Cin + 15) / 16 * 2;
    g.CoutP = (a.Cout + 31) / 32 * 32;
    if (a.KW == 3) {
        if (g.CoutP == 64) launch_bf16p<2, 3>(a, g, wq, s);      //  64 co x 512 positions, 4 waves, two workgroups per CU
        else launch_bf16p<4, 3>(a, g, wq, s);                    // 128 co x 512 positions, 8 waves
    } else {
        if (g.CoutP == 64) launch_bf16p<2, 1>(a, g, wq, s);
        else launch_bf16p<4, 1>(a, g, wq, s);
    }
    return 0;
}

/* Same conv with the activations already in bf16 "units" (babe_scale_gelu_units): a->in = unit tensor, a->in_bs / a->in_cs =
 * units per batch item / per 8-channel group; no second source, no in_scale.  Both operands travel by LDS-DMA. */
extern "C" int babe_conv2d_bf16_units_supported(const babe_conv_args* ap) {
    if (!ap) return 0;
    const babe_conv_args& a = *ap;
    static const char* ov = getenv("BABE_CONV_BF16U");
    if (ov && ov[0] == '0') return 0;
    if (a.KH != 5 || a.KW != 3 || (a.T & 3) || (a.Cin & 7) || a.Cout <= 32 || a.in2 || a.in_scale) return 0;
    auto al16 = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
    if (!al16(a.in) || !al16(a.out) || (a.out_bs & 3) || (a.out_cs & 3)) return 0;
    if (a.res && (!al16(a.res) || (a.res_bs & 3) || (a.res_cs & 3))) return 0;
    if ((long)(a.Cin >> 3) * a.in_cs * 16 >= 0x7fffffffL) return 0;
    if ((long)a.KH * 3 * ((a.Cin + 15) / 16 * 2) * ((a.Cout + 31) / 32 * 32) * 16 >= 0x7fffffffL) return 0;
    return 1;
}

extern "C" int babe_conv2d_bf16_units(const babe_conv_args* ap, const void* w_bf16, void* stream) {
    BABE_CHECK_ARG(ap && w_bf16, "conv2d_bf16_units: null args");
    BABE_CHECK_ARG(babe_conv2d_bf16_units_supported(ap), "conv2d_bf16_units: unsupported problem (see babe_conv2d_bf16_units_supported)");
    const babe_conv_args& a = *ap;
    BABE_CHECK_ARG(a.in_cs == (long)a.F * 4 * (a.T / 4 + 1), "conv2d_bf16_units: in_cs %ld is not F*4*(T/4+1)", a.in_cs);
    Bf16pGeom g;
    g.GP = (a.Cin + 15) / 16 * 2;
    g.CoutP = (a.Cout + 31) / 32 * 32;
    const double flops = babe_conv_flops(a);
    BabeProfScope prof(BABE_SLOT_CONV_BF16P, babe_conv_bytes(a) - 2.0 * a.B * a.Cin * (double)a.F * a.T, flops, flops, stream);
    if (g.CoutP == 64) launch_bf16p<2, 3, true>(a, g, (const unsigned short*)w_bf16, (hipStream_t)stream);
    else launch_bf16p<4, 3, true>(a, g, (const unsigned short*)w_bf16, (hipStream_t)stream);
    BABE_LAUNCH_CHECK();
    return BABE_OK;
}
