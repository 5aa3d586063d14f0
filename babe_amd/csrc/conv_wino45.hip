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
#include <type_traits>

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ABL: compile-time ablation bits for profiling builds (tools/ab/abl_build.sh conv_wino45 <mask>): 1 no activation loads,
// 2 no MFMA, 4 no operand LDS reads, 8 no input transform / LDS stores, 16 no barrier, 32 no weight DMA, 64 every slab
// loads channel 0 (cache-resident loads: same instructions, no memory-side traffic).  NB: with the loads or the transform
// removed hipcc also removes whatever became dead (ablation 1 drops the transform arithmetic too) (cache-resident loads: same instructions, no memory-side traffic)
#ifndef ABL
#define ABL 0
#endif

namespace {

struct Wino45Geom {
    int CinP, CoutP, tiles_t, groups;      // groups: tiles along the flattened (residue class, row pair) list
    int npairs;   // row pairs per residue class (ceil(ceil(F / dil) / 2))
    int tsh;      // tile shape: 4 >> tsh row pairs x (64 << tsh) time steps (the 64 units = 4 segments of 16; tsh = 0, 1, 2)
};

// Tile shape of a launch: segments (16 time units = 64 steps of one row pair) are dealt as (4 >> tsh) row pairs x (1 << tsh)
// time blocks.  Chosen per launch for the fullest tiles (time: T = 128 fills 64- and 128-step tiles, not 256-step ones; rows:
// the pairs of all residue classes are one list, so only its last tile and the odd row of a class are lost).
static inline double wino45_fill(const babe_conv_args& a, int tsh) {
    const int ppt = 4 >> tsh, tlen = 64 << tsh;
    const long npall = (long)a.dil * (((a.F + a.dil - 1) / a.dil + 1) / 2);     // pairs of all residue classes (one list)
    const long tiles = (npall + ppt - 1) / ppt;
    return ((double)a.F / (2.0 * ppt * tiles)) * ((double)a.T / ((double)tlen * ((a.T + tlen - 1) / tlen)));
}
static inline int wino45_best_tsh(const babe_conv_args& a) {
    int best = 0;
    for (int t = 1; t <= 2; ++t)
        if (wino45_fill(a, t) > wino45_fill(a, best) + 1e-9) best = t;
    return best;
}

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
constexpr unsigned OOBH = 0xC0000000u;     // invalid offsets start here: +-(a source view < 1 GiB) stays >= 2^31

// Optional register cap (-DBABE_W45_NUM_VGPR=112; the attribute counts in halves of the unified file on gfx90a+, so 112 means
// 224 registers per wave).  Idea: two waves of this kernel per SIMD would then leave 64 of the SIMD's 512 registers, room
// for one wave of an element-wise kernel (GroupNorm, GELU, axpby, resample: 8..62 VGPRs) of the sampler's other lane.
// Measured (round 3, same box, same session): 25 spilled registers, kernels 1 % slower, headline 1.636 vs 1.640 - the
// two-lane overlap is not limited by registers.  Default: no cap (248 / 243 registers, no spills).
#ifndef BABE_W45_NUM_VGPR
#define BABE_W45_NUM_VGPR 128
#endif
template <bool HAS_ISC, bool PADC>
__global__ __launch_bounds__(512, 1) __attribute__((amdgpu_num_vgpr(BABE_W45_NUM_VGPR))) void conv_wino45_kernel(babe_conv_args a, Wino45Geom g, const float* __restrict__ wq) {
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
    // (cw, uw): waves w and w + 4 share a SIMD; they get channel tiles cw and cw ^ 2, so that on a tile whose upper 32 output
    // channels are padding (Cout = 96: the second tile of every 96-channel layer) each SIMD hosts ONE wave with matrix work
    // - the waves of the padded half skip their MFMAs and operand reads (wave-uniform), and the tile costs its transforms +
    // half its MFMAs instead of a full tile
    const int uw = wave >> 2, cw = (wave + 2 * uw) & 3;
    const int l15 = lane & 15, lk = lane >> 4;
    const int b = blockIdx.z;
    const int co0 = blockIdx.y * BN;
    // PADC (launches whose last channel tile is padded): does this wave's 16-channel tile hold any real channel?
    const bool mact = !PADC || co0 + cw * 16 < a.Cout;   // wave-uniform; compile-time true without PADC
    const int tile_t = blockIdx.x % g.tiles_t;
    const int rest = blockIdx.x / g.tiles_t;
    // Row pairs of ALL residue classes form one list, pair P = class * npairs + p (rows class + 2p dil, class + (2p+1) dil);
    // a tile takes ppt consecutive entries - a class boundary may fall inside a tile (the segments of a tile are independent:
    // 6 rows per class = 3 pairs filled 3 of 4 slots when tiles stayed inside a class).  Beyond the list: row >= F.
    const int P0 = rest * (4 >> g.tsh);
    const int npall = a.dil * g.npairs;
    auto pair_row = [&](int lp) __attribute__((always_inline)) {        // first output row of local pair lp of this tile
        const int P = P0 + lp;
        const int c = P / g.npairs;
        return P < npall ? c + 2 * (P - c * g.npairs) * a.dil : a.F + 4 * a.dil;
    };
    const int t0 = tile_t * (64 << g.tsh);
    const int tbm = (1 << g.tsh) - 1;                     // mask of the time-block index of a segment
    const int nci = g.CinP / KC;
    const int nslab = 3 * nci;

    // ---- descriptors (ONE source: every (5,3) conv of the UNet reads a single tensor - the two-source inputs of the decoder
    // go through (1,1) convs - and a `second ? a : b` on pointers captured by the staging lambdas put them in scratch memory)
    const float* p1 = a.in + (long)b * a.in_bs;
    const int cs1 = (int)a.in_cs;
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)p1, 0, a.Cin * cs1 * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)wq, 0, 3 * g.CinP * g.CoutP * 48, 0x00020000);

    // ---- per-thread staging constants: thread = (ci = wave, unit = lane), unit = rp * 16 + tu.
    // fp32 MFMA and vector-ALU instructions do NOT overlap on this hardware (tools/mfma_valu_coexec.hip: a loop of
    // v_mfma_f32_16x16x4_f32 slows down by the full issue time of every v_fma / v_mov put beside it, 3-4 ns per wave
    // instruction at two waves per SIMD; ds_read and SALU do overlap), so every vector instruction of the staging path is
    // paid in matrix-pipe time.  Hence: the channel and slab part of a load address travels in the SCALAR offset of the buffer
    // instruction (one input channel per wave), the per-lane part is a loop constant, coefficients are SGPRs, neighbour
    // samples come by DPP, and the slab body is one straight line (branches made hipcc shuffle 100+ registers per slab).
    // segment q = lane >> 4 (a 16-lane DPP row): row pair q >> tsh of the tile, time block q & tbm
    const int s_tu = lane & 15, s_rp = lane >> 4;
    const int s_t = t0 + 64 * (s_rp & tbm) + 4 * s_tu;
    const int s_fa = pair_row(s_rp >> g.tsh);                        // first output row of the pair
    // Neighbour samples t-1 and t+4 come from the adjacent lanes (same row, tu -+ 1) by DPP row shifts; only the first lane of
    // a 16-lane DPP row needs t-1 from memory and only the last one t+4.  ONE dword load per slab fetches all 48 of them for
    // the wave - lane L < 48 loads the sample of (row L >> 3, segment (L >> 1) & 3, side L & 1) - and ds_bpermute hands
    // each to the lane that needs it.  A vector-memory instruction costs about 8 ns of CU time next to fp32 MFMAs whether it
    // moves 1 KB or nothing (ablations in DESIGN.md 8), so their NUMBER is what the staging path minimises.
    unsigned er[6];                                                     // byte offset of (row r, t), or OOBH
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        const int fr = s_fa + (r - 2) * a.dil;
        const bool ok = fr >= 0 && fr < a.F && s_t < a.T;
        er[r] = ok ? (unsigned)((fr * a.T + s_t) * 4) : OOBH;
    }
    unsigned ehalo = OOBH;
    {
        const int hr = lane >> 3, hg = (lane >> 1) & 3, hs = lane & 1;
        const int fr = pair_row(hg >> g.tsh) + (hr - 2) * a.dil;
        const int th = t0 + 64 * (hg & tbm) + (hs ? 64 : -1);
        if (lane < 48 && fr >= 0 && fr < a.F && th >= 0 && th < a.T) ehalo = (unsigned)((fr * a.T + th) * 4);
    }
    // bpermute source (byte index) of row 0 for this lane: its row pair's left sample, or the right one for the last lane of
    // the DPP row; row r adds 32 bytes through the instruction's offset field
    const int hsrc = (2 * s_rp + (s_tu == 15 ? 1 : 0)) * 4;
    const int xlds = tid * 3;                                           // float4 index of this thread's 12 floats
    const int wvo = lane * 16;                                          // weight DMA: this lane's 16 bytes inside a 1 KB chunk

    f32x4 acc[2][12];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int p = 0; p < 12; ++p) acc[i][p] = f32x4{0.f, 0.f, 0.f, 0.f};

    // staging registers: raw loads of the slab that is next to be transformed.  Rows 0 and 5 of the patch are read by pass 2
    // (phases 0, 5) only: they are loaded behind a wave-uniform branch in that pass and hold zeros before it (their
    // coefficients are 0 in passes 0 and 1 anyway).
    // ONE set, loaded one slab before its transform.  (A second set - loads issued two slabs ahead - measured within noise:
    // what the staging path costs is the ISSUE of its ~125 vector instructions per wave and slab, not memory latency; and
    // its 17 registers put the kernel at 256 VGPRs with 10 spilled.)
    f32x4 xvs[1][4], xve[2];
    float xhls[1] = {0.f}, xscs[1] = {1.f};               // xhls: the wave's 48 halo samples, one per lane
    int pSs[1] = {0};                                    // pass of the data held in the set
    xve[0] = xve[1] = f32x4{0.f, 0.f, 0.f, 0.f};
    // rows r0 .. r1-1 of the patch of slab (ps, ci0).  The loads of a slab are issued a few at a time BETWEEN the MFMA groups:
    // issued back to back by all eight waves they queue up in front of the CU's one address unit and every wave sits in
    // its load-issue phase at the same time (ablation: 240 of 750 us on the 256-channel layers).
    auto issue_rows = [&](int set, int ps, int ci0, int r0, int r1) __attribute__((always_inline)) {
        if (ABL & 1) return;
        const int so = (ABL & 64) ? 0 : (ci0 + wave) * cs1 * 4;       // scalar: channel of this wave
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            if (r < r0 || r >= r1) continue;
            const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs1, er[r], so, 0));
            if (r == 0) xve[0] = v;
            else if (r == 5) xve[1] = v;
            else xvs[set][r - 1] = v;
        }
    };
    auto issue_halo = [&](int set, int ci0) __attribute__((always_inline)) {
        if (ABL & 1) return;
        const int so = (ABL & 64) ? 0 : (ci0 + wave) * cs1 * 4;
        xhls[set] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs1, ehalo, so, 0));
    };
    auto issue_isc = [&](int set, int ci0) __attribute__((always_inline)) {
        if (HAS_ISC) xscs[set] = a.in_scale[(long)b * a.Cin + ci0 + wave];    // one channel per wave: a scalar load
    };
    auto issue_main = [&](int set, int ps, int ci0) __attribute__((always_inline)) {    // (prologue: rows 1-4 at once)
        issue_rows(set, ps, ci0, 1, 5);
        issue_halo(set, ci0);
        issue_isc(set, ci0);
        pSs[set] = ps;
    };
    // rows 0 and 5: pass 2 only.  Always issued - outside pass 2 with an out-of-range offset, which costs an instruction slot
    // but no traffic: behind a branch hipcc cannot count the outstanding loads any more and waits for vmcnt(0) at the next
    // transform, i.e. for the loads of the OTHER register set too.
    auto issue_edge = [&](int ps, int ci0) __attribute__((always_inline)) {
        if (ABL & 1) return;
        const int so = (ABL & 64) ? 0 : (ci0 + wave) * cs1 * 4;
        const unsigned edge = ps == 2 ? 0u : OOBH;
        xve[0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs1, er[0] | edge, so, 0));
        xve[1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs1, er[5] | edge, so, 0));
    };
    // lane i <- src of lane i-1 / i+1 inside its 16-lane row; the row's first / last lane keeps `old` (its own halo load).
    // Inline asm: hipcc 7.2 miscompiles __builtin_amdgcn_update_dpp on element 3 of a loaded vector (it reads element 0).
    // NOT `asm volatile`: a side-effecting asm statement inside these by-reference lambdas keeps the captured locals (source
    // pointers, strides) in scratch memory, and every buffer instruction then gets a waterfall loop around its descriptor.
    auto dpp_shr1 = [](float old, float src) __attribute__((always_inline)) {
        asm("s_nop 1\n\tv_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(old) : "v"(src));
        return old;
    };
    auto dpp_shl1 = [](float old, float src) __attribute__((always_inline)) {
        asm("s_nop 1\n\tv_mov_b32_dpp %0, %1 row_shl:1 row_mask:0xf bank_mask:0xf" : "+v"(old) : "v"(src));
        return old;
    };
    // frequency phases of pass ps from the 6 rows d0..d5 of one column:
    //   e = d4 - k2 d2,  o = k3 d3 - k1 d1,  Ea = x0 d0 + e + za o,  Eb = x5 d5 + ye e - o
    //   (1,2): k2 4, k1 4, k3 1, x0 0, za 1, x5 0, ye 1      (3,4): k2 1, k1 2, k3 2, x0 0, za 1, x5 0, ye 1
    //   (0,5): k2 5, k1 4, k3 5, x0 4, za 0, x5 1, ye 0      (Ea = 4 d0 - 5 d2 + d4,  Eb = 4 d1 - 5 d3 + d5)
    // (coefficients selected as integers so that they stay in scalar registers)
    auto fsel = [](int ps, unsigned c0, unsigned c1, unsigned c2) __attribute__((always_inline)) {
        return __builtin_bit_cast(float, ps == 0 ? c0 : (ps == 1 ? c1 : c2));
    };
    // time transform (same B^T): U0 = 4E0-5E2+E4, U1/U2 = (E4-4E2) +- (E3-4E1), U3/U4 = (E4-E2) +- 2(E3-E1), U5 = 4E1-5E3+E5
    auto tt = [](const float (&E)[6], float (&U)[6]) {
        const float e = E[4] - 4.f * E[2], o = E[3] - 4.f * E[1];
        const float e2 = E[4] - E[2], o2 = E[3] - E[1];
        U[0] = 4.f * E[0] + (E[4] - 5.f * E[2]);
        U[1] = e + o;
        U[2] = e - o;
        U[3] = e2 + 2.f * o2;
        U[4] = e2 - 2.f * o2;
        U[5] = 4.f * E[1] + (E[5] - 5.f * E[3]);
    };
    auto store_act = [&](int set, f32x4* buf) __attribute__((always_inline)) {
        if (ABL & 8) return;
        const int ps = pSs[set];
        const f32x4 xv[6] = {xve[0], xvs[set][0], xvs[set][1], xvs[set][2], xvs[set][3], xve[1]};
        const float xhl = xhls[set], xsc = xscs[set];
        const float k2 = fsel(ps, 0x40800000u, 0x3f800000u, 0x40a00000u);        // 4 1 5
        const float k1 = fsel(ps, 0x40800000u, 0x40000000u, 0x40800000u);        // 4 2 4
        const float k3 = fsel(ps, 0x3f800000u, 0x40000000u, 0x40a00000u);        // 1 2 5
        const float x0 = fsel(ps, 0u, 0u, 0x40800000u), za = fsel(ps, 0x3f800000u, 0x3f800000u, 0u);
        const float x5 = fsel(ps, 0u, 0u, 0x3f800000u), ye = za;
        // rows 1-4 of the patch: every pass
        float xh[6];                                      // row r: left sample in the first lane of a DPP row, right one in the last
#pragma unroll
        for (int r = 1; r < 5; ++r)
            asm("ds_bpermute_b32 %0, %1, %2 offset:%3" : "=v"(xh[r]) : "v"(hsrc), "v"(xhl), "n"(32 * r));
        // hipcc does not know that the statements above complete asynchronously: every later use of xh[] (register copies
        // included) is made to depend on this wait
        asm("s_waitcnt lgkmcnt(0)" : "+v"(xh[1]), "+v"(xh[2]), "+v"(xh[3]), "+v"(xh[4]));
        float Ea[6], Eb[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            float d[6];
#pragma unroll
            for (int r = 1; r < 5; ++r)
                d[r] = j == 0 ? dpp_shr1(xh[r], xv[r][3]) : (j == 5 ? dpp_shl1(xh[r], xv[r][0]) : xv[r][j - 1]);
            const float e = d[4] - k2 * d[2];
            const float o = k3 * d[3] - k1 * d[1];
            Ea[j] = za * o + e;
            Eb[j] = ye * e - o;
        }
        // rows 0 and 5: the last pass only (x0 = 4, x5 = 1; their coefficients are 0 in the other two).  A wave-uniform
        // branch around 2 ds_bpermute + 4 DPP moves + 12 FMAs - a seventh of the slab's vector instructions, and every one
        // of them costs matrix-pipe time (the fp32 MFMA does not co-issue with the vector ALU).  It only touches transform
        // temporaries: no accumulator is live-modified across it, so hipcc has nothing to copy at the merge.
        if (ps == 2) {
            asm("ds_bpermute_b32 %0, %1, %2" : "=v"(xh[0]) : "v"(hsrc), "v"(xhl));
            asm("ds_bpermute_b32 %0, %1, %2 offset:160" : "=v"(xh[5]) : "v"(hsrc), "v"(xhl));
            asm("s_waitcnt lgkmcnt(0)" : "+v"(xh[0]), "+v"(xh[5]));
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const float d0 = j == 0 ? dpp_shr1(xh[0], xv[0][3]) : (j == 5 ? dpp_shl1(xh[0], xv[0][0]) : xv[0][j - 1]);
                const float d5 = j == 0 ? dpp_shr1(xh[5], xv[5][3]) : (j == 5 ? dpp_shl1(xh[5], xv[5][0]) : xv[5][j - 1]);
                Ea[j] = x0 * d0 + Ea[j];
                Eb[j] = x5 * d5 + Eb[j];
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
        tt(Ea, Ua);
        tt(Eb, Ub);
        buf[xlds] = f32x4{Ua[0], Ua[1], Ua[2], Ua[3]};
        buf[xlds + 1] = f32x4{Ua[4], Ua[5], Ub[0], Ub[1]};
        buf[xlds + 2] = f32x4{Ub[2], Ub[3], Ub[4], Ub[5]};
    };
    // weight slab [8 ci][64 co][12] = 24 chunks of 1 KB (one wave instruction each), 3 per input channel; wave w moves chunks
    // w, w + 8, w + 16: the chunk part of the address is scalar, the LDS image is the slab as it lies in memory
    auto dma_w = [&](int ps, int ci0, f32x4* buf, int j0, int j1) __attribute__((always_inline)) {
        if (ABL & 32) return;
#pragma unroll
        for (int jj = j0; jj < j1; ++jj) {
            const int c = wave + 8 * jj;
            const int so = ((ps * g.CinP + ci0 + c / 3) * g.CoutP + co0) * 48 + (c % 3) * 1024;     // bytes, scalar
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, LDS_PTR(buf + XSZ + c * 64), 16, wvo, so, 0, 0);
        }
    };
    auto advance = [&](int& ps, int& ci0) __attribute__((always_inline)) {             // next slab, clamped at the last one
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
    int pA = 0, cA = 0;                          // next slab to load for
    int pW = 0, cW = 0;                          // next slab to DMA weights for
    issue_edge(pA, cA);
    issue_main(0, pA, cA);
    dma_w(pW, cW, smem, 0, WJ);
    advance(pW, cW);
    dma_w(pW, cW, smem + BUF, 0, WJ);
    advance(pW, cW);
    store_act(0, smem);
    advance(pA, cA);
    issue_edge(pA, cA);
    issue_main(0, pA, cA);                        // slab 1 (or a clamped copy of the last slab)
    store_act(0, smem + BUF);
    advance(pA, cA);
    issue_edge(pA, cA);
    issue_main(0, pA, cA);                        // slab 2
    __syncthreads();

    f32x4 av[1], bv[1][2];                        // (one operand set: the register file has no room for a prefetch set)
    av[0] = smem[aoff];
    bv[0][0] = smem[boff];
    bv[0][1] = smem[boff + 16 * 3];

    int rb = 0;                                   // ring slot of the slab being multiplied
    int pM = 0, cM = 0;                           // slab being multiplied (for the pass boundaries)
#define MFMA_GRP(c, pg)                                                                                               \
    if (!(ABL & 2) && mact) _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                            \
        acc[0][4 * (pg) + i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c][i], bv[c][0][i], acc[0][4 * (pg) + i], 0, 0, 0); \
        acc[1][4 * (pg) + i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c][i], bv[c][1][i], acc[1][4 * (pg) + i], 0, 0, 0); \
    }
#define READ_GRP(c, base, ks, pg)                                                   \
    if ((!(ABL & 4) || j == 0) && mact) {                                           \
        av[c] = (base)[aoff + (ks) * 4 * BN * 3 + (pg)];                            \
        bv[c][0] = (base)[boff + (ks) * 4 * NU * 3 + (pg)];                         \
        bv[c][1] = (base)[boff + (ks) * 4 * NU * 3 + 16 * 3 + (pg)];                \
    }
#define GROUP(cr, base, ks, pg, cm, pgm)       \
    READ_GRP(cr, base, ks, pg)                 \
    __builtin_amdgcn_sched_barrier(0);         \
    MFMA_GRP(cm, pgm)                          \
    __builtin_amdgcn_sched_barrier(0);
    // One slab: staging of slab j+2 from the register set (transform + write; its weights by DMA), then - spread over the
    // MFMA groups - the loads of slab j+3 into the freed set, the MFMAs of slab j.
#define SLAB(SET)                                                                                          \
    {                                                                                                      \
        const int rn = rb == 2 ? 0 : rb + 1;                                                               \
        const int rw = rn == 2 ? 0 : rn + 1;                                                               \
        const f32x4* Xs = smem + rb * BUF;                                                                 \
        const f32x4* Xn = smem + rn * BUF;                                                                 \
        f32x4* Xw = smem + rw * BUF;                                                                       \
        store_act(SET, Xw);                                                                                \
        advance(pA, cA);                                                                                   \
        issue_isc(SET, cA);                                                                                \
        pSs[SET] = pA;                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                 \
        issue_edge(pA, cA);                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                                 \
        dma_w(pW, cW, Xw, 0, WJ);                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                 \
        issue_rows(SET, pA, cA, 1, 3);                                                                     \
        MFMA_GRP(0, 0)                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                 \
        issue_rows(SET, pA, cA, 3, 5);                                                                     \
        GROUP(0, Xs, 0, 1, 0, 1)                                                                           \
        issue_halo(SET, cA);                                                                               \
        GROUP(0, Xs, 0, 2, 0, 2)                                                                           \
        GROUP(0, Xs, 1, 0, 0, 0)                                                                           \
        GROUP(0, Xs, 1, 1, 0, 1)                                                                           \
        advance(pW, cW);                                                                                   \
        GROUP(0, Xs, 1, 2, 0, 2)                                                                           \
        READ_GRP(0, Xn, 0, 0)                                                                              \
        /* slab j+2 complete (DMA + ds_write), slab j's buffer free.  Not __syncthreads(): its fence waits for vmcnt(0), */ \
        /* i.e. for the loads of slab j+3 issued during this slab.  Memory operations complete in order, so the slab      */ \
        /* issues: rows 0 / 5 of slab j+3, the three weight DMAs, then its 4 rows + halo: vmcnt(5) leaves those five in    */ \
        /* flight.                                                                                                         */ \
        if (ABL & 16) {                                                                                    \
        } else if (ABL & 1) {                                                                              \
            __syncthreads();                                                                               \
        } else {                                                                                           \
            asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)");                                                 \
            __builtin_amdgcn_s_barrier();                                                                  \
        }                                                                                                  \
        rb = rn;                                                                                           \
    }
    // pass boundary: carry the finished phases into the accumulators of the next pass (see the header)
    auto carry = [&](int pm) __attribute__((always_inline)) {
        if (pm == 0) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int p = 0; p < 6; ++p) {
                    const f32x4 m1 = acc[i][p], m2 = acc[i][6 + p];
                    acc[i][p] = 0.75f * m1 + 0.25f * m2;
                    acc[i][6 + p] = 0.25f * m1 + 0.75f * m2;
                }
        } else if (pm == 1) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int p = 0; p < 6; ++p) {
                    const f32x4 m3 = acc[i][p], m4 = acc[i][6 + p];
                    acc[i][p] = m3 + m4;
                    acc[i][6 + p] = 2.f * (m3 - m4);
                }
        }
    };
    for (int j = 0; j < nslab; ++j) {
        SLAB(0)
        cM += KC;
        if (cM >= g.CinP) {
            cM = 0;
            carry(pM);
            ++pM;
        }
    }
#undef MFMA_GRP
#undef READ_GRP
#undef GROUP
#undef SLAB

    // ---- output: rows fa (from M_0) and fa + dil (from M_5), time transform A4^T; lane = (unit l15, channels 4 lk .. 4 lk + 3)
    const bool has_os = a.oscale != nullptr, has_res = a.res != nullptr;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int rp = uw * 2 + i;                                   // segment
        const int t = t0 + 64 * (rp & tbm) + 4 * l15;
        const int fa = pair_row(rp >> g.tsh);
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

// ---------------------------------------------------------------------------------------------------------------------
// WIDE variant (128 / 256 output channels): tile = 128 output channels x 32 units (2 segments of 16 time units: 2 row pairs
// x 64 steps, or 1 row pair x 128 steps), wave w = 16-channel tile w x both segments: the same 2 x 12 accumulators.
// Why: the 2D input transform is recomputed by every channel tile of a layer and its ~100 vector instructions per wave
// and slab are paid in matrix-pipe time (fp32 MFMA and the vector ALU do not co-issue).  With 128-channel tiles a layer
// has half as many channel tiles, and the transform runs over SIXTEEN input channels at a time (thread = (ci, unit), all
// 512 threads busy) once per TWO 8-channel weight slabs: half the transform instructions per MFMA.
//   LDS 144 KB: activations X[2] of 16 ci x 32 units x 48 B = 24 KB each, weights W[2] of 8 ci x 128 co x 48 B = 48 KB.
//   step k (8 ci, 48 MFMAs per wave): MFMAs on X[S & 1] (S = k / 2) half k & 1 with W[k & 1]; at its top the DMA of
//   W(k + 1) goes to the buffer step k - 1 released; EVEN steps also transform super-slab S + 1 from the staging registers
//   into X[(S + 1) & 1] (free since the barrier of step k - 1) and then issue the loads of super-slab S + 2 into the same
//   registers - two steps before their transform; ODD steps have no staging work.  One barrier per step.
// BN = 96 (the 96-channel layers): 6 channel tiles x 2 segments = 12 (tile, segment) items for 8 waves - waves 0-3 take a
// channel tile with both segments, waves 4-7 one (tile, segment) each, so that the two waves of a SIMD (w, w + 4) hold three
// items: every SIMD does 3/4 of the matrix work of a 128-channel tile, none idles.
template <bool HAS_ISC, int BN>
__global__ __launch_bounds__(512, 1) void conv_wino45w_kernel(babe_conv_args a, Wino45Geom g, const float* __restrict__ wq) {
#if __HIP_DEVICE_COMPILE__
    constexpr int NTH = 512, KC = 8, KS = 16, NU = 32;
    static_assert(BN == 128 || BN == 96, "tile width");
    constexpr int XSZ = KS * NU * 3;                    // float4 per activation super-slab (16 ci x 32 units x 12 floats)
    constexpr int WSZ = KC * BN * 3;                    // float4 per weight slab
    constexpr int WJ = (WSZ + NTH - 1) / NTH;           // 6 (5 for BN = 96: 36 chunks) weight float4 per thread and step
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    f32x4* smem = reinterpret_cast<f32x4*>(smem_f);
    f32x4* const Xb = smem;                             // X[2]
    f32x4* const Wb = smem + 2 * XSZ;                   // W[2]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // items of this wave: channel tile cw, segment sg0 (and segment 1 when `two`)
    const int cw = (BN == 128 || wave < 4) ? wave : 4 + ((wave - 4) >> 1);
    const int sg0 = (BN == 128 || wave < 4) ? 0 : (wave - 4) & 1;
    const bool two = BN == 128 || wave < 4;              // wave-uniform
    const int l15 = lane & 15, lk = lane >> 4;
    const int b = blockIdx.z;
    const int co0 = blockIdx.y * BN;
    const int tile_t = blockIdx.x % g.tiles_t;
    const int rest = blockIdx.x / g.tiles_t;
    const int t0 = tile_t * (64 << g.tsh);
    const int tbm = (1 << g.tsh) - 1;
    const int P0 = rest * (2 >> g.tsh);
    const int npall = a.dil * g.npairs;
    auto pair_row = [&](int lp) __attribute__((always_inline)) {
        const int P = P0 + lp;
        const int c = P / g.npairs;
        return P < npall ? c + 2 * (P - c * g.npairs) * a.dil : a.F + 4 * a.dil;
    };
    const int nk = 3 * (g.CinP / KC);                   // steps (8-channel weight slabs)

    const float* p1 = a.in + (long)b * a.in_bs;
    const int cs1 = (int)a.in_cs;
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)p1, 0, a.Cin * cs1 * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)wq, 0, 3 * g.CinP * g.CoutP * 48, 0x00020000);

    // ---- staging constants: thread = (ci = 2 wave + (lane >> 5), unit = lane & 31 = segment (lane >> 4) & 1, time unit lane & 15).
    // The wave's first channel travels in the scalar offset, the second one is a per-lane loop constant (+ one channel stride).
    const int s_tu = lane & 15, s_sg = (lane >> 4) & 1, s_ch = lane >> 5;
    const int s_t = t0 + 64 * (s_sg & tbm) + 4 * s_tu;
    const int s_fa = pair_row(s_sg >> g.tsh);
    const unsigned chb = (unsigned)(s_ch * cs1 * 4);
    unsigned er[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        const int fr = s_fa + (r - 2) * a.dil;
        const bool ok = fr >= 0 && fr < a.F && s_t < a.T;
        er[r] = ok ? (unsigned)((fr * a.T + s_t) * 4) + chb : OOBH;
    }
    // halo: lane L < 48 loads the sample of (row L >> 3, channel (L >> 2) & 1, segment (L >> 1) & 1, side L & 1)
    unsigned ehalo = OOBH;
    {
        const int hr = lane >> 3, hc = (lane >> 2) & 1, hg = (lane >> 1) & 1, hs = lane & 1;
        const int fr = pair_row(hg >> g.tsh) + (hr - 2) * a.dil;
        const int th = t0 + 64 * (hg & tbm) + (hs ? 64 : -1);
        if (lane < 48 && fr >= 0 && fr < a.F && th >= 0 && th < a.T) ehalo = (unsigned)((fr * a.T + th) * 4 + hc * cs1 * 4);
    }
    const int hsrc = (4 * s_ch + 2 * s_sg + (s_tu == 15 ? 1 : 0)) * 4;     // row r adds 32 bytes (8 lanes per row)
    const int xlds = tid * 3;                                              // (ci_local * 32 + unit) * 3 with ci_local * 32 + unit = tid
    const int wvo = lane * 16;
    // BN = 96: a weight row (96 co x 48 B) is not a whole number of 1 KB chunks: per-lane source offsets (slot s of the slab
    // image [8 ci][96 co][3] lies at ci * CoutP * 48 + (s % 288) * 16 of the slab's first row), chunks beyond the 36 read nothing
    unsigned wvl[BN == 96 ? WJ : 1];
    if (BN == 96) {
#pragma unroll
        for (int jj = 0; jj < WJ; ++jj) {
            const int sl = (wave + 8 * jj) * 64 + lane;
            const int ci = sl / (BN * 3);
            wvl[BN == 96 ? jj : 0] = wave + 8 * jj < WSZ / 64 ? (unsigned)(ci * g.CoutP * 48 + (sl - ci * BN * 3) * 16) : OOBH;
        }
    }

    f32x4 acc[2][12];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int p = 0; p < 12; ++p) acc[i][p] = f32x4{0.f, 0.f, 0.f, 0.f};

    f32x4 xvm[4], xve[2];
    float xhl = 0.f, xsc = 1.f;
    int pS = 0;                                           // pass of the data held in the staging registers
    xve[0] = xve[1] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto issue_rows = [&](int ci0, int r0, int r1) __attribute__((always_inline)) {
        const int so = (ci0 + 2 * wave) * cs1 * 4;
#pragma unroll
        for (int r = 1; r < 5; ++r) {
            if (r < r0 || r >= r1) continue;
            xvm[r - 1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs1, er[r], so, 0));
        }
    };
    auto issue_halo = [&](int ci0) __attribute__((always_inline)) {
        const int so = (ci0 + 2 * wave) * cs1 * 4;
        xhl = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs1, ehalo, so, 0));
    };
    auto issue_isc = [&](int ci0) __attribute__((always_inline)) {
        if (HAS_ISC) {
            const float s0 = a.in_scale[(long)b * a.Cin + ci0 + 2 * wave], s1 = a.in_scale[(long)b * a.Cin + ci0 + 2 * wave + 1];
            xsc = s_ch ? s1 : s0;
        }
    };
    auto issue_edge = [&](int ps, int ci0) __attribute__((always_inline)) {
        const int so = (ci0 + 2 * wave) * cs1 * 4;
        const unsigned edge = ps == 2 ? 0u : OOBH;
        xve[0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs1, er[0] | edge, so, 0));
        xve[1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs1, er[5] | edge, so, 0));
    };
    auto dpp_shr1 = [](float old, float src) __attribute__((always_inline)) {
        asm("s_nop 1\n\tv_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(old) : "v"(src));
        return old;
    };
    auto dpp_shl1 = [](float old, float src) __attribute__((always_inline)) {
        asm("s_nop 1\n\tv_mov_b32_dpp %0, %1 row_shl:1 row_mask:0xf bank_mask:0xf" : "+v"(old) : "v"(src));
        return old;
    };
    auto fsel = [](int ps, unsigned c0, unsigned c1, unsigned c2) __attribute__((always_inline)) {
        return __builtin_bit_cast(float, ps == 0 ? c0 : (ps == 1 ? c1 : c2));
    };
    auto tt = [](const float (&E)[6], float (&U)[6]) {
        const float e = E[4] - 4.f * E[2], o = E[3] - 4.f * E[1];
        const float e2 = E[4] - E[2], o2 = E[3] - E[1];
        U[0] = 4.f * E[0] + (E[4] - 5.f * E[2]);
        U[1] = e + o;
        U[2] = e - o;
        U[3] = e2 + 2.f * o2;
        U[4] = e2 - 2.f * o2;
        U[5] = 4.f * E[1] + (E[5] - 5.f * E[3]);
    };
    // (same arithmetic, in the same order, as conv_wino45_kernel::store_act: the two kernels give bit-identical products)
    auto store_act = [&](f32x4* buf) __attribute__((always_inline)) {
        const int ps = pS;
        const f32x4 xv[6] = {xve[0], xvm[0], xvm[1], xvm[2], xvm[3], xve[1]};
        const float k2 = fsel(ps, 0x40800000u, 0x3f800000u, 0x40a00000u);
        const float k1 = fsel(ps, 0x40800000u, 0x40000000u, 0x40800000u);
        const float k3 = fsel(ps, 0x3f800000u, 0x40000000u, 0x40a00000u);
        const float x0 = fsel(ps, 0u, 0u, 0x40800000u), za = fsel(ps, 0x3f800000u, 0x3f800000u, 0u);
        const float x5 = fsel(ps, 0u, 0u, 0x3f800000u), ye = za;
        float xh[6];
#pragma unroll
        for (int r = 1; r < 5; ++r)
            asm("ds_bpermute_b32 %0, %1, %2 offset:%3" : "=v"(xh[r]) : "v"(hsrc), "v"(xhl), "n"(32 * r));
        asm("s_waitcnt lgkmcnt(0)" : "+v"(xh[1]), "+v"(xh[2]), "+v"(xh[3]), "+v"(xh[4]));
        float Ea[6], Eb[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            float d[6];
#pragma unroll
            for (int r = 1; r < 5; ++r)
                d[r] = j == 0 ? dpp_shr1(xh[r], xv[r][3]) : (j == 5 ? dpp_shl1(xh[r], xv[r][0]) : xv[r][j - 1]);
            const float e = d[4] - k2 * d[2];
            const float o = k3 * d[3] - k1 * d[1];
            Ea[j] = za * o + e;
            Eb[j] = ye * e - o;
        }
        if (ps == 2) {
            asm("ds_bpermute_b32 %0, %1, %2" : "=v"(xh[0]) : "v"(hsrc), "v"(xhl));
            asm("ds_bpermute_b32 %0, %1, %2 offset:160" : "=v"(xh[5]) : "v"(hsrc), "v"(xhl));
            asm("s_waitcnt lgkmcnt(0)" : "+v"(xh[0]), "+v"(xh[5]));
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const float d0 = j == 0 ? dpp_shr1(xh[0], xv[0][3]) : (j == 5 ? dpp_shl1(xh[0], xv[0][0]) : xv[0][j - 1]);
                const float d5 = j == 0 ? dpp_shr1(xh[5], xv[5][3]) : (j == 5 ? dpp_shl1(xh[5], xv[5][0]) : xv[5][j - 1]);
                Ea[j] = x0 * d0 + Ea[j];
                Eb[j] = x5 * d5 + Eb[j];
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
        tt(Ea, Ua);
        tt(Eb, Ub);
        buf[xlds] = f32x4{Ua[0], Ua[1], Ua[2], Ua[3]};
        buf[xlds + 1] = f32x4{Ua[4], Ua[5], Ub[0], Ub[1]};
        buf[xlds + 2] = f32x4{Ub[2], Ub[3], Ub[4], Ub[5]};
    };
    // weight slab [8 ci][128 co][12] = 48 chunks of 1 KB, 6 per input channel; wave w moves chunks w, w + 8, ..., w + 40
    auto dma_w = [&](int ps, int ci0, f32x4* buf) __attribute__((always_inline)) {
#pragma unroll
        for (int jj = 0; jj < WJ; ++jj) {
            const int c = wave + 8 * jj;
            if constexpr (BN == 128) {
                const int so = ((ps * g.CinP + ci0 + c / 6) * g.CoutP + co0) * 48 + (c % 6) * 1024;     // bytes, scalar
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, LDS_PTR(buf + c * 64), 16, wvo, so, 0, 0);
            } else {
                const int so = ((ps * g.CinP + ci0) * g.CoutP + co0) * 48;
                if (c < WSZ / 64)              // wave-uniform: the image has 36 chunks, waves 4-7 move four of them
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, LDS_PTR(buf + c * 64), 16, wvl[jj], so, 0, 0);
            }
        }
    };
    auto advance = [&](int& ps, int& ci0, int step) __attribute__((always_inline)) {    // next (super-)slab, clamped at the last one
        int nc = ci0 + step, np = ps;
        if (nc >= g.CinP) {
            nc = 0;
            ++np;
        }
        if (np <= 2) {
            ps = np;
            ci0 = nc;
        }
    };

    // operand addresses (float4 units): A in a weight slab, B in an activation super-slab (half h adds 8 * NU * 3)
    const int aoff = (lk * BN + cw * 16 + l15) * 3;
    const int boff = (lk * NU + sg0 * 16 + l15) * 3;

    // ---- prologue: super-slab 0 transformed into X[0], loads of super-slab 1 in flight, weight slab 0 in W[0]
    int pA = 0, cA = 0;                          // next super-slab to load
    int pW = 0, cW = 0;                          // next weight slab to DMA
    issue_edge(pA, cA);
    issue_rows(cA, 1, 5);
    issue_halo(cA);
    issue_isc(cA);
    pS = pA;
    dma_w(pW, cW, Wb);
    advance(pW, cW, KC);
    store_act(Xb);
    advance(pA, cA, KS);
    issue_edge(pA, cA);
    issue_rows(cA, 1, 5);
    issue_halo(cA);
    issue_isc(cA);
    pS = pA;
    __syncthreads();

    f32x4 av, bv[2];
    int pM = 0, cM = 0;                           // weight slab being multiplied (for the pass boundaries)
#define W_MFMA(pg)                                                                                   \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                  \
        acc[0][4 * (pg) + i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[0][i], acc[0][4 * (pg) + i], 0, 0, 0); \
        if (two) acc[1][4 * (pg) + i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[1][i], acc[1][4 * (pg) + i], 0, 0, 0); \
    }
#define W_READ(ks, pg)                                              \
    av = Ws[aoff + (ks) * 4 * BN * 3 + (pg)];                       \
    bv[0] = Xs[boff + (ks) * 4 * NU * 3 + (pg)];                    \
    if (two) bv[1] = Xs[boff + (ks) * 4 * NU * 3 + 16 * 3 + (pg)];
#define W_GROUP(ks, pg, pgm)                   \
    W_READ(ks, pg)                             \
    __builtin_amdgcn_sched_barrier(0);         \
    W_MFMA(pgm)                                \
    __builtin_amdgcn_sched_barrier(0);
    for (int k = 0; k < nk; k += 2) {
        // ---- even step: S = k / 2, half 0
        {
            const f32x4* Xs = Xb + ((k >> 1) & 1) * XSZ;
            const f32x4* Ws = Wb;                                    // W[k & 1] = W[0]
            f32x4* Xw = Xb + (((k >> 1) + 1) & 1) * XSZ;
            dma_w(pW, cW, Wb + WSZ);                                 // weight slab k + 1 -> W[1]
            advance(pW, cW, KC);
            __builtin_amdgcn_sched_barrier(0);
            W_READ(0, 0)
            store_act(Xw);                                           // super-slab S + 1
            advance(pA, cA, KS);
            issue_isc(cA);
            pS = pA;
            __builtin_amdgcn_sched_barrier(0);
            issue_edge(pA, cA);                                      // loads of super-slab S + 2: two steps ahead
            __builtin_amdgcn_sched_barrier(0);
            issue_rows(cA, 1, 3);
            W_MFMA(0)
            __builtin_amdgcn_sched_barrier(0);
            issue_rows(cA, 3, 5);
            W_GROUP(0, 1, 1)
            issue_halo(cA);
            W_GROUP(0, 2, 2)
            W_GROUP(1, 0, 0)
            W_GROUP(1, 1, 1)
            W_GROUP(1, 2, 2)
            // weight slab k + 1 landed (6 DMAs, older than the 7 loads of this step), X[(S + 1) & 1] written
            asm volatile("s_waitcnt vmcnt(7) lgkmcnt(0)");               /* (BN = 96: 5 DMAs; the count is of the loads after them) */
            __builtin_amdgcn_s_barrier();
        }
        // ---- odd step: half 1 of the same super-slab, W[1]
        {
            const f32x4* Xs = Xb + ((k >> 1) & 1) * XSZ + KC * NU * 3;
            const f32x4* Ws = Wb + WSZ;
            dma_w(pW, cW, Wb);                                       // weight slab k + 2 -> W[0]
            advance(pW, cW, KC);
            __builtin_amdgcn_sched_barrier(0);
            W_READ(0, 0)
            __builtin_amdgcn_sched_barrier(0);
            W_MFMA(0)
            __builtin_amdgcn_sched_barrier(0);
            W_GROUP(0, 1, 1)
            W_GROUP(0, 2, 2)
            W_GROUP(1, 0, 0)
            W_GROUP(1, 1, 1)
            W_GROUP(1, 2, 2)
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)");
            __builtin_amdgcn_s_barrier();
        }
        cM += KS;
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
#undef W_MFMA
#undef W_READ
#undef W_GROUP

    // ---- output (as conv_wino45_kernel): lane = (unit l15 of segment i, channels 4 lk .. 4 lk + 3 of the wave's 16)
    const bool has_os = a.oscale != nullptr, has_res = a.res != nullptr;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        if (i == 1 && !two) break;
        const int sg = i == 0 ? sg0 : 1;                              // segment of item i
        const int t = t0 + 64 * (sg & tbm) + 4 * l15;
        const int fa = pair_row(sg >> g.tsh);
#pragma unroll
        for (int row = 0; row < 2; ++row) {
            const int f = fa + row * a.dil;
            const bool pv = f < a.F && t < a.T;
            const long sp = pv ? (long)f * a.T + t : 0;
            float os[4];
            f32x4 rr[4];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int co = co0 + cw * 16 + 4 * lk + kk;           // < Cout: Cout % BN == 0
                os[kk] = has_os ? a.oscale[b * a.Cout + co] : 1.f;
                rr[kk] = has_res ? *reinterpret_cast<const f32x4*>(a.res + (long)b * a.res_bs + (long)co * a.res_cs + sp)
                                 : f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int co = co0 + cw * 16 + 4 * lk + kk;
                const float m0 = acc[i][6 * row + 0][kk], m1 = acc[i][6 * row + 1][kk], m2 = acc[i][6 * row + 2][kk];
                const float m3 = acc[i][6 * row + 3][kk], m4 = acc[i][6 * row + 4][kk], m5 = acc[i][6 * row + 5][kk];
                const float s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
                f32x4 y = {m0 + s12 + s34, d12 + 2.f * d34, s12 + 4.f * s34, d12 + 8.f * d34 + m5};
                const float sc = a.alpha * os[kk];
                y = y * sc + a.rbeta * rr[kk];
                if (pv) *reinterpret_cast<f32x4*>(a.out + (long)b * a.out_bs + (long)co * a.out_cs + sp) = y;
            }
        }
    }
#endif
}

// ---------------------------------------------------------------------------------------------------------------------
// WIDE variant, second generation (round 4): same tile (BN output channels x 32 units), same arithmetic in the same order as
// conv_wino45w_kernel (bit-identical results), different pipeline.  What the round-3 counters said about the first one: the
// matrix pipe is busy 0.54-0.62 of the time, the vector ALU work accounts for ~0.1 more, and for the remaining ~0.3 BOTH waves
// of a SIMD are parked.  Three causes, all structural:
//  (1) ONE operand register set: a wave reads the operands of MFMA group g + 1 only after the last MFMA of group g has issued
//      and then waits a full LDS latency; the two waves of a SIMD are released together by the step barrier, alternate their
//      MFMAs and therefore reach their read-and-wait at the same time, every group (6 per step);
//  (2) a barrier per 8-channel step, needed only because the weight slab is shared workgroup-wide in LDS - although in this
//      tiling wave w is the ONLY reader of the weights of channel tile w;
//  (3) all 8 waves issue their 6 weight DMAs back to back at the top of every step.
// Here: (1) two operand sets - the reads of group g + 1 are issued before the MFMAs of group g; (2) WAVE-PRIVATE weights: every
// wave DMAs exactly the 16-channel slice it multiplies with (4 input channels x 16 output channels x 48 B = 3 chunks of 1 KB
// per slot, per-lane gather offsets) into its own part of a 4-slot ring and waits for it with a counted s_waitcnt - no barrier
// for weights, three slots (72 MFMAs per wave, ~2 us) of DMA latency budget instead of one step; the only shared data are the
// transformed activations X[2], written once per 16-channel super-slab: ONE barrier per 96 MFMAs per wave, placed before the
// last MFMA group of the super-slab, whose operands are in registers by then, so that the first operand reads of the next
// super-slab run behind those MFMAs; (3) one DMA instruction per MFMA group.  The halo samples' ds_bpermute are issued one
// MFMA group before the transform that consumes them.  LDS: X[2] x 24 KB + ring 4 x 8 waves x 3 KB = 144 KB.
// vmcnt bookkeeping (vector-memory operations complete in order).  Per super-slab and wave, in issue order:
//   g0: D E E R R | g1: D R R | g2: D H | g3 .. g11: D          (D weight DMA chunk, E edge row, R row, H halo: 19 operations)
// the DMAs of group 3i + j move chunk j of weight slot 4S + i + 3 into ring position (i + 3) & 3.  Operand reads of a new
// slot are issued in the last group of the previous one: slot 1 at g2 (its last chunk was issued at g8 of the previous
// super-slab: 11 younger operations without the edge rows), slot 2 at g5 (g11 of the previous one: 11), slot 3 at g8 (g2: 7), slot 0 of the next
// super-slab at g11 (g5: 6).
// (-DBABE_W45X_NUM_VGPR=96 caps the kernel at 192 registers - the attribute counts halves of the unified file - so that two
// 64-register waves of the other lane's element-wise kernels would fit beside the two conv waves of a SIMD.  Measured, round 4,
// same box: 76-136 spilled registers, a few of them inside the loop, whole-job throughput 1.44 instead of 2.06 audio-sec/s.
// Default 128 = no cap.)
#ifndef BABE_W45X_NUM_VGPR
#define BABE_W45X_NUM_VGPR 128
#endif
template <bool HAS_ISC, int BN>
__global__ __launch_bounds__(512, 1) __attribute__((amdgpu_num_vgpr(BABE_W45X_NUM_VGPR))) void conv_wino45x_kernel(babe_conv_args a, Wino45Geom g, const float* __restrict__ wq) {
#if __HIP_DEVICE_COMPILE__
    constexpr int KS = 16, KQ = 4, NU = 32;
    static_assert(BN == 128 || BN == 96, "tile width");
    constexpr int XSZ = KS * NU * 3;                    // float4 per activation super-slab (16 ci x 32 units x 12 floats)
    constexpr int WWV = KQ * 16 * 3;                    // float4 per wave and ring slot (4 ci x 16 co x 12 floats = 3 KB)
    constexpr int WSL = 8 * WWV;                        // float4 per ring slot (8 waves)
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    f32x4* smem = reinterpret_cast<f32x4*>(smem_f);
    f32x4* const Xb = smem;                             // X[2]
    f32x4* const Wb = smem + 2 * XSZ;                   // ring[4][8 waves][4 ci][16 co][3]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cw = (BN == 128 || wave < 4) ? wave : 4 + ((wave - 4) >> 1);
    const int sg0 = (BN == 128 || wave < 4) ? 0 : (wave - 4) & 1;
    const bool two = BN == 128 || wave < 4;              // wave-uniform
    const int l15 = lane & 15, lk = lane >> 4;
    const int b = blockIdx.z;
    const int co0 = blockIdx.y * BN;
    const int tile_t = blockIdx.x % g.tiles_t;
    const int rest = blockIdx.x / g.tiles_t;
    const int t0 = tile_t * (64 << g.tsh);
    const int tbm = (1 << g.tsh) - 1;
    const int P0 = rest * (2 >> g.tsh);
    const int npall = a.dil * g.npairs;
    auto pair_row = [&](int lp) __attribute__((always_inline)) {
        const int P = P0 + lp;
        const int c = P / g.npairs;
        return P < npall ? c + 2 * (P - c * g.npairs) * a.dil : a.F + 4 * a.dil;
    };
    const int NS = 3 * (g.CinP / KS);                   // super-slabs

    const float* p1 = a.in + (long)b * a.in_bs;
    const int cs1 = (int)a.in_cs;
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)p1, 0, a.Cin * cs1 * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)wq, 0, 3 * g.CinP * g.CoutP * 48, 0x00020000);

    // ---- staging constants (as conv_wino45w_kernel): thread = (ci = 2 wave + (lane >> 5), unit = lane & 31)
    const int s_tu = lane & 15, s_sg = (lane >> 4) & 1, s_ch = lane >> 5;
    const int s_t = t0 + 64 * (s_sg & tbm) + 4 * s_tu;
    const int s_fa = pair_row(s_sg >> g.tsh);
    const unsigned chb = (unsigned)(s_ch * cs1 * 4);
    unsigned er[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        const int fr = s_fa + (r - 2) * a.dil;
        const bool ok = fr >= 0 && fr < a.F && s_t < a.T;
        er[r] = ok ? (unsigned)((fr * a.T + s_t) * 4) + chb : OOBH;
    }
    unsigned ehalo = OOBH;
    {
        const int hr = lane >> 3, hc = (lane >> 2) & 1, hg = (lane >> 1) & 1, hs = lane & 1;
        const int fr = pair_row(hg >> g.tsh) + (hr - 2) * a.dil;
        const int th = t0 + 64 * (hg & tbm) + (hs ? 64 : -1);
        if (lane < 48 && fr >= 0 && fr < a.F && th >= 0 && th < a.T) ehalo = (unsigned)((fr * a.T + th) * 4 + hc * cs1 * 4);
    }
    const int hsrc = (4 * s_ch + 2 * s_sg + (s_tu == 15 ? 1 : 0)) * 4;
    const int xlds = tid * 3;
    // weight DMA: chunk j of a slot = bytes [1024 j, 1024 j + 1024) of the wave's image [4 ci][16 co][48 B]; lane -> byte
    // o = 1024 j + 16 lane = (ci, 768-byte row position) -> source offset ci * CoutP * 48 + position; the slot / channel-tile part
    // of the address is the scalar offset sW (+ 4 CoutP 48 per slot; beyond the last slot: out of range, zeros)
    unsigned wvl[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int o = j * 1024 + lane * 16;
        const int ci = o / 768;
        wvl[j] = (unsigned)(ci * g.CoutP * 48 + (o - ci * 768));
    }
    const int wstep = KQ * g.CoutP * 48;
    const int sWend = 3 * g.CinP * g.CoutP * 48;
    int sW = (co0 + cw * 16) * 48;
    auto w_next = [&]() __attribute__((always_inline)) {
        sW += wstep;
        sW = sW < sWend ? sW : sWend;
    };
    f32x4* const Ww = Wb + wave * WWV;                   // this wave's part of ring position 0
    // (ablation bits of this kernel, profiling builds only - results are wrong: 0x2000 no weight DMA, 0x4000 no row / edge / halo
    // loads, 0x8000 no transform (halo permutes, vector work, LDS stores), 0x10000 no barrier)
#define X_DMA(rp, j) \
    if (!(ABL & 0x2000)) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, LDS_PTR(Ww + (rp) * WSL + (j) * 64), 16, wvl[j], sW, 0, 0);
#define X_FENCE0 __builtin_amdgcn_sched_barrier(0);
#if (ABL & 0x100000)
#define X_WAITVM(n)                                      /* (ablation: no counted waits at all - racy, timing only) */
#elif (ABL & 256)
#define X_WAITVM(n) asm volatile("s_waitcnt vmcnt(0)");
#elif (ABL & 512)
#define X_WAITVM(n) asm volatile("s_waitcnt vmcnt(0)"); __builtin_amdgcn_s_barrier();
#else
#define X_WAITVM(n) asm volatile("s_waitcnt vmcnt(" #n ")");
#endif
#if (ABL & 0x20000)
#define X_WAITVM13 X_WAITVM(13)
#else
#define X_WAITVM13 X_WAITVM(11)
#endif

    f32x4 xvm[4], xve[2];
    float xhl = 0.f, xsc = 1.f;
    float xh[6];
    int pS = 0;
    xve[0] = xve[1] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto issue_rows = [&](int ci0, int r0, int r1) __attribute__((always_inline)) {
        if (ABL & 0x4000) return;
        const int so = (ci0 + 2 * wave) * cs1 * 4;
#pragma unroll
        for (int r = 1; r < 5; ++r) {
            if (r < r0 || r >= r1) continue;
            xvm[r - 1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs1, er[r], so, 0));
        }
    };
    auto issue_halo = [&](int ci0) __attribute__((always_inline)) {
        if (ABL & 0x4000) return;
        const int so = (ci0 + 2 * wave) * cs1 * 4;
        xhl = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs1, ehalo, so, 0));
    };
    auto issue_isc = [&](int ci0) __attribute__((always_inline)) {
        if (HAS_ISC) {
            const float s0 = a.in_scale[(long)b * a.Cin + ci0 + 2 * wave], s1 = a.in_scale[(long)b * a.Cin + ci0 + 2 * wave + 1];
            xsc = s_ch ? s1 : s0;
        }
    };
    auto issue_edge = [&](int ps, int ci0) __attribute__((always_inline)) {
        if (ABL & 0x4000) return;
        const int so = (ci0 + 2 * wave) * cs1 * 4;
#if !(ABL & 0x20000)
        // the two edge-row loads only in the pass that uses them (a wave-uniform branch; the hand-counted waits assume they are
        // absent: 11 where the always-issued form - ABL bit 0x20000 - has 13).  A vector-memory instruction costs this kernel
        // 1-2 % of its time whether it moves data or not (ablations): +2 % over all layers, +5 % on the 96-channel ones.
        if (ps == 2) {
            xve[0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs1, er[0], so, 0));
            xve[1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs1, er[5], so, 0));
        }
#else
        const unsigned edge = ps == 2 ? 0u : OOBH;
        xve[0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs1, er[0] | edge, so, 0));
        xve[1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs1, er[5] | edge, so, 0));
#endif
    };
    auto dpp_shr1 = [](float old, float src) __attribute__((always_inline)) {
        asm("s_nop 1\n\tv_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(old) : "v"(src));
        return old;
    };
    auto dpp_shl1 = [](float old, float src) __attribute__((always_inline)) {
        asm("s_nop 1\n\tv_mov_b32_dpp %0, %1 row_shl:1 row_mask:0xf bank_mask:0xf" : "+v"(old) : "v"(src));
        return old;
    };
    auto fsel = [](int ps, unsigned c0, unsigned c1, unsigned c2) __attribute__((always_inline)) {
        return __builtin_bit_cast(float, ps == 0 ? c0 : (ps == 1 ? c1 : c2));
    };
    auto tt = [](const float (&E)[6], float (&U)[6]) {
        const float e = E[4] - 4.f * E[2], o = E[3] - 4.f * E[1];
        const float e2 = E[4] - E[2], o2 = E[3] - E[1];
        U[0] = 4.f * E[0] + (E[4] - 5.f * E[2]);
        U[1] = e + o;
        U[2] = e - o;
        U[3] = e2 + 2.f * o2;
        U[4] = e2 - 2.f * o2;
        U[5] = 4.f * E[1] + (E[5] - 5.f * E[3]);
    };
    // halo samples of the rows held in the staging registers -> xh[] (issued one MFMA group before the transform; the
    // statement that consumes them waits).  Rows 0 / 5 are used by pass 2 only; their two bpermutes are issued in every pass
    // (an LDS instruction does not cost matrix-pipe time, a branch here would cost hipcc's register shuffling)
    auto halo_permute = [&]() __attribute__((always_inline)) {
        if (ABL & 0x8000) return;
#pragma unroll
        for (int r = 0; r < 6; ++r)
            asm volatile("ds_bpermute_b32 %0, %1, %2 offset:%3" : "=v"(xh[r]) : "v"(hsrc), "v"(xhl), "n"(32 * r));
    };
    // (same arithmetic, in the same order, as conv_wino45_kernel::store_act: bit-identical products)
    auto store_act = [&](f32x4* buf) __attribute__((always_inline)) {
        if (ABL & 0x8000) return;
        const int ps = pS;
        const f32x4 xv[6] = {xve[0], xvm[0], xvm[1], xvm[2], xvm[3], xve[1]};
        const float k2 = fsel(ps, 0x40800000u, 0x3f800000u, 0x40a00000u);
        const float k1 = fsel(ps, 0x40800000u, 0x40000000u, 0x40800000u);
        const float k3 = fsel(ps, 0x3f800000u, 0x40000000u, 0x40a00000u);
        const float x0 = fsel(ps, 0u, 0u, 0x40800000u), za = fsel(ps, 0x3f800000u, 0x3f800000u, 0u);
        const float x5 = fsel(ps, 0u, 0u, 0x3f800000u), ye = za;
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xh[0]), "+v"(xh[1]), "+v"(xh[2]), "+v"(xh[3]), "+v"(xh[4]), "+v"(xh[5]));
        float Ea[6], Eb[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            float d[6];
#pragma unroll
            for (int r = 1; r < 5; ++r)
                d[r] = (ABL & 0x80000) ? xv[r][j == 0 ? 0 : (j == 5 ? 3 : j - 1)]
                                        : (j == 0 ? dpp_shr1(xh[r], xv[r][3]) : (j == 5 ? dpp_shl1(xh[r], xv[r][0]) : xv[r][j - 1]));
            const float e = d[4] - k2 * d[2];
            const float o = k3 * d[3] - k1 * d[1];
            Ea[j] = za * o + e;
            Eb[j] = ye * e - o;
        }
        if (ps == 2) {
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const float d0 = j == 0 ? dpp_shr1(xh[0], xv[0][3]) : (j == 5 ? dpp_shl1(xh[0], xv[0][0]) : xv[0][j - 1]);
                const float d5 = j == 0 ? dpp_shr1(xh[5], xv[5][3]) : (j == 5 ? dpp_shl1(xh[5], xv[5][0]) : xv[5][j - 1]);
                Ea[j] = x0 * d0 + Ea[j];
                Eb[j] = x5 * d5 + Eb[j];
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
        if (ABL & 0x80000) {                              // (ablation: no time transform, no halo assembly - what a producer-side
#pragma unroll                                            //  time transform would leave in this kernel; results are wrong)
            for (int j = 0; j < 6; ++j) {
                Ua[j] = Ea[j];
                Ub[j] = Eb[j];
            }
        } else {
            tt(Ea, Ua);
            tt(Eb, Ub);
        }
        buf[xlds] = f32x4{Ua[0], Ua[1], Ua[2], Ua[3]};
        buf[xlds + 1] = f32x4{Ua[4], Ua[5], Ub[0], Ub[1]};
        buf[xlds + 2] = f32x4{Ub[2], Ub[3], Ub[4], Ub[5]};
    };
    auto advance = [&](int& ps, int& ci0) __attribute__((always_inline)) {    // next super-slab, clamped at the last one
        int nc = ci0 + KS, np = ps;
        if (nc >= g.CinP) {
            nc = 0;
            ++np;
        }
        if (np <= 2) {
            ps = np;
            ci0 = nc;
        }
    };

    // operand addresses (float4 units): A in the wave's ring part (slot i of a super-slab = ring position i), B in X
    const int aoff = (lk * 16 + l15) * 3;
    const int boff = (lk * NU + sg0 * 16 + l15) * 3;

    // ---- prologue: super-slab 0 transformed into X[0], rows of super-slab 1 in flight, weight slots 0-2 in ring 0-2
    int pA = 0, cA = 0;
    issue_edge(pA, cA);
    issue_rows(cA, 1, 5);
    issue_halo(cA);
    issue_isc(cA);
    pS = pA;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        X_DMA(i, 0)
        X_DMA(i, 1)
        X_DMA(i, 2)
        w_next();
    }
    X_FENCE0
    halo_permute();
    store_act(Xb);
    X_FENCE0
    advance(pA, cA);
    issue_edge(pA, cA);
    issue_rows(cA, 1, 5);
    issue_halo(cA);
    issue_isc(cA);
    pS = pA;
    X_FENCE0
    // In flight here, oldest first: 9 weight DMAs (slots 0-2), then the 5 loads of super-slab 1 (4 rows + halo; the edge rows are
    // issued in pass 2 only, which the prologue never reaches).  vmcnt(7) = everything but the 7 youngest: slot 0, slot 1 and the
    // first chunk of slot 2 have landed - slot 0 is what the first operand read below needs; slots 1 and 2 are covered by the
    // loop's own counted waits at g2 / g5 (13 resp. 11 younger operations by then).
    asm volatile("s_waitcnt vmcnt(7) lgkmcnt(0)");
    __builtin_amdgcn_s_barrier();

    X_FENCE0
    f32x4 acc[2][12];                                     // (zeroed after the prologue's transform: 96 registers less while it runs)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int p = 0; p < 12; ++p) acc[i][p] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 av[2], bv[2][2];
#define X_MFMA(c, pg)                                                                                \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                  \
        acc[0][4 * (pg) + i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c][i], bv[c][0][i], acc[0][4 * (pg) + i], 0, 0, 0); \
        if (two) acc[1][4 * (pg) + i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c][i], bv[c][1][i], acc[1][4 * (pg) + i], 0, 0, 0); \
    }
#define X_MFMA1(c, pg, i)                                                                            \
    acc[0][4 * (pg) + (i)] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c][i], bv[c][0][i], acc[0][4 * (pg) + (i)], 0, 0, 0); \
    if (two) acc[1][4 * (pg) + (i)] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c][i], bv[c][1][i], acc[1][4 * (pg) + (i)], 0, 0, 0);
#define X_READ(c, Xp, sl, pg)                                       \
    av[c] = Ww[(sl) * WSL + aoff + (pg)];                           \
    bv[c][0] = (Xp)[boff + (sl) * KQ * NU * 3 + (pg)];              \
    if (two) bv[c][1] = (Xp)[boff + (sl) * KQ * NU * 3 + 16 * 3 + (pg)];
#define X_FENCE __builtin_amdgcn_sched_barrier(0);
    // group g: [weight DMA chunk] [staging loads] [wait for a new slot's weights] [reads of group g + 1 into the other operand
    // set] [transform] [8 MFMAs of group g].  STAGGER: the two waves of a SIMD (w, w + 4) do their staging work - transform
    // (~90 vector instructions and two LDS round trips during which the wave issues no MFMA) and row loads - half a super-slab
    // apart: waves 0-3 in groups 0-2 (EARLY), waves 4-7 in groups 6-8 (LATE), so that one partner multiplies while the other
    // transforms - OPTIONAL, see below.  Two copies of the loop (generic lambda): every count is a compile-time constant.
    //   vmcnt, EARLY (see the header): 13, 13, 6, 6 at g2, g5, g8, g11;
    //   LATE issues  g0-g5: D | g6: D E E R R | g7: D R R | g8: D H | g9-g11: D  ->  6, 6, 13, 13.
#define X_G(g, rp, j, cr, sl, pgr, cm, pgm)                                                                     \
    X_DMA(rp, j)                                                                                                \
    X_FENCE                                                                                                     \
    if constexpr ((g) == SA + 1) {                                                                              \
        issue_rows(cA, 3, 5);                                                                                   \
        X_FENCE                                                                                                 \
    }                                                                                                           \
    if constexpr ((g) == SA + 2) {                                                                              \
        issue_halo(cA);                                                                                         \
        X_FENCE                                                                                                 \
    }                                                                                                           \
    if constexpr ((j) == 2) {                                                                                   \
        w_next();                                                                                               \
        if constexpr (((g) < 6) != LATE) { X_WAITVM13 } else { X_WAITVM(6) }                                    \
    }                                                                                                           \
    if constexpr (!SPREAD || (g) == SA) {                                                                       \
        X_READ(cr, Xs, sl, pgr)                                                                                 \
        X_FENCE                                                                                                 \
    }                                                                                                           \
    if constexpr ((g) == SA) {                                                                                  \
        /* transform of super-slab S + 1 (rows loaded one super-slab ago), then the loads of super-slab S + 2 */ \
        store_act(Xw);                                                                                          \
        advance(pA, cA);                                                                                        \
        issue_isc(cA);                                                                                          \
        pS = pA;                                                                                                \
        X_FENCE                                                                                                 \
        issue_edge(pA, cA);                                                                                     \
        issue_rows(cA, 1, 3);                                                                                   \
        X_FENCE                                                                                                 \
    }                                                                                                           \
    if constexpr (LATE && (g) == 5) {                                                                           \
        halo_permute();                                                                                         \
        X_FENCE                                                                                                 \
    }                                                                                                           \
    if constexpr (!SPREAD || (g) == SA) {                                                                       \
        X_MFMA(cm, pgm)                                                                                         \
    } else {                                                                                                    \
        /* the three operand reads of group g + 1 one at a time BETWEEN the MFMA pairs of group g (tools/mfma_feed.hip: the */ \
        /* bare loop runs at 0.991 of the peak this way, 0.969 with the reads clustered in front of the eight MFMAs) */ \
        X_MFMA1(cm, pgm, 0)                                                                                     \
        X_FENCE                                                                                                 \
        av[cr] = Ww[(sl) * WSL + aoff + (pgr)];                                                                 \
        X_FENCE                                                                                                 \
        X_MFMA1(cm, pgm, 1)                                                                                     \
        X_FENCE                                                                                                 \
        bv[cr][0] = Xs[boff + (sl) * KQ * NU * 3 + (pgr)];                                                      \
        X_FENCE                                                                                                 \
        X_MFMA1(cm, pgm, 2)                                                                                     \
        X_FENCE                                                                                                 \
        if (two) bv[cr][1] = Xs[boff + (sl) * KQ * NU * 3 + 16 * 3 + (pgr)];                                    \
        X_FENCE                                                                                                 \
        X_MFMA1(cm, pgm, 3)                                                                                     \
    }                                                                                                           \
    X_FENCE
    auto run = [&](auto late_c) __attribute__((always_inline)) {
        constexpr bool LATE = decltype(late_c)::value;
        constexpr int SA = LATE ? 6 : 0;
        constexpr bool SPREAD = (ABL & 0x40000) != 0;
        X_READ(0, Xb, 0, 0)
        if constexpr (!LATE) halo_permute();              // (super-slab 1's halo: waits for its load)
        int cM = 0, pM = 0;
        for (int S = 0; S < NS; ++S) {
            const f32x4* Xs = Xb + (S & 1) * XSZ;
            f32x4* Xw = Xb + ((S + 1) & 1) * XSZ;
            X_G(0, 3, 0, 1, 0, 1, 0, 0)
            X_G(1, 3, 1, 0, 0, 2, 1, 1)
            X_G(2, 3, 2, 1, 1, 0, 0, 2)                   // first operands of slot 1
            X_G(3, 0, 0, 0, 1, 1, 1, 0)
            X_G(4, 0, 1, 1, 1, 2, 0, 1)
            X_G(5, 0, 2, 0, 2, 0, 1, 2)                   // first operands of slot 2
            X_G(6, 1, 0, 1, 2, 1, 0, 0)
            X_G(7, 1, 1, 0, 2, 2, 1, 1)
            X_G(8, 1, 2, 1, 3, 0, 0, 2)                   // first operands of slot 3
            X_G(9, 2, 0, 0, 3, 1, 1, 0)
            X_G(10, 2, 1, 1, 3, 2, 0, 1)
            // g11: X[(S + 1) & 1] is complete and nobody reads X[S & 1] any more (the operands of this group are in
            // registers): barrier, then the first operand reads of the next super-slab (+ the EARLY waves' halo permutes of
            // super-slab S + 2) run behind this group's MFMAs
            X_DMA(2, 2)
            w_next();
            X_FENCE
            if constexpr (LATE) { X_WAITVM13 } else { X_WAITVM(6) }
            asm volatile("s_waitcnt lgkmcnt(0)");
            if (!(ABL & 0x10000)) __builtin_amdgcn_s_barrier();
            X_FENCE
            X_READ(0, Xw, 0, 0)
            if constexpr (!LATE) halo_permute();
            X_FENCE
            X_MFMA(1, 2)
            X_FENCE
            cM += KS;
            if (cM >= g.CinP) {
                // pass boundary (twice per tile): the finished phases are carried IN PLACE.  Written as inline assembly with
                // tied operands: as C++ expressions hipcc computed the 96 results into 96 fresh registers and copied them back
                // on the loop's back edge - 255 registers for a kernel whose loop needs 175.  The rounding is that of the
                // expressions of conv_wino45_kernel::carry as hipcc contracts them: bit-identical.  s_nop: the MFMAs of the last
                // group may still be writing the accumulators, and the hazard recogniser does not look inside inline assembly.
                cM = 0;
                asm volatile("s_nop 15\n\ts_nop 15");
                if (pM == 0) {
                    float c75 = 0.75f;
                    asm volatile("" : "+s"(c75));
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int p = 0; p < 6; ++p)
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                float t0;
                                // m1' = fma(0.75, m1, 0.25 m2)  (0.25 m2 exact);  m2' = fma(0.25, m1, round(0.75 m2)): the
                                // contraction hipcc chose for the C++ expressions of the other two kernels - kept, bit for bit
                                asm volatile("v_mul_f32 %2, 0x3e800000, %1\n\tv_mul_f32 %1, 0x3f400000, %1\n\t"
                                             "v_fmac_f32 %1, 0x3e800000, %0\n\tv_fma_f32 %0, %3, %0, %2"
                                             : "+v"(acc[i][p][e]), "+v"(acc[i][6 + p][e]), "=&v"(t0) : "s"(c75));
                            }
                } else if (pM == 1) {
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int p = 0; p < 6; ++p)
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                float t0;
                                asm volatile("v_sub_f32 %2, %0, %1\n\tv_add_f32 %0, %0, %1\n\tv_mul_f32 %1, 2.0, %2"
                                             : "+v"(acc[i][p][e]), "+v"(acc[i][6 + p][e]), "=&v"(t0));
                            }
                }
                asm volatile("s_nop 4");
                ++pM;
            }
        }
    };
    // Measured (round 4, same box, six layer shapes): staggered 2.42 ms, not staggered 2.40 ms - no gain, as round 3 found for
    // the first-generation kernel (the vector work costs matrix-pipe time whichever wave does it, and the LDS round trips it
    // exposes are short).  Default: every wave EARLY (one copy of the loop); -DABL=1024 builds the staggered form.
#if (ABL & 1024)
    if (wave >= 4) run(std::true_type{});
    else run(std::false_type{});
#else
    run(std::false_type{});
#endif
#undef X_G
#undef X_MFMA
#undef X_MFMA1
#undef X_READ
#undef X_FENCE
#undef X_DMA
#undef X_FENCE0
#undef X_WAITVM
#undef X_WAITVM13
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)");          // (the tail's clamped loads / out-of-range DMAs: nothing may land after the exit)

    // ---- output (as conv_wino45w_kernel)
    const bool has_os = a.oscale != nullptr, has_res = a.res != nullptr;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        if (i == 1 && !two) break;
        const int sg = i == 0 ? sg0 : 1;
        const int t = t0 + 64 * (sg & tbm) + 4 * l15;
        const int fa = pair_row(sg >> g.tsh);
#pragma unroll
        for (int row = 0; row < 2; ++row) {
            const int f = fa + row * a.dil;
            const bool pv = f < a.F && t < a.T;
            const long sp = pv ? (long)f * a.T + t : 0;
            float os[4];
            f32x4 rr[4];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int co = co0 + cw * 16 + 4 * lk + kk;           // < Cout: Cout % BN == 0
                os[kk] = has_os ? a.oscale[b * a.Cout + co] : 1.f;
                rr[kk] = has_res ? *reinterpret_cast<const f32x4*>(a.res + (long)b * a.res_bs + (long)co * a.res_cs + sp)
                                 : f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int co = co0 + cw * 16 + 4 * lk + kk;
                const float m0 = acc[i][6 * row + 0][kk], m1 = acc[i][6 * row + 1][kk], m2 = acc[i][6 * row + 2][kk];
                const float m3 = acc[i][6 * row + 3][kk], m4 = acc[i][6 * row + 4][kk], m5 = acc[i][6 * row + 5][kk];
                const float s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
                f32x4 y = {m0 + s12 + s34, d12 + 2.f * d34, s12 + 4.f * s34, d12 + 8.f * d34 + m5};
                const float sc = a.alpha * os[kk];
                y = y * sc + a.rbeta * rr[kk];
                if (pv) *reinterpret_cast<f32x4*>(a.out + (long)b * a.out_bs + (long)co * a.out_cs + sp) = y;
            }
        }
    }
#endif
}

// (Round 4 also built a THIRD generation - the weights not through LDS at all: wave w is the only consumer of its tile's weights
// and its A fragment of a slot is exactly three coalesced 16-byte loads per lane, so the A operands were loaded global ->
// registers into a ring, LDS holding X only.  With hipcc's own waitcnt insertion the ring got vmcnt(0) / vmcnt(1) in front of
// the transform (write-after-write and branch-merge conservatism): slower (96-channel layers 678 vs 575 us).  With every loop
// load as inline assembly and hand-counted waits, a 4-slot ring spilled inside the loop and a 2-slot ring was 2 % faster than
// this kernel at best: the 10 % the weight DMA costs - ablation, same box: DMA off -10 %, row loads + transform off -22 %, barrier
// off 0 %, all off -25 % = the MFMA + operand-read floor - is the ISSUE of its vector-memory instructions, not its LDS traffic,
// and their number per MFMA is fixed by the tile: 128 B of weights per MFMA.  Removed again.)
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
    auto al16 = [](const void* p) __attribute__((always_inline)) { return ((uintptr_t)p & 15) == 0; };
    if (a.KH != 5 || a.KW != 3 || a.T % 4 != 0 || a.T < 64 || a.dil < 1) return 0;   // (tiles are 64 time steps wide)
    if (a.Cin < 16 || a.Cin % 16 != 0 || a.Cout < 33) return 0;   // (the channel part of a load address is a scalar offset,
    // which the buffer range check does not cover: no padded input channels; an even number of 8-channel slabs per pass: the
    // slab loop is unrolled by two; few-channel convs run on conv_fewco / direct)
    if (!al16(a.in) || a.in_bs % 4 || a.in_cs % 4) return 0;
    if (a.in2) return 0;                                     // one source only (csrc/conv_wino45.hip, descriptors)
    if (!al16(a.out) || a.out_bs % 4 || a.out_cs % 4) return 0;
    if (a.res && (!al16(a.res) || a.res_bs % 4 || a.res_cs % 4)) return 0;
    const long lim = 0x3fffffffL / 4;                        // source views below 1 GiB per batch item (OOBH arithmetic)
    if ((long)a.Cin * a.in_cs >= lim) return 0;
    if ((long)a.F * a.T >= lim) return 0;
    if (36L * ((a.Cin + 7) / 8 * 8) * ((a.Cout + 63) / 64 * 64) * 4 >= 0x7fffffffL) return 0;
    return 1;
}

// wide variant (128-channel tiles x 32 units): eligible shapes and its tile shape (2 row pairs x 64 steps or 1 x 128)
static inline double wino45w_fill(const babe_conv_args& a, int tsh) {
    const int ppt = 2 >> tsh, tlen = 64 << tsh;
    const long npall = (long)a.dil * (((a.F + a.dil - 1) / a.dil + 1) / 2);
    const long tiles = (npall + ppt - 1) / ppt;
    return ((double)a.F / (2.0 * ppt * tiles)) * ((double)a.T / ((double)tlen * ((a.T + tlen - 1) / tlen)));
}
static inline int wino45w_ok(const babe_conv_args& a) {
    static const char* ov = getenv("BABE_CONV_WINO45W");
    if (ov && ov[0] == '0') return 0;
    if ((a.Cout % 128 != 0 && a.Cout % 96 != 0) || a.Cin % 16 != 0) return 0;
    const double f0 = wino45w_fill(a, 0), f1 = wino45w_fill(a, 1);
    // as full as the 64-channel tiling of the same launch (which has 64 units to deal)
    return (f0 > f1 ? f0 : f1) + 1e-9 >= wino45_fill(a, wino45_best_tsh(a)) ? 1 : 0;
}

/* 1 if the nested kernel is also the FASTER choice (what the dispatcher asks).  Tiles are 64 output channels x 64 units, the
 * units dealt as 4 / 2 / 1 row pairs of one residue class x 64 / 128 / 256 time steps (best fill per launch); the kernel is
 * worth its 0.6x matrix work only while the tiles are reasonably full (measured 1.25-1.3x over conv_wino4p on full tiles):
 * row-pair slots x time steps at least 80 % used (7 rows per class = 4 pairs, one half empty: 87.5 %; 6 rows per class at
 * T = 128: 75 % in every shape, left to conv_wino4p), channels at least 7/8 of the (cost-weighted) channel tiles. */
extern "C" int babe_conv2d_wino45_preferred(const babe_conv_args* ap) {
    if (!babe_conv2d_wino45_supported(ap)) return 0;
    const babe_conv_args& a = *ap;
    const double u_rt = wino45_fill(a, wino45_best_tsh(a));      // rows x time steps, best of the three tile shapes
    // channel tiles: a tile whose waves partly hold padding still pays its transforms (~0.4 of a tile, measured share of the
    // vector work) but only the MFMAs of the waves with real channels (PADC variant of the kernel)
    const int full = a.Cout / 64, rem = a.Cout % 64;
    const double tiles_cost = full + (rem ? 0.4 + 0.6 * ((rem + 15) / 16) / 4.0 : 0.0);
    const double u_c = (double)a.Cout / (64.0 * tiles_cost);
    return (u_rt >= 0.8 && u_c >= 0.875) ? 1 : 0;
}

extern "C" int babe_conv2d_wino45(const babe_conv_args* ap, const float* w_wino45, void* stream) {
    BABE_CHECK_ARG(ap && w_wino45, "conv2d_wino45: null args");
    BABE_CHECK_ARG(babe_conv2d_wino45_supported(ap), "conv2d_wino45: unsupported problem (use babe_conv2d_wino4 / babe_conv2d)");
    const babe_conv_args& a = *ap;
    Wino45Geom g;
    g.CinP = (a.Cin + 7) / 8 * 8;
    g.CoutP = (a.Cout + 63) / 64 * 64;
    g.tsh = wino45_best_tsh(a);
    g.tiles_t = cdiv(a.T, 64 << g.tsh);
    const int n = cdiv(a.F, a.dil);                  // rows per residue class (at most)
    g.npairs = cdiv(n, 2);
    g.groups = cdiv(a.dil * g.npairs, 4 >> g.tsh);   // 4 >> tsh row pairs per tile, from the list of all classes' pairs
    hipStream_t s = (hipStream_t)stream;
    const double flops = babe_conv_flops(a);         // 36 multiplies per 8 outputs instead of 120: 0.3 of the direct count
    BabeProfScope prof(BABE_SLOT_CONV53_WINO45, babe_conv_bytes(a), flops, flops * 0.3, stream);
    if (wino45w_ok(a)) {
        g.tsh = wino45w_fill(a, 1) > wino45w_fill(a, 0) + 1e-9 ? 1 : 0;
        g.tiles_t = cdiv(a.T, 64 << g.tsh);
        g.npairs = cdiv(cdiv(a.F, a.dil), 2);
        g.groups = cdiv(a.dil * g.npairs, 2 >> g.tsh);
        const int bnw = a.Cout % 128 == 0 ? 128 : 96;
        dim3 gridw(g.tiles_t * g.groups, a.Cout / bnw, a.B);
        static const char* ovx = getenv("BABE_CONV_WINO45X");
        if (!(ovx && ovx[0] == '0')) {                  // second-generation pipeline (wave-private weights, two operand sets)
            const size_t ldsx = (size_t)(2 * 16 * 32 * 3 + 4 * 8 * 4 * 16 * 3) * 16;   // 144 KB
            static std::atomic<unsigned long long> attr_x{0};
            if (babe_lds_optin(attr_x, {reinterpret_cast<const void*>(&conv_wino45x_kernel<true, 128>),
                                        reinterpret_cast<const void*>(&conv_wino45x_kernel<false, 128>),
                                        reinterpret_cast<const void*>(&conv_wino45x_kernel<true, 96>),
                                        reinterpret_cast<const void*>(&conv_wino45x_kernel<false, 96>)}, (int)ldsx) == hipSuccess) {
                if (bnw == 128) {
                    if (a.in_scale) hipLaunchKernelGGL((conv_wino45x_kernel<true, 128>), gridw, dim3(512), ldsx, s, a, g, w_wino45);
                    else hipLaunchKernelGGL((conv_wino45x_kernel<false, 128>), gridw, dim3(512), ldsx, s, a, g, w_wino45);
                } else {
                    if (a.in_scale) hipLaunchKernelGGL((conv_wino45x_kernel<true, 96>), gridw, dim3(512), ldsx, s, a, g, w_wino45);
                    else hipLaunchKernelGGL((conv_wino45x_kernel<false, 96>), gridw, dim3(512), ldsx, s, a, g, w_wino45);
                }
            }
            BABE_LAUNCH_CHECK();
            return BABE_OK;
        }
        const size_t ldsw = (size_t)(2 * 16 * 32 * 3 + 2 * 8 * bnw * 3) * 16;       // 144 KB (120 KB for 96-channel tiles)
        static std::atomic<unsigned long long> attr_w{0};
        if (babe_lds_optin(attr_w, {reinterpret_cast<const void*>(&conv_wino45w_kernel<true, 128>),
                                    reinterpret_cast<const void*>(&conv_wino45w_kernel<false, 128>),
                                    reinterpret_cast<const void*>(&conv_wino45w_kernel<true, 96>),
                                    reinterpret_cast<const void*>(&conv_wino45w_kernel<false, 96>)},
                           (int)((size_t)(2 * 16 * 32 * 3 + 2 * 8 * 128 * 3) * 16)) == hipSuccess) {
            if (bnw == 128) {
                if (a.in_scale) hipLaunchKernelGGL((conv_wino45w_kernel<true, 128>), gridw, dim3(512), ldsw, s, a, g, w_wino45);
                else hipLaunchKernelGGL((conv_wino45w_kernel<false, 128>), gridw, dim3(512), ldsw, s, a, g, w_wino45);
            } else {
                if (a.in_scale) hipLaunchKernelGGL((conv_wino45w_kernel<true, 96>), gridw, dim3(512), ldsw, s, a, g, w_wino45);
                else hipLaunchKernelGGL((conv_wino45w_kernel<false, 96>), gridw, dim3(512), ldsw, s, a, g, w_wino45);
            }
        }
        BABE_LAUNCH_CHECK();
        return BABE_OK;
    }
    dim3 grid(g.tiles_t * g.groups, g.CoutP / 64, a.B);
    const size_t lds = 3 * (size_t)(8 * 64 * 3 + 8 * 64 * 3) * 16;           // 144 KB
    static std::atomic<unsigned long long> attr_done{0};
    // PADC variant: the last channel tile has whole 16-channel wave tiles of padding (Cout = 96: two of four)
    const bool padc = g.CoutP - a.Cout >= 16;
    if (babe_lds_optin(attr_done, {reinterpret_cast<const void*>(&conv_wino45_kernel<true, false>),
                                   reinterpret_cast<const void*>(&conv_wino45_kernel<false, false>),
                                   reinterpret_cast<const void*>(&conv_wino45_kernel<true, true>),
                                   reinterpret_cast<const void*>(&conv_wino45_kernel<false, true>)}, (int)lds) == hipSuccess) {
        if (a.in_scale && padc) hipLaunchKernelGGL((conv_wino45_kernel<true, true>), grid, dim3(512), lds, s, a, g, w_wino45);
        else if (a.in_scale) hipLaunchKernelGGL((conv_wino45_kernel<true, false>), grid, dim3(512), lds, s, a, g, w_wino45);
        else if (padc) hipLaunchKernelGGL((conv_wino45_kernel<false, true>), grid, dim3(512), lds, s, a, g, w_wino45);
        else hipLaunchKernelGGL((conv_wino45_kernel<false, false>), grid, dim3(512), lds, s, a, g, w_wino45);
    }
    BABE_LAUNCH_CHECK();
    return BABE_OK;
}
