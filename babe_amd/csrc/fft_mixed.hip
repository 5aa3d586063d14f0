// Mixed-radix length-L real FFT for the CQT (round 4): L = N1 * N2 (368368 = 572 * 644 = (4*11*13) * (4*7*23)) as a four-step
// transform whose two stages are REAL FFTs of length N1 / N2 - Stockham passes of radix 2, 3, 4, 5, 7, 11, 13, 23 in LDS -
// instead of the two dense DFT matrices on the MFMA (1,1) kernel (rounds 1-3: O(L (N1 + N2)) flops, 2.0 GFLOP per transform,
// MFMA-bound from B = 8 on, 16 launches of 32 us per score evaluation).  Replaces torch.fft.rfft / its transpose inside
// cqt_nsgt_pytorch.CQT_nsgt.fwd / .bwd / .apply_hpf_DC (call sites networks/cqtdiff+.py:743,841; testing/blind_bwe_sampler.py:156).
//
// ONE kernel, used four ways: "FFT of length N along the rows of a [N][ld] matrix, for a tile of C = 8 columns per workgroup":
//   forward  stage 1: x[n1][n2] real, N = N1 over n1, columns n2 -> times tw[k1][n2] = e^{-2 pi i k1 n2 / L}, written TRANSPOSED
//                     At[n2][k1] (rows of N1 contiguous k1: coalesced)
//            stage 2: At[n2][k1], N = N2 over n2, columns k1 -> spec[k2][k1] for k2 < K2 (bin k = k1 + N1 k2, natural order)
//   transpose step 1: spec[k2][k1] (zero for k2 >= K2), N = N2 over k2 with e^{+...}, columns k1 -> times conj(tw[k1][n2]), written
//                     transposed Z[k1][n2]
//            step 2: Z[k1][n2], N = N1 over k1, columns n2 -> x[n1][n2], real part
// (the adjoint of the forward map as a real-linear map = real part of the unnormalised inverse DFT of the zero-padded half
// spectrum, which is what the VJP of CQT.fwd and CQT.bwd itself need).
// A pass of radix R (Stockham autosort, natural order in and out): butterfly j < N / R reads rows j + t N / R, multiplies by
// w_N^(t (j mod Ns) N / (Ns R)), does an R-point DFT and writes rows expand(j, Ns, R) + t Ns; two LDS images, one barrier per
// pass.  Odd radices use the symmetric form (sums and differences of v[t], v[R - t]: 4 ((R-1)/2)^2 real multiply-adds per
// butterfly instead of 4 R^2).  Workgroup -> tile mapping keeps the four tiles that share 128-byte lines on one XCD.
// HBM-bound by design: algorithmic bytes per transform and clip = 4 L (signal) + 8 KX (spectrum); the two stages add one
// round trip of the 8 L-byte intermediate, which at the benchmark's batch sizes stays in the Infinity Cache.
#include "common.h"
#include "../../include/babe_hip.h"
#include "prof.h"
#include <cstdlib>

namespace {

// Round 5: 7 columns per workgroup, row pitch 7 float2 (odd: conflict-free along rows and along columns without a pad element).
// Rounds 3-4 ran 8 columns at pitch 9: two 46 KB images + the twiddle table = 98 KB at N = 644 - ONE workgroup per CU, each a
// chain of dependent Stockham passes (~12 us) with nothing beside it to hide them (0.73 TB/s at B = 32).  At 7
// columns the images are 2 x 36 KB (77 KB with the table): TWO workgroups per CU for every batch size, the same arithmetic per
// column in the same order (a column's transform never depended on its neighbours: bit-identical spectra), 644 = 7 * 92 tiles
// without a remainder.  Global rows are read / written in 28-byte pieces instead of 32-byte ones; the four-to-five tiles that
// share a 128-byte line still go to one XCD.
constexpr int FC = 7;            // columns per workgroup
constexpr int FCP = 7;           // row pitch of the LDS images (float2)
constexpr int FNT = 512;         // 8 waves per workgroup, two workgroups per CU

struct ColFFTArgs {
    const float* in_re;
    const float* in_im;          // nullptr: real input
    long in_bs;                  // batch stride (floats)
    int in_ld, in_rows;          // row stride; rows >= in_rows read as zero
    float* out_re;
    float* out_im;               // nullptr: real part only
    long out_bs;
    int out_ld, out_rows, out_transposed;   // natural: out[row * ld + col], rows < out_rows; transposed: out[col * ld + row]
    int N, ncols, nrad, rad[6];
    const float2* wN;            // exp(-2 pi i j / N), j < N
    const float2* big_tw;        // nullptr or [N1][N2] exp(-2 pi i k1 n2 / L)
    int tw_ld, tw_mode;          // 1: tw[row * tw_ld + col], 2: tw[col * tw_ld + row]
    int tiles;
    // Real-input forms of the FORWARD transform (round 5; both halve a stage's work and bytes):
    int pack_real;               // stage 1: the tile's 2 FC real columns travel as FC complex ones (a + i b), unpacked at the store
    int half_rows;               // stage 1: only rows k <= N/2 of the (Hermitian) column spectra are written
    int mirror_cols;             // stage 2: N1 (0 = off): columns k1 <= N1/2 are transformed; column N1 - k1 is written as the
                                 // conjugate of rows N - 1 - k2 of column k1 (X[L - k] = conj X[k])
};

__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ float2 cconj(float2 a) { return make_float2(a.x, -a.y); }

// R-point DFT in registers; w[m] = exp(-2 pi i m / R) for m = 1 .. (R-1)/2 (forward) - the inverse uses the conjugates
template <int R, bool INV>
__device__ __forceinline__ void dft_r(float2 (&v)[R], const float2* w) {
    if constexpr (R == 2) {
        const float2 a = v[0], b = v[1];
        v[0] = make_float2(a.x + b.x, a.y + b.y);
        v[1] = make_float2(a.x - b.x, a.y - b.y);
    } else if constexpr (R == 4) {
        const float2 a = make_float2(v[0].x + v[2].x, v[0].y + v[2].y), b = make_float2(v[0].x - v[2].x, v[0].y - v[2].y);
        const float2 c = make_float2(v[1].x + v[3].x, v[1].y + v[3].y), d = make_float2(v[1].x - v[3].x, v[1].y - v[3].y);
        // forward: X1 = b - i d, X3 = b + i d;  -i d = (d.y, -d.x)
        const float2 jd = INV ? make_float2(-d.y, d.x) : make_float2(d.y, -d.x);
        v[0] = make_float2(a.x + c.x, a.y + c.y);
        v[2] = make_float2(a.x - c.x, a.y - c.y);
        v[1] = make_float2(b.x + jd.x, b.y + jd.y);
        v[3] = make_float2(b.x - jd.x, b.y - jd.y);
    } else {
        constexpr int H = (R - 1) / 2;
        float2 s[H + 1], d[H + 1];
#pragma unroll
        for (int t = 1; t <= H; ++t) {
            s[t] = make_float2(v[t].x + v[R - t].x, v[t].y + v[R - t].y);
            d[t] = make_float2(v[t].x - v[R - t].x, v[t].y - v[R - t].y);
        }
        const float2 v0 = v[0];
        float2 acc0 = v0;
#pragma unroll
        for (int t = 1; t <= H; ++t) acc0 = make_float2(acc0.x + s[t].x, acc0.y + s[t].y);
        v[0] = acc0;
#pragma unroll
        for (int q = 1; q <= H; ++q) {
            float2 A = v0, B = make_float2(0.f, 0.f);
#pragma unroll
            for (int t = 1; t <= H; ++t) {
                constexpr int dummy = 0;
                (void)dummy;
                const int m = (t * q) % R;                       // compile-time after unrolling
                const int mm = m <= H ? m : R - m;
                const float c = w[mm].x;                          // cos(2 pi m / R)
                const float sn = m <= H ? -w[mm].y : w[mm].y;     // sin(2 pi m / R)  (w = cos - i sin)
                A = make_float2(A.x + c * s[t].x, A.y + c * s[t].y);
                B = make_float2(B.x + sn * d[t].x, B.y + sn * d[t].y);
            }
            // forward: X[q] = v0 + A' - i B, X[R-q] = v0 + A' + i B;  -i B = (B.y, -B.x)
            const float2 jb = INV ? make_float2(-B.y, B.x) : make_float2(B.y, -B.x);
            v[q] = make_float2(A.x + jb.x, A.y + jb.y);
            v[R - q] = make_float2(A.x - jb.x, A.y - jb.y);
        }
    }
}

template <int R, bool INV>
__device__ __forceinline__ void stockham_pass(const float2* __restrict__ src, float2* __restrict__ dst,
                                              const float2* __restrict__ wl, int N, int Ns) {
    const int NR = N / R;
    float2 w[(R - 1) / 2 + 1];
    if constexpr (R != 2 && R != 4) {
#pragma unroll
        for (int m = 1; m <= (R - 1) / 2; ++m) w[m] = wl[m * NR];
    }
    const int tstride = N / (Ns * R);
    const float inv_ns = 1.f / (float)Ns;                 // (j + 0.5) / Ns truncated is exact for j < 2^22: no integer division
    for (int item = threadIdx.x; item < NR * FC; item += FNT) {
        const int j = item / FC, c = item - j * FC;
        const int blk = (int)(((float)j + 0.5f) * inv_ns), k = j - blk * Ns;
        float2 v[R];
#pragma unroll
        for (int t = 0; t < R; ++t) v[t] = src[(j + t * NR) * FCP + c];
        if (Ns > 1) {
#pragma unroll
            for (int t = 1; t < R; ++t) {
                float2 tw = wl[t * k * tstride];
                if (INV) tw.y = -tw.y;
                v[t] = cmul(v[t], tw);
            }
        }
        dft_r<R, INV>(v, w);
        const int o = blk * Ns * R + k;
#pragma unroll
        for (int t = 0; t < R; ++t) dst[(o + t * Ns) * FCP + c] = v[t];
    }
}

template <bool INV>
__global__ __launch_bounds__(FNT) void colfft_kernel(ColFFTArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    float2* buf0 = reinterpret_cast<float2*>(smem_f);
    float2* buf1 = buf0 + a.N * FCP;
    float2* wl = buf1 + a.N * FCP;                    // exp(-2 pi i j / N)
    const int b = blockIdx.y;
    // tile of this workgroup: consecutive workgroups go to consecutive XCDs (8 of them); the four tiles that share the 128-byte
    // lines of a row are given to four consecutive workgroups OF ONE XCD
    const int wg = blockIdx.x;
    const int q = wg >> 3, x = wg & 7;
    const int tile = ((q >> 2) * 8 + x) * 4 + (q & 3);   // (grid padded to a multiple of 32: a bijection onto [0, padded))
    if (tile >= a.tiles) return;                            // surplus tiles of the padding (whole workgroup)
    const int col0 = tile * (a.pack_real ? 2 * FC : FC);
    const int N = a.N;
    // (the table's loads are issued first and stored after the tile's first loads have been issued: one DRAM latency, not two)
    float2 wreg[3];
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const int i = threadIdx.x + u * FNT;
        wreg[u] = i < N ? a.wN[i] : make_float2(0.f, 0.f);
    }
    // ---- load: rows x 8 columns (c fastest: 32-byte segments per plane).  Four items per thread and round: the four (eight)
    // global loads are issued back to back before their LDS stores - a one-item loop body serialises a DRAM latency per item
    // (10 per thread at N = 644: measured 15 us per workgroup for 41 KB)
    {
        const float* re = a.in_re + (long)b * a.in_bs;
        const float* im = a.in_im ? a.in_im + (long)b * a.in_bs : nullptr;
        if (INV && a.pack_real) {
            // Transpose of the forward unpack (last stage of the transposed transform: two REAL output columns per complex
            // inverse FFT).  Rows k <= N/2 of the two columns' spectra Za, Zb are given; W = Ha + i Hb with the Hermitian halves
            // Ha[k] = Za[k] / 2, Ha[N - k] = conj(Za[k]) / 2 (0 < k < N/2), Ha[0] = Re Za[0], Ha[N/2] = Re Za[N/2]: then
            // IDFT(W) = a' + i b', the two real outputs.
            if (threadIdx.x < (unsigned)N) {
#pragma unroll
                for (int u = 0; u < 3; ++u)
                    if (threadIdx.x + u * FNT < N) wl[threadIdx.x + u * FNT] = wreg[u];
            }
            const int Nh = N / 2 + 1;
            for (int base = threadIdx.x; base < Nh * FC; base += 2 * FNT) {
                float ar[2], ai[2], br[2], bi[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int item = base + u * FNT;
                    const int r = item / FC, c = item - r * FC;
                    const int ca = col0 + 2 * c;
                    const bool oka = item < Nh * FC && r < a.in_rows && ca < a.ncols, okb = oka && ca + 1 < a.ncols;
                    const long o = oka ? (long)r * a.in_ld + ca : 0;
                    ar[u] = oka ? re[o] : 0.f;
                    ai[u] = oka ? im[o] : 0.f;
                    br[u] = okb ? re[o + 1] : 0.f;
                    bi[u] = okb ? im[o + 1] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int item = base + u * FNT;
                    if (item >= Nh * FC) continue;
                    const int r = item / FC, c = item - r * FC;
                    if (r == 0 || 2 * r == N) {
                        buf0[r * FCP + c] = make_float2(ar[u], br[u]);
                    } else {
                        buf0[r * FCP + c] = make_float2(0.5f * (ar[u] - bi[u]), 0.5f * (ai[u] + br[u]));
                        buf0[(N - r) * FCP + c] = make_float2(0.5f * (ar[u] + bi[u]), 0.5f * (br[u] - ai[u]));
                    }
                }
            }
        } else
        for (int base = threadIdx.x; base < N * FC; base += 4 * FNT) {
            float vr[4], vi[4];
            if (base == (int)threadIdx.x) {
#pragma unroll
                for (int u = 0; u < 3; ++u)
                    if (threadIdx.x + u * FNT < N) wl[threadIdx.x + u * FNT] = wreg[u];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int item = base + u * FNT;
                const int r = item / FC, c = item - r * FC;
                if (!INV && a.pack_real) {
                    // two neighbouring REAL columns as one complex column: z = x[:, 2c] + i x[:, 2c + 1]
                    const int ca = col0 + 2 * c;
                    const bool oka = item < N * FC && r < a.in_rows && ca < a.ncols, okb = oka && ca + 1 < a.ncols;
                    const long o = oka ? (long)r * a.in_ld + ca : 0;
                    vr[u] = oka ? re[o] : 0.f;
                    vi[u] = okb ? re[o + 1] : 0.f;
                    continue;
                }
                const bool okc = item < N * FC && col0 + c < a.ncols;
                const bool ok = okc && r < a.in_rows;
                const long o = ok ? (long)r * a.in_ld + col0 + c : 0;
                vr[u] = ok ? re[o] : 0.f;
                vi[u] = (ok && im) ? im[o] : 0.f;
                if (INV && a.mirror_cols) {
                    // transpose of the forward mirror store: row r >= N - in_rows of column k1 also receives
                    // conj(spec[N - 1 - r][N1 - k1]) (columns the transformed half does not hold itself)
                    const int col = col0 + c, mc = a.mirror_cols - col, q = N - 1 - r;
                    if (okc && q < a.in_rows && col >= 1 && mc >= a.mirror_cols / 2 + 1) {
                        const long om = (long)q * a.in_ld + mc;
                        vr[u] += re[om];
                        vi[u] -= im[om];
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int item = base + u * FNT;
                if (item < N * FC) buf0[(item / FC) * FCP + (item % FC)] = make_float2(vr[u], vi[u]);
            }
        }
    }
    __syncthreads();
    float2 *src = buf0, *dst = buf1;
    int Ns = 1;
    for (int p = 0; p < a.nrad; ++p) {
        const int R = a.rad[p];
        switch (R) {
            case 2: stockham_pass<2, INV>(src, dst, wl, N, Ns); break;
            case 3: stockham_pass<3, INV>(src, dst, wl, N, Ns); break;
            case 4: stockham_pass<4, INV>(src, dst, wl, N, Ns); break;
            case 5: stockham_pass<5, INV>(src, dst, wl, N, Ns); break;
            case 7: stockham_pass<7, INV>(src, dst, wl, N, Ns); break;
            case 11: stockham_pass<11, INV>(src, dst, wl, N, Ns); break;
            case 13: stockham_pass<13, INV>(src, dst, wl, N, Ns); break;
            default: stockham_pass<23, INV>(src, dst, wl, N, Ns); break;
        }
        Ns *= R;
        float2* t = src;
        src = dst;
        dst = t;
        __syncthreads();
    }
    // ---- store (+ the four-step twiddle)
    float* ore = a.out_re + (long)b * a.out_bs;
    float* oim = a.out_im ? a.out_im + (long)b * a.out_bs : nullptr;
    if (!INV && a.pack_real) {
        // Unpack Z = FFT(a + i b):  A[k] = (Z[k] + conj Z[N - k]) / 2,  B[k] = -i (Z[k] - conj Z[N - k]) / 2, then the four-step
        // twiddle, written transposed out[col][k] for the tile's 2 FC real columns; rows k <= N / 2 only with half_rows (the
        // rest is the conjugate mirror and the second stage does not read it).  One index space item = c2 Nh + k, k fastest.
        const int Nh = a.half_rows ? N / 2 + 1 : N;
        const float inv_nh = 1.f / (float)Nh;
        for (int base = threadIdx.x; base < Nh * 2 * FC; base += 4 * FNT) {
            float2 tw[4];
            int rr[4], cc[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int item = base + u * FNT;
                cc[u] = (int)(((float)item + 0.5f) * inv_nh);
                rr[u] = item - cc[u] * Nh;
                tw[u] = make_float2(1.f, 0.f);
                if (a.big_tw && item < Nh * 2 * FC && col0 + cc[u] < a.ncols)
                    tw[u] = a.tw_mode == 1 ? a.big_tw[(long)rr[u] * a.tw_ld + col0 + cc[u]] : a.big_tw[(long)(col0 + cc[u]) * a.tw_ld + rr[u]];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int item = base + u * FNT;
                if (item >= Nh * 2 * FC || col0 + cc[u] >= a.ncols) continue;
                const int c = cc[u] >> 1;
                const int rm = rr[u] == 0 ? 0 : N - rr[u];
                const float2 z1 = src[rr[u] * FCP + c], z2 = src[rm * FCP + c];
                float2 v = (cc[u] & 1) ? make_float2(0.5f * (z1.y + z2.y), 0.5f * (z2.x - z1.x))
                                       : make_float2(0.5f * (z1.x + z2.x), 0.5f * (z1.y - z2.y));
                if (a.big_tw) v = cmul(v, tw[u]);
                const long o = (long)(col0 + cc[u]) * a.out_ld + rr[u];
                ore[o] = v.x;
                if (oim) oim[o] = v.y;
            }
        }
    } else if (a.out_transposed) {
        // out[col][row]: rows fastest across the threads (contiguous runs of N floats per column and plane); ONE index space
        // item = c N + r over the tile, four items per thread and round, their twiddle loads issued together (a loop over the
        // 8 columns paid a DRAM latency per column)
        const float inv_n = 1.f / (float)N;
        for (int base = threadIdx.x; base < N * FC; base += 4 * FNT) {
            float2 tw[4];
            int rr[4], cc[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int item = base + u * FNT;
                cc[u] = (int)(((float)item + 0.5f) * inv_n);
                rr[u] = item - cc[u] * N;
                tw[u] = make_float2(1.f, 0.f);
                if (a.big_tw && item < N * FC && col0 + cc[u] < a.ncols)
                    tw[u] = a.tw_mode == 1 ? a.big_tw[(long)rr[u] * a.tw_ld + col0 + cc[u]] : a.big_tw[(long)(col0 + cc[u]) * a.tw_ld + rr[u]];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int item = base + u * FNT;
                if (item >= N * FC || col0 + cc[u] >= a.ncols) continue;
                float2 v = src[rr[u] * FCP + cc[u]];
                if (a.big_tw) {
                    if (INV) tw[u].y = -tw[u].y;
                    v = cmul(v, tw[u]);
                }
                const long o = (long)(col0 + cc[u]) * a.out_ld + rr[u];
                ore[o] = v.x;
                if (oim) oim[o] = v.y;
            }
        }
    } else if (INV && a.pack_real) {
        // the two real outputs of a packed inverse transform: real part -> column 2c, imaginary part -> column 2c + 1
        const int rows = a.out_rows < N ? a.out_rows : N;
        for (int item = threadIdx.x; item < rows * 2 * FC; item += FNT) {
            const int r = item / (2 * FC), c2 = item - r * (2 * FC);
            if (col0 + c2 >= a.ncols) continue;
            const float2 v = src[r * FCP + (c2 >> 1)];
            ore[(long)r * a.out_ld + col0 + c2] = (c2 & 1) ? v.y : v.x;
        }
    } else {
        const int rows = a.out_rows < N ? a.out_rows : N;
        for (int item = threadIdx.x; item < rows * FC; item += FNT) {
            const int r = item / FC, c = item - r * FC;
            if (col0 + c >= a.ncols) continue;
            float2 v = src[r * FCP + c];
            if (a.big_tw) {
                float2 tw = a.tw_mode == 1 ? a.big_tw[(long)r * a.tw_ld + col0 + c] : a.big_tw[(long)(col0 + c) * a.tw_ld + r];
                if (INV) tw.y = -tw.y;
                v = cmul(v, tw);
            }
            const long o = (long)r * a.out_ld + col0 + c;
            ore[o] = v.x;
            if (oim) oim[o] = v.y;
        }
        if (!INV && a.mirror_cols) {
            // Hermitian half: X[(N1 - k1) + N1 k2'] = conj X[k1 + N1 (N - 1 - k2')] - rows N - rows .. N - 1 of this tile's columns
            // k1 >= 1 become rows rows - 1 .. 0 of columns N1 - k1 (only columns the transformed half does not hold itself)
            const int N1 = a.mirror_cols, nh1 = N1 / 2 + 1;
            for (int item = threadIdx.x; item < rows * FC; item += FNT) {
                const int q = item / FC, c = item - q * FC;          // q: target row k2'
                const int col = col0 + c, mc = N1 - col;
                if (col < 1 || col >= a.ncols || mc < nh1) continue;
                const float2 v = src[(N - 1 - q) * FCP + c];
                const long o = (long)q * a.out_ld + mc;
                ore[o] = v.x;
                if (oim) oim[o] = -v.y;
            }
        }
    }
}

int launch(const ColFFTArgs& a, int B, int inverse, hipStream_t s) {
    const size_t lds = (size_t)(2 * a.N * FCP + a.N) * sizeof(float2);
    static std::atomic<unsigned long long> attr{0};
    if (babe_lds_optin(attr, {reinterpret_cast<const void*>(&colfft_kernel<false>), reinterpret_cast<const void*>(&colfft_kernel<true>)},
                       160 * 1024) != hipSuccess) {
        babe_set_error("fft_mixed: cannot opt in to %zu bytes of LDS", lds);
        return BABE_ERR_HIP;
    }
    const dim3 grid((a.tiles + 31) / 32 * 32, B);
    if (inverse) hipLaunchKernelGGL(colfft_kernel<true>, grid, dim3(FNT), lds, s, a);
    else hipLaunchKernelGGL(colfft_kernel<false>, grid, dim3(FNT), lds, s, a);
    return BABE_OK;
}

bool radices_ok(const int* rad, int n, int N) {
    long p = 1;
    for (int i = 0; i < n; ++i) {
        const int r = rad[i];
        if (!(r == 2 || r == 3 || r == 4 || r == 5 || r == 7 || r == 11 || r == 13 || r == 23)) return false;
        p *= r;
    }
    return n >= 1 && n <= 6 && p == N;
}

}  // namespace

/* Length-L = N1*N2 real FFT, planar half spectrum [B][2][K2*N1] (bin k = k1 + N1*k2), and its transpose.  rad1 / rad2: the
 * radices of N1 / N2 (each in {2,3,4,5,7,11,13,23}, at most 6, product = N).  w1 / w2: exp(-2 pi i j / N1|N2) (float2), tw:
 * [N1][N2] exp(-2 pi i k1 n2 / L) (float2).  work: [B][2][N1*N2] floats of scratch (the transposed intermediate).
 * direction 0: x [B][L] -> spec;  1: spec -> x (transpose of direction 0: real part of the unnormalised inverse DFT of the
 * zero-padded half spectrum). */
extern "C" int babe_rfft_mixed(const float* x_in, float* spec_out, const float* spec_in, float* x_out, float* work, int B,
                               int N1, int N2, int K2, const int* rad1, int nrad1, const int* rad2, int nrad2,
                               const float* w1, const float* w2, const float* tw, int direction, void* stream) {
    BABE_CHECK_ARG(work && B > 0 && N1 > 0 && N2 > 0 && K2 > 0 && K2 <= N2 && rad1 && rad2 && w1 && w2 && tw,
                   "rfft_mixed: bad arguments");
    BABE_CHECK_ARG(radices_ok(rad1, nrad1, N1) && radices_ok(rad2, nrad2, N2), "rfft_mixed: unsupported radices");
    BABE_CHECK_ARG((size_t)(2 * (N1 > N2 ? N1 : N2) * FCP + (N1 > N2 ? N1 : N2)) * 8 <= 160 * 1024, "rfft_mixed: factor too long for LDS");
    hipStream_t s = (hipStream_t)stream;
    const long L = (long)N1 * N2;
    const double bytes = (double)B * (4.0 * L + 8.0 * K2 * N1);
    BabeProfScope prof(BABE_SLOT_CQT_GATHER, bytes, 0, 0, stream);
    ColFFTArgs a1{}, a2{};
    if (direction == 0) {
        BABE_CHECK_ARG(x_in && spec_out, "rfft_mixed: null input / output");
        // stage 1: x[n1][n2] -> At[n2][k1] * tw[k1][n2]
        a1.in_re = x_in; a1.in_im = nullptr; a1.in_bs = L; a1.in_ld = N2; a1.in_rows = N1;
        a1.out_re = work; a1.out_im = work + L; a1.out_bs = 2 * L; a1.out_ld = N1; a1.out_rows = N1; a1.out_transposed = 1;
        a1.N = N1; a1.ncols = N2; a1.nrad = nrad1;
        for (int i = 0; i < nrad1; ++i) a1.rad[i] = rad1[i];
        a1.wN = reinterpret_cast<const float2*>(w1); a1.big_tw = reinterpret_cast<const float2*>(tw); a1.tw_ld = N2; a1.tw_mode = 1;
        // real input: two columns per complex transform, and only rows k1 <= N1/2 of the intermediate are produced; the second
        // stage transforms those columns and writes the others as conjugate mirrors (half the butterflies and a third less
        // traffic per transform).  BABE_FFT_REAL=0: the complex form of round 4 on both stages (A/B switch).
        static const char* ovr = getenv("BABE_FFT_REAL");
        const bool real_form = !(ovr && ovr[0] == '0');
        a1.pack_real = real_form; a1.half_rows = real_form;
        a1.tiles = real_form ? cdiv(N2, 2 * FC) : cdiv(N2, FC);
        // stage 2: At[n2][k1] -> spec[k2][k1], k2 < K2
        a2.in_re = work; a2.in_im = work + L; a2.in_bs = 2 * L; a2.in_ld = N1; a2.in_rows = N2;
        a2.out_re = spec_out; a2.out_im = spec_out + (long)K2 * N1; a2.out_bs = 2L * K2 * N1; a2.out_ld = N1; a2.out_rows = K2;
        a2.out_transposed = 0;
        a2.N = N2; a2.ncols = N1; a2.nrad = nrad2;
        for (int i = 0; i < nrad2; ++i) a2.rad[i] = rad2[i];
        a2.wN = reinterpret_cast<const float2*>(w2); a2.big_tw = nullptr; a2.tiles = cdiv(N1, FC);
        if (real_form) {
            a2.ncols = N1 / 2 + 1;
            a2.mirror_cols = N1;
            a2.tiles = cdiv(a2.ncols, FC);
        }
        if (int e = launch(a1, B, 0, s)) return e;
        if (int e = launch(a2, B, 0, s)) return e;
    } else {
        BABE_CHECK_ARG(spec_in && x_out, "rfft_mixed: null input / output");
        // step 1: spec[k2][k1] (k2 < K2) -> Z[k1][n2] * conj(tw[k1][n2])
        a1.in_re = spec_in; a1.in_im = spec_in + (long)K2 * N1; a1.in_bs = 2L * K2 * N1; a1.in_ld = N1; a1.in_rows = K2;
        a1.out_re = work; a1.out_im = work + L; a1.out_bs = 2 * L; a1.out_ld = N2; a1.out_rows = N2; a1.out_transposed = 1;
        a1.N = N2; a1.ncols = N1; a1.nrad = nrad2;
        for (int i = 0; i < nrad2; ++i) a1.rad[i] = rad2[i];
        a1.wN = reinterpret_cast<const float2*>(w2); a1.big_tw = reinterpret_cast<const float2*>(tw); a1.tw_ld = N2; a1.tw_mode = 2;
        a1.tiles = cdiv(N1, FC);
        // the exact transposes of the real-input forms of direction 0 (same switch): step 1 transforms the columns k1 <= N1/2
        // with the mirrored half of the spectrum added in conjugated, step 2 produces two real columns per inverse transform
        static const char* ovr = getenv("BABE_FFT_REAL");
        const bool real_form = !(ovr && ovr[0] == '0');
        if (real_form) {
            a1.ncols = N1 / 2 + 1;
            a1.mirror_cols = N1;
            a1.tiles = cdiv(a1.ncols, FC);
        }
        // step 2: Z[k1][n2] -> x[n1][n2] (real part)
        a2.in_re = work; a2.in_im = work + L; a2.in_bs = 2 * L; a2.in_ld = N2; a2.in_rows = N1;
        a2.out_re = x_out; a2.out_im = nullptr; a2.out_bs = L; a2.out_ld = N2; a2.out_rows = N1; a2.out_transposed = 0;
        a2.N = N1; a2.ncols = N2; a2.nrad = nrad1;
        for (int i = 0; i < nrad1; ++i) a2.rad[i] = rad1[i];
        a2.wN = reinterpret_cast<const float2*>(w1); a2.big_tw = nullptr; a2.tiles = cdiv(N2, FC);
        if (real_form) {
            a2.pack_real = 1;
            a2.in_rows = N1 / 2 + 1;
            a2.tiles = cdiv(N2, 2 * FC);
        }
        if (int e = launch(a1, B, 1, s)) return e;
        if (int e = launch(a2, B, 1, s)) return e;
    }
    BABE_LAUNCH_CHECK();
    return BABE_OK;
}
