// Denoiser pre-pass ("next" row 3): kernels for the two-stage STFT-domain U-Net of networks/denoiser.py:232-321 and
// its STFT / inverse-STFT wrapper (testing/denoise_and_bwe_tester.py:146-165).  Inference only (the reference runs it
// under no_grad).
//
// dn_conv_kernel: general small-kernel Conv2d on v_mfma_f32_32x32x2_f32 for [B,C,H=frames,W=bins] tensors:
//   KH x KW <= 7x7, stride 1 or 2, reflect or zero padding, bias + optional ELU + optional residual in the epilogue,
//   strided / offset output mapping.  The 4x4 stride-2 ConvTranspose2d runs on the same kernel as four 2x2 stride-1
//   convolutions (one per output parity) whose outputs interleave (out index = 2*m + parity - crop).
//   Tile: 64 output channels x (4 rows x 64 columns); wave w owns row w: 2 x 2 accumulator tiles.  Per 8-channel
//   chunk the (reflect-)padded input patch [8][PH][PW] is staged once in LDS and reused by all KH*KW taps; the
//   weights [KH'][KW][8][64] follow in slabs of KH' kernel rows.
#include "common.h"
#include "../../include/babe_hip.h"
#include "prof.h"
#include "fft_lds.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int DN_BN = 64, DN_TH = 4, DN_TW = 64;

// KH x KW kernel, stride S; a step = (KC input channels) x (KHC kernel rows): LDS holds the input rows those kernel
// rows touch, [KC][(TH-1)*S+KHC][(TW-1)*S+KW], and the weight slab [KHC][KW][KC][64].  The next step's global loads are
// issued (all of a thread's loads back to back, fully unrolled) before the MFMAs of the current step and written to LDS
// after them, so their latency hides behind the matrix work.
template <int KH, int KW, int S, int KHC, int KC>
__global__ __launch_bounds__(256, 2) void dn_conv_kernel(babe_dnconv_args a, const float* __restrict__ wq, int CinP,
                                                         int CoutP, int tiles_w) {
    constexpr int PH = (DN_TH - 1) * S + KHC, PW = (DN_TW - 1) * S + KW;
    constexpr int NX = KC * PH * PW;                         // patch floats
    constexpr int XJ = (NX + 255) / 256;
    constexpr int NW4 = KHC * KW * KC * (DN_BN / 4);         // weight float4 per slab
    constexpr int WJ = (NW4 + 255) / 256;
    constexpr int NKG = (KH + KHC - 1) / KHC;                // kernel-row groups
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Xs = smem;
    float* Ws = smem + (NX + 3) / 4 * 4;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, l31 = lane & 31;
    const int tile_w = blockIdx.x % tiles_w, tile_h = blockIdx.x / tiles_w;
    const int oh0 = tile_h * DN_TH, ow0 = tile_w * DN_TW;
    const int co0 = blockIdx.y * DN_BN;
    const int ksplit = a.ksplit > 1 ? a.ksplit : 1;       // split-K over input-channel chunks (small planes)
    const int b = blockIdx.z / ksplit, ks = blockIdx.z - b * ksplit;
    const float* inb = a.in + (long)b * a.in_bs;
    const int ih0 = oh0 * S - a.pad_t, iw0 = ow0 * S - a.pad_l;
    const unsigned in_cs = (unsigned)a.in_cs;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // per-thread patch slots: element e = tid + 256*j -> (channel, patch row, patch column); the column part of the
    // source index does not change between steps
    constexpr bool CACHE_COL = XJ <= 16;             // keep the reflected column index in registers when it is cheap
    auto col_of = [&](int j) {
        const int e = tid + 256 * j;
        int iw = iw0 + e % PW;
        if (a.pad_mode) {
            if (iw < 0) iw = -iw;
            if (iw >= a.IW) iw = 2 * (a.IW - 1) - iw;
        }
        return (e < NX && iw >= 0 && iw < a.IW) ? iw : -1;
    };
    int xoff[CACHE_COL ? XJ : 1];                    // iw (reflected) or -1
    if constexpr (CACHE_COL) {
#pragma unroll
        for (int j = 0; j < XJ; ++j) xoff[j] = col_of(j);
    }
    float xr[XJ];
    f32x4 wr[WJ];
    auto load_step = [&](int ci0, int kg) {
        const int kh0 = kg * KHC;
#pragma unroll
        for (int j = 0; j < XJ; ++j) {
            const int e = tid + 256 * j;
            const int rr = e / PW;
            const int r = rr % PH, ch = rr / PH;
            int ih = ih0 + kh0 + r;
            if (a.pad_mode) {
                if (ih < 0) ih = -ih;
                if (ih >= a.IH) ih = 2 * (a.IH - 1) - ih;
            }
            int iw;
            if constexpr (CACHE_COL) iw = xoff[j];
            else iw = col_of(j);
            const bool ok = iw >= 0 && ih >= 0 && ih < a.IH && ci0 + ch < a.Cin;
            const unsigned off = ok ? (unsigned)(ci0 + ch) * in_cs + (unsigned)(ih * a.IW + iw) : 0u;   // < 2^32 elements
            const float v = inb[off];
            xr[j] = ok ? v : 0.f;
        }
#pragma unroll
        for (int j = 0; j < WJ; ++j) {
            int i = tid + 256 * j;
            if (i > NW4 - 1) i = NW4 - 1;
            const int c4 = i & 15;
            const int rest = i >> 4;                              // (khl*KW + kw)*KC + ci_l
            const int ci_l = rest % KC;
            const int tap = rest / KC;
            const int khl = tap / KW, kw = tap - khl * KW;
            const int kh = kh0 + khl < KH ? kh0 + khl : KH - 1;   // rows past KH are never used by the MFMA loop
            wr[j] = *reinterpret_cast<const f32x4*>(wq + ((unsigned)((kh * KW + kw) * CinP + ci0 + ci_l) * (unsigned)CoutP + co0 + c4 * 4));
        }
    };
    auto store_step = [&]() {
#pragma unroll
        for (int j = 0; j < XJ; ++j)
            if (tid + 256 * j < NX) Xs[tid + 256 * j] = xr[j];
#pragma unroll
        for (int j = 0; j < WJ; ++j)
            if (tid + 256 * j < NW4) *reinterpret_cast<f32x4*>(Ws + (tid + 256 * j) * 4) = wr[j];
    };

    const int nchunks = CinP / KC;
    const int per = (nchunks + ksplit - 1) / ksplit;
    const int c_lo = ks * per, c_hi = c_lo + per < nchunks ? c_lo + per : nchunks;
    const int nsteps = (c_hi > c_lo ? c_hi - c_lo : 0) * NKG;
    int ci0 = c_lo * KC, kg = 0;
    if (nsteps > 0) load_step(ci0, 0);
    for (int step = 0; step < nsteps; ++step) {
        __syncthreads();                                          // everyone finished reading the previous slab
        store_step();
        __syncthreads();
        int nci = ci0, nkg = kg + 1;
        if (nkg == NKG) {
            nkg = 0;
            nci += KC;
        }
        if (step + 1 < nsteps) load_step(nci, nkg);
        const int khn = KH - kg * KHC < KHC ? KH - kg * KHC : KHC;
#pragma unroll
        for (int khl = 0; khl < KHC; ++khl) {
            if (khl < khn) {
                const float* xrow = Xs + (wave * S + khl) * PW + l31 * S;
#pragma unroll
                for (int kw = 0; kw < KW; ++kw) {
                    const float* wt = Ws + (khl * KW + kw) * KC * DN_BN + l31;
#pragma unroll
                    for (int st = 0; st < KC / 2; ++st) {
                        const int ch = 2 * st + h;
                        const float a0 = wt[ch * DN_BN], a1 = wt[ch * DN_BN + 32];
                        const float b0 = xrow[ch * PH * PW + kw], b1 = xrow[ch * PH * PW + kw + 32 * S];
                        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
                    }
                }
            }
        }
        ci0 = nci;
        kg = nkg;
    }

    const int oh = oh0 + wave;
    if (ksplit > 1) {
        // raw partial sums [ks][b][co][OH][OW]; dn_splitk_finish_kernel adds them up and applies the epilogue
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int bt = 0; bt < 2; ++bt) {
                const int ow = ow0 + bt * 32 + l31;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = co0 + nt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (oh < a.OH && ow < a.OW && co < a.Cout)
                        a.ws[(((long)(ks * a.B + b) * a.Cout + co) * a.OH + oh) * a.OW + ow] = acc[nt][bt][r];
                }
            }
        return;
    }
    // ---- epilogue: bias, ELU, residual; output index = logical index * out_step + out_off (both axes).
    // Round 6: the bias and residual loads of a 32 x 32 tile are issued back to back inside ONE wave-uniform branch per operand, from
    // clamped addresses (an element outside the output reads a valid one and is masked at the store).  Under the per-element
    // `if (ok) { v += a.bias ? a.bias[co] : 0; if (a.res) v += a.res[..]; }` this replaces, every one of a lane's 64 outputs loaded and
    // waited vmcnt(0) on its own: 64 - 128 exposed memory latencies per workgroup (the same finding as conv_wino85.hip's epilogue).
    const int fh = oh * a.out_hstep + a.out_h0;
    const bool rowok = oh < a.OH && fh >= 0 && fh < a.out_H;
    const bool has_b = a.bias != nullptr, has_r = a.res != nullptr;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        int cc[16];
        float bs[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + nt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            cc[r] = co < a.Cout ? co : a.Cout - 1;
            bs[r] = 0.f;
        }
        if (has_b) {
#pragma unroll
            for (int r = 0; r < 16; ++r) bs[r] = a.bias[cc[r]];
        }
#pragma unroll
        for (int bt = 0; bt < 2; ++bt) {
            const int ow = ow0 + bt * 32 + l31;
            const int fw = ow * a.out_wstep + a.out_w0;
            const bool ok = rowok && ow < a.OW && fw >= 0 && fw < a.out_W;
            const long sp = ok ? (long)fh * a.out_W + fw : 0;
            float rv[16];
            if (has_r) {
#pragma unroll
                for (int r = 0; r < 16; ++r) rv[r] = a.res[(long)b * a.res_bs + (long)cc[r] * a.res_cs + sp];
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + nt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                float v = acc[nt][bt][r] + bs[r];
                if (a.act) v = v > 0.f ? v : expm1f(v);
                if (has_r) v += rv[r];
                if (ok && co < a.Cout) a.out[(long)b * a.out_bs + (long)co * a.out_cs + sp] = v;
            }
        }
    }
}

// sum of the split-K partials + the conv epilogue
__global__ void dn_splitk_finish_kernel(babe_dnconv_args a, long total) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int ow = (int)(i % a.OW);
    long r = i / a.OW;
    const int oh = (int)(r % a.OH);
    r /= a.OH;
    const int co = (int)(r % a.Cout), b = (int)(r / a.Cout);
    float v = a.bias ? a.bias[co] : 0.f;
    for (int ks = 0; ks < a.ksplit; ++ks) v += a.ws[(long)ks * total + i];
    if (a.act) v = v > 0.f ? v : expm1f(v);
    const int fh = oh * a.out_hstep + a.out_h0, fw = ow * a.out_wstep + a.out_w0;
    if (fh < 0 || fh >= a.out_H || fw < 0 || fw >= a.out_W) return;
    const long sp = (long)fh * a.out_W + fw;
    if (a.res) v += a.res[(long)b * a.res_bs + (long)co * a.res_cs + sp];
    a.out[(long)b * a.out_bs + (long)co * a.out_cs + sp] = v;
}

template <int KH, int KW, int S, int KHC, int KC>
void dn_launch(const babe_dnconv_args& a, const float* wq, hipStream_t s) {
    constexpr int PH = (DN_TH - 1) * S + KHC, PW = (DN_TW - 1) * S + KW;
    const int CinP = (a.Cin + 31) / 32 * 32, CoutP = (a.Cout + 63) / 64 * 64;
    const int tiles_w = cdiv(a.OW, DN_TW), tiles_h = cdiv(a.OH, DN_TH);
    const size_t lds = ((size_t)(KC * PH * PW + 3) / 4 * 4 + (size_t)KHC * KW * KC * DN_BN) * sizeof(float);
    const int ksplit = a.ksplit > 1 ? a.ksplit : 1;
    dim3 grid(tiles_w * tiles_h, CoutP / DN_BN, a.B * ksplit);
    hipLaunchKernelGGL((dn_conv_kernel<KH, KW, S, KHC, KC>), grid, dim3(256), lds, s, a, wq, CinP, CoutP, tiles_w);
    if (ksplit > 1) {
        const long total = (long)a.B * a.Cout * a.OH * a.OW;
        hipLaunchKernelGGL(dn_splitk_finish_kernel, dim3(cdiv(total, 256)), dim3(256), 0, s, a, total);
    }
}

// mode 0: w [Cout][Cin][KH][KW] -> dst [KH][KW][CinP][CoutP]
// mode 1: ConvTranspose2d weight [Cin][Cout][4][4], output parity (ph, pw) -> dst [2][2][CinP][CoutP] with
//         dst[kh'][kw'] = w[ci][co][ph + 2(1-kh')][pw + 2(1-kw')]
__global__ void dn_pack_kernel(const float* __restrict__ w, float* __restrict__ dst, int Cout, int Cin, int KH, int KW,
                               int CinP, int CoutP, int mode, int ph, int pw, long total) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int co = (int)(i % CoutP);
    long r = i / CoutP;
    const int ci = (int)(r % CinP);
    r /= CinP;
    const int kw = (int)(r % KW), kh = (int)(r / KW);
    float v = 0.f;
    if (co < Cout && ci < Cin) {
        if (mode == 0) v = w[(((long)co * Cin + ci) * KH + kh) * KW + kw];
        else v = w[(((long)ci * Cout + co) * 4 + (ph + 2 * (1 - kh))) * 4 + (pw + 2 * (1 - kw))];
    }
    dst[i] = v;
}

// out[b][c][h][w] += low[b][c][(h+dh)>>1][(w+dw)>>1]       (nearest x2 upsampling, cropped, added in place)
__global__ void dn_upsample_add_kernel(float* __restrict__ out, long out_bs, long out_cs, const float* __restrict__ low,
                                       long low_bs, long low_cs, int C, int H, int W, int LH, int LW, int dh, int dw,
                                       long total) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int w = (int)(i % W);
    long r = i / W;
    const int hh = (int)(r % H);
    r /= H;
    const int c = (int)(r % C), b = (int)(r / C);
    const int lh = (hh + dh) >> 1, lw = (w + dw) >> 1;
    if (lh < LH && lw < LW)
        out[(long)b * out_bs + (long)c * out_cs + (long)hh * W + w] += low[(long)b * low_bs + (long)c * low_cs + (long)lh * LW + lw];
}

// SAM gate (denoiser.py:126-130): out = x1 * sigmoid(m) + feats, per (b, c) planes of hw contiguous floats
__global__ void dn_sam_gate_kernel(const float* __restrict__ x1, const float* __restrict__ m, const float* __restrict__ f,
                                   long f_bs, long f_cs, float* __restrict__ out, long out_bs, long out_cs, int C, long hw,
                                   long total) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const long p = i % hw;
    const long bc = i / hw;
    const int c = (int)(bc % C), b = (int)(bc / C);
    const float g = 1.f / (1.f + expf(-m[i]));
    out[(long)b * out_bs + (long)c * out_cs + p] = x1[i] * g + f[(long)b * f_bs + (long)c * f_cs + p];
}

// network input with the 10 frequency-embedding channels appended (AddFreqEncoding, denoiser.py:159-169)
__global__ void dn_fill_input_kernel(const float* __restrict__ X, const float* __restrict__ femb, float* __restrict__ out,
                                     int T, int F, int nemb, long total) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int f = (int)(i % F);
    long r = i / F;
    r /= T;
    const int C = 2 + nemb;
    const int c = (int)(r % C), b = (int)(r / C);
    const long tf = i % ((long)T * F);
    out[i] = c < 2 ? X[((long)b * 2 + c) * T * F + tf] : femb[(long)f * nemb + (c - 2)];
}

__device__ __forceinline__ float dn_hamming(int i, int n) { return 0.54f - 0.46f * cospif(2.0f * (float)i / (float)n); }

// STFT, center=False, periodic Hamming, hop = n/4: X[b][0|1][t][k], k <= n/2.  grid (frames, B)
__global__ __launch_bounds__(256) void dn_stft_kernel(const float* __restrict__ x, long x_bs, int L, float* __restrict__ X,
                                                      int log2n, int hop, int frames, const float2* __restrict__ tw) {
    __shared__ float2 a[FFT_LDS_LEN(1024)];
    const int n = 1 << log2n, nb = n / 2 + 1;
    const int t = blockIdx.x, b = blockIdx.y;
    const float* xb = x + (long)b * x_bs;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const long sidx = (long)t * hop + i;
        const float v = sidx < L ? xb[sidx] : 0.f;
        a[fft_at(bitrev_n(i, log2n))] = make_float2(v * dn_hamming(i, n), 0.f);
    }
    fft_lds_inplace(a, log2n, tw, -1);
    float* re = X + (((long)b * 2 + 0) * frames + t) * nb;
    float* im = X + (((long)b * 2 + 1) * frames + t) * nb;
    for (int k = threadIdx.x; k < nb; k += blockDim.x) {
        const float2 v = a[fft_at(k)];
        re[k] = v.x;
        im[k] = v.y;
    }
}

// frames[b][t][i] = w[i] * irfft(P[b][:][t][:])[i]
__global__ __launch_bounds__(256) void dn_istft_frames_kernel(const float* __restrict__ P, float* __restrict__ fr,
                                                              int log2n, int frames, const float2* __restrict__ tw) {
    __shared__ float2 a[FFT_LDS_LEN(1024)];
    const int n = 1 << log2n, nb = n / 2 + 1;
    const int t = blockIdx.x, b = blockIdx.y;
    const float* re = P + (((long)b * 2 + 0) * frames + t) * nb;
    const float* im = P + (((long)b * 2 + 1) * frames + t) * nb;
    for (int k = threadIdx.x; k < nb; k += blockDim.x) {
        if (k == 0 || k == n / 2) {
            a[fft_at(bitrev_n(k, log2n))] = make_float2(re[k], 0.f);          // C2R ignores Im at DC / Nyquist
        } else {
            a[fft_at(bitrev_n(k, log2n))] = make_float2(re[k], im[k]);
            a[fft_at(bitrev_n(n - k, log2n))] = make_float2(re[k], -im[k]);
        }
    }
    fft_lds_inplace(a, log2n, tw, +1);
    float* o = fr + ((long)b * frames + t) * n;
    const float inv = 1.f / (float)n;
    for (int i = threadIdx.x; i < n; i += blockDim.x) o[i] = a[fft_at(i)].x * inv * dn_hamming(i, n);
}

// y[b][j] = sum_t frames[b][t][j - t*hop] / sum_t w^2[j - t*hop]      (torch.istft, center=False), j < Lout
__global__ void dn_istft_ola_kernel(const float* __restrict__ fr, float* __restrict__ y, long y_bs, int n, int hop,
                                    int frames, int Lout, long total) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int j = (int)(i % Lout), b = (int)(i / Lout);
    int t1 = j / hop;
    if (t1 > frames - 1) t1 = frames - 1;
    float num = 0.f, den = 0.f;
    for (int t = t1; t >= 0 && j - t * hop < n; --t) {
        const int k = j - t * hop;
        const float w = dn_hamming(k, n);
        num += fr[((long)b * frames + t) * n + k];
        den += w * w;
    }
    y[(long)b * y_bs + j] = den > 1e-11f ? num / den : 0.f;
}

}  // namespace

extern "C" long babe_dn_packed_size(int Cout, int Cin, int KH, int KW) {
    return (long)KH * KW * ((Cin + 31) / 32 * 32) * ((Cout + 63) / 64 * 64);
}

extern "C" int babe_dn_pack_weights(const float* w, float* dst, int Cout, int Cin, int KH, int KW, int mode, int ph,
                                    int pw, void* stream) {
    BABE_CHECK_ARG(w && dst && Cout > 0 && Cin > 0, "dn_pack_weights: null/empty");
    BABE_CHECK_ARG(mode == 0 || (mode == 1 && KH == 2 && KW == 2 && (ph | 1) == 1 && (pw | 1) == 1),
                   "dn_pack_weights: mode 1 packs one parity of a 4x4 stride-2 transposed conv as a 2x2 kernel");
    BABE_CHECK_ARG(KH >= 1 && KH <= 7 && KW >= 1 && KW <= 7, "dn_pack_weights: kernel %dx%d unsupported", KH, KW);
    const int CinP = (Cin + 31) / 32 * 32, CoutP = (Cout + 63) / 64 * 64;
    const long total = (long)KH * KW * CinP * CoutP;
    hipLaunchKernelGGL(dn_pack_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, w, dst, Cout, Cin, KH,
                       KW, CinP, CoutP, mode, ph, pw, total);
    BABE_LAUNCH_CHECK();
    return BABE_OK;
}

extern "C" int babe_dn_conv2d(const babe_dnconv_args* ap, const float* w_packed, void* stream) {
    BABE_CHECK_ARG(ap && w_packed, "dn_conv2d: null args");
    const babe_dnconv_args& a = *ap;
    BABE_CHECK_ARG(a.in && a.out && a.B > 0 && a.Cin > 0 && a.Cout > 0 && a.OH > 0 && a.OW > 0, "dn_conv2d: bad shapes");
    BABE_CHECK_ARG(a.KH >= 1 && a.KH <= 7 && a.KW >= 1 && a.KW <= 7 && (a.stride == 1 || a.stride == 2),
                   "dn_conv2d: kernel %dx%d stride %d unsupported", a.KH, a.KW, a.stride);
    BABE_CHECK_ARG(a.pad_mode == 0 || (a.pad_t < a.IH && a.pad_l < a.IW && a.KH - 1 - a.pad_t < a.IH &&
                                       a.KW - 1 - a.pad_l < a.IW),
                   "dn_conv2d: reflect padding needs pad < input size (%dx%d)", a.IH, a.IW);
    BABE_CHECK_ARG(a.out_hstep >= 1 && a.out_wstep >= 1 && a.out_H > 0 && a.out_W > 0, "dn_conv2d: bad output mapping");
    BABE_CHECK_ARG(a.ksplit <= 1 || a.ws, "dn_conv2d: ksplit=%d needs a workspace of ksplit*B*Cout*OH*OW floats", a.ksplit);
    BABE_CHECK_ARG((double)a.in_cs * a.Cin < 4.0e9, "dn_conv2d: one batch item of the input must stay below 2^32 elements");
    hipStream_t s = (hipStream_t)stream;
    const double dn_flops = 2.0 * a.B * (double)a.Cout * a.Cin * a.KH * a.KW * (double)a.OH * a.OW;
    BabeProfScope prof(BABE_SLOT_DENOISER, 4.0 * a.B * ((double)a.Cin * a.IH * a.IW + (double)a.Cout * a.OH * a.OW),
                       dn_flops, dn_flops, stream);
    const int key = a.KH * 100 + a.KW * 10 + a.stride;
    switch (key) {
        case 331: dn_launch<3, 3, 1, 3, 8>(a, w_packed, s); break;
        case 771: dn_launch<7, 7, 1, 1, 8>(a, w_packed, s); break;
        case 442: dn_launch<4, 4, 2, 1, 8>(a, w_packed, s); break;
        case 221: dn_launch<2, 2, 1, 2, 8>(a, w_packed, s); break;
        case 111: dn_launch<1, 1, 1, 1, 32>(a, w_packed, s); break;
        default:
            babe_set_error("dn_conv2d: kernel %dx%d stride %d is not one of the denoiser's shapes (3x3, 7x7, 1x1, 2x2 "
                           "stride 1; 4x4 stride 2)", a.KH, a.KW, a.stride);
            return BABE_ERR_UNSUPPORTED;
    }
    BABE_LAUNCH_CHECK();
    return BABE_OK;
}

extern "C" int babe_dn_upsample_add(float* out, long out_bs, long out_cs, const float* low, long low_bs, long low_cs,
                                    int B, int C, int H, int W, int LH, int LW, int dh, int dw, void* stream) {
    BABE_CHECK_ARG(out && low && B > 0 && C > 0 && H > 0 && W > 0 && dh >= 0 && dw >= 0, "dn_upsample_add: bad args");
    const long total = (long)B * C * H * W;
    hipLaunchKernelGGL(dn_upsample_add_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, out, out_bs,
                       out_cs, low, low_bs, low_cs, C, H, W, LH, LW, dh, dw, total);
    BABE_LAUNCH_CHECK();
    return BABE_OK;
}

extern "C" int babe_dn_sam_gate(const float* x1, const float* m, const float* feats, long f_bs, long f_cs, float* out,
                                long out_bs, long out_cs, int B, int C, long hw, void* stream) {
    BABE_CHECK_ARG(x1 && m && feats && out && B > 0 && C > 0 && hw > 0, "dn_sam_gate: bad args");
    const long total = (long)B * C * hw;
    hipLaunchKernelGGL(dn_sam_gate_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, x1, m, feats, f_bs,
                       f_cs, out, out_bs, out_cs, C, hw, total);
    BABE_LAUNCH_CHECK();
    return BABE_OK;
}

extern "C" int babe_dn_fill_input(const float* X, const float* femb, float* out, int B, int T, int F, int nemb,
                                  void* stream) {
    BABE_CHECK_ARG(X && out && B > 0 && T > 0 && F > 0 && nemb >= 0 && (nemb == 0 || femb), "dn_fill_input: bad args");
    const long total = (long)B * (2 + nemb) * T * F;
    hipLaunchKernelGGL(dn_fill_input_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, X, femb, out, T, F,
                       nemb, total);
    BABE_LAUNCH_CHECK();
    return BABE_OK;
}

static int dn_log2(int n) {
    int l = 0;
    while ((1 << l) < n) ++l;
    return (1 << l) == n ? l : -1;
}

extern "C" int babe_dn_stft(const float* x, long x_bs, int L, float* X, int B, int nfft, int hop, int frames,
                            const float* tw4096, void* stream) {
    const int lg = dn_log2(nfft);
    BABE_CHECK_ARG(x && X && tw4096 && B > 0, "dn_stft: null args");
    BABE_CHECK_ARG(lg >= 6 && lg <= 10 && hop > 0 && hop <= nfft, "dn_stft: nfft=%d hop=%d unsupported", nfft, hop);
    BABE_CHECK_ARG(L >= nfft && frames == 1 + (L - nfft) / hop, "dn_stft: frames=%d inconsistent with L=%d", frames, L);
    hipLaunchKernelGGL(dn_stft_kernel, dim3(frames, B), dim3(256), 0, (hipStream_t)stream, x, x_bs, L, X, lg, hop, frames,
                       reinterpret_cast<const float2*>(tw4096));
    BABE_LAUNCH_CHECK();
    return BABE_OK;
}

extern "C" int babe_dn_istft(const float* P, float* frames_ws, float* y, long y_bs, int Lout, int B, int nfft, int hop,
                             int frames, const float* tw4096, void* stream) {
    const int lg = dn_log2(nfft);
    BABE_CHECK_ARG(P && frames_ws && y && tw4096 && B > 0, "dn_istft: null args");
    BABE_CHECK_ARG(lg >= 6 && lg <= 10 && hop > 0 && hop <= nfft, "dn_istft: nfft=%d hop=%d unsupported", nfft, hop);
    BABE_CHECK_ARG(Lout > 0 && Lout <= nfft + hop * (frames - 1), "dn_istft: Lout=%d exceeds the synthesised length", Lout);
    hipLaunchKernelGGL(dn_istft_frames_kernel, dim3(frames, B), dim3(256), 0, (hipStream_t)stream, P, frames_ws, lg, frames,
                       reinterpret_cast<const float2*>(tw4096));
    const long total = (long)B * Lout;
    hipLaunchKernelGGL(dn_istft_ola_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, frames_ws, y, y_bs,
                       nfft, hop, frames, Lout, total);
    BABE_LAUNCH_CHECK();
    return BABE_OK;
}
