/* babe_hip.h — C-ABI of the MI355X (gfx950) kernels behind the BABE blind-BWE hot path.
 *
 * The reference (eloimoliner/BABE) has no FFI: its "plug-in" boundary is Python classes
 * (utils/setup.py:47-96).  Each entry point below replaces the ATen call(s) cited beside it.
 * Conventions: raw device pointers (fp32 unless noted), explicit shapes/strides (in ELEMENTS),
 * hipStream_t passed as void*, every call is asynchronous on that stream, returns 0 on success
 * or a negative code (message: babe_last_error()).  The library never allocates on the hot
 * path; the caller owns all buffers and workspaces.
 */
#ifndef BABE_HIP_H
#define BABE_HIP_H
#ifdef __cplusplus
extern "C" {
#endif

const char* babe_version(void);
const char* babe_last_error(void);

/* Strided 4-D tensor view [B][C][F][T], T contiguous, row stride = T. */
typedef struct { float* p; long bs; long cs; } babe_view;

/* ---- Conv2d "same", no bias: networks/cqtdiff+.py:79-88 (F.conv2d) ------------------------
 * w_packed: [KH][KW][CinP][CoutP] from babe_conv_pack_weights (CinP = ceil8(Cin), CoutP = ceil32(Cout)).
 * Input channels [0,cin_split) come from `in`, [cin_split,Cin) from `in2` (torch.cat dim=1, :814).
 * out = alpha * acc * oscale[b,co] * (in_scale folded per input channel) + rbeta * res      */
typedef struct {
    const float* in;  long in_bs,  in_cs;
    const float* in2; long in2_bs, in2_cs; int cin_split;
    const float* w_packed;
    float* out;       long out_bs, out_cs;
    const float* res; long res_bs, res_cs;
    const float* in_scale;   /* [B][Cin] or NULL  */
    const float* oscale;     /* [B][Cout] or NULL */
    float alpha, rbeta;
    int B, Cin, Cout, F, T, KH, KW, dil;
    /* Optional reduction fused into the EPILOGUE of the F(4,5) x F(4,3) kernels (babe_conv2d_wino85; every other kernel ignores
     * these fields - zero them): the sums a GroupNorm pass would otherwise re-read the freshly written output for (round 6).
     *   stat_mode 1: stat_part[((b*G + g)*S + slot)*2 + {0,1}] = sum, sum of squares of the slot's outputs (the values stored in
     *                out) - the partial sums of the NEXT layer's GroupNorm, in the layout babe_scale_gelu_fin / babe_gn_finalize read
     *                with that S (what babe_gn_partial would re-read out for).
     *   stat_mode 2: stat_part[(b*G + g)*S + slot] = sum over the slot's outputs y of y * u * gelu'(u), u = stat_scale[b][c] *
     *                stat_x[b][c][f][t] - the S_g partial sums of the GroupNorm / FiLM / GELU input-VJP (babe_gn_bwd_partial) for
     *                the transposed conv that writes da; stat_x: dense [B][Cout][F][T] (the layer's saved input), stat_scale [B][Cout].
     * G = Cout / stat_cg groups of stat_cg channels (a multiple of 4); S = babe_conv2d_wino85_stat_slots(a) slots per group, each
     * written exactly once (deterministic), in the layout babe_gn_bwd_apply reads with that S. */
    int stat_mode, stat_cg;
    const float* stat_x; const float* stat_scale; double* stat_part;
} babe_conv_args;
/* slots per GroupNorm group the fused reduction writes: (row-quad x time tiles of one batch element) * stat_cg / sg, sg = the largest
 * of 16, 8, 4 that divides stat_cg */
int babe_conv2d_wino85_stat_slots(const babe_conv_args* a);
int babe_conv2d(const babe_conv_args* a, void* stream);
/* w: [Cout][Cin][KH][KW] (reference layout) -> packed; transpose_flip=1 builds the bwd-data weights
 * (w'[ci][co][KH-1-kh][KW-1-kw]) so that babe_conv2d computes the input-VJP (replaces autograd's
 * convolution_backward for the input).  dst size = KH*KW*ceil8(Cin')*ceil32(Cout') floats. */
int babe_conv_pack_weights(const float* w, float* dst, int Cout, int Cin, int KH, int KW,
                           int transpose_flip, void* stream);
long babe_conv_packed_size(int Cout, int Cin, int KH, int KW, int transpose_flip);
/* Same with an explicit row-tile count per workgroup (nt x 32 output channels; 0 = the default, the largest of 4..1 that
 * divides ceil32(Cout)/32).  nt = 1 gives the most workgroups: used for the dense DFT stages of the length-L real FFT,
 * whose "positions" are only a few hundred (networks/cqtdiff+.py:743 -> CQT_nsgt.fwd -> torch.fft).  Weights packed with
 * a given nt must be used with the same nt. */
int babe_conv2d_nt(const babe_conv_args* a, int nt, void* stream);
int babe_conv_pack_weights_nt(const float* w, float* dst, int Cout, int Cin, int KH, int KW, int transpose_flip, int nt,
                              void* stream);
/* bf16-MFMA variants (fp32 tensors in HBM, bf16 operands, fp32 accumulate): splits=1 plain bf16 (configs #3-#5),
 * splits=2 "bf16x3" (hi/lo split of both operands, three products: 16-bit-mantissa multiplies).
 * w_bf16 from babe_conv_pack_weights_bf16: [splits][KH][KW][ceil16(Cin)/8][ceil32(Cout)][8] bf16. */
int babe_conv2d_bf16(const babe_conv_args* a, const void* w_bf16, int splits, void* stream);
int babe_conv_pack_weights_bf16(const float* w, void* dst, int Cout, int Cin, int KH, int KW, int transpose_flip,
                                int splits, void* stream);
long babe_conv_packed_size_bf16(int Cout, int Cin, int KH, int KW, int transpose_flip, int splits);
/* bf16 "units" path for the forward (5,3) layers under precision='bf16' (csrc/conv_bf16p.hip, UNITS variant).  The
 * producer of a conv's input - GroupNorm scale * GELU, networks/cqtdiff+.py:472-482 - writes it as bf16 units instead of
 * fp32: a unit = 8 consecutive channels of one (f, t) = 16 bytes = one lane's MFMA operand fragment; tensor layout
 * [B][C/8][F][4 planes][T/4 + 1] units, plane p entry j = time step 4j + p - 1 (t = -1 and t >= T stored as zeros).
 * babe_units_size = units per batch item.  babe_conv2d_bf16_units takes such a tensor as a->in (a->in_bs / a->in_cs = units
 * per batch item / per 8-channel group = F*4*(T/4+1); no second source, no in_scale) and moves BOTH operands by LDS-DMA;
 * result identical to babe_conv2d_bf16 (splits = 1) on the fp32 tensor.  Needs KH x KW = 5 x 3, T % 4 == 0, Cin % 8 == 0,
 * Cout > 32, 16-byte aligned out / res rows (babe_conv2d_bf16_units_supported). */
long babe_units_size(int C, int F, int T);
int babe_scale_gelu_units(const float* x, const float* scale, void* units, int B, int C, int F, int T, void* stream);
int babe_conv2d_bf16_units_supported(const babe_conv_args* a);
int babe_conv2d_bf16_units(const babe_conv_args* a, const void* w_bf16, void* stream);
/* Winograd F(2,3)-along-time variant of the exact-fp32 conv for KW == 3 (csrc/conv_wino.hip): 2/3 of the MFMA work,
 * same result up to fp32 rounding.  w_wino from babe_conv_pack_weights_wino: [KH][ceil8(Cin)][ceil32(Cout)][4].
 * babe_conv2d_wino_supported() tells whether a problem qualifies (even T, aligned rows, tileable Cout). */
int babe_conv2d_wino(const babe_conv_args* a, const float* w_wino, void* stream);
int babe_conv2d_wino_supported(const babe_conv_args* a);
int babe_conv_pack_weights_wino(const float* w, float* dst, int Cout, int Cin, int KH, int KW, int transpose_flip,
                                void* stream);
long babe_conv_packed_size_wino(int Cout, int Cin, int KH, int transpose_flip);
/* Winograd F(4,3)-along-time variant (csrc/conv_wino4.hip): half of the direct kernel's MFMA work, fp32, about 2.5x the
 * rounding error of F(2,3) (1e-6 relative).  w_wino4 from babe_conv_pack_weights_wino4:
 * [KH][ceil8(Cin)][2][ceil32(Cout)][4].  Needs T % 4 == 0, 16-byte aligned in/out/res rows, Cout in 64, 96 or a multiple
 * of 128 (after rounding up to 32). */
int babe_conv2d_wino4(const babe_conv_args* a, const float* w_wino4, void* stream);
int babe_conv2d_wino4_supported(const babe_conv_args* a);
int babe_conv_pack_weights_wino4(const float* w, float* dst, int Cout, int Cin, int KH, int KW, int transpose_flip,
                                 void* stream);
long babe_conv_packed_size_wino4(int Cout, int Cin, int KH, int transpose_flip);
/* NESTED Winograd variant (csrc/conv_wino45.hip, round 3): F(2,5) along frequency x F(4,3) along time, 4.5 multiplies per
 * output instead of 7.5 (0.6 of the F(4,3) kernel's matrix work, 0.3 of the direct kernel's), fp32 MFMA, rounding error
 * about 3.4x the F(4,3) kernel's (2-3e-6 relative).  Replaces the same F.conv2d call (cqtdiff+.py:85) and its input-VJP.
 * w_wino45 from babe_conv_pack_weights_wino45: [3 passes][ceil8(Cin)][ceil64(Cout)][12].  Needs KH x KW = 5 x 3,
 * T % 4 == 0, T >= 64, Cin % 16 == 0, Cout > 32, 16-byte aligned in/out/res rows, ONE source (no in2), views < 1 GiB. */
/* Two tilings behind the one entry point: 64-channel tiles x 64 units (any Cout; a PADC variant skips the MFMAs of waves whose
 * channels are padding, e.g. Cout = 96) and, for Cout % 128 == 0, 128-channel tiles x 32 units that transform 16 input
 * channels at a time (BABE_CONV_WINO45W=0 turns the second one off).  Same arithmetic in the same order in both. */
int babe_conv2d_wino45(const babe_conv_args* a, const float* w_wino45, void* stream);
int babe_conv2d_wino45_supported(const babe_conv_args* a);   /* the kernel CAN run this problem */
int babe_conv2d_wino45_preferred(const babe_conv_args* a);   /* ... and its tiles are full enough to beat the F(4,3) kernel */
int babe_conv_pack_weights_wino45(const float* w, float* dst, int Cout, int Cin, int KH, int KW, int transpose_flip,
                                  void* stream);
long babe_conv_packed_size_wino45(int Cout, int Cin, int transpose_flip);
/* (KH,3) conv with at most 4 OUTPUT channels on the vector ALU (csrc/conv_fewco.hip): the input-VJP of the UNet's 2-channel
 * pyramid projections (cqtdiff+.py:676, 794).  w is in the REFERENCE layout, not packed: [Cout][Cin][KH][3] for
 * transpose_flip = 0; for transpose_flip = 1 the weights [Cin][Cout][KH][3] of the conv whose input-VJP is computed (a->Cin,
 * a->Cout describe the op as executed).  One source, no in_scale. */
int babe_conv2d_fewco(const babe_conv_args* a, const float* w, int transpose_flip, void* stream);
int babe_conv2d_fewco_supported(const babe_conv_args* a);
/* Measurement hook (bench.py; csrc/prof.h lists the slots): when enabled EVERY entry point of this library brackets
 * its launch with HIP events on the launch stream and tallies, per slot, the kernel time, the ALGORITHMIC bytes and flops
 * (conv: 2*B*Cout*Cin*KH*KW*F*T with the unpadded channel counts) and the flops the matrix pipe executes (Winograd
 * F(2,3): 2/3, F(4,3): 1/2 of the algorithmic count).  babe_prof_read() waits for the events, fills arrays of
 * babe_prof_nslots() entries and resets.  babe_prof_dispatch_counts() is always on: launches per slot, i.e. which
 * kernel a conv call really dispatched to.  Diagnostics only; single host thread. */
int babe_prof_nslots(void);
const char* babe_prof_slot_name(int slot);
int babe_prof_enable(int on);          /* on < 0: query, returns 1 / 0 */
int babe_prof_conv_slot(int slot);      /* >= 0: tally conv launches in that slot (the CQT's dense DFT stages); -1: off */
int babe_prof_read(double* ms, double* bytes, double* flops, double* exec_flops, long* launches);
int babe_prof_dispatch_counts(long* counts, int reset);
/* per-launch timeline of the records pending since the last babe_prof_read (call before it): event times in ms relative to the
 * first record, slot, lane (stream index in order of first appearance), tallied flops; returns the count written (<= cap) */
long babe_prof_timeline(double* t0_ms, double* t1_ms, int* slot, int* lane, double* flops, long cap);
long babe_prof_pending(void);

/* ---- BiasFreeGroupNorm + FiLM + GELU: cqtdiff+.py:147-163, :472-482 ------------------------ */
/* partial sums (double) of x and x^2 per (b,group,split): part[(b*G+g)*S+s] = {sum, sumsq} */
int babe_gn_partial(const float* x, double* part, int B, int G, long n_per_group, int S, void* stream);
/* babe_gn_partial + babe_gn_finalize in one launch (the workgroup that finishes a group last finalises it; bit-identical
 * results).  ticket: B*G ints, zero before the first use and left zero by every call; one buffer per concurrently used stream. */
int babe_gn_stats(const float* x, double* part, int* ticket, const float* gamma, const float* film, long film_bs,
                  float* stats, float* scale, int B, int C, int G, long n_per_group, int S, float eps, void* stream);
/* stats[b*G+g] = {mean, std, 1/(std+eps)}; scale[b][c] = gamma[c]*(film[b][c]+1)/(std+eps) */
int babe_gn_finalize(const double* part, const float* gamma, const float* film, long film_bs,
                     float* stats, float* scale, int B, int C, int G, long n_per_group, int S,
                     float eps, void* stream);
/* babe_gn_finalize + babe_scale_gelu in one launch (every workgroup re-derives its channel's scale from the partial sums in
 * gn_finalize's order; the first one stores scale / stats for the VJP): bit-identical to the two calls. */
int babe_scale_gelu_fin(const float* x, const double* part, const float* gamma, const float* film, long film_bs, float* stats,
                        float* scale, float* a, int B, int C, int G, long hw, int S, float eps, void* stream);
/* a = gelu(x * scale[b][c]) */
int babe_scale_gelu(const float* x, const float* scale, float* a, int B, int C, long hw, void* stream);
/* VJP pass 1: with du = da * gelu'(x*scale) (never stored): part[(b*G+g)*S+s] = sum(du * scale*(std+eps) * x) */
int babe_gn_bwd_partial(const float* x, const float* da, const float* scale, double* part,
                        int B, int C, int G, long hw, int S, void* stream);
/* VJP pass 2: gx = rbeta*gy + scale*du - (x-mean)*coef_g with du recomputed from da; coef from part and stats */
int babe_gn_bwd_apply(const float* x, const float* da, const float* gy, const float* scale,
                      const float* stats, const double* part, float* gx, float rbeta,
                      int B, int C, int G, long hw, int S, float eps, void* stream);
/* the same pass with a ResnetBlock's VJP tail merged in: gx_out = ca*acc + cb*(the gx babe_gn_bwd_apply would store) - the residual
 * path's rs2*g_out plus the main path's rs2*gz of an N -> N block, without storing gz and re-reading it (acc dense, 16-byte aligned) */
int babe_gn_bwd_apply_merge(const float* x, const float* da, const float* gy, const float* scale,
                            const float* stats, const double* part, float* gx, float rbeta,
                            int B, int C, int G, long hw, int S, float eps, void* stream,
                            const float* acc, float ca, float cb);

/* ---- UpDownResample ('cubic', reflect): cqtdiff+.py:549-580 (conv1d / conv_transpose1d with a
 * dense diagonal weight) as a depth-wise 8-tap polyphase FIR.  mode: 0 down, 1 up, 2 down^T, 3 up^T.
 * T = time length of the INPUT of the forward op (for the adjoints: of the forward op's input too). */
int babe_resample(const float* in, long in_bs, long in_cs, float* out, long out_bs, long out_cs,
                  int B, int C, int F, int T, int mode, float alpha, float beta, void* stream);
/* out = alpha*resample(in) + beta*res with res a separate tensor of out's shape (16-byte aligned views): the encoder VJP's
 * g_H = g_skip + rs2 * down^T(g_P) (cqtdiff+.py:776-794 backwards) without first copying g_skip into out */
int babe_resample_res(const float* in, long in_bs, long in_cs, const float* res, long res_bs, long res_cs, float* out,
                      long out_bs, long out_cs, int B, int C, int F, int T, int mode, float alpha, float beta, void* stream);

/* ---- sample-rate conversion of the file-level flows: torchaudio.functional.resample as published (Hann-windowed sinc,
 * lowpass_filter_width 6, rolloff 0.99), testing/blind_bwe_tester.py:410,744,930, testing/denoise_and_bwe_tester.py:282-289.
 * orig / new_: the two rates divided by their gcd; kernel [new_][2 width + orig] (host-built, babe_amd/resample.py);
 * krange [new_][2] = first and one-past-last non-zero tap of each phase.  out[b][i new_ + j] = sum_k kernel[j][k] xpad[b][i orig + k]
 * with xpad = x padded by (width, width + orig) zeros; L_out = ceil(new_ L_in / orig). */
int babe_resample_sinc(const float* x, long x_bs, float* out, long out_bs, int B, long L_in, long L_out, const float* kernel,
                       const int* krange, int orig, int new_, int width, void* stream);

/* ---- nested Winograd F(4,5) along frequency x F(4,3) along time for the same (5,3) convs as babe_conv2d_wino45 (Conv2d at
 * networks/cqtdiff+.py:79-88, 433-436): 3.0 multiplies per output instead of 4.5 (csrc/conv_wino85.hip).  Output-channel tiles of
 * 128, 96 or 64 (Cout a multiple of one of them), Cin % 16 == 0, one source; weights [2 passes][Cin/4][2][Cout/16][3][4][16][4] from
 * babe_conv_pack_weights_wino85.  _preferred: the problem is supported AND its row quads x time tiles are >= 80 % full - what
 * babe_conv2d_auto and babe_amd/ops.py::conv2d dispatch on (BABE_CONV_F45=0 in the Python host leaves everything to wino45). */
long babe_conv_packed_size_wino85(int Cout, int Cin, int transpose_flip);
int babe_conv_pack_weights_wino85(const float* w, float* dst, int Cout, int Cin, int KH, int KW, int transpose_flip, void* stream);
int babe_conv2d_wino85_supported(const babe_conv_args* a);
int babe_conv2d_wino85_preferred(const babe_conv_args* a);
int babe_conv2d_wino85(const babe_conv_args* a, const float* w_wino85, void* stream);
/* Form of the kernel's 128-channel tile: 12 (default; 8 multiplying + 4 transform waves, three per SIMD) or 8 (round 5: waves 0-3
 * transform and multiply).  Bit-identical outputs; BABE_W85_12W=0 selects 8 for the process. */
int babe_conv2d_wino85_set_waves(int waves);

/* ---- the whole UNet body from one call: networks/cqtdiff+.py:746-839 (forward), ResnetBlock :452-493, and its input-VJP
 * (the autograd pass of testing/blind_bwe_sampler.py:120).  csrc/unet_engine.hip sequences the op-level functions of this header
 * exactly as babe_amd/networks/unet_engine.py does (bit-identical results); fp32 convs.  All device memory is the caller's. */
typedef struct {                       /* every image babe_amd/ops.py::PackedConv builds for one Conv2d; NULL = not packed */
    int Cout, Cin, KH, KW, nt, splits; /* Cout == 0: the layer does not exist; splits: 0 fp32, 1 bf16, 2 bf16x3 */
    const void *fwd, *bwd;             /* babe_conv_pack_weights_nt (fp32) or babe_conv_pack_weights_bf16 images, transpose_flip 0 / 1 */
    const float *fwd_wino, *bwd_wino, *fwd_wino4, *bwd_wino4, *fwd_wino45, *bwd_wino45;
    const float* w_raw;                /* reference layout, for babe_conv2d_fewco (<= 4 channels on one side) */
    const float *fwd_wino85, *bwd_wino85;  /* babe_conv_pack_weights_wino85 images, or NULL */
} babe_packed_conv;
/* conv with the kernel chosen by the library; a->w_packed, Cin, Cout, KH, KW are filled in from pc / transpose */
int babe_conv2d_auto(babe_conv_args* a, const babe_packed_conv* pc, int transpose, void* stream);
typedef struct {                       /* one ResnetBlock */
    int N, nd, k53;                    /* channels of the dilated stack, dilation layers (dilation 2^d when k53), (5,3) kernels? */
    babe_packed_conv proj_in, res_conv, proj_out, H[8];
    const float* gamma[8];             /* BiasFreeGroupNorm gamma of layer d, [N] */
    int film_aff[8], film_gate[8];     /* offsets of layer d's affine / gate vectors in the FiLM row */
} babe_unet_block;
typedef struct {
    int nocts, bpo;                    /* octaves (7), bins per octave (64) */
    int Ns[8];
    babe_unet_block init_blk[8], main_blk[8], up_out[8], up_blk[8], mid_blk, mid_out;
    babe_packed_conv pyr_conv[8];
} babe_unet_plan_desc;
void* babe_unet_plan_create(const babe_unet_plan_desc* desc);      /* copies the descriptor (pointers stay the caller's) */
void babe_unet_plan_destroy(void* plan);
void* babe_unet_state_create(void);                                 /* one evaluation's bookkeeping; one per concurrent stream */
void babe_unet_state_destroy(void* state);
long babe_unet_workspace_bytes(const void* plan, int B, const int* T_oct);
int babe_unet_fwd(const void* plan, void* state, const float* const* C_in, const float* film, long film_bs, int B,
                  const int* T_oct, void* workspace, long workspace_bytes, float* const* outs, void* stream);
int babe_unet_vjp(const void* plan, void* state, const float* const* gouts, float* const* gC, void* stream);

/* ---- strided copy / axpby: out = alpha*in + beta*out on [B][C][F][T] views (torch.cat / slicing /
 * (a+b)/sqrt2 residual merges, cqtdiff+.py:769-774,794,814-822) */
int babe_axpby4d(const float* in, long in_bs, long in_cs, float* out, long out_bs, long out_cs,
                 int B, int C, int F, int T, float alpha, float beta, void* stream);
/* out[b][c][:, :] = value on a [B][C][F][T] view */
int babe_fill4d(float* out, long out_bs, long out_cs, int B, int C, int F, int T, float value, void* stream);
/* out = alpha*x + beta*y in one pass (ResnetBlock's (x + h)/sqrt2 without res_conv, cqtdiff+.py:493); every view 16-byte aligned,
 * F*T % 4 == 0 */
int babe_axpby2_4d(const float* x, long x_bs, long x_cs, const float* y, long y_bs, long y_cs, float* out, long out_bs,
                   long out_cs, int B, int C, int F, int T, float alpha, float beta, void* stream);

/* ---- small dense: out[b][j] = act(sum_k x[b][k] W[j][k] + bias[j]); Linear :36-40, RFF_MLP :184-211 */
int babe_linear(const float* x, const float* W, const float* bias, float* out, int B, int K, int J,
                int relu, void* stream);
/* table = 2*pi*cnoise[b]*freq[j]; out[b] = [sin(table), cos(table)]  (:199-211) */
int babe_rff(const float* cnoise, const float* freq, float* out, int B, int R, void* stream);

/* ---- constant-Q transform (NSGT, mode "oct"): cqt_nsgt_pytorch.CQT_nsgt.fwd/.bwd/.apply_hpf_DC, call
 * sites networks/cqtdiff+.py:743,841 and testing/blind_bwe_sampler.py:156.  The length-L real DFT is a
 * four-step N1 x N2 transform whose two dense DFT stages run on babe_conv2d (1x1 convs with DFT matrices);
 * the entry points below are the remaining pieces.  Spectra are planar: spec[b][0][k] = Re, spec[b][1][k] = Im,
 * k = k1 + N1*k2 stored as [K2][N1] (natural order), KX = K2*N1 >= L/2+1. */
/* forward: in [B][2*N1][N2] (Re rows k1 then Im rows) -> out [B][2*N2][N1] = transpose(in * tw[k1][n2]);
 * adjoint=1: in [B][2*N2][N1] -> out [B][2*N1][N2] = transpose(in) * conj(tw). tw: [N1][N2] float2. */
/* Mixed-radix variant (csrc/fft_mixed.hip, round 4): the same four-step transform with both stages as Stockham FFTs of
 * length N1 / N2 in LDS (radices 2,3,4,5,7,11,13,23) - two launches per transform, no dense DFT matrices.  direction 0:
 * x_in [B][N1*N2] -> spec_out planar [B][2][K2*N1];  direction 1: the transpose, spec_in -> x_out (real part of the unnormalised
 * inverse DFT of the zero-padded half spectrum).  rad1 / rad2: radices whose product is N1 / N2 (at most 6 each); w1 / w2:
 * exp(-2 pi i j / N1|N2) as float2[N]; tw: [N1][N2] float2 exp(-2 pi i k1 n2 / L); work: [B][2][N1*N2] floats of scratch. */
int babe_rfft_mixed(const float* x_in, float* spec_out, const float* spec_in, float* x_out, float* work, int B, int N1,
                    int N2, int K2, const int* rad1, int nrad1, const int* rad2, int nrad2, const float* w1, const float* w2,
                    const float* tw, int direction, void* stream);
int babe_fft_twiddle_transpose(const float* in, float* out, const float* tw, int B, int N1, int N2,
                               int adjoint, void* stream);
typedef struct {
    int nbands; int L; int KX;
    const int* c;        /* [nbands] centre bin                               */
    const int* M;        /* [nbands] window length                            */
    const int* woff;     /* [nbands] offset of the band in the window tables  */
    const int* log2T;    /* [nbands] log2 of the band's coefficient count     */
    const int* oct;      /* [nbands] octave index (into coef[])               */
    const int* binoct;   /* [nbands] bin index inside the octave              */
    const float* tw4096; /* [2048] float2 exp(-2 pi i q/4096)                 */
    int nocts; int binsoct;
    float* coef[8];      /* per octave planar [B][2][binsoct][T_oct]          */
    /* workgroup table: workgroup w transforms bands wg_first[w] .. wg_first[w]+wg_count[w]-1, all of one octave
     * (same T), wg_count[w] * T <= 4096 points */
    const int* wg_first; const int* wg_count; int nwg;
    /* the same tables packed for the kernel (one 16-byte load each instead of chains of dependent 4-byte loads - a
     * workgroup lives ~6 us and spent a third of that fetching its own description):
     * wg_rec[w] = {wg_first, wg_count, log2T, oct | binoct << 8},  band_rec[k] = {c, M, woff, 0}   (int4 each) */
    const int* wg_rec; const int* band_rec;
    int abl;             /* timing-only ablation bits (1: skip the FFT passes); read only by -DBABE_CQT_ABL builds */
    /* host-side summary of the DEVICE tables above, validated by the entry points (the kernel keeps wg_count bands of
     * 3 ints in LDS and dispatches on log2T): max of wg_count (<= 64), min / max of log2T (2..12) */
    int max_wg_count, min_log2T, max_log2T;
    long sum_T, sum_M;   /* sum over bands of T_k and M_k (for the measurement hook's algorithmic bytes) */
    double sum_TlogT;    /* sum over bands of T_k*log2(T_k) (algorithmic FFT flops = 5*that)            */
    /* Analytic Kaiser window (round 6): with win == NULL the band kernels evaluate the window g_k(m) / T_k themselves,
     * g_k(m) = I0(beta sqrt(a)) / I0(beta), a = 1 - (2 m / M_k)^2, as the polynomial sum_j kpoly[j] a^j, j <= kdeg
     * (kpoly[j] = (beta^2/4)^j / (j!)^2 / I0(beta): the power series of I0, truncated below 1e-9) - the table read was
     * 4 bytes per point and clip, 6 of the 45 us of the analysis launch at 32 clips.  kdeg = 0: not available (large
     * beta), the table has to be passed. */
    int kdeg; float kpoly[12];
} babe_cqt_bands;
/* analysis-type: coef_k = IFFT_T(fold(spec[(c_k+m) mod L] * win[woff_k+m']))  (win carries 1/T and any scale).
 * Used for CQT.fwd (win = g/T, or NULL: evaluated analytically, bands->kdeg > 0) and for the adjoint of CQT.bwd
 * (win = (2/L) T^2 gd). */
int babe_cqt_band_analysis(const babe_cqt_bands* bands, const float* spec, const float* win, int B, void* stream);
/* synthesis-type: bs[b][woff_k+m'] = FFT_T(coef_k)[m mod T] * win[woff_k+m']  (float2).
 * Used for CQT.bwd (win = T gd) and for the adjoint of CQT.fwd (win = g/T, or NULL: analytic). */
int babe_cqt_band_synthesis(const babe_cqt_bands* bands, float* bs, const float* win, long bs_stride, int B,
                            void* stream);
/* spec[b][:, n] = scale * sum over CSR entries of n of bs (conjugated when the entry's sign bit is set);
 * optionally multiplied by mul[n] (real, e.g. the DC/Nyquist high-pass) ; n > L/2 -> 0.
 * rec (optional, [L/2+1] int4): the same CSR as fixed records {src0, src1, src2, count}, count <= 3 for every bin; when
 * given, the record kernel runs (same sums in the same order), otherwise the CSR walk. */
int babe_cqt_gather(const float* bs, long bs_stride, const int* rowptr, const int* src, const int* rec, float* spec,
                    int KX, int L, float scale, const float* mul, int B, void* stream);
/* spec_out = spec_in * mul[n] * scale for n <= L/2, 0 above (apply_hpf_DC in the frequency domain);
 * optional second term: spec_out += spec2 * mul[n] * scale2. */
int babe_spec_scale(const float* spec_in, const float* spec2, float* spec_out, const float* mul, int KX, int L,
                    float scale, float scale2, int B, void* stream);

/* ---- the whole constant-Q transform from a plan handle (csrc/cqt_plan.hip, round 6): what a non-Python host binds instead of
 * cqt_nsgt_pytorch.CQT_nsgt(numocts, binsoct, mode="oct", window=("kaiser", beta), fs, audio_len) - constructed at
 * networks/cqtdiff+.py:620, .fwd :743, .bwd :841, .apply_hpf_DC testing/blind_bwe_sampler.py:156.  The band design (NSGT LogScale
 * grid, band lengths, Kaiser windows, painless-case dual frame incl. the mirrored bands, power-of-two rasterisation per octave,
 * the CSR of the overlap-add, the mixed-radix plan of the length-L real FFT) is computed by the library in float64 with the
 * arithmetic of babe_amd/cqt.py::design_bands (integer tables equal, float tables to a few ulp: tests/test_cqt_plan_cpu.py).
 *   design: host only, no GPU call.  babe_cqt_design_get copies table `name` (int64: M c T woff idx rowptr src; float64: f Om g
 *   gdual Tw hpf kpoly; int32: rad1 rad2 wg_first wg_count T_oct; one int64: nb nwin M_dc N1 N2 K2 KX kdeg) into out and returns
 *   its size in bytes (out = NULL: size only), -1 for an unknown name.
 *   plan: design + device tables (one allocation on the current device).  Coefficients are planar, one tensor per octave:
 *   coef[j] = [B][2][binsoct][T_oct[j]], j = 0 the LOWEST octave (the [B,2,64,T] tensors the UNet consumes).  ws: device scratch of
 *   babe_cqt_workspace_bytes(plan, B) bytes, owned by the caller (one per concurrent stream).  NULL / negative on failure
 *   (babe_last_error()). */
void* babe_cqt_design_create(double fs, int audio_len, int numocts, int binsoct, double beta);
void babe_cqt_design_destroy(void* design);
long babe_cqt_design_get(const void* design, const char* name, void* out, long capacity_bytes);
void* babe_cqt_plan_create(double fs, int audio_len, int numocts, int binsoct, double beta);
void babe_cqt_plan_destroy(void* plan);
const void* babe_cqt_plan_design(const void* plan);                  /* the plan's design (owned by the plan), for babe_cqt_design_get */
long babe_cqt_workspace_bytes(const void* plan, int B);
int babe_cqt_fwd(const void* plan, const float* x, float* const* coef, float* ws, int B, void* stream);            /* .fwd: x [B][L] -> coef */
int babe_cqt_bwd(const void* plan, float* const* coef, float* x, float* ws, int B, void* stream);                  /* .bwd: coef -> x [B][L] */
int babe_cqt_fwd_adjoint(const void* plan, float* const* gcoef, float* gx, float* ws, int B, void* stream);        /* (.fwd)^T, for the VJP */
int babe_cqt_bwd_adjoint(const void* plan, const float* gx, float* const* gcoef, float* ws, int B, void* stream);  /* (.bwd)^T, for the VJP */
int babe_cqt_hpf(const void* plan, const float* x, float* out, float* ws, int B, void* stream);                    /* .apply_hpf_DC (self-adjoint) */

/* ---- STFT-domain degradation model: utils/blind_bwe_utils.py:6-39 (apply_stft / apply_filter_istft /
 * apply_filter) and testing/blind_bwe_sampler.py:518-595.  nfft in {256..4096} (power of two), hop nfft/2,
 * periodic Hamming window, NFFT zeros appended, frames = 1 + L/hop.  spec: [B][frames][nfft/2+1] float2. */
/* spec = rfft(frame * w); if pre (length nfft+hop*(frames-1)) != NULL the signal is multiplied by pre[n] first
 * (1/envelope, used by the adjoint of the iSTFT normalisation). */
int babe_stft_fwd(const float* x, long x_bs, int L, const float* pre, float* spec, int B, int nfft, int frames,
                  const float* tw4096, void* stream);
/* frames_out[b][t][:] = w * irfft(spec[b][t] * H[b or 0][:]) ; H stride H_bs (0 = one filter for the batch) */
int babe_spec_filter_istft(const float* spec, const float* H, long H_bs, float* frames_out, int B, int nfft,
                           int frames, const float* tw4096, void* stream);
/* overlap-add of frames, times post[n] if given (1/envelope), cropped to L.
 * y == NULL: out = ola.   y != NULL: out = y - ola (the residual) and part[b][blk] = sum of squares (double). */
int babe_ola(const float* frames_in, const float* post, const float* y, long y_bs, float* out, long out_bs,
             double* part, int nblk, int B, int L, int nfft, int frames, void* stream);
/* out[b][n] = -r[b][n] / ||r_b|| * post[n]  (seed of the guidance VJP: d||y-Ax|| / d(pre-normalisation OLA)).
 * shared_norm=1 is NOT used for the norm (the reference's norm is per item, blind_bwe_sampler.py:117). */
int babe_residual_seed(const float* r, long r_bs, const double* part, int nblk, const float* post, float* out,
                       long out_bs, int B, int L, void* stream);
/* per-bin sufficient statistics of the magnitude fit: stats[b][0..2][k] (double) =
 * sum_t |X|^2, sum_t |X||Y|, sum_t |Y|^2.  shared=1 sums over the batch too (reference batch semantics). */
int babe_stft_mag_stats(const float* specX, const float* specY, double* stats, int B, int nbins, int frames,
                        int shared, void* stream);
/* design_filter (utils/blind_bwe_utils.py:82-119): params [P][2][K] (fc row, A row) -> H [P][nbins];
 * bin frequencies are float32(k) * float32(fs/nfft) and masks are evaluated in float32 like the reference. */
int babe_design_filter(const float* params, float* H, int P, int K, int nbins, float fs, int nfft, void* stream);
typedef struct {
    float mu_fc, mu_A, tol_fc, tol_A, fcmin, fcmax, Amin, Amax;
    int max_iter, clamp_fc, clamp_A, only_negative_A, weighting; /* 0 None, 1 sqrt, 2 linear, 3 log */
    int kernel;  /* 0: filter_fit_fast_kernel (register-resident, v_exp/v_log segments; default), 1: filter_fit_kernel, the first
                  * kernel, which keeps the reference's operation order inside an iteration; both are pinned against the same
                  * goldens, and against each other over full runs (tests/test_gpu_stft.py) */
} babe_fit_cfg;
/* BlindSampler.fit_params (:533-595): projected gradient descent on (fc, A) from the statistics.
 * params [P][2][K] updated in place; n_iter[P] receives the iteration count. K <= 8. */
int babe_filter_fit(const double* stats, float* params, int* n_iter, int P, int K, int nbins, float fs, int nfft,
                    const babe_fit_cfg* cfg, void* stream);
/* The objective of that fit and its gradient at given parameters, no descent step (BlindSampler.optimizer_func + autograd,
 * :523-533, as compute_sweep :598-616 evaluates them on a grid): lossgrad [P][1 + 2K] = loss, d loss / d fc_j, d loss / d A_j for
 * each of the P parameter sets params [P][2][K]; stats_pstride: doubles between the statistics of consecutive sets (3 * nbins for
 * one set each, 0 for ONE set of statistics shared by all P).  Reference-order kernel (cfg.kernel is ignored). */
int babe_filter_loss_grad(const double* stats, long stats_pstride, const float* params, float* lossgrad, int P, int K,
                          int nbins, float fs, int nfft, const babe_fit_cfg* cfg, void* stream);

/* ---- known FIR degradation (config #1): F.conv1d(y[B,1,L], taps[1,1,ntaps], padding="same"),
 * testing/edm_sampler.py:245-252, utils/bandwidth_extension.py:76-95.  PyTorch pads (ntaps-1)/2 on the left and the
 * rest on the right (even ntaps: 249 | 250) and does not flip the kernel: out[n] = sum_k taps[k] x[n+k-padl].
 * adjoint=1 computes the transpose (the input-VJP). */
int babe_fir_same(const float* x, long x_bs, const float* taps, int ntaps, float* out, long out_bs, int B, int L,
                  int adjoint, void* stream);

/* ---- sampler element-wise steps: testing/blind_bwe_sampler.py:503-516, :125-135, :701-761; edm.py:144-159 */
/* out = a*x + b*y + c*z (y, z optional) over n elements */
int babe_lincomb3(float* out, float a, const float* x, float b, const float* y, float c, const float* z, long n,
                  void* stream);
/* y[b] += sqrt(var(y[b]) / snr) * noise[b] in place, var = unbiased sample variance over the n samples of clip b; snr is the
 * LINEAR ratio 10^(SNR_observations / 10).  Replaces the observation-noise lines of get_rec_grads and fit_params
 * (testing/blind_bwe_sampler.py:80-86, :542-548; conf/tester/blind_bwe_2.yaml SNR_observations: 50). */
int babe_add_obs_noise(float* y, long y_bs, const float* noise, long noise_bs, float snr, int B, long n, void* stream);
/* out[b][i] = m[i]*a[b][i] + (1-m[i])*b_[b][i]; a or b_ may be NULL (= 0); mask stride mask_bs (0: shared).
 * Mask-mixed degradation and the replacement data-consistency step of predict_bwe_AR (blind_bwe_sampler.py:63-73,280-300) */
int babe_mask_blend(float* out, const float* mask, long mask_bs, const float* a, const float* b_, int B, long n,
                    void* stream);
/* part[b][blk] = sum of squares of g[b][blk-th slice] (double) */
int babe_sumsq_partial(const float* g, long g_bs, double* part, int nblk, int B, long n, void* stream);
/* STFT-domain guidance distances (get_rec_grads :105-115 -> utils/blind_bwe_utils.py:148-247; conf/tester/blind_bwe_2.yaml):
 * X = STFT(rec), R = STFT(y) as [B][frames][nbins] complex (babe_stft_fwd), w[nbins] = frequency weighting.  mode 0:
 * ||w (X - R)||_2, 1: ||w|X| - w|R|||_2, 2: ||log10(w|X| + 1e-8) - log10(w|R| + 1e-8)||_2.  babe_stft_dist_partial writes
 * partial sums of squares ([B][nblk] doubles), babe_stft_dist_grad G = dD/dX with ONE D over the batch (shared = 1, the
 * reference) or per item; d/d(rec) = STFT^T(G) = babe_ola(babe_spec_filter_istft(G, H = nfft * [1, 1/2, ..., 1/2, 1])).
 * babe_residual_seed_alt mode 3 multiplies such a ready gradient by the overlap-add normalisation. */
int babe_stft_dist_partial(const float* X, const float* R, const float* w, double* part, int nblk, int B, int nbins,
                           int frames, int mode, void* stream);
int babe_stft_dist_grad(const float* X, const float* R, const float* w, const double* part, int nblk, float* G, int B,
                        int nbins, int frames, int mode, int shared, void* stream);
/* Alternative guidance distances of get_rec_grads (testing/blind_bwe_sampler.py:99-103; conf/tester/blind_bwe_cossim.yaml):
 * seed = d(distance)/d(rec) [* post] from r = y - rec.  mode 1: smooth_l1_loss(y, rec, reduction='sum', beta):
 * -clamp(r / beta, -1, 1).  mode 2: clamp(1 - CosineSimilarity(rec, y), min 0) per batch item, with the three sums
 * (rec.rec, rec.y, y.y) from babe_cos_partial (part: [B][nblk][3] doubles).  mode 3: r is already the gradient: out = r * post. */
int babe_cos_partial(const float* r, long r_bs, const float* y, long y_bs, double* part, int nblk, int B, long n,
                     void* stream);
int babe_residual_seed_alt(const float* r, long r_bs, const float* y, long y_bs, const double* part, int nblk,
                           const float* post, float* out, long out_bs, int B, int L, int mode, float beta, void* stream);
/* mode 0 (blind_bwe_sampler.py:125-135,701): d = -t*((xden-xhat)/t^2 - s*g/t), s = xi/(||g||/sqrt(audio_len) + 1e-6)
 * mode 1 (edm_sampler.py:78-92):              d = -t*((xden-xhat)/t^2 - s*g),   s = xi/(||g||/sqrt(audio_len)*t + 1e-6)
 * shared_norm=1 uses the norm over the whole batch (reference semantics). */
int babe_score_direction(const float* xden, const float* xhat, const float* g, const double* part, int nblk,
                         float* d, float t, float xi, float audio_len, int shared_norm, int mode, int B, long n,
                         void* stream);

/* ---- one whole SCORE EVALUATION from plan handles (csrc/score_eval.hip, round 6): what BlindSampler.evaluate sequences from
 * Python (testing/blind_bwe_sampler.py:75-170, 503-595, 687-761; diff_params/edm.py:144-159), as one call for a non-Python host:
 * preconditioned denoiser (CQT.fwd -> UNet -> CQT.bwd) -> apply_hpf_DC -> STFT -> [filter fit] -> filter design -> residual
 * y - A(x_den) and the gradient of its norm through iSTFT o H o STFT, the high-pass and the network -> score direction.
 * Default configuration only (L2 guidance norm, STFT-domain low-pass, no observation noise / data-consistency / AR mask / FIR);
 * bit-identical to the Python sequencer on it (tests/test_gpu_eval_c.py).  No allocation, no synchronisation. */
typedef struct {
    const void* unet_plan; void* unet_state;      /* babe_unet_plan_create / babe_unet_state_create (one state per stream) */
    const void* cqt_plan;                         /* babe_cqt_plan_create for (fs, L) */
    int L;                                        /* exp.audio_len */
    /* noise embedding (RFF_MLP_Block cqtdiff+.py:184-211) and every FiLM Linear (:36-40) stacked into one matrix */
    const float* rff_freq; int rff_n;             /* embedding.RFF_freq [rff_n] */
    const float* emb_W[3]; const float* emb_b[3]; /* embedding.MLP.i weight [emb_dim[i+1]][emb_dim[i]], bias */
    int emb_dim[4];                               /* emb_dim[0] = 2 * rff_n */
    const float* film_W; const float* film_b; int film_J;   /* [film_J][emb_dim[3]]: affine / gate rows in the UNet plan's order */
    /* STFT-domain degradation model (utils/blind_bwe_utils.py): nfft, periodic-Hamming OLA envelope 1 / sum w^2
     * [nfft + hop (frames - 1)], exp(-2 pi i q / 4096) [2048] float2 */
    int nfft; float fs; const float* env_inv; const float* tw4096;
    int K;                                        /* break points of the piecewise filter (<= 8) */
    babe_fit_cfg fit;
    int blind;                                    /* 1: fit the filter parameters in this evaluation (predict_blind_bwe) */
    int shared;                                   /* 1: the reference's batch coupling (one filter, whole-batch norms); 0: per clip */
    int hpf;                                      /* tester.filter_out_cqt_DC_Nyq */
    float xi; int score_mode; float audio_len_norm;   /* babe_score_direction's xi, mode and audio_len */
} babe_eval_desc;
long babe_eval_workspace_bytes(const babe_eval_desc* desc, int B);
int babe_score_eval(const babe_eval_desc* desc, const float* x, float t, float cskip, float cout, float cin, float cnoise,
                    const float* y, const float* specY, float* params, float* d, float* x_den, int* n_iter, void* ws,
                    long ws_bytes, int B, void* stream);
int babe_fill(float* out, float v, int n, void* stream);            /* out[0..n) = v (the [B][1] cnoise input of the embedding) */

/* ---- Denoiser pre-pass: networks/denoiser.py:232-321 (MultiStage_denoise, inference only) and
 * testing/denoise_and_bwe_tester.py:146-165 (STFT 1024/256 -> network -> inverse STFT).  csrc/denoiser.hip --------- */
/* General small-kernel Conv2d on [B][C][H][W] tensors with contiguous rows (replaces nn.Conv2d with
 * padding_mode='reflect' denoiser.py:40-46, the 4x4 stride-2 down conv :358-363 and, as four 2x2 parity kernels, the
 * ConvTranspose2d :383-388):  v = conv(in)[co][oh][ow] + bias[co];  act=1: v = ELU(v);  res: v += res[...];
 * out[b][co][oh*out_hstep + out_h0][ow*out_wstep + out_w0] = v  where that index lies inside [out_H][out_W]
 * (res is indexed like out).  Input index = o*stride - pad + k, reflected (pad_mode 1) or zero (0) outside. */
typedef struct {
    const float* in; long in_bs, in_cs; int IH, IW;
    const float* bias;           /* [Cout] or NULL */
    float* out; long out_bs, out_cs; int out_H, out_W, out_hstep, out_h0, out_wstep, out_w0;
    const float* res; long res_bs, res_cs;
    int B, Cin, Cout, OH, OW, KH, KW, stride, pad_t, pad_l, pad_mode, act;
    int ksplit;                  /* > 1: split the input channels over ksplit workgroups per tile (small planes); the */
    float* ws;                   /* partial sums go through ws [ksplit][B][Cout][OH][OW] and a finishing pass */
} babe_dnconv_args;
int babe_dn_conv2d(const babe_dnconv_args* a, const float* w_packed, void* stream);
/* mode 0: w [Cout][Cin][KH][KW] -> [KH][KW][ceil32(Cin)][ceil64(Cout)].  mode 1 (KH=KW=2): parity (ph,pw) of a
 * ConvTranspose2d weight [Cin][Cout][4][4] with stride 2: out[2m+p] = sum_a in[m-a] w[p+2a]. */
int babe_dn_pack_weights(const float* w, float* dst, int Cout, int Cin, int KH, int KW, int mode, int ph, int pw,
                         void* stream);
long babe_dn_packed_size(int Cout, int Cin, int KH, int KW);
/* out[b][c][h][w] += low[b][c][(h+dh)>>1][(w+dw)>>1]: nn.Upsample(2,'nearest') + CropAdd/CropConcat, denoiser.py:400-407 */
int babe_dn_upsample_add(float* out, long out_bs, long out_cs, const float* low, long low_bs, long low_cs, int B, int C,
                         int H, int W, int LH, int LW, int dh, int dw, void* stream);
/* out = x1 * sigmoid(m) + feats (SAM, denoiser.py:126-130); x1, m contiguous [B][C][hw] */
int babe_dn_sam_gate(const float* x1, const float* m, const float* feats, long f_bs, long f_cs, float* out, long out_bs,
                     long out_cs, int B, int C, long hw, void* stream);
/* out[B][2+nemb][T][F] = cat(X[B][2][T][F], femb[F][nemb] broadcast over b, t)  (AddFreqEncoding :159-169) */
int babe_dn_fill_input(const float* X, const float* femb, float* out, int B, int T, int F, int nemb, void* stream);
/* torch.stft(x, nfft, hop, hamming_window(nfft), center=False) as X[B][2][frames][nfft/2+1]; frames = 1+(L-nfft)/hop */
int babe_dn_stft(const float* x, long x_bs, int L, float* X, int B, int nfft, int hop, int frames, const float* tw4096,
                 void* stream);
/* torch.istft(P, nfft, hop, hamming_window(nfft), center=False)[..., :Lout]; frames_ws: [B][frames][nfft] scratch */
int babe_dn_istft(const float* P, float* frames_ws, float* y, long y_bs, int Lout, int B, int nfft, int hop, int frames,
                  const float* tw4096, void* stream);


#ifdef __cplusplus
}
#endif
#endif
