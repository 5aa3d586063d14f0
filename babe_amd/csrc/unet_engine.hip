// CQTDiff+ UNet body, forward and hand-wired input-VJP, sequenced INSIDE the library from a plan handle: one C call per direction
// (babe_unet_fwd / babe_unet_vjp) instead of ~1100 op-level calls from the host language.  Same wiring, same kernels, same order
// as the Python engine (babe_amd/networks/unet_engine.py; /root/reference/networks/cqtdiff+.py:746-839 forward, ResnetBlock :452-493;
// the VJP replaces torch.autograd through the network, testing/blind_bwe_sampler.py:120) - the two are bit-identical
// (tests/test_gpu_unet_c.py) - so a non-Python host gets the whole network through the C-ABI, and the Python host sheds its
// per-evaluation enqueue time.  fp32 arithmetic only (precision 'bf16' / 'bf16x3' stay on the Python engine).
//
// Memory: the caller owns everything.  Activations, saved tensors and scratch are carved from ONE workspace by a bump allocator
// (256-byte granules) that is reset by babe_unet_fwd; what the VJP needs (one tensor per dilation layer + per-layer statistics)
// stays where the forward pass put it, the VJP's own buffers are carved after it.  babe_unet_workspace_bytes runs both passes
// dry.  A state (babe_unet_state_*) holds one evaluation's bookkeeping: two clips on two streams = two states + two workspaces
// over one plan.  Host code only: every tensor operation is one of this library's extern "C" functions.
#include "common.h"
#include "../../include/babe_hip.h"
#include <cmath>
#include <cstdlib>
#include <cstring>

/* out = alpha*conv(in[,in2]; W)*oscale + rbeta*res with the kernel chosen by the library (what babe_amd/ops.py::conv2d did in
 * Python): few-output-channel vector kernel, nested Winograd F(2,5)xF(4,3) where its tiles are full, F(4,3), F(2,3), direct /
 * pipelined (1,1).  The caller fills every field of *a except w_packed, Cin, Cout, KH, KW (taken from pc and transpose).  A requested
 * fused reduction (a->stat_mode) is formed only by the F(4,5) kernels: on return a->stat_mode is 0 if it was NOT produced. */
extern "C" int babe_conv2d_auto(babe_conv_args* a, const babe_packed_conv* pc, int transpose, void* stream) {
    BABE_CHECK_ARG(a && pc, "conv2d_auto: null arguments");
    a->Cin = transpose ? pc->Cout : pc->Cin;
    a->Cout = transpose ? pc->Cin : pc->Cout;
    a->KH = pc->KH;
    a->KW = pc->KW;
    const void* wq = transpose ? pc->bwd : pc->fwd;
    BABE_CHECK_ARG(wq, "conv2d_auto: weights not packed for this direction");
    if (!a->in2) a->cin_split = a->Cin;
    if (pc->splits) {
        a->w_packed = nullptr;
        a->stat_mode = 0;
        return babe_conv2d_bf16(a, wq, pc->splits, stream);
    }
    a->w_packed = (const float*)wq;
    const float* w45 = transpose ? pc->bwd_wino45 : pc->fwd_wino45;
    if (pc->w_raw && a->Cout <= 4 && babe_conv2d_fewco_supported(a)) {
        a->stat_mode = 0;
        return babe_conv2d_fewco(a, pc->w_raw, transpose, stream);
    }
    const float* w85 = transpose ? pc->bwd_wino85 : pc->fwd_wino85;
    if (w85 && !a->in2 && babe_conv2d_wino85_preferred(a)) return babe_conv2d_wino85(a, w85, stream);
    a->stat_mode = 0;                  // only the F(4,5) kernels form the fused reduction: tell the caller it was not produced
    if (w45 && babe_conv2d_wino45_preferred(a)) return babe_conv2d_wino45(a, w45, stream);
    if (pc->fwd_wino4 && babe_conv2d_wino4_supported(a)) return babe_conv2d_wino4(a, transpose ? pc->bwd_wino4 : pc->fwd_wino4, stream);
    if (pc->fwd_wino && babe_conv2d_wino_supported(a)) return babe_conv2d_wino(a, transpose ? pc->bwd_wino : pc->fwd_wino, stream);
    return babe_conv2d_nt(a, pc->nt, stream);
}

namespace {

constexpr float RS2 = 0.70710678118654752440f;
constexpr int G_GROUPS = 8;
constexpr float GN_EPS = 1e-7f;

struct View {                     // [B][C][F][T] fp32, rows contiguous (a frequency sub-range is a view)
    float* p = nullptr;
    long bs = 0, cs = 0;
    int C = 0, F = 0, T = 0;
    bool dense() const { return cs == (long)F * T && bs == (long)C * F * T; }
};
View sub_f(const View& v, int f0, int nf) { View r = v; r.p = v.p + (long)f0 * v.T; r.F = nf; return r; }
View sub_c(const View& v, int c0, int nc) { View r = v; r.p = v.p + (long)c0 * v.cs; r.C = nc; return r; }

struct Saved { View z; float* stats; float* scale; float* gate; };

}  // namespace

struct babe_unet_state_s {
    int B = 0, Ts[8] = {};
    float* ws = nullptr;
    size_t cap = 0, off = 0, high = 0;
    bool dry = false;
    float* scr_a = nullptr;
    size_t scr_a_n = 0;
    Saved saved[6][8][8];          // [kind][index][layer]: kinds 0 init, 1 main, 2 up_out, 3 up_blk, 4 mid_blk, 5 mid_out
    View hs[8];
    const float* film = nullptr;
    long film_bs = 0;
    bool have_fwd = false;
};

namespace {

struct Ctx {
    const babe_unet_plan_desc* P;
    babe_unet_state_s* S;
    hipStream_t st;
    int err = 0;
    int B() const { return S->B; }
    bool dry() const { return S->dry; }
    float* alloc(size_t nfloat) {
        // (+ an odd number of 256-byte granules: tensor sizes here are large powers of two times small integers, and operands
        // that start at multiples of the same power of two walk the HBM channels in lockstep)
        // (BABE_UNET_SKEW overrides the 4352 bytes; rounded to a multiple of 256 so that every carved buffer keeps the alignment the
        // float4 / LDS-DMA kernels need, negative values read as 0)
        static const size_t skew = [] {
            const char* e = getenv("BABE_UNET_SKEW");
            long v = e ? atol(e) : 4352L;
            if (v < 0) v = 0;
            return (size_t)((v + 255) / 256 * 256);
        }();
        const size_t bytes = ((nfloat * 4 + 255) & ~(size_t)255) + skew;
        float* p = S->ws ? reinterpret_cast<float*>(reinterpret_cast<char*>(S->ws) + S->off) : nullptr;
        S->off += bytes;
        if (S->off > S->high) S->high = S->off;
        if (!S->dry && S->off > S->cap && !err) {
            babe_set_error("unet: workspace too small (%zu bytes needed so far, %zu given; ask babe_unet_workspace_bytes)", S->off, S->cap);
            err = BABE_ERR_ARG;
        }
        return p;
    }
    View buf(int C, int F, int T) {
        View v;
        v.p = alloc((size_t)B() * C * F * T);
        v.C = C; v.F = F; v.T = T; v.cs = (long)F * T; v.bs = (long)C * F * T;
        return v;
    }
    // stack discipline for the VJP's temporaries: what a level needs only while it runs is carved after a mark and given back, so
    // the next level reuses the same (cache-warm) bytes - what the caching allocator did for the Python-sequenced engine
    size_t mark() const { return S->off; }
    void release(size_t m) { S->off = m; }
    float* scratch_a(size_t n) {
        if (n > S->scr_a_n) { S->scr_a = alloc(n); S->scr_a_n = n; }
        return S->scr_a;
    }
    void ck(int e) { if (e && !err) err = e; }

    // ---- ops (each = one extern "C" call of this library; nothing runs in a dry pass or after an error)
    // stat_cg / stat_part: also form the GroupNorm sums of the output in the conv's epilogue (babe_conv_args::stat_mode 1); returns
    // true if the kernel that ran the conv produced them (only the F(4,5) kernels do)
    bool conv(const View& x, const babe_packed_conv& pc, const View& out, int dil, bool transpose, const View* x2, const View* res,
              const float* in_scale, const float* oscale, float alpha, float rbeta, int stat_cg = 0, double* stat_part = nullptr) {
        if (dry() || err) return false;
        babe_conv_args a;
        memset(&a, 0, sizeof a);
        a.in = x.p; a.in_bs = x.bs; a.in_cs = x.cs;
        if (x2) { a.in2 = x2->p; a.in2_bs = x2->bs; a.in2_cs = x2->cs; a.cin_split = x.C; }
        a.out = out.p; a.out_bs = out.bs; a.out_cs = out.cs;
        if (res) { a.res = res->p; a.res_bs = res->bs; a.res_cs = res->cs; }
        a.in_scale = in_scale; a.oscale = oscale; a.alpha = alpha; a.rbeta = rbeta;
        a.B = B(); a.F = x.F; a.T = x.T; a.dil = dil;
        if (stat_part) { a.stat_mode = 1; a.stat_cg = stat_cg; a.stat_part = stat_part; }
        ck(babe_conv2d_auto(&a, &pc, transpose ? 1 : 0, st));
        return a.stat_mode == 1;
    }
    // slots per group of the fused reduction for a [C][F][T] output at this dilation (babe_conv2d_wino85_stat_slots)
    static int stat_slots(int cg, int F, int T, int dil) {
        babe_conv_args a;
        memset(&a, 0, sizeof a);
        a.stat_cg = cg; a.F = F; a.T = T; a.dil = dil;
        return babe_conv2d_wino85_stat_slots(&a);
    }
    void axpby(const View& x, const View& out, float alpha = 1.f, float beta = 0.f) {
        if (dry() || err) return;
        ck(babe_axpby4d(x.p, x.bs, x.cs, out.p, out.bs, out.cs, B(), x.C, x.F, x.T, alpha, beta, st));
    }
    static bool al16(const View& v) { return ((uintptr_t)v.p & 15) == 0 && v.bs % 4 == 0 && v.cs % 4 == 0; }
    void axpby2(const View& x, const View& y, const View& out, float alpha, float beta) {
        if (dry() || err) return;
        if (((long)x.F * x.T) % 4 == 0 && al16(x) && al16(y) && al16(out))
            ck(babe_axpby2_4d(x.p, x.bs, x.cs, y.p, y.bs, y.cs, out.p, out.bs, out.cs, B(), x.C, x.F, x.T, alpha, beta, st));
        else {
            axpby(x, out, alpha);
            axpby(y, out, beta, 1.f);
        }
    }
    // mode 0 down, 1 up, 2 down^T, 3 up^T; T argument of the C function = the forward op's input length
    void resample(const View& x, const View& out, int mode, float alpha = 1.f, float beta = 0.f, const View* res = nullptr) {
        if (dry() || err) return;
        const int T = mode == 0 || mode == 1 ? x.T : (mode == 2 ? x.T * 2 : x.T / 2);
        if (res) {
            if (al16(*res) && al16(out)) {
                ck(babe_resample_res(x.p, x.bs, x.cs, res->p, res->bs, res->cs, out.p, out.bs, out.cs, B(), x.C, x.F, T, mode, alpha, beta, st));
                return;
            }
            axpby(*res, out);
        }
        ck(babe_resample(x.p, x.bs, x.cs, out.p, out.bs, out.cs, B(), x.C, x.F, T, mode, alpha, beta, st));
    }
    static int splits(long n) { long s = n / 16384; return (int)(s < 1 ? 1 : (s > 64 ? 64 : s)); }
    const float* film_at(int off) const { return S->film + off; }

    // ---- ResnetBlock (unet_engine.py block_fwd / block_vjp)
    View block_fwd(const babe_unet_block& blk, Saved* saved, const View& x, const View& out, const View* x2) {
        const int N = blk.N, Fq = x.F, T = x.T;
        View z;
        if (blk.proj_in.Cout) {
            z = buf(N, Fq, T);
            conv(x, blk.proj_in, z, 1, false, x2, nullptr, nullptr, nullptr, 1.f, 0.f);
        } else if (x.dense()) {
            z = x;
        } else {
            z = buf(N, Fq, T);
            axpby(x, z);
        }
        double* zpart = nullptr;                 // GroupNorm sums of z formed by the conv that wrote it (zS slots per group; 0: none)
        int zS = 0;
        for (int d = 0; d < blk.nd; ++d) {
            // gate = film[:, goff : goff + N] made contiguous ([B][N]: the conv's oscale / the VJP's in_scale)
            float* gate;
            if (B() == 1) gate = const_cast<float*>(film_at(blk.film_gate[d]));       // one row IS contiguous (the caller keeps film alive until the VJP)
            else {
                gate = alloc((size_t)B() * N);
                if (!dry() && !err) ck(babe_axpby4d(film_at(blk.film_gate[d]), S->film_bs, 0, gate, N, 0, B(), 1, 1, N, 1.f, 0.f, st));
            }
            View znew = buf(N, Fq, T);
            float* a = scratch_a((size_t)B() * N * Fq * T);
            const long n = (long)(N / G_GROUPS) * Fq * T;
            const int Sp = splits(n);
            double* part = reinterpret_cast<double*>(alloc((size_t)B() * G_GROUPS * Sp * 2 * 2));
            float* stats = alloc((size_t)B() * G_GROUPS * 3);
            float* scale = alloc((size_t)B() * N);
            if (!dry() && !err) {
                if (!zS) ck(babe_gn_partial(z.p, part, B(), G_GROUPS, n, Sp, st));
                ck(babe_scale_gelu_fin(z.p, zS ? zpart : part, blk.gamma[d], film_at(blk.film_aff[d]), S->film_bs, stats, scale, a, B(), N,
                                       G_GROUPS, (long)Fq * T, zS ? zS : Sp, GN_EPS, st));
            }
            View av = z; av.p = a; av.cs = (long)Fq * T; av.bs = (long)N * Fq * T;
            // the next layer's GroupNorm reads znew: its sums come out of this conv's epilogue when the F(4,5) kernel runs it
            // (unet_engine.py: ops.conv2d(..., fwd_stat=))
            const int dil = blk.k53 ? (1 << d) : 1, cg = N / G_GROUPS;
            double* npart = nullptr;
            int nS = 0;
            if (fuse_gn_fwd() && d + 1 < blk.nd && blk.k53 && cg % 4 == 0) {
                nS = stat_slots(cg, Fq, T, dil);
                npart = reinterpret_cast<double*>(alloc((size_t)B() * G_GROUPS * nS * 2 * 2));
            }
            const bool made = conv(av, blk.H[d], znew, dil, false, nullptr, &z, nullptr, gate, RS2, RS2, cg, npart);
            zS = made ? nS : 0;
            zpart = npart;
            saved[d] = Saved{z, stats, scale, gate};
            z = znew;
        }
        if (blk.proj_out.Cout) {
            View zo = buf(blk.proj_out.Cout, Fq, T);
            conv(z, blk.proj_out, zo, 1, false, nullptr, nullptr, nullptr, nullptr, 1.f, 0.f);
            z = zo;
        }
        if (blk.res_conv.Cout) conv(x, blk.res_conv, out, 1, false, x2, &z, nullptr, nullptr, RS2, RS2);
        else axpby2(z, x, out, RS2, RS2);
        return out;
    }
    // the transposed conv of dilation layer d, with the GroupNorm-VJP partial sums formed in its epilogue when it takes the F(4,5)
    // kernel (unet_engine.py: ops.conv2d(..., vjp_stat=)): returns the slot count S (0: not fused) and the buffer in *part
    int conv_vjp(const View& src, const babe_unet_block& blk, int d, const View& da, const Saved& sv, double** part) {
        const View& z = sv.z;
        const int dil = blk.k53 ? (1 << d) : 1, cg = z.C / G_GROUPS;
        const int Sf = cg % 4 == 0 ? stat_slots(cg, z.F, z.T, dil) : 0;
        const long n = (long)cg * z.F * z.T;
        const int Sp = splits(n);
        *part = reinterpret_cast<double*>(alloc((size_t)B() * G_GROUPS * (Sf > Sp ? Sf : Sp) * 2));     // (floats: 2 per double)
        if (dry() || err) return 0;
        babe_conv_args a;
        memset(&a, 0, sizeof a);
        a.in = src.p; a.in_bs = src.bs; a.in_cs = src.cs;
        a.out = da.p; a.out_bs = da.bs; a.out_cs = da.cs;
        a.in_scale = sv.gate; a.alpha = RS2; a.rbeta = 0.f;
        a.B = B(); a.F = src.F; a.T = src.T; a.dil = dil;
        if (fuse_gn() && blk.k53 && cg % 4 == 0 && z.dense() && da.dense()) {
            a.stat_mode = 2; a.stat_cg = cg; a.stat_x = z.p; a.stat_scale = sv.scale; a.stat_part = *part;
        }
        ck(babe_conv2d_auto(&a, &blk.H[d], 1, st));
        return a.stat_mode == 2 ? Sf : 0;
    }
    static bool fuse_gn_fwd() {
        static const bool on = [] { const char* e = getenv("BABE_FUSE_GN_FWD"); return !(e && atoi(e) == 0); }();   // (on by default: ops.py FUSE_GN_FWD)
        return on;
    }
    static bool fuse_gn() {
        static const bool on = [] { const char* e = getenv("BABE_FUSE_GN"); return e && atoi(e) != 0; }();   // (off by default: ops.py FUSE_GN)
        return on;
    }
    void gn_bwd(const Saved& sv, const View& da, const View& gy, const View& gx, float rbeta, const View* acc, double* part, int Sf) {
        const View& z = sv.z;
        const long n = (long)(z.C / G_GROUPS) * z.F * z.T;
        const int Sp = Sf > 0 ? Sf : splits(n);
        if (dry() || err) return;
        if (Sf <= 0) ck(babe_gn_bwd_partial(z.p, da.p, sv.scale, part, B(), z.C, G_GROUPS, (long)z.F * z.T, Sp, st));
        if (acc)          // the block's tail merged in: gx = RS2*acc + RS2*(this layer's gradient)
            ck(babe_gn_bwd_apply_merge(z.p, da.p, gy.p, sv.scale, sv.stats, part, gx.p, rbeta, B(), z.C, G_GROUPS, (long)z.F * z.T, Sp, GN_EPS, st,
                                       acc->p, RS2, RS2));
        else
            ck(babe_gn_bwd_apply(z.p, da.p, gy.p, sv.scale, sv.stats, part, gx.p, rbeta, B(), z.C, G_GROUPS, (long)z.F * z.T, Sp, GN_EPS, st));
    }
    // g_in (+)= VJP of the block w.r.t. its (concatenated) input; consume: g_out is a dense buffer that may be overwritten
    View block_vjp(const babe_unet_block& blk, const Saved* saved, const View& g_out, const View& g_in, bool accumulate, bool consume) {
        const int N = blk.N, Fq = g_out.F, T = g_out.T;
        const float beta = accumulate ? 1.f : 0.f;
        auto da_view = [&]() { View v; v.p = scratch_a((size_t)B() * N * Fq * T); v.C = N; v.F = Fq; v.T = T; v.cs = (long)Fq * T; v.bs = (long)N * Fq * T; return v; };
        if (!blk.res_conv.Cout && !blk.proj_out.Cout && !blk.proj_in.Cout && !accumulate && blk.nd > 0 && g_out.dense()) {
            View gz;
            if (blk.nd > 1) gz = buf(N, Fq, T);
            View da = da_view();
            View src = g_out;
            const bool merged = g_in.dense() && al16(g_in) && al16(g_out);
            for (int d = blk.nd - 1; d >= 0; --d) {
                double* part;
                const int Sf = conv_vjp(src, blk, d, da, saved[d], &part);
                if (d == 0 && merged) {
                    gn_bwd(saved[d], da, src, g_in, RS2, &g_out, part, Sf);
                } else {
                    if (!gz.p && !gz.C) gz = buf(N, Fq, T);
                    gn_bwd(saved[d], da, src, gz, RS2, nullptr, part, Sf);
                    src = gz;
                }
            }
            if (!merged) axpby2(g_out, gz, g_in, RS2, RS2);
            return g_in;
        }
        if (blk.res_conv.Cout) conv(g_out, blk.res_conv, g_in, 1, true, nullptr, accumulate ? &g_in : nullptr, nullptr, nullptr, RS2, beta);
        else axpby(g_out, g_in, RS2, beta);
        float c = 1.f;
        View gz;
        if (blk.proj_out.Cout) {
            gz = buf(N, Fq, T);
            conv(g_out, blk.proj_out, gz, 1, true, nullptr, nullptr, nullptr, nullptr, RS2, 0.f);
        } else if (consume && g_out.dense()) {
            gz = g_out;
            c = RS2;
        } else {
            gz = buf(N, Fq, T);
            axpby(g_out, gz, RS2);
        }
        View da = da_view();
        for (int d = blk.nd - 1; d >= 0; --d) {
            double* part;
            const int Sf = conv_vjp(gz, blk, d, da, saved[d], &part);
            gn_bwd(saved[d], da, gz, gz, RS2, nullptr, part, Sf);
        }
        if (blk.proj_in.Cout) conv(gz, blk.proj_in, g_in, 1, true, nullptr, &g_in, nullptr, nullptr, c, 1.f);
        else axpby(gz, g_in, c, 1.f);
        return g_in;
    }

    // ---- the network (unet_engine.py forward / vjp).  C_in[j]: [B][2][bpo][T_j], index 0 = lowest octave
    void forward(const float* const* C_in, float** outs) {
        const int n = P->nocts, bpo = P->bpo;
        const int* Ns = P->Ns;
        const int* Ts = S->Ts;                     // level i time length (level 0 = the highest octave's rate... = C_in[n-1])
        auto cview = [&](int lvl) { View v; v.p = const_cast<float*>(C_in[n - 1 - lvl]); v.C = 2; v.F = bpo; v.T = Ts[lvl]; v.cs = (long)bpo * Ts[lvl]; v.bs = 2 * v.cs; return v; };
        View XC = buf(Ns[0], bpo, Ts[0]);
        View pyr_prev, X;
        for (int i = 0; i < n; ++i) {
            const View C = cview(i);
            const int Fi = bpo * (i + 1);
            block_fwd(P->init_blk[i], S->saved[0][i], C, sub_f(XC, 0, bpo), nullptr);
            View pyr;
            if (i == 0) {
                pyr = buf(2, bpo, Ts[0] / 2);
                resample(C, pyr, 0);
            } else if (i < n - 1) {
                pyr = buf(2, Fi, Ts[i] / 2);
                resample(C, sub_f(pyr, 0, bpo), 0);
                resample(pyr_prev, sub_f(pyr, bpo, Fi - bpo), 0);
            } else {
                pyr = buf(2, Fi, Ts[i]);
                axpby(C, sub_f(pyr, 0, bpo));
                axpby(pyr_prev, sub_f(pyr, bpo, Fi - bpo));
            }
            pyr_prev = pyr;
            View H = buf(Ns[i], Fi, Ts[i]);
            block_fwd(P->main_blk[i], S->saved[1][i], XC, H, nullptr);
            S->hs[i] = H;
            if (i < n - 1) {
                View XCn = buf(Ns[i], Fi + bpo, Ts[i + 1]);
                View sub = sub_f(XCn, bpo, Fi);
                resample(H, sub, 0);
                conv(pyr, P->pyr_conv[i], sub, 1, false, nullptr, &sub, nullptr, nullptr, RS2, RS2);
                XC = XCn;
            } else {
                X = buf(Ns[i], Fi, Ts[i]);
                conv(pyr, P->pyr_conv[i], X, 1, false, nullptr, &H, nullptr, nullptr, RS2, RS2);
            }
        }
        {
            View Xm = buf(X.C, X.F, X.T);
            block_fwd(P->mid_blk, S->saved[4][0], X, Xm, nullptr);
            X = Xm;
        }
        View Xout = buf(2, bpo * n, Ts[n - 1]);
        block_fwd(P->mid_out, S->saved[5][0], X, Xout, nullptr);
        for (int i = 0; i < n; ++i) {
            const int j = n - 1 - i;
            const int Fj = bpo * (j + 1);
            const int Nout = Ns[j > 0 ? j - 1 : 0];
            View R = buf(Nout, Fj, Ts[j]);
            block_fwd(P->up_blk[i], S->saved[3][i], X, R, &S->hs[j]);
            View O = buf(2, Fj, Ts[j]);
            block_fwd(P->up_out[i], S->saved[2][i], R, O, nullptr);
            axpby(O, Xout, RS2, RS2);
            View o; o.p = outs[i]; o.C = 2; o.F = bpo; o.T = Ts[j]; o.cs = (long)bpo * Ts[j]; o.bs = 2 * o.cs;
            axpby(sub_f(Xout, 0, bpo), o);
            if (j > 0) {
                View Xn = buf(Nout, Fj - bpo, Ts[j - 1]);
                resample(sub_f(R, bpo, Fj - bpo), Xn, 1);
                View Xo = buf(2, Fj - bpo, Ts[j - 1]);
                resample(sub_f(Xout, bpo, Fj - bpo), Xo, 1);
                X = Xn;
                Xout = Xo;
            }
        }
    }

    void vjp(const float* const* gouts, float** gC) {
        const int n = P->nocts, bpo = P->bpo;
        const int* Ns = P->Ns;
        const int* Ts = S->Ts;
        View gH[8];
        View gX_prev, gXO_prev;
        for (int j = 0; j < n; ++j) {
            const int i = n - 1 - j;
            const int Fj = bpo * (j + 1);
            const int Nout = Ns[j > 0 ? j - 1 : 0];
            // (what outlives the level first: the next level reads gXOn / gcat[:, :N], the encoder VJP gcat[:, N:])
            View gXOn = buf(2, Fj, Ts[j]);
            View gcat = buf(2 * Ns[j], Fj, Ts[j]);
            const size_t lvl = mark();
            View gXOp = buf(2, Fj, Ts[j]);
            View go; go.p = const_cast<float*>(gouts[i]); go.C = 2; go.F = bpo; go.T = Ts[j]; go.cs = (long)bpo * Ts[j]; go.bs = 2 * go.cs;
            axpby(go, sub_f(gXOp, 0, bpo));
            View gR = buf(Nout, Fj, Ts[j]);
            bool accumulate = false;
            if (j > 0) {
                resample(gXO_prev, sub_f(gXOp, bpo, Fj - bpo), 3);
                if (!dry() && !err) ck(babe_fill4d(gR.p, gR.bs, gR.cs, B(), Nout, bpo, Ts[j], 0.f, st));      // gR[:, :, :bpo, :] = 0
                resample(gX_prev, sub_f(gR, bpo, Fj - bpo), 3);
                accumulate = true;
            }
            View gO = buf(2, Fj, Ts[j]);
            axpby(gXOp, gO, RS2);
            block_vjp(P->up_out[i], S->saved[2][i], gO, gR, accumulate, true);
            axpby(gXOp, gXOn, RS2);
            gXO_prev = gXOn;
            block_vjp(P->up_blk[i], S->saved[3][i], gR, gcat, false, true);
            gX_prev = sub_c(gcat, 0, Ns[j]);
            gH[j] = sub_c(gcat, Ns[j], Ns[j]);
            release(lvl);
        }
        View gM = buf(Ns[n - 1], bpo * n, Ts[n - 1]);
        axpby(gX_prev, gM);
        block_vjp(P->mid_out, S->saved[5][0], gXO_prev, gM, true, true);
        View gXm = buf(gM.C, gM.F, gM.T);
        block_vjp(P->mid_blk, S->saved[4][0], gM, gXm, false, true);
        View gpyr_next, gP;
        bool have_next = false;
        for (int i = n - 1; i >= 0; --i) {
            const int Fi = bpo * (i + 1);
            const int Nin = Ns[i > 0 ? i - 1 : 0];
            // (read by the next level: gXC[:, :, bpo:] = g_P, the pyramid gradient; the level's own temporaries come after the mark)
            View gXC = buf(Nin, Fi, Ts[i]);
            View gpyr = i == n - 1 ? buf(2, Fi, Ts[i]) : buf(2, Fi, Ts[i] / 2);
            View nx;
            if (i > 0 && i < n - 1) nx = buf(2, Fi - bpo, Ts[i]);
            const size_t lvl = mark();
            View gHi = buf(Ns[i], Fi, Ts[i]);
            if (i == n - 1) {
                axpby2(gH[i], gXm, gHi, 1.f, RS2);
                conv(gXm, P->pyr_conv[i], gpyr, 1, true, nullptr, nullptr, nullptr, nullptr, RS2, 0.f);
            } else {
                resample(gP, gHi, 2, RS2, 1.f, &gH[i]);
                conv(gP, P->pyr_conv[i], gpyr, 1, true, nullptr, have_next ? &gpyr_next : nullptr, nullptr, nullptr, RS2, have_next ? 1.f : 0.f);
            }
            block_vjp(P->main_blk[i], S->saved[1][i], gHi, gXC, false, true);
            View gCi; gCi.p = gC[n - 1 - i]; gCi.C = 2; gCi.F = bpo; gCi.T = Ts[i]; gCi.cs = (long)bpo * Ts[i]; gCi.bs = 2 * gCi.cs;
            block_vjp(P->init_blk[i], S->saved[0][i], sub_f(gXC, 0, bpo), gCi, false, false);
            release(lvl);
            if (i > 0) gP = sub_f(gXC, bpo, Fi - bpo);
            if (i == n - 1) {
                axpby(sub_f(gpyr, 0, bpo), gCi, 1.f, 1.f);
                gpyr_next = sub_f(gpyr, bpo, Fi - bpo);
                have_next = true;
            } else if (i > 0) {
                resample(sub_f(gpyr, 0, bpo), gCi, 2, 1.f, 1.f);
                resample(sub_f(gpyr, bpo, Fi - bpo), nx, 2);
                gpyr_next = nx;
                have_next = true;
            } else {
                resample(gpyr, gCi, 2, 1.f, 1.f);
            }
        }
    }
};

// A descriptor arrives from ANY C-ABI host: everything the sequencer indexes with (saved[kind][i][8], H[8], gamma[8], the FiLM
// offsets) and every pointer it dereferences unconditionally is checked here, for every block of the plan.
bool block_ok(const babe_unet_block& b, bool required) {
    if (b.nd == 0 && !required) return true;
    if (b.nd < 0 || b.nd > 8) return false;
    if (required && b.nd < 1) return false;
    if (b.N < G_GROUPS || b.N % G_GROUPS) return false;
    for (int d = 0; d < b.nd; ++d) {
        if (!b.gamma[d] || b.film_aff[d] < 0 || b.film_gate[d] < 0) return false;
        const babe_packed_conv& h = b.H[d];
        if (h.Cout != b.N || h.Cin != b.N || h.KH < 1 || h.KW < 1) return false;
        if (!h.fwd && !h.w_raw && !h.fwd_wino45 && !h.fwd_wino4 && !h.fwd_wino) return false;      // no packed image at all
    }
    for (const babe_packed_conv* pc : {&b.proj_in, &b.res_conv, &b.proj_out})
        if (pc->Cout != 0 && (pc->Cout < 0 || pc->Cin < 1 || (!pc->fwd && !pc->w_raw))) return false;
    return true;
}
bool desc_ok(const babe_unet_plan_desc* d) {
    if (!d || d->nocts < 1 || d->nocts > 8 || d->bpo < 1) return false;
    for (int i = 0; i < d->nocts; ++i) {
        if (d->Ns[i] < G_GROUPS || d->Ns[i] % G_GROUPS) return false;
        if (!block_ok(d->init_blk[i], true) || !block_ok(d->main_blk[i], true) || !block_ok(d->up_blk[i], true) ||
            !block_ok(d->up_out[i], true))
            return false;
    }
    return block_ok(d->mid_blk, true) && block_ok(d->mid_out, true);
}

}  // namespace

extern "C" void* babe_unet_plan_create(const babe_unet_plan_desc* desc) {
    if (!desc_ok(desc)) {
        babe_set_error("unet_plan_create: bad descriptor (1..8 octaves; every block: 1..8 dilation layers, channels a positive "
                       "multiple of 8, gamma / FiLM offsets / packed weights present)");
        return nullptr;
    }
    auto* p = static_cast<babe_unet_plan_desc*>(malloc(sizeof(babe_unet_plan_desc)));
    if (p) memcpy(p, desc, sizeof *p);
    return p;
}
extern "C" void babe_unet_plan_destroy(void* plan) { free(plan); }
extern "C" void* babe_unet_state_create(void) { return new babe_unet_state_s(); }
extern "C" void babe_unet_state_destroy(void* st) { delete static_cast<babe_unet_state_s*>(st); }

static int unet_set_shape(babe_unet_state_s* S, const babe_unet_plan_desc* P, int B, const int* T_oct) {
    BABE_CHECK_ARG(B > 0 && T_oct, "unet: bad batch / octave lengths");
    S->B = B;
    for (int i = 0; i < P->nocts; ++i) {
        S->Ts[i] = T_oct[P->nocts - 1 - i];                 // level i = octave n-1-i (index 0 of T_oct = lowest octave)
        BABE_CHECK_ARG(S->Ts[i] >= 8 && S->Ts[i] % 2 == 0, "unet: octave length %d unsupported (even, >= 8: the resamplers)", S->Ts[i]);
    }
    return BABE_OK;
}

/* bytes of workspace one evaluation (forward + VJP) needs for batch B and octave lengths T_oct[nocts] (index 0 = lowest octave) */
extern "C" long babe_unet_workspace_bytes(const void* plan, int B, const int* T_oct) {
    if (!plan) return -1;
    babe_unet_state_s S;
    const auto* P = static_cast<const babe_unet_plan_desc*>(plan);
    if (unet_set_shape(&S, P, B, T_oct)) return -1;
    S.dry = true;
    Ctx c{P, &S, nullptr};
    float* dummy[8] = {};
    const float* cdummy[8] = {};
    c.forward(cdummy, dummy);
    c.vjp(cdummy, dummy);
    return (long)S.high + 256;
}

/* outs[i] / C_in[j]: caller-owned [B][2][bpo][T] tensors (index 0 = lowest octave); film: [B][J] FiLM vector of every layer
 * (babe_linear of the embedding), row stride film_bs.  The state keeps what babe_unet_vjp needs inside `workspace`, which must
 * stay untouched until the VJP (or the next forward) has run. */
extern "C" int babe_unet_fwd(const void* plan, void* state, const float* const* C_in, const float* film, long film_bs, int B,
                             const int* T_oct, void* workspace, long workspace_bytes, float* const* outs, void* stream) {
    BABE_CHECK_ARG(plan && state && C_in && film && outs && workspace, "unet_fwd: null arguments");
    BABE_CHECK_ARG(((uintptr_t)workspace & 255) == 0, "unet_fwd: workspace must be 256-byte aligned");
    const auto* P = static_cast<const babe_unet_plan_desc*>(plan);
    auto* S = static_cast<babe_unet_state_s*>(state);
    if (int e = unet_set_shape(S, P, B, T_oct)) return e;
    S->ws = static_cast<float*>(workspace); S->cap = (size_t)workspace_bytes; S->off = 0; S->high = 0; S->dry = false;
    S->scr_a = nullptr; S->scr_a_n = 0; S->film = film; S->film_bs = film_bs; S->have_fwd = false;
    Ctx c{P, S, (hipStream_t)stream};
    c.forward(C_in, const_cast<float**>(outs));
    if (c.err) return c.err;
    S->have_fwd = true;
    return BABE_OK;
}

/* gC[j] (+)= gradient w.r.t. C_in[j] given gouts[i] = gradient w.r.t. outs[i]; overwrites gC.  One VJP per forward. */
extern "C" int babe_unet_vjp(const void* plan, void* state, const float* const* gouts, float* const* gC, void* stream) {
    BABE_CHECK_ARG(plan && state && gouts && gC, "unet_vjp: null arguments");
    const auto* P = static_cast<const babe_unet_plan_desc*>(plan);
    auto* S = static_cast<babe_unet_state_s*>(state);
    BABE_CHECK_ARG(S->have_fwd, "unet_vjp: no forward pass in this state");
    Ctx c{P, S, (hipStream_t)stream};
    c.vjp(gouts, const_cast<float**>(gC));
    S->have_fwd = false;
    return c.err;
}
