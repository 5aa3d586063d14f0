// One SCORE EVALUATION of the blind bandwidth-extension sampler from plan handles - the sequence BlindSampler.evaluate
// (babe_amd/testing/blind_bwe_sampler.py) issues from Python, as ONE C-ABI call for a non-Python host:
//   EDM.denoiser preconditioning (diff_params/edm.py:144-159) around the CQTDiff+ network (CQT.fwd -> UNet -> CQT.bwd,
//   networks/cqtdiff+.py:730-845), apply_hpf_DC (testing/blind_bwe_sampler.py:152-157), the STFT of the denoised estimate, the
//   filter fit (fit_params :533-595), design_filter / apply_filter (utils/blind_bwe_utils.py:82-119, 6-39), the reconstruction-
//   guidance residual norm and its gradient through iSTFT o H o STFT and through the network (get_rec_grads :75-135), the score
//   direction (:125-135, :701).
// Scope: the default blind / known-fc_A configuration - L2 guidance norm, STFT-domain low-pass, no observation noise, no
// data-consistency replacement, no AR mask, no FIR degradation (the options of the other tester YAMLs stay with the Python
// sequencer, which this call equals bit for bit on the default path: tests/test_gpu_eval_c.py).
// Every intermediate lives in the caller's workspace (babe_eval_workspace_bytes); nothing is allocated, nothing synchronises.
#include "common.h"
#include "../../include/babe_hip.h"

namespace {

__global__ void fill_kernel(float* out, float v, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = v;
}

struct Carver {
    char* p;
    char* end;
    bool ok = true;
    template <typename T>
    T* take(size_t n) {
        const size_t bytes = (n * sizeof(T) + 255) / 256 * 256;
        if (p + bytes > end) {
            ok = false;
            return nullptr;
        }
        T* r = reinterpret_cast<T*>(p);
        p += bytes;
        return r;
    }
};

struct EvalGeom {
    int nocts, binsoct, T_oct[8];
    long coef_floats;                 // per clip, all octaves
    int hop, frames, nbins;
    long ntot;
};

bool geom(const babe_eval_desc* e, EvalGeom& g) {
    const void* d = babe_cqt_plan_design(e->cqt_plan);
    if (!d) return false;
    long nb = 0;
    int Toct[8] = {0};
    babe_cqt_design_get(d, "nb", &nb, 8);
    const long bytes = babe_cqt_design_get(d, "T_oct", nullptr, 0);
    if (bytes <= 0 || bytes > 32) return false;
    babe_cqt_design_get(d, "T_oct", Toct, 32);
    g.nocts = (int)(bytes / 4);
    g.binsoct = (int)(nb / g.nocts);
    g.coef_floats = 0;
    for (int j = 0; j < g.nocts; ++j) {
        g.T_oct[j] = Toct[j];
        g.coef_floats += 2L * g.binsoct * Toct[j];
    }
    g.hop = e->nfft / 2;
    g.frames = 1 + e->L / g.hop;
    g.nbins = g.hop + 1;
    g.ntot = e->nfft + (long)g.hop * (g.frames - 1);
    return true;
}

long unet_ws_bytes(const babe_eval_desc* e, const EvalGeom& g, int B) {
    int Tlvl[8];
    for (int j = 0; j < g.nocts; ++j) Tlvl[j] = g.T_oct[j];
    return babe_unet_workspace_bytes(e->unet_plan, B, Tlvl);
}

}  // namespace

extern "C" int babe_fill(float* out, float v, int n, void* stream) {
    BABE_CHECK_ARG(out && n > 0, "fill: bad arguments");
    hipLaunchKernelGGL(fill_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, out, v, n);
    BABE_LAUNCH_CHECK();
    return BABE_OK;
}

extern "C" long babe_eval_workspace_bytes(const babe_eval_desc* e, int B) {
    if (!e || !e->unet_plan || !e->cqt_plan || B < 1 || e->nfft < 256 || e->L < e->nfft) {
        babe_set_error("eval_workspace_bytes: bad descriptor");
        return -1;
    }
    EvalGeom g;
    if (!geom(e, g)) {
        babe_set_error("eval_workspace_bytes: the CQT plan has no design");
        return -1;
    }
    const long cq = babe_cqt_workspace_bytes(e->cqt_plan, B), un = unet_ws_bytes(e, g, B);
    if (cq < 0 || un < 0) return -1;
    const long L = e->L, spec = 2L * g.frames * g.nbins, fr = (long)g.frames * e->nfft;
    long fl = 0;                       // floats, every piece rounded up to 256 bytes by the carver: 64 floats of slack each
    const int pieces = 40;
    fl += 9L * B * L;                                              // xin net xd x_den r seed g_den g_net g_xin (g_x reuses xd)
    fl += 2L * B * g.coef_floats;                                  // coefficients in / out (and their gradients: reused)
    fl += 2L * B * spec + (long)B * fr;                            // specX, spec of the seed, frames
    fl += (long)B * (1 + 2 * e->rff_n + e->emb_dim[1] + e->emb_dim[2] + e->emb_dim[3] + e->film_J);
    fl += (long)B * g.nbins + 64;                                  // H
    const long dbl = (long)B * 3 * g.nbins + 2L * B * 64;          // statistics, two sets of partial sums
    return cq + un + 4 * fl + 8 * dbl + 4L * B + 256L * pieces + 4096;
}

/* x [B][L]: the noisy state at noise level t; (cskip, cout, cin, cnoise): EDM preconditioning of t (diff_params/edm.py:46-60,
 * computed by the host, so that this call contains no formula of its own); y [B][L]: the observations; specY = babe_stft_fwd(y);
 * params [P][2][K], P = B (per-clip semantics) or 1 (desc.shared: the reference's batch coupling): filter parameters, updated IN
 * PLACE when desc.blind; d [B][L] receives -t * score, x_den [B][L] the denoised estimate (after the DC / Nyquist high-pass);
 * n_iter [P] (may be NULL) the fit's iteration counts. */
extern "C" int babe_score_eval(const babe_eval_desc* e, const float* x, float t, float cskip, float cout, float cin, float cnoise,
                               const float* y, const float* specY, float* params, float* d, float* x_den, int* n_iter, void* ws,
                               long ws_bytes, int B, void* stream) {
    BABE_CHECK_ARG(e && x && y && specY && params && d && x_den && ws && B > 0, "score_eval: bad arguments");
    BABE_CHECK_ARG(e->unet_plan && e->unet_state && e->cqt_plan && e->rff_freq && e->film_W && e->film_b && e->env_inv && e->tw4096,
                   "score_eval: incomplete descriptor");
    BABE_CHECK_ARG(e->K >= 1 && e->K <= 8 && e->rff_n > 0 && e->emb_dim[0] == 2 * e->rff_n, "score_eval: K = %d, rff_n = %d, emb_dim[0] = %d",
                   e->K, e->rff_n, e->emb_dim[0]);
    EvalGeom g;
    BABE_CHECK_ARG(geom(e, g), "score_eval: the CQT plan has no design");
    BABE_CHECK_ARG(ws_bytes >= babe_eval_workspace_bytes(e, B), "score_eval: workspace of %ld bytes, need %ld", ws_bytes,
                   babe_eval_workspace_bytes(e, B));
    const long L = e->L, n = (long)B * L;
    const int P = e->shared ? 1 : B;
    Carver c{static_cast<char*>(ws), static_cast<char*>(ws) + ws_bytes};
    float* cqws = reinterpret_cast<float*>(c.take<char>((size_t)babe_cqt_workspace_bytes(e->cqt_plan, B)));
    const long unb = unet_ws_bytes(e, g, B);
    void* unws = c.take<char>((size_t)unb);
    float *xin = c.take<float>(n), *net = c.take<float>(n), *xd = c.take<float>(n), *r = c.take<float>(n), *seed = c.take<float>(n),
          *g_den = c.take<float>(n), *g_net = c.take<float>(n), *g_xin = c.take<float>(n), *hp = c.take<float>(n);
    float* coA[8];
    float* coB[8];
    for (int j = 0; j < g.nocts; ++j) coA[j] = c.take<float>((size_t)B * 2 * g.binsoct * g.T_oct[j]);
    for (int j = 0; j < g.nocts; ++j) coB[j] = c.take<float>((size_t)B * 2 * g.binsoct * g.T_oct[j]);
    const long spec = 2L * g.frames * g.nbins;
    float *specX = c.take<float>((size_t)B * spec), *specS = c.take<float>((size_t)B * spec), *fr = c.take<float>((size_t)B * g.frames * e->nfft);
    float *cn = c.take<float>(B), *h0 = c.take<float>((size_t)B * 2 * e->rff_n), *h1 = c.take<float>((size_t)B * e->emb_dim[1]),
          *h2 = c.take<float>((size_t)B * e->emb_dim[2]), *h3 = c.take<float>((size_t)B * e->emb_dim[3]), *film = c.take<float>((size_t)B * e->film_J);
    float* H = c.take<float>((size_t)P * g.nbins);
    double *stats = c.take<double>((size_t)P * 3 * g.nbins), *part = c.take<double>((size_t)B * 64), *gpart = c.take<double>((size_t)B * 64);
    int* nit = c.take<int>(P);
    BABE_CHECK_ARG(c.ok, "score_eval: workspace carve failed (internal size formula)");
    int Tlvl[8];
    for (int j = 0; j < g.nocts; ++j) Tlvl[j] = g.T_oct[j];
#define EV(call)                     \
    do {                             \
        const int rc__ = (call);     \
        if (rc__ != BABE_OK) return rc__; \
    } while (0)
    // ---- denoiser: cskip x + cout net(cin x, cnoise), then the DC / Nyquist high-pass
    EV(babe_lincomb3(xin, cin, x, 0.f, nullptr, 0.f, nullptr, n, stream));
    EV(babe_fill(cn, cnoise, B, stream));
    EV(babe_rff(cn, e->rff_freq, h0, B, e->rff_n, stream));
    EV(babe_linear(h0, e->emb_W[0], e->emb_b[0], h1, B, e->emb_dim[0], e->emb_dim[1], 1, stream));
    EV(babe_linear(h1, e->emb_W[1], e->emb_b[1], h2, B, e->emb_dim[1], e->emb_dim[2], 1, stream));
    EV(babe_linear(h2, e->emb_W[2], e->emb_b[2], h3, B, e->emb_dim[2], e->emb_dim[3], 1, stream));
    EV(babe_linear(h3, e->film_W, e->film_b, film, B, e->emb_dim[3], e->film_J, 0, stream));
    EV(babe_cqt_fwd(e->cqt_plan, xin, coA, cqws, B, stream));
    EV(babe_unet_fwd(e->unet_plan, e->unet_state, coA, film, e->film_J, B, Tlvl, unws, unb, coB, stream));
    EV(babe_cqt_bwd(e->cqt_plan, coB, net, cqws, B, stream));
    EV(babe_lincomb3(xd, cskip, x, cout, net, 0.f, nullptr, n, stream));
    if (e->hpf) EV(babe_cqt_hpf(e->cqt_plan, xd, x_den, cqws, B, stream));
    else EV(babe_lincomb3(x_den, 1.f, xd, 0.f, nullptr, 0.f, nullptr, n, stream));
    // ---- filter fit on STFT magnitudes, filter design
    EV(babe_stft_fwd(x_den, L, (int)L, nullptr, specX, B, e->nfft, g.frames, e->tw4096, stream));
    if (e->blind) {
        EV(babe_stft_mag_stats(specX, specY, stats, B, g.nbins, g.frames, e->shared, stream));
        EV(babe_filter_fit(stats, params, n_iter ? n_iter : nit, P, e->K, g.nbins, e->fs, e->nfft, &e->fit, stream));
    }
    EV(babe_design_filter(params, H, P, e->K, g.nbins, e->fs, e->nfft, stream));
    const long H_bs = P == B ? g.nbins : 0;            // per-clip filters, or one filter for the batch
    // ---- reconstruction guidance: residual y - A(x_den), its norm's gradient through iSTFT o H o STFT
    EV(babe_spec_filter_istft(specX, H, H_bs, fr, B, e->nfft, g.frames, e->tw4096, stream));
    EV(babe_ola(fr, e->env_inv, y, L, r, L, part, 64, B, (int)L, e->nfft, g.frames, stream));
    EV(babe_residual_seed(r, L, part, 64, e->env_inv, seed, L, B, (int)L, stream));
    EV(babe_stft_fwd(seed, L, (int)L, nullptr, specS, B, e->nfft, g.frames, e->tw4096, stream));
    EV(babe_spec_filter_istft(specS, H, H_bs, fr, B, e->nfft, g.frames, e->tw4096, stream));
    EV(babe_ola(fr, nullptr, nullptr, 0, e->hpf ? hp : g_den, L, nullptr, 64, B, (int)L, e->nfft, g.frames, stream));
    if (e->hpf) EV(babe_cqt_hpf(e->cqt_plan, hp, g_den, cqws, B, stream));
    // ---- through the network: g_x = cskip g_den + cin net^T(cout g_den)
    EV(babe_lincomb3(g_net, cout, g_den, 0.f, nullptr, 0.f, nullptr, n, stream));
    EV(babe_cqt_bwd_adjoint(e->cqt_plan, g_net, coA, cqws, B, stream));
    EV(babe_unet_vjp(e->unet_plan, e->unet_state, coA, coB, stream));
    EV(babe_cqt_fwd_adjoint(e->cqt_plan, coB, g_xin, cqws, B, stream));
    EV(babe_lincomb3(xd, cskip, g_den, cin, g_xin, 0.f, nullptr, n, stream));                 // g_x (xd is free again)
    EV(babe_sumsq_partial(xd, L, gpart, 64, B, L, stream));
    EV(babe_score_direction(x_den, x, xd, gpart, 64, d, t, e->xi, (float)e->audio_len_norm, e->shared, e->score_mode, B, L, stream));
#undef EV
    return BABE_OK;
}
