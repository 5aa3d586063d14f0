"""Guided stochastic-Heun EDM sampler for (blind) bandwidth extension on the babe_hip kernels.

Drop-in for the reference's ``testing.blind_bwe_sampler.BlindSampler``
(/root/reference/testing/blind_bwe_sampler.py): same constructor ``(model, diff_params, args,
rid=False)`` and the same ``predict_blind_bwe`` :619-769 / ``predict_bwe(..., 'fc_A')`` :306-364
call surface and return values.  Per score evaluation (:685-761):

  x_den  = hpf_DC( c_skip x + c_out F(c_in x, ln(sigma)/4) )            get_denoised_estimate :152-157
  params = projected GD on the STFT-magnitude fit                      fit_params :533-595
  g      = d/dx || y - iSTFT(H(params) STFT(x_den)) ||_2  (through F)  get_rec_grads :75-123
  score  = (x_den - x)/t^2 - xi/(||g||/sqrt(L)+1e-6) g/t               :125-135, :701

No autograd: the guidance gradient is the hand-wired VJP (STFT/iSTFT adjoints, self-adjoint
high-pass, UNet input-VJP).  All tensor work is C-ABI calls; the host only sequences them.

``batch_semantics``:
  'reference' - one shared filter for the batch and a whole-batch guidance norm, exactly what the
                reference does for B>1 (blind_bwe_utils.py:295 flattens B; blind_bwe_sampler.py:125).
  'per_clip'  - every clip has its own filter and guidance norm (== the reference run at B=1 per
                clip); this is what independent-clip batching/sharding uses.  Identical at B=1.
"""
import os

import torch

from ..stft import STFTOps, add_obs_noise, fir_same, lincomb, make_fit_cfg, mask_blend
from .._lib import check, lib, ptr, stream

# Every score evaluation of the default configuration is ONE library call (csrc/score_eval.hip through testing/eval_c.py: the path
# a non-Python host takes; the default since round 6: +1.2 % on the whole job, the host enqueues 2 x 69 calls per clip instead of
# 2 x 69 x ~1200).  BABE_EVAL_C=0: this class sequences the kernels itself (bit-identical on the library's CQT plan); the options
# outside the default configuration always do.
EVAL_C = os.environ.get("BABE_EVAL_C", "1") == "1"


class BlindSampler:
    NBLK = 64
    SCORE_MODE = 0          # guidance scaling of blind_bwe_sampler.py:125-135

    obs_snr, sigma_den = None, 0.0        # (class defaults: testing/edm_sampler.Sampler has no observation-noise options)

    def __init__(self, model, diff_params, args, rid=False, batch_semantics="per_clip", noise_device="cpu",
                 max_segments_in_flight=32):
        """max_segments_in_flight: with per-clip semantics a larger batch is restored in sub-batches of at most this many
        segments, one after the other (clips are independent; the UNet keeps 3.6 GB of activations per 368368-sample
        segment between forward and VJP, so 32 in flight = 115 GB of the 288 GB).  Ignored for 'reference' semantics,
        where the clips of a batch are coupled through the shared filter."""
        self.max_in_flight = int(max_segments_in_flight)
        self.model = model
        self.diff_params = diff_params
        self.args = args
        if not args.tester.diff_params.same_as_training:
            self.update_diff_params()
        ps = args.tester.posterior_sampling
        self.order = args.tester.order
        self.xi = ps.xi
        self.data_consistency = ps.data_consistency
        self._dc_cfg = bool(ps.data_consistency)       # the blind loop reads the CONFIG value (:704, :748), never the
        # attribute that predict_bwe_AR flips for good (:300); the known-filter loop reads the attribute (:178)
        self.nb_steps = args.tester.T
        bb = args.tester.blind_bwe
        self.mu = [bb.optimization.mu[0], bb.optimization.mu[1]]
        self.fcmin = bb.fcmin
        self.fcmax = args.exp.sample_rate // 2 if bb.fcmax == "nyquist" else bb.fcmax
        self.Amin, self.Amax = bb.Amin, bb.Amax
        self.tol = bb.optimization.tol
        self.start_sigma = None if ps.start_sigma == "None" else ps.start_sigma
        if ps.norm not in (2, "smoothl1", "cosine"):
            raise NotImplementedError("posterior_sampling.norm must be 2, 'smoothl1' or 'cosine'")
        self.norm = ps.norm
        self.stft_dist = None
        # (older tester YAMLs - the conf/tester files written for a `testing.blind_bwe.*` package the reference no longer
        # ships - have no stft_distance / SNR_observations keys: absent means off, tests/test_config_surface.py)
        sdist = ps.get("stft_distance", None)
        if ps.norm == 2 and sdist is not None and sdist.use:          # same precedence as get_rec_grads :99-117
            if sdist.get("use_multires", False):
                raise NotImplementedError("posterior_sampling.stft_distance.use_multires: dead code in the reference (get_rec_grads :108 calls "
                                          "self.norm, which the sampler never defines)")
            mode = (2 if sdist.get("logmag", False) else 1) if sdist.mag else 0
            self.stft_dist = dict(nfft=int(sdist.nfft), mode=mode, weight=ps.freq_weighting)
        self.smoothl1_beta = ps.get("smoothl1_beta", 1.0)
        # observation-noise regularisation (get_rec_grads :80-86, fit_params :542-552; conf/tester/blind_bwe_2.yaml sets
        # SNR_observations: 50): the observations receive fresh noise IN PLACE before every fit and before every guidance
        # evaluation, and the fit may see a noisy copy of the denoised estimate.  The draws interleave with the step noise, so
        # these modes run the single-stream loop (no lanes) and draw in the reference's order.
        snr_db = ps.get("SNR_observations", "None")
        self.obs_snr = None if snr_db == "None" else 10.0 ** (float(snr_db) / 10.0)
        self.sigma_den = float(bb.get("sigma_den_estimate", 0) or 0)

        assert batch_semantics in ("per_clip", "reference")
        self.batch_semantics = batch_semantics
        self.noise_device = noise_device
        self.fit_cfg = make_fit_cfg(mu=self.mu, tol=self.tol, max_iter=bb.optimization.max_iter, fcmin=self.fcmin,
                                    fcmax=self.fcmax, Amin=self.Amin, Amax=self.Amax,
                                    clamp_fc=bb.optimization.clamp_fc, clamp_A=bb.optimization.clamp_A,
                                    only_negative_A=bb.optimization.only_negative_A,
                                    weighting=ps.freq_weighting_filter)
        self._stft = None
        self._ceval = {}               # BABE_EVAL_C=1: one library-side evaluation descriptor per lane (testing/eval_c.py)
        self.fir_taps = None
        self.ar_mask = None            # predict_bwe_AR: degradation(x) = mask*x + (1-mask)*A(x)
        self.dc = None                 # (smooth_mask, y_smooth_masked) of the replacement data-consistency step

    def update_diff_params(self):
        dp, src = self.diff_params, self.args.tester.diff_params
        dp.sigma_min, dp.sigma_max, dp.ro, dp.sigma_data = src.sigma_min, src.sigma_max, src.ro, src.sigma_data
        dp.Schurn, dp.Stmin, dp.Stmax, dp.Snoise = src.Schurn, src.Stmin, src.Stmax, src.Snoise

    # ------------------------------------------------------------------ helpers
    def stft_ops(self, L, device):
        if self._stft is None or self._stft.L != L:
            self._stft = STFTOps(self.args.tester.blind_bwe.NFFT, L, self.args.exp.sample_rate, device)
        return self._stft

    def _randn(self, shape, device):
        if self.noise_device == "cpu":
            return torch.randn(shape).to(device)          # reference: CPU draw then copy (:513, edm.py:105)
        return torch.randn(shape, device=device)

    def _sumsq(self, g):
        B, n = g.shape
        part = torch.empty(B, self.NBLK, device=g.device, dtype=torch.float64)
        check(lib().babe_sumsq_partial(ptr(g), g.stride(0), ptr(part), self.NBLK, B, n, stream()), "sumsq_partial")
        return part

    def _seed(self, st, r, y, part, post):
        """d(distance)/d(rec) for the configured guidance distance (get_rec_grads :99-117)."""
        if self.stft_dist is not None:
            sd = self.stft_dist
            if sd.get("ops") is None or sd["ops"].L != r.shape[1] or sd["ops"].dev != r.device:
                from ..stft import STFTOps
                from ..stft import freq_weights
                sd["ops"] = STFTOps(sd["nfft"], r.shape[1], self.args.exp.sample_rate, r.device)
                sd["w"] = freq_weights(sd["ops"].nbins, sd["weight"]).to(r.device)
            rec = lincomb(torch.empty_like(r), 1.0, y, -1.0, r)             # r = y - rec
            g = sd["ops"].distance_grad(rec, y, sd["w"], sd["mode"], shared=self.batch_semantics == "reference")
            return st.residual_seed(g, None, post=post, norm="ready") if post else g
        return st.residual_seed(r, part, post=post, norm=self.norm, y=y, beta=self.smoothl1_beta)

    def _lane_kw(self, lane):
        return {"lane": lane} if (lane is not None and getattr(self.model, "supports_lanes", False)) else {}

    def get_denoised_estimate(self, x, t, lane=None, hpf=True):
        """x [B,L] device, t host float -> hpf_DC(denoiser(x))  (:152-157); keeps the UNet context for the VJP.
        hpf=False: without the DC / Nyquist high-pass (the unguided replacement branch of edm_sampler.py:124-130 has none)."""
        dp = self.diff_params
        s = torch.as_tensor(t, dtype=torch.float32)
        self._c = (float(dp.cskip(s)), float(dp.cout(s)), float(dp.cin(s)))
        cskip, cout, cin = self._c
        B = x.shape[0]
        xin = lincomb(torch.empty_like(x), cin, x)
        cn = torch.full((B, 1), float(dp.cnoise(s)), device=x.device, dtype=torch.float32)   # (no host-to-device copy: a
        # pageable H2D copy in the loop would stall the host on this stream and starve the other lanes)
        net = self.model.fwd_nograd(xin, cn, **self._lane_kw(lane))
        xd = lincomb(torch.empty_like(x), cskip, x, cout, net)
        if hpf and self.args.tester.filter_out_cqt_DC_Nyq:
            xd = self.model.CQTransform.apply_hpf_DC(xd)
        return xd

    def fit_params(self, specX, specY, filter_params):
        """filter_params [P,2,K] (device) updated by the projected GD (:533-595); returns (params, n_iter)."""
        st = self._stft
        stats = st.mag_stats(specX, specY, shared=(self.batch_semantics == "reference"))
        p = filter_params.clone()
        nit = st.filter_fit(stats, p, self.fit_cfg)
        return p, nit

    def evaluate(self, x, t, y, specY, filter_params, blind, lane=None):
        """One score evaluation at noise level t. Returns (d = -t*score, x_den, filter_params).
        lane: engine state of the network to use (the caller runs this call chain on a stream of its own)."""
        st = self._stft
        cq = self.model.CQTransform
        B, L = x.shape
        if EVAL_C:
            from . import eval_c
            if eval_c.supported(self, y, blind) and x.is_contiguous() and y.is_contiguous():
                ce = self._ceval.get(lane)
                if ce is None or ce._keep[1] is not st:
                    ce = self._ceval[lane] = eval_c.CEval(self, lane)
                return ce(self, x, t, y, specY, filter_params, blind)
        x_den = self.get_denoised_estimate(x, t, lane)
        cskip, cout, cin = self._c
        specX_fit = None
        if y is not None and (self.obs_snr is not None or (blind and self.sigma_den)):
            # reference order of the draws inside one evaluation: fit_params (y, then the denoised estimate), get_rec_grads (y)
            if blind:
                if self.obs_snr is not None:
                    add_obs_noise(y, self._randn(y.shape, y.device).contiguous(), self.obs_snr)
                    specY = st.stft(y)
                if self.sigma_den:
                    den_fit = lincomb(torch.empty_like(x_den), 1.0, x_den, self.sigma_den, self._randn(x_den.shape, x_den.device).contiguous())
                    specX_fit = st.stft(den_fit)
                else:
                    specX_fit = st.stft(x_den)
                filter_params, self.last_n_iter = self.fit_params(specX_fit, specY, filter_params)
            if self.obs_snr is not None:
                add_obs_noise(y, self._randn(y.shape, y.device).contiguous(), self.obs_snr)
        if y is None:
            # unconditional (get_score :160-170): d = -t*(x_den - x)/t^2
            return lincomb(torch.empty_like(x), 1.0 / float(t), x, -1.0 / float(t), x_den), x_den, filter_params
        if self.ar_mask is not None:
            # mask-mixed degradation of predict_bwe_AR (:280-288): mask*x + (1-mask)*A(x), A = fc_A filter or FIR
            m = self.ar_mask
            if self.fir_taps is not None:
                rec0 = fir_same(x_den, self.fir_taps)
            else:
                H = st.design_filter(filter_params)
                Hq = H if H.shape[0] == B else H[0]
                rec0 = st.ola(st.filter_frames(st.stft(x_den), Hq), normalise=True)
            rec = mask_blend(m, x_den, rec0)
            r = lincomb(torch.empty_like(y), 1.0, y, -1.0, rec)
            part = self._sumsq(r)
            seed_raw = self._seed(st, r, y, part, post=False)                # -r/||r||
            if self.fir_taps is not None:
                gA = fir_same(mask_blend(m, None, seed_raw), self.fir_taps, adjoint=True)
            else:
                u = mask_blend(m, None, self._seed(st, r, y, part, post=True))
                gA = st.ola(st.filter_frames(st.stft(u), Hq), normalise=False)
            g_den = lincomb(torch.empty_like(gA), 1.0, mask_blend(m, seed_raw, None), 1.0, gA)
        elif getattr(self, "inpaint_mask", None) is not None:
            # masking degradation of edm_sampler.Sampler.predict_inpainting (edm_sampler.py:231-243): A(x) = mask * x, self-adjoint
            m = self.inpaint_mask
            r = lincomb(torch.empty_like(y), 1.0, y, -1.0, mask_blend(m, x_den, None))
            g_den = mask_blend(m, self._seed(st, r, y, self._sumsq(r), post=False), None)
        elif self.fir_taps is not None:
            # known FIR degradation (edm_sampler.py:245-252): residual, then the transpose FIR
            rec = fir_same(x_den, self.fir_taps)
            r = lincomb(torch.empty_like(y), 1.0, y, -1.0, rec)
            seed = self._seed(st, r, y, self._sumsq(r), post=False)
            g_den = fir_same(seed, self.fir_taps, adjoint=True)
        else:
            specX = st.stft(x_den)
            if blind and specX_fit is None:
                filter_params, self.last_n_iter = self.fit_params(specX, specY, filter_params)
            H = st.design_filter(filter_params)                     # [P,nbins]
            Hq = H if H.shape[0] == B else H[0]
            # reconstruction guidance: forward residual and hand-wired VJP
            r, part = st.ola(st.filter_frames(specX, Hq), normalise=True, y=y)
            seed = self._seed(st, r, y, part, post=True)
            g_den = st.ola(st.filter_frames(st.stft(seed), Hq), normalise=False)
        if self.args.tester.filter_out_cqt_DC_Nyq:
            g_den = cq.apply_hpf_DC(g_den)                      # zero-phase real filter: self-adjoint
        g_net = lincomb(torch.empty_like(g_den), cout, g_den)
        g_xin = self.model.vjp(g_net, **self._lane_kw(lane))
        g_x = lincomb(torch.empty_like(g_den), cskip, g_den, cin, g_xin)
        gpart = self._sumsq(g_x)
        d = torch.empty_like(x)
        check(lib().babe_score_direction(ptr(x_den), ptr(x), ptr(g_x), ptr(gpart), self.NBLK, ptr(d), float(t),
                                         float(self.xi), float(self.args.exp.audio_len),
                                         int(self.batch_semantics == "reference"), self.SCORE_MODE, B, L, stream()),
              "score_direction")
        if self.dc is not None:
            # replacement data-consistency step (:63-73, :178-188): x0 <- y_sm + x0 - smooth_mask*x0 on the Tweedie
            # estimate x0 = x + t^2*score = x - t*d, then back to a direction
            sm, y_sm = self.dc
            x0 = lincomb(torch.empty_like(x), 1.0, x, -float(t), d)
            x0 = lincomb(torch.empty_like(x), 1.0, mask_blend(sm, None, x0), 1.0, y_sm)
            d = lincomb(torch.empty_like(x), 1.0 / float(t), x, -1.0 / float(t), x0)
        elif (self._dc_cfg if blind else self.data_consistency):
            # posterior_sampling.data_consistency (conf/tester/blind_bwe_DC.yaml, bwe_formal_1000_DC.yaml): the classic
            # replacement x0 <- y + x0 - A(x0) with the CURRENT degradation (:63-73; :178-188, :704-709, :748-753)
            x0 = lincomb(torch.empty_like(x), 1.0, x, -float(t), d)
            if getattr(self, "inpaint_mask", None) is not None:
                a0 = mask_blend(self.inpaint_mask, x0, None)
            elif self.fir_taps is not None:
                a0 = fir_same(x0, self.fir_taps)
            else:
                H = st.design_filter(filter_params)
                a0 = st.ola(st.filter_frames(st.stft(x0), H if H.shape[0] == B else H[0]), normalise=True)
            x0 = lincomb(torch.empty_like(x), 1.0, x0, 1.0, y, -1.0, a0)
            d = lincomb(torch.empty_like(x), 1.0 / float(t), x, -1.0 / float(t), x0)
        return d, x_den, filter_params

    # ------------------------------------------------------------------ sampling loops
    def step(self, x, t_i, gamma_i, t_next, eps, y, specY, filter_params, blind, snoise=1.0, lane=None):
        """ONE stochastic Heun step of the reverse diffusion (:687-761 / predict :439-480): noise injection
        (move_timestep :509-516), score evaluation, 2nd-order correction unless t_next == 0 or order == 1.
        Returns (x_next, filter_params, rec) with rec = dict(x_hat, t_hat, x_den, d) of the first evaluation.
        Exposed so that a step can be teacher-forced from recorded reference state (tests/test_gpu_sampler.py)."""
        t_hat = t_i + gamma_i * t_i
        x_hat = lincomb(torch.empty_like(x), 1.0, x, float((t_hat ** 2 - t_i ** 2) ** (1 / 2)) * float(snoise), eps)
        d, x_den, filter_params = self.evaluate(x_hat, float(t_hat), y, specY, filter_params, blind, lane)
        rec = dict(x_hat=x_hat, t_hat=float(t_hat), x_den=x_den, d=d, filter_params=filter_params)
        h = float(t_next - t_hat)
        if float(t_next) != 0 and self.order == 2:
            x_prime = lincomb(torch.empty_like(x), 1.0, x_hat, h, d)
            d2, _, filter_params = self.evaluate(x_prime, float(t_next), y, specY, filter_params, blind, lane)
            x = lincomb(torch.empty_like(x), 1.0, x_hat, 0.5 * h, d, 0.5 * h, d2)
        else:
            x = lincomb(torch.empty_like(x), 1.0, x_hat, h, d)
        return x, filter_params, rec

    def _sample(self, y, filter_params, blind, rid, snoise=1.0, shape=None, device=None, diag=(False, False)):
        """y None: unconditional sampling of `shape` on `device` (predict_unconditional :366-374).
        diag = (test_filter_fit, compute_sweep): the two diagnostics of the known-degradation loop (predict :419-466) - per step, on
        the guided Tweedie estimate: a filter fit from the INITIAL conditions (the reference never feeds the estimate back) and
        the fit objective + gradient on a (fc, A) grid.  They do not touch the trajectory; with rid their records are appended."""
        test_fit, sweep = diag if (y is not None and not blind) else (False, False)
        dp = self.diff_params
        if y is not None:
            device = y.device
            y = y.contiguous().float()
            if self.obs_snr is not None:
                y = y.clone()                  # the observation noise is added in place: never to the caller's tensor
            shape = y.shape
        B, L = shape
        with torch.cuda.device(device):
            st = self.stft_ops(L, device)
            specY = st.stft(y) if (y is not None and self.fir_taps is None) else None
            T = self.nb_steps
            if rid:
                data_denoised = torch.zeros((T, B, L))
                if blind:
                    data_filters = torch.zeros((T, *filter_params.shape[-2:])) if filter_params.shape[0] == 1 else \
                        torch.zeros((T, *filter_params.shape))
                else:
                    data_score = torch.zeros((T, B, L))
            if self.start_sigma is None or y is None:
                t = dp.create_schedule(T)
                x = lincomb(torch.empty(B, L, device=device), float(t[0]), self._randn((B, L), device).contiguous())
            else:
                t = dp.create_schedule_from_initial_t(self.start_sigma, T)
                x = lincomb(torch.empty_like(y), 1.0, y, float(t[0]), self._randn((B, L), device).contiguous())
            gamma = dp.get_gamma(t)
            if sweep:
                self.fc_s, self.A_s = torch.logspace(2.5, 4, 15), torch.linspace(-80, -5, 12)      # (:439-440)
            if rid and test_fit:
                data_fit = []
            if rid and sweep:
                data_norms = torch.zeros((T, 15, 12))
                data_grads = torch.zeros((T, 15, 12, 2))
            if self._use_lanes(B, y, rid, filter_params) and not (test_fit or sweep):
                x, filter_params = self._sample_lanes(x, y, specY, filter_params, blind, snoise, t, gamma)
                T = 0                                        # (loop below skipped)
            for i in range(T):
                eps = self._randn((B, L), device).contiguous()
                x, filter_params, rec = self.step(x, t[i], gamma[i], t[i + 1], eps, y, specY, filter_params, blind, snoise)
                if rid and blind:
                    # predict_blind_bwe records the denoised estimate BEFORE guidance (x_den_2, :720) and the filter
                    data_denoised[i] = rec["x_den"].cpu()
                    fp1 = rec["filter_params"]
                    data_filters[i] = (fp1[0] if fp1.shape[0] == 1 else fp1).cpu()
                elif rid:
                    # predict records the guided Tweedie estimate score*t^2 + x_hat = x_hat - t*d and the score (:452-454)
                    th = rec["t_hat"]
                    data_denoised[i] = lincomb(torch.empty_like(x), 1.0, rec["x_hat"], -th, rec["d"]).cpu()
                    data_score[i] = lincomb(torch.empty_like(x), -1.0 / th, rec["d"]).cpu()
                if test_fit or sweep:
                    den = lincomb(torch.empty_like(x), 1.0, rec["x_hat"], -rec["t_hat"], rec["d"])       # score2denoised (:452, :456)
                    if test_fit:
                        est = self.fit_params_signal(den, y, self._init_params(B, device))
                        if rid:
                            data_fit.append((est[0] if est.shape[0] == 1 else est).cpu())
                    if sweep:
                        norms, grads = self.compute_sweep(den, y)
                        if rid:
                            data_norms[i], data_grads[i] = norms, grads
        if blind:
            fp_out = filter_params[0] if (filter_params.shape[0] == 1) else filter_params
            return (x, fp_out, data_denoised, t, data_filters) if rid else (x, fp_out)
        if not rid:
            return (x,)
        out = (x, data_denoised, data_score, t)
        if test_fit:
            out = out + (torch.stack(data_fit),)
        if sweep:
            out = out + (data_norms, data_grads)
        return out

    # ---- the reference's helper methods under their own names and signatures (third-party code that pokes at the sampler)
    def move_timestep(self, x, t, gamma, Snoise=1):
        """(:509-516) x_hat = x + sqrt(t_hat^2 - t^2) * Snoise * eps, t_hat = t + gamma t; the noise is drawn here, like there."""
        t_hat = t + gamma * t
        eps = self._randn(tuple(x.shape), x.device).contiguous()
        x_hat = lincomb(torch.empty_like(x), 1.0, x.contiguous(), float((t_hat ** 2 - t ** 2) ** (1 / 2)) * float(Snoise), eps)
        return x_hat, t_hat

    def apply_filter_fcA(self, x, filter_params):
        """(:518-520) x [B,L] through the piecewise filter filter_params [2,K]."""
        st = self.stft_ops(x.shape[-1], x.device)
        return st.apply_filter(x.contiguous(), st.design_filter(torch.as_tensor(filter_params, dtype=torch.float32, device=x.device)))

    def fit_params_signal(self, denoised_estimate, y, filter_params):
        """The reference's fit_params(denoised_estimate, y, filter_params) (:533-595) on SIGNALS: both STFTs are taken here.
        filter_params [2,K] or [P,2,K]; returned in the same form.  (fit_params above takes the spectra - the sampling loop has them.)"""
        st = self.stft_ops(y.shape[-1], y.device)
        fp = torch.as_tensor(filter_params, dtype=torch.float32, device=y.device)
        single = fp.dim() == 2
        out, self.last_n_iter = self.fit_params(st.stft(denoised_estimate.contiguous()), st.stft(y.contiguous()),
                                                (fp.unsqueeze(0) if single else fp).contiguous())
        return out[0] if single else out

    def get_rec_grads(self, x_hat, y, x, t_i, degradation=None, filter_params=None):
        """(:75-135) s * grad_x ||y - A(x_hat)|| / t_i with s = xi / (||grad|| / sqrt(audio_len) + 1e-6), for x_hat = the estimate
        the LAST get_denoised_estimate(x, t) call returned: where the reference differentiates through the network with autograd,
        this runs the network's hand-wired VJP on the state that call left (so it must directly follow it, as in the reference's
        get_score).  L2 norm and STFT-domain low-pass A = filter_params [2,K] (or the FIR taps set by predict_bwe(..., 'firwin'));
        `degradation` is accepted for the signature and not called.  The sampling loop does the same inside evaluate()."""
        if self.norm != 2 or self.stft_dist is not None or self.obs_snr is not None:
            raise NotImplementedError("get_rec_grads helper: default guidance distance only (the sampling loop handles the others)")
        st = self.stft_ops(y.shape[-1], y.device)
        cskip, cout, cin = self._c
        x_hat, y = x_hat.contiguous(), y.contiguous()
        if self.fir_taps is not None:
            r = lincomb(torch.empty_like(y), 1.0, y, -1.0, fir_same(x_hat, self.fir_taps))
            g_den = fir_same(self._seed(st, r, y, self._sumsq(r), post=False), self.fir_taps, adjoint=True)
        else:
            H = st.design_filter(torch.as_tensor(filter_params, dtype=torch.float32, device=y.device))
            Hq = H[0] if (H.dim() == 2 and H.shape[0] != x_hat.shape[0]) else H
            r, part = st.ola(st.filter_frames(st.stft(x_hat), Hq), normalise=True, y=y)
            g_den = st.ola(st.filter_frames(st.stft(self._seed(st, r, y, part, post=True)), Hq), normalise=False)
        if self.args.tester.filter_out_cqt_DC_Nyq:
            g_den = self.model.CQTransform.apply_hpf_DC(g_den)
        g_xin = self.model.vjp(lincomb(torch.empty_like(g_den), cout, g_den))
        g_x = lincomb(torch.empty_like(g_den), cskip, g_den, cin, g_xin)
        normguide = g_x.double().pow(2).sum().sqrt() / self.args.exp.audio_len ** 0.5          # (whole batch, like torch.linalg.norm there)
        return g_x * float(self.xi / (normguide + 1e-6) / float(t_i))

    def compute_sweep(self, denoised_estimate, y):
        """(:598-616) the fit objective ||w (|X| H(fc, A) - |Y|)|| and its gradient w.r.t. (fc, A) on the grid fc_s x A_s
        (15 x 12, one break point): returns (norms [15,12], grads [15,12,2]) on the host.  One launch evaluates all 180 points
        from the per-bin sufficient statistics (babe_filter_loss_grad); the reference runs 180 autograd passes."""
        st = self.stft_ops(y.shape[-1], y.device)
        if getattr(self, "fc_s", None) is None:
            self.fc_s, self.A_s = torch.logspace(2.5, 4, 15), torch.linspace(-80, -5, 12)
        stats = st.mag_stats(st.stft(denoised_estimate.contiguous()), st.stft(y.contiguous()), shared=True)
        nf, na = self.fc_s.numel(), self.A_s.numel()
        grid = torch.stack([self.fc_s.float().cpu().repeat_interleave(na), self.A_s.float().cpu().repeat(nf)], 1).unsqueeze(-1)
        lg = st.filter_loss_grad(stats, grid.contiguous().to(y.device), self.fit_cfg).cpu()
        return lg[:, 0].reshape(nf, na), lg[:, 1:3].reshape(nf, na, 2)

    # ---- clip-level pipelining -------------------------------------------------------------------------------------------
    # Clips are independent (per-clip semantics), so their whole evaluation chains - UNet forward, CQT, STFT, the
    # single-workgroup filter fit, guidance, UNet VJP, Heun update - run on separate HIP streams, one engine state of the
    # network per lane.  The kernels that cannot fill the GPU on their own (filter fit: one workgroup for 1.8 ms; the dense
    # DFT stages; the per-clip CQT / STFT launches) then overlap the other lane's convolutions instead of serialising
    # behind a join on the caller's stream.  Enqueue order alternates between the lanes once per score evaluation so that
    # neither queue runs dry; the noise is drawn for the whole batch in the reference's order BEFORE the loop, so results
    # are identical to the single-stream loop.
    LANES = 2

    def _use_lanes(self, B, y, rid, filter_params):
        return (self.LANES > 1 and B >= 2 and y is not None and not rid and self.batch_semantics == "per_clip" and
                getattr(self.model, "supports_lanes", False) and
                self.ar_mask is None and self.dc is None and self.obs_snr is None and not self.sigma_den and
                self.fir_taps is None and filter_params.shape[0] == B)

    def _lane_step(self, ln, i, t, gamma, noise, blind, snoise, half):
        """One half of stochastic Heun step i on lane `ln` (enqueued on the current stream): half 0 = noise injection +
        first score evaluation (+ the Euler update when the step has no correction), half 1 = the Heun correction."""
        t_hat = t[i] + gamma[i] * t[i]
        h = float(t[i + 1] - t_hat)
        heun = float(t[i + 1]) != 0 and self.order == 2
        if half == 0:
            eps = noise[ln["sl"]]
            x_hat = lincomb(torch.empty_like(ln["x"]), 1.0, ln["x"].contiguous(),
                            float((t_hat ** 2 - t[i] ** 2) ** (1 / 2)) * float(snoise), eps)
            d, _, ln["fp"] = self.evaluate(x_hat, float(t_hat), ln["y"], ln["specY"], ln["fp"], blind, ln["k"])
            ln["x_hat"], ln["d"] = x_hat, d
            if not heun:
                ln["x"] = lincomb(torch.empty_like(x_hat), 1.0, x_hat, h, d)
        elif heun:
            x_prime = lincomb(torch.empty_like(ln["x_hat"]), 1.0, ln["x_hat"], h, ln["d"])
            d2, _, ln["fp"] = self.evaluate(x_prime, float(t[i + 1]), ln["y"], ln["specY"], ln["fp"], blind, ln["k"])
            ln["x"] = lincomb(torch.empty_like(x_prime), 1.0, ln["x_hat"], 0.5 * h, ln["d"], 0.5 * h, d2)

    # (Rounds 2-4 carried an opt-in HIP-graph replay of the lanes' Heun steps, BABE_SAMPLER_GRAPHS=1.  Measured again on the
    # driver's command in round 5, same box: 2.129 (graphs) vs 2.140 / 2.140 (eager) audio-sec/s - the eager loop has no host
    # sync and enqueues an evaluation in 17 ms of Python against 34 ms of GPU time per evaluation and lane, so there is no launch
    # gap for a graph to close.  Removed, with its pointer-lifetime bookkeeping: one way to sequence the sampler from Python -
    # this loop - and one for a non-Python host, the library-side UNet sequencer csrc/unet_engine.hip.)
    def _sample_lanes(self, x, y, specY, filter_params, blind, snoise, t, gamma):
        B, L = x.shape
        T = self.nb_steps
        dev = x.device
        # (a network that must not run two chains at once - precision='bf16' without the opt-in, see lanes_ok_for - gets ONE lane
        # with the whole batch: same launches as the plain loop, on one stream)
        ok_for = getattr(self.model, "lanes_ok_for", None)
        lanes_ok = ok_for(self.noise_device) if ok_for is not None else getattr(self.model, "concurrent_lanes_ok", True)
        nl = min(self.LANES, B) if lanes_ok else 1
        per = -(-B // nl)
        main = torch.cuda.current_stream(dev)
        if getattr(self, "_lane_streams", None) is None or len(self._lane_streams) < nl:
            self._lane_streams = [torch.cuda.Stream(device=dev) for _ in range(nl)]
        noises = [self._randn((B, L), dev).contiguous() for _ in range(T)]          # reference order: one draw per step
        lanes = []
        for k in range(nl):
            b0, b1 = k * per, min(B, (k + 1) * per)
            if b0 < b1:
                lanes.append(dict(k=k, st=self._lane_streams[k], sl=slice(b0, b1), x=x[b0:b1], y=y[b0:b1],
                                  specY=specY[b0:b1], fp=filter_params[b0:b1].contiguous()))
        ready = torch.cuda.Event()
        ready.record(main)                                     # everything the lanes read has been enqueued on `main`
        for ln in lanes:
            ln["st"].wait_event(ready)
        for i in range(T):
            for half in (0, 1):                                # first evaluation of step i lane after lane, then the second
                for ln in lanes:
                    with torch.cuda.stream(ln["st"]):
                        self._lane_step(ln, i, t, gamma, noises[i], blind, snoise, half)
        for ln in lanes:
            done = torch.cuda.Event()
            done.record(ln["st"])
            main.wait_event(done)
            ln["x"].record_stream(main)
            ln["fp"].record_stream(main)
        for n_ in noises:
            for ln in lanes:
                n_.record_stream(ln["st"])
        return torch.cat([ln["x"] for ln in lanes], 0), torch.cat([ln["fp"] for ln in lanes], 0)

    def _init_params(self, B, device):
        ic = self.args.tester.blind_bwe.initial_conditions
        p = torch.tensor([list(ic.fc), list(ic.A)], dtype=torch.float32)
        if p.dim() == 1:
            p = p.unsqueeze(1)
        P = 1 if self.batch_semantics == "reference" else B
        return p.unsqueeze(0).repeat(P, 1, 1).contiguous().to(device)

    def predict_blind_bwe(self, y, rid=False, compute_sweep=False):
        """y [B,L] observations on the GPU -> (x, filter_params[, data_denoised, t, data_filters])  (:619-769)."""
        # (compute_sweep is accepted and ignored, as in the reference: its blind loop :619-769 never reads the flag)
        B = y.shape[0]
        if self.batch_semantics == "per_clip" and B > self.max_in_flight and not rid:
            outs = [self._sample(y[i:i + self.max_in_flight], self._init_params(min(self.max_in_flight, B - i), y.device),
                                 blind=True, rid=False) for i in range(0, B, self.max_in_flight)]
            return torch.cat([o[0] for o in outs], 0), torch.cat([o[1].reshape(-1, *o[1].shape[-2:]) for o in outs], 0)
        return self._sample(y, self._init_params(B, y.device), blind=True, rid=rid)

    def predict_bwe(self, ylpf, filt, filt_type, rid=False, test_filter_fit=False, compute_sweep=False):
        """Known-degradation variant (:306-364 -> predict_conditional :387-404 -> predict :406-498).
        filt_type 'fc_A' (filt = [2,K] breakpoints) or 'firwin' / 'firwin_hpf' (filt = FIR taps applied with
        conv1d(padding="same"), :211-218).  rid=True returns (x, data_denoised, data_score, t) like predict."""
        dev = ylpf.device
        if filt_type == "fc_A":
            p = torch.as_tensor(filt, dtype=torch.float32)
            if p.dim() == 1:
                p = p.unsqueeze(1)
            params = p.unsqueeze(0).contiguous().to(dev)
            self.fir_taps = None
        elif filt_type in ("firwin", "firwin_hpf"):
            self.fir_taps = torch.as_tensor(filt, dtype=torch.float32).reshape(-1).contiguous().to(dev)
            params = torch.zeros(1, 2, 1, device=dev)
        else:
            raise NotImplementedError(f"filt_type={filt_type!r}: 'fc_A' and 'firwin' run on the HIP path (the IIR / "
                                      f"resample degradations are torchaudio paths no target config uses)")
        try:
            res = self._sample(ylpf, params, blind=False, rid=rid, snoise=self.diff_params.Snoise,
                               diag=(bool(test_filter_fit), bool(compute_sweep)))
        finally:
            self.fir_taps = None
        return res if rid else res[0]

    def predict_unconditional(self, shape, device, rid=False):
        """Unguided sampling from the prior (:366-374): score = (D(x) - x)/t^2, no observations."""
        res = self._sample(None, None, blind=False, rid=rid, snoise=self.diff_params.Snoise, shape=tuple(shape),
                           device=torch.device(device))
        return res if rid else res[0]

    # ------------------------------------------------------------------ autoregressive out-painting ("next" row 2)
    @staticmethod
    def prepare_smooth_mask(mask, size=10):
        """Hann-smoothed copy of a 0/1 mask [B,N] (:232-257): every 1->0 edge gets the falling half window just
        before it, every 0->1 edge the rising half just after it.  Vectorised form of the reference's Python loop."""
        hann = torch.hann_window(size * 2)
        left, right = hann[0:size], hann[size:]
        B, N = mask.shape
        m = mask[0].detach().cpu()
        new = m.clone()
        prev = torch.cat((torch.ones(1), m[:-1]))
        for i in torch.nonzero(m != prev).flatten().tolist():
            if m[i] == 0:
                new[i - size:i] = right
            else:
                new[i:i + size] = left
        return new.unsqueeze(0).expand(B, -1).contiguous().to(mask.device)

    def predict_bwe_AR(self, ylpf, y_masked, filt, filt_type, rid=False, test_filter_fit=False, compute_sweep=False,
                       mask=None):
        """(:259-303) observations = mask*y_masked + (1-mask)*ylpf; degradation(x) = mask*x + (1-mask)*A(x)."""
        assert mask is not None
        if test_filter_fit or compute_sweep:
            raise NotImplementedError("test_filter_fit / compute_sweep (logging only)")
        dev = ylpf.device
        ylpf = ylpf.contiguous().float()
        mask = mask.to(dev).float().contiguous()
        y_masked = y_masked.to(dev).float().contiguous()
        B, L = ylpf.shape
        if filt_type == "fc_A":
            p = torch.as_tensor(filt, dtype=torch.float32)
            if p.dim() == 1:
                p = p.unsqueeze(1)
            params = p.unsqueeze(0).contiguous().to(dev)
            self.fir_taps = None
        elif filt_type == "firwin":
            self.fir_taps = torch.as_tensor(filt, dtype=torch.float32).reshape(-1).contiguous().to(dev)
            params = torch.zeros(1, 2, 1, device=dev)
        else:
            raise NotImplementedError(filt_type)
        y = mask_blend(mask, y_masked, ylpf)
        self.ar_mask = mask
        self.dc = None
        if self.args.tester.complete_recording.inpaint_DC:
            sm = self.prepare_smooth_mask(mask, 50)
            self.dc = (sm, mask_blend(sm, y_masked, None))
            self.data_consistency = True                      # the reference flips this permanently (:300)
        try:
            res = self._sample(y, params, blind=False, rid=rid, snoise=self.diff_params.Snoise)
        finally:
            self.ar_mask, self.dc, self.fir_taps = None, None, None
        return res if rid else res[0]
