"""Known-degradation EDM sampler (config #1) on the babe_hip kernels.

Drop-in for the reference's ``testing.edm_sampler.Sampler`` (/root/reference/testing/edm_sampler.py:8-305):
``Sampler(model, diff_params, args, rid=False)``, ``predict_bwe(ylpf, filt, 'firwin')`` :266-305 ->
``predict_conditional`` -> ``predict`` :166-229 with ``get_score_rec_guidance`` :56-94.  Differences from
BlindSampler that matter for parity: the schedule always starts from sigma_max with a pure
prior sample (:177-179); no noise is drawn on steps with gamma == 0 (:187-198); Snoise scales the noise;
the guidance scale is xi/(||g||/sqrt(L)*t + 1e-6) without the extra 1/t (:78-92).
Only the FIR degradation ('firwin' / 'firwin_hpf') runs on the HIP path; IIR/biquad/resample/decimate
are torchaudio paths that no target config uses.
"""
import torch

from ..stft import lincomb
from .blind_bwe_sampler import BlindSampler


class Sampler(BlindSampler):
    SCORE_MODE = 1

    def __init__(self, model, diff_params, args, rid=False, batch_semantics="per_clip", noise_device="cpu"):
        self.model = model
        self.diff_params = diff_params
        self.args = args
        if not args.tester.diff_params.same_as_training:
            self.update_diff_params()
        self.order = args.tester.order
        self.xi = args.tester.posterior_sampling.xi
        # posterior_sampling.data_consistency (:47-54, :113-122): after the guided score, the replacement step
        # x0 <- y + x0 - A(x0) on the Tweedie estimate - BlindSampler.evaluate's classic branch with the FIR as A.
        # xi = 0 (:124-130): no guidance at all, the replacement step on the plain denoised estimate (evaluate below).
        self.data_consistency = bool(args.tester.posterior_sampling.data_consistency)
        self._dc_cfg = self.data_consistency
        self.nb_steps = args.tester.T
        self.rid = rid
        self.batch_semantics = batch_semantics
        self.noise_device = noise_device
        self._stft = None
        self.norm, self.smoothl1_beta, self.stft_dist = 2, 1.0, None      # edm_sampler.py:71: torch.linalg.norm(y - den_rec, ord=2) only
        self.fir_taps = None
        self.ar_mask = None
        self.dc = None
        self.inpaint_mask = None

    def stft_ops(self, L, device):
        if self._stft is None or self._stft.L != L:
            from ..stft import STFTOps
            self._stft = STFTOps(4096, L, self.args.exp.sample_rate, device)     # only its residual_seed helper is used
        return self._stft

    def evaluate(self, x, t, y, specY, filter_params, blind, lane=None):
        if self.xi > 0 or y is None:
            return super().evaluate(x, t, y, specY, filter_params, blind, lane)
        # xi = 0: x0 = D(x) (no high-pass in this branch of the reference, :126), x0 <- y + x0 - A(x0), d = (x - x0) / t
        from ..stft import fir_same
        x_den = self.get_denoised_estimate(x, t, lane, hpf=False)
        from ..stft import mask_blend
        a0 = mask_blend(self.inpaint_mask, x_den, None) if self.inpaint_mask is not None else fir_same(x_den, self.fir_taps)
        x0 = lincomb(torch.empty_like(x), 1.0, x_den, 1.0, y, -1.0, a0)
        return lincomb(torch.empty_like(x), 1.0 / float(t), x, -1.0 / float(t), x0), x_den, filter_params

    def predict_bwe(self, ylpf, filt, filt_type):
        if filt_type not in ("firwin", "firwin_hpf"):
            raise NotImplementedError(f"filt_type={filt_type!r}: only FIR degradations run on the HIP path")
        self.fir_taps = torch.as_tensor(filt, dtype=torch.float32).reshape(-1).contiguous().to(ylpf.device)
        return self.predict_conditional(ylpf)

    def predict_inpainting(self, y_masked, mask):
        """Masking degradation A(x) = mask * x (edm_sampler.py:231-243 -> predict_conditional): y_masked [B,L], mask [L] or [B,L]
        (1 = observed).  Guidance gradient through the mask, data-consistency replacement x0 <- y + x0 - mask * x0 if configured."""
        self.fir_taps = None
        self.inpaint_mask = torch.as_tensor(mask, dtype=torch.float32).contiguous().to(y_masked.device)
        try:
            return self.predict_conditional(y_masked)
        finally:
            self.inpaint_mask = None

    def predict_unconditional(self, shape, device):
        """Unguided sampling (edm_sampler.py:231-243 -> predict :166-229 with y = None)."""
        return self.predict_conditional(None, shape=tuple(shape), device=torch.device(device))

    def predict_conditional(self, y, shape=None, device=None):
        dp = self.diff_params
        if y is not None:
            y = y.contiguous().float()
            shape, device = y.shape, y.device
        B, L = shape
        self.stft_ops(L, device)
        T = self.nb_steps
        if self.rid:
            data_denoised = torch.zeros((T, B, L))
        t = dp.create_schedule(T)
        x = (self._randn((B, L), device) * float(t[0])).contiguous()
        gamma = dp.get_gamma(t)
        for i in range(T):
            if float(gamma[i]) == 0:
                t_hat, x_hat = t[i], x
            else:
                t_hat = t[i] + gamma[i] * t[i]
                eps = self._randn((B, L), device).contiguous()
                x_hat = lincomb(torch.empty_like(x), 1.0, x, float((t_hat ** 2 - t[i] ** 2) ** (1 / 2)) * float(dp.Snoise), eps)
            d, _, _ = self.evaluate(x_hat, float(t_hat), y, None, None, blind=False)
            h = float(t[i + 1] - t_hat)
            if float(t[i + 1]) != 0 and self.order == 2:
                x_prime = lincomb(torch.empty_like(x), 1.0, x_hat, h, d)
                d2, _, _ = self.evaluate(x_prime, float(t[i + 1]), y, None, None, blind=False)
                x = lincomb(torch.empty_like(x), 1.0, x_hat, 0.5 * h, d, 0.5 * h, d2)
            else:
                x = lincomb(torch.empty_like(x), 1.0, x_hat, h, d)
            if self.rid:
                data_denoised[i] = x.cpu()
        if self.rid:
            return x, data_denoised, t
        return x
