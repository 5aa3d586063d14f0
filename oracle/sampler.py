"""ORACLE (test infrastructure) — guided stochastic Heun sampler for blind BWE.

Follows /root/reference/testing/blind_bwe_sampler.py: predict_blind_bwe
:619-769, move_timestep :509-516, get_denoised_estimate :152-157,
get_rec_grads :75-135 (norm: 2, 'smoothl1' :99-100 and 'cosine' :101-103), fit_params :533-595, and the known-filter
variant predict_bwe('fc_A') :351-364 + predict :406-498.
Noise is injected (list of tensors in the reference's draw order: prior first,
then one per step) so runs are reproducible against the golden records (G8).
"""
import torch

from . import bwe_utils as U
from . import edm as E


class OracleBlindSampler:
    def __init__(self, net, cqt, edm_params, *, fs, audio_len, T=35, order=2, xi=0.2, start_sigma=0.2,
                 nfft=4096, fc_init=(280, 285, 290, 295, 300), A_init=(-15, -17, -20, -25, -30),
                 mu=(1000.0, 10.0), tol=(5e-3, 5e-3), max_iter=100, fcmin=20.0, Amin=-50.0,
                 weighting="sqrt", filter_out_cqt_DC_Nyq=True, norm=2, smoothl1_beta=1.0, stft_distance=None,
                 data_consistency=False, SNR_observations=None, sigma_den_estimate=0.0):
        """stft_distance: None or dict(nfft=, weight=, mag=, logmag=) = posterior_sampling.stft_distance.use (:105-115).
        SNR_observations (dB) / sigma_den_estimate: the observation-noise regularisation of get_rec_grads :80-86 and
        fit_params :542-552 - y receives fresh noise IN PLACE before every fit and every guidance evaluation."""
        self.net, self.cqt, self.p = net, cqt, edm_params
        self.obs_snr = None if SNR_observations is None else 10.0 ** (SNR_observations / 10.0)
        self.sigma_den = sigma_den_estimate
        self.norm, self.smoothl1_beta, self.stft_distance = norm, smoothl1_beta, stft_distance
        self.data_consistency = data_consistency          # posterior_sampling.data_consistency (conf/tester/blind_bwe_DC.yaml)
        self.fs, self.audio_len, self.T, self.order, self.xi = fs, audio_len, T, order, xi
        self.start_sigma, self.nfft = start_sigma, nfft
        self.fc_init, self.A_init = fc_init, A_init
        self.fit_kw = dict(fs=fs, nfft=nfft, mu=mu, tol=tol, max_iter=max_iter, fcmin=fcmin, Amin=Amin,
                           weighting=weighting)
        self.hpf = filter_out_cqt_DC_Nyq
        self.freqs = U.bin_freqs(nfft, fs)

    def distance(self, y, rec):
        """posterior_sampling.norm of get_rec_grads (:99-117), per batch item (smooth-L1: one scalar, 'sum' reduction)."""
        if self.norm == "smoothl1":
            return torch.nn.functional.smooth_l1_loss(y, rec, reduction="sum", beta=self.smoothl1_beta)
        if self.norm == "cosine":
            return (1 - torch.nn.functional.cosine_similarity(rec, y, dim=1, eps=1e-6)).clamp(min=0)
        if self.stft_distance is not None:
            return U.stft_distance(y, rec, **self.stft_distance)
        return torch.linalg.norm(y - rec, dim=1, ord=self.norm)

    def denoised(self, x, t):
        xd = E.denoiser(self.p, self.net, x, t.reshape(1, 1).expand(x.shape[0], 1) if t.dim() == 0 else t)
        if self.hpf:
            xd = self.cqt.apply_hpf_DC(xd)
        return xd

    def rec_grads(self, x_den, y, x, t, params):
        H = U.design_filter(params[0], params[1], self.freqs)
        rec = U.apply_filter(x_den, H, self.nfft)
        norm = self.distance(y, rec)
        g, = torch.autograd.grad(norm.sum(), x)
        s = self.xi / (torch.linalg.norm(g) / self.audio_len ** 0.5 + 1e-6)
        return s * g / t

    def _obs_noise(self, y, draw):
        sigma = torch.sqrt(torch.var(y, -1) / self.obs_snr).unsqueeze(-1)          # :81-86, :543-548
        y += sigma * draw()

    def evaluate(self, x, t, y, params, blind=True, timers=None, draw=None):
        """One score evaluation. Returns score, x_den (detached), new params.
        timers: optional dict accumulating wall seconds of the components (bench.py's cpu_baseline split):
        'unet_fwd' (denoiser forward incl. CQT + high-pass), 'fit' (fit_params), 'filter' (filter apply + norm),
        'vjp' (autograd through iSTFT/STFT/high-pass/UNet)."""
        import time
        tick = time.perf_counter
        x = x.detach().requires_grad_(True)
        t0 = tick()
        x_den = self.denoised(x, t)
        t1 = tick()
        xd2 = x_den.detach().clone()
        if blind:
            if self.obs_snr is not None:
                self._obs_noise(y, draw)
            den_fit = xd2 + draw() * self.sigma_den if self.sigma_den else xd2          # :551-552
            params, _ = U.fit_params(den_fit, y, params, **self.fit_kw)
        if self.obs_snr is not None:
            self._obs_noise(y, draw)
        t2 = tick()
        if timers is None:
            rg = self.rec_grads(x_den, y, x, t, params)
        else:
            H = U.design_filter(params[0], params[1], self.freqs)
            rec = U.apply_filter(x_den, H, self.nfft)
            norm = self.distance(y, rec)
            t3 = tick()
            g, = torch.autograd.grad(norm.sum(), x)
            t4 = tick()
            rg = self.xi / (torch.linalg.norm(g) / self.audio_len ** 0.5 + 1e-6) * g / t
            for k, v in (("unet_fwd", t1 - t0), ("fit", t2 - t1), ("filter", t3 - t2), ("vjp", t4 - t3)):
                timers[k] = timers.get(k, 0.0) + v
        score = (xd2 - x.detach()) / t ** 2 - rg
        if self.data_consistency:
            # replacement step on the Tweedie estimate (data_consistency_step_classic :63-73; :178-188, :704-709, :748-753)
            x3 = score * t ** 2 + x.detach()
            H = U.design_filter(params[0], params[1], self.freqs)
            x3 = y + x3 - U.apply_filter(x3, H, self.nfft)
            score = (x3 - x.detach()) / t ** 2
        return score, xd2, params

    def predict_blind_bwe(self, y, noises, blind=True, params=None, record=None):
        p = self.p
        if params is None:
            params = torch.tensor([list(self.fc_init), list(self.A_init)], dtype=torch.float32)
        it = iter(noises)                       # the reference's draw order: prior, then per step the step noise followed by
        draw = lambda: next(it)                 # the observation-noise draws of that step's evaluations (if enabled)
        if self.obs_snr is not None:
            y = y.clone()                       # (noise is added in place)
        if self.start_sigma is None:
            t = E.schedule(p, self.T)
            x = draw() * t[0]
        else:
            t = E.schedule(p, self.T, self.start_sigma)
            x = y + draw() * t[0]
        gam = E.gamma(p, t)
        for i in range(self.T):
            t_hat = t[i] + gam[i] * t[i]
            x_hat = x + ((t_hat ** 2 - t[i] ** 2) ** 0.5) * draw()
            score, xden, params = self.evaluate(x_hat, t_hat, y, params, blind, draw=draw)
            d = -t_hat * score
            h = t[i + 1] - t_hat
            if record is not None:
                record.append(dict(x_hat=x_hat.clone(), t_hat=t_hat.clone(), x_den=xden.clone(), params=params.clone()))
            if t[i + 1] != 0 and self.order == 2:
                x_prime = x_hat + h * d
                score2, _, params = self.evaluate(x_prime, t[i + 1], y, params, blind, draw=draw)
                d2 = -t[i + 1] * score2
                x = x_hat + h * (0.5 * d + 0.5 * d2)
            else:
                x = x_hat + h * d
        return x.detach(), params.detach()


class OracleEDMSampler:
    """Known-FIR-degradation sampler: /root/reference/testing/edm_sampler.py Sampler.predict :166-229,
    get_score_rec_guidance :56-94, apply_FIR_filter :245-252 (conv1d padding='same', no kernel flip)."""

    def __init__(self, net, cqt, edm_params, *, audio_len, T=35, order=2, xi=0.25, filter_out_cqt_DC_Nyq=True,
                 data_consistency=False):
        self.net, self.cqt, self.p = net, cqt, edm_params
        self.audio_len, self.T, self.order, self.xi, self.hpf = audio_len, T, order, xi, filter_out_cqt_DC_Nyq
        self.data_consistency = data_consistency          # posterior_sampling.data_consistency (edm_sampler.py:47-54, :113-130)

    def score(self, x, t, y, taps):
        # taps: FIR taps, or a callable degradation (predict_inpainting: v -> mask * v, edm_sampler.py:231-243)
        fir = taps if callable(taps) else (lambda v: torch.nn.functional.conv1d(v.unsqueeze(1), taps.view(1, 1, -1), padding="same").squeeze(1))
        if self.xi <= 0:
            # :124-130 - no guidance: the denoised estimate with the replacement step, always
            with torch.no_grad():
                xd = E.denoiser(self.p, self.net, x, t.reshape(1, 1).expand(x.shape[0], 1))   # (no hpf here in the reference)
                xd = y + xd - fir(xd)
                return (xd - x) / t ** 2
        sc = self._guided(x, t, y, taps)
        if self.data_consistency:
            # :113-122 - Tweedie estimate, replacement x0 <- y + x0 - A(x0), back to a score
            x0 = sc * t ** 2 + x
            x0 = y + x0 - fir(x0)
            sc = (x0 - x) / t ** 2
        return sc

    def _guided(self, x, t, y, taps):
        x = x.detach().requires_grad_(True)
        xd = E.denoiser(self.p, self.net, x, t.reshape(1, 1).expand(x.shape[0], 1))
        if self.hpf:
            xd = self.cqt.apply_hpf_DC(xd)
        rec = taps(xd) if callable(taps) else torch.nn.functional.conv1d(xd.unsqueeze(1), taps.view(1, 1, -1), padding="same").squeeze(1)
        norm = torch.linalg.norm(y - rec, dim=1, ord=2)
        g, = torch.autograd.grad(norm.sum(), x)
        s = self.xi / (torch.linalg.norm(g) / self.audio_len ** 0.5 * t + 1e-6)
        return (xd.detach() - x.detach()) / t ** 2 - s * g

    def predict_bwe(self, y, taps, noises):
        p = self.p
        t = E.schedule(p, self.T)
        gam = E.gamma(p, t)
        it = iter(noises)
        x = next(it) * t[0]
        for i in range(self.T):
            if gam[i] == 0:
                t_hat, x_hat = t[i], x
            else:
                t_hat = t[i] + gam[i] * t[i]
                x_hat = x + ((t_hat ** 2 - t[i] ** 2) ** 0.5) * (next(it) * p.Snoise)
            d = -t_hat * self.score(x_hat, t_hat, y, taps)
            h = t[i + 1] - t_hat
            if t[i + 1] != 0 and self.order == 2:
                d2 = -t[i + 1] * self.score(x_hat + h * d, t[i + 1], y, taps)
                x = x_hat + h * (0.5 * d + 0.5 * d2)
            else:
                x = x_hat + h * d
        return x.detach()


def edm_predict_inpainting(smp, y_masked, mask, noises):
    """OracleEDMSampler with the masking degradation of Sampler.predict_inpainting (edm_sampler.py:231-243)."""
    return smp.predict_bwe(y_masked, lambda v: mask * v, noises)


def smooth_mask(mask, size):
    """prepare_smooth_mask, /root/reference/testing/blind_bwe_sampler.py:232-257 (same element-by-element walk)."""
    hann = torch.hann_window(size * 2)
    left, right = hann[0:size], hann[size:]
    B, N = mask.shape
    m = mask[0]
    prev = 1
    new = m.clone()
    for i in range(len(m)):
        if m[i] != prev:
            if m[i] == 0:
                new[i - size:i] = right
            if m[i] == 1:
                new[i:i + size] = left
        prev = m[i]
    return new.unsqueeze(0).expand(B, -1)


def predict_bwe_AR(smp, ylpf, y_masked, params, mask, noises, inpaint_DC=True):
    """OracleBlindSampler + the AR degradation / data-consistency step: blind_bwe_sampler.py:259-303 ('fc_A'),
    get_score :160-188, data_consistency_step_classic :63-73, predict :406-498."""
    p = smp.p
    y = mask * y_masked + (1 - mask) * ylpf
    if inpaint_DC:
        sm = smooth_mask(mask, 50)
        y_sm = sm * y_masked
    t = E.schedule(p, smp.T, smp.start_sigma)
    it = iter(noises)
    x = y + next(it) * t[0]
    gam = E.gamma(p, t)

    def score_fn(xx, tt):
        xx = xx.detach().requires_grad_(True)
        x_den = smp.denoised(xx, tt)
        H = U.design_filter(params[0], params[1], smp.freqs)
        rec = mask * x_den + (1 - mask) * U.apply_filter(x_den, H, smp.nfft)
        norm = torch.linalg.norm(y - rec, dim=1, ord=2)
        g, = torch.autograd.grad(norm.sum(), xx)
        s = smp.xi / (torch.linalg.norm(g) / smp.audio_len ** 0.5 + 1e-6)
        score = (x_den.detach() - xx.detach()) / tt ** 2 - s * g / tt
        if inpaint_DC:
            x0 = score * tt ** 2 + xx.detach()
            x0 = y_sm + x0 - sm * x0
            score = (x0 - xx.detach()) / tt ** 2
        return score

    for i in range(smp.T):
        t_hat = t[i] + gam[i] * t[i]
        x_hat = x + ((t_hat ** 2 - t[i] ** 2) ** 0.5) * (next(it) * p.Snoise)
        d = -t_hat * score_fn(x_hat, t_hat)
        h = t[i + 1] - t_hat
        if t[i + 1] != 0 and smp.order == 2:
            d2 = -t[i + 1] * score_fn(x_hat + h * d, t[i + 1])
            x = x_hat + h * (0.5 * d + 0.5 * d2)
        else:
            x = x_hat + h * d
    return x.detach()
