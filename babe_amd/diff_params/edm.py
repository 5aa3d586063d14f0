"""Karras-EDM parameterisation, host side.  Same surface as /root/reference/diff_params/edm.py
(EDM :7-159): create_schedule :55-64, create_schedule_from_initial_t :66-75, get_gamma :38-53,
sample_prior :98-106, cskip/cout/cin/cnoise :108-139, denoiser :144-159, mutable sigma_* / S* fields
(mutated by BlindSampler.update_diff_params, blind_bwe_sampler.py:50-60).  Scalars stay float32
torch CPU tensors so schedules are bit-identical to the reference; tensors on the GPU go through
the babe_hip element-wise kernel."""
import torch

from ..stft import lincomb


class EDM:
    def __init__(self, args):
        self.args = args
        dp = args.diff_params
        self.sigma_min, self.sigma_max = dp.sigma_min, dp.sigma_max
        self.P_mean, self.P_std = dp.get("P_mean", -1.2), dp.get("P_std", 1.2)
        self.ro, self.ro_train = dp.ro, dp.get("ro_train", dp.ro)
        self.sigma_data = dp.sigma_data
        self.Schurn, self.Stmin, self.Stmax, self.Snoise = dp.Schurn, dp.Stmin, dp.Stmax, dp.Snoise
        if dp.get("aweighting", {}).get("use_aweighting", False):
            raise NotImplementedError("A-weighting is a training-only option (edm.py:33-34)")

    def get_gamma(self, t):
        N = t.shape[0]
        gamma = torch.zeros(t.shape)
        sel = torch.logical_and(t > self.Stmin, t < self.Stmax)
        gamma[sel] = gamma[sel] + torch.min(torch.Tensor([self.Schurn / N, 2 ** (1 / 2) - 1]))
        return gamma

    def _sched(self, s0, nb_steps):
        i = torch.arange(0, nb_steps + 1)
        t = (s0 ** (1 / self.ro) + i / (nb_steps - 1) * (self.sigma_min ** (1 / self.ro) - s0 ** (1 / self.ro))) ** self.ro
        t[-1] = 0
        return t

    def create_schedule(self, nb_steps):
        return self._sched(self.sigma_max, nb_steps)

    def create_schedule_from_initial_t(self, initial_t, nb_steps):
        return self._sched(initial_t, nb_steps)

    def sample_prior(self, shape, sigma):
        return torch.randn(shape) * sigma

    def cskip(self, sigma):
        return self.sigma_data ** 2 * (sigma ** 2 + self.sigma_data ** 2) ** -1

    def cout(self, sigma):
        return sigma * self.sigma_data * (self.sigma_data ** 2 + sigma ** 2) ** (-0.5)

    def cin(self, sigma):
        return (self.sigma_data ** 2 + sigma ** 2) ** (-0.5)

    def cnoise(self, sigma):
        return (1 / 4) * torch.log(torch.as_tensor(sigma, dtype=torch.float32))

    def denoiser(self, xn, net, sigma):
        """cskip*x + cout*net(cin*x, cnoise) for ONE sigma shared by the batch (like the reference's use)."""
        s = torch.as_tensor(sigma, dtype=torch.float32).reshape(-1)[0].cpu()
        B = xn.shape[0]
        xin = lincomb(torch.empty_like(xn), float(self.cin(s)), xn.contiguous())
        cn = self.cnoise(s).reshape(1, 1).expand(B, 1).contiguous().to(xn.device)
        out = net(xin, cn)
        return lincomb(torch.empty_like(xn), float(self.cskip(s)), xn.contiguous(), float(self.cout(s)), out.contiguous())
