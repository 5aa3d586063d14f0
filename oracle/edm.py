"""ORACLE (test infrastructure) — Karras-EDM scalars and preconditioning.

Follows /root/reference/diff_params/edm.py: create_schedule :55-64,
create_schedule_from_initial_t :66-75, get_gamma :38-53, cskip/cout/cin/cnoise
:108-139, denoiser :144-159.  float32 torch arithmetic like the reference.
Pinned by tests/golden/edm.npz (G1).
"""
import math

import torch


class EDMParams:
    def __init__(self, sigma_data, sigma_min, sigma_max, ro, Schurn=0.0, Stmin=0.0, Stmax=50.0, Snoise=1.0):
        self.sigma_data, self.sigma_min, self.sigma_max, self.ro = sigma_data, sigma_min, sigma_max, ro
        self.Schurn, self.Stmin, self.Stmax, self.Snoise = Schurn, Stmin, Stmax, Snoise


def schedule(p, nb_steps, initial_t=None):
    s0 = p.sigma_max if initial_t is None else initial_t
    i = torch.arange(0, nb_steps + 1)
    t = (s0 ** (1 / p.ro) + i / (nb_steps - 1) * (p.sigma_min ** (1 / p.ro) - s0 ** (1 / p.ro))) ** p.ro
    t[-1] = 0
    return t


def gamma(p, t):
    N = t.shape[0]
    g = torch.zeros_like(t)
    sel = torch.logical_and(t > p.Stmin, t < p.Stmax)
    g[sel] = g[sel] + min(torch.tensor(p.Schurn / N, dtype=torch.float32).item(), 2 ** 0.5 - 1)
    return g


def cskip(p, s):
    return p.sigma_data ** 2 * (s ** 2 + p.sigma_data ** 2) ** -1


def cout(p, s):
    return s * p.sigma_data * (p.sigma_data ** 2 + s ** 2) ** (-0.5)


def cin(p, s):
    return (p.sigma_data ** 2 + s ** 2) ** (-0.5)


def cnoise(p, s):
    return 0.25 * torch.log(s)


def denoiser(p, net, xn, sigma):
    """cskip*x + cout*net(cin*x, cnoise); sigma tensor [B,1] or [1]."""
    if sigma.dim() == 1:
        sigma = sigma.unsqueeze(-1)
    return cskip(p, sigma) * xn + cout(p, sigma) * net(cin(p, sigma) * xn, cnoise(p, sigma))
