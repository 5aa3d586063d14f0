"""Segmented application of the denoiser pre-pass.

Mirrors BlindTester.apply_denoiser / apply_denoiser_model of /root/reference/testing/denoise_and_bwe_tester.py:109-165
(same method names, same arithmetic): fixed-size segments (`sample_rate_denoiser * segment_size` samples) hopping by
segment - 1024, each zero-padded by one STFT window, STFT 1024/256 (periodic Hamming, center=False) -> network -> inverse
STFT, cross-faded with the halves of a 2048-point Hamming window.  STFT, inverse STFT and the network run on the
babe_hip kernels (csrc/denoiser.hip); device tensors only.
"""
import numpy as np
import torch

from .._lib import check, lib, ptr, stream
from ..networks.denoiser import _register


class DenoiserPrepass:
    def __init__(self, denoiser, dargs, device="cuda"):
        """denoiser: babe_amd.networks.denoiser.MultiStage_denoise (on the GPU); dargs: the `tester.denoiser` config node
        (sample_rate_denoiser, segment_size [s], stft_win_size, stft_hop_size, num_stages)."""
        _register()
        self.denoiser = denoiser
        self.dargs = dargs
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("the denoiser pre-pass runs on the GPU only (no CPU fallback)")
        q = np.arange(2048)
        self.tw4096 = torch.tensor(np.stack([np.cos(2 * np.pi * q / 4096), -np.sin(2 * np.pi * q / 4096)], -1),
                                   dtype=torch.float32, device=self.device).contiguous()

    def _get(self, k):
        return self.dargs[k] if isinstance(self.dargs, dict) else getattr(self.dargs, k)

    def apply_denoiser_model(self, x):
        """x [B, n] -> denoised [B, n + win] (the caller crops), denoise_and_bwe_tester.py:146-165."""
        win, hop = int(self._get("stft_win_size")), int(self._get("stft_hop_size"))
        x = x.contiguous().float()
        B, n = x.shape
        Lp = n + win                                             # one window of zeros appended
        frames = 1 + (Lp - win) // hop
        nb = win // 2 + 1
        xp = torch.zeros(B, Lp, device=self.device)
        xp[:, :n] = x
        X = torch.empty(B, 2, frames, nb, device=self.device)
        L = lib()
        check(L.babe_dn_stft(ptr(xp), xp.stride(0), Lp, ptr(X), B, win, hop, frames, ptr(self.tw4096), stream()), "dn_stft")
        pred = self.denoiser(X)
        if int(self._get("num_stages")) > 1:
            pred = pred[0]
        pred = pred.contiguous()
        Lout = min(win + hop * (frames - 1), Lp)
        ws = torch.empty(B, frames, win, device=self.device)
        y = torch.empty(B, Lout, device=self.device)
        check(L.babe_dn_istft(ptr(pred), ptr(ws), ptr(y), y.stride(0), Lout, B, win, hop, frames, ptr(self.tw4096),
                              stream()), "dn_istft")
        return y

    def apply_denoiser(self, x):
        """x [B, n] -> [B, n], denoise_and_bwe_tester.py:109-142 (the last, zero-padded segment is not faded out; like the
        reference this needs the final remainder to be at most segment_size - 1024 samples)."""
        segment_size = int(self._get("sample_rate_denoiser") * self._get("segment_size"))
        overlapsize = 1024
        n = x.shape[-1]
        window = torch.hamming_window(window_length=2 * overlapsize).to(self.device)
        wl, wr = window[:overlapsize], window[overlapsize:]
        out = torch.zeros_like(x)
        pointer = 0
        while True:
            if pointer + segment_size < n:
                y = self.apply_denoiser_model(x[:, pointer:pointer + segment_size])[:, :segment_size].clone()
                if pointer != 0:
                    y[:, :overlapsize] *= wl
                y[:, segment_size - overlapsize:] *= wr
                out[:, pointer:pointer + segment_size] += y
                pointer += segment_size - overlapsize
            else:
                nl = n - pointer
                seg = torch.zeros(x.shape[0], segment_size, device=self.device)
                seg[:, :nl] = x[:, pointer:]
                y = self.apply_denoiser_model(seg)[:, :segment_size].clone()
                if pointer != 0:
                    y[:, :overlapsize] *= wl
                    if nl > segment_size - overlapsize:
                        raise ValueError("last segment longer than segment_size - 1024 samples (the reference fails here too)")
                out[:, pointer:] += y[:, :nl]
                return out
