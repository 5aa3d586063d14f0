"""Command line: blind bandwidth extension of one wav file on the MI355X path.

    python -m babe_amd.restore in.wav out_dir [--ckpt weights.pt] [--precision f32|bf16x3|bf16] [--T 35]
                               [--denoise [--denoiser-ckpt denoiser.pt]]

Follows the reference's file-level flow (testing/blind_bwe_tester.py:321-577 formal_test_bwe, blind mode):
read -> resample to exp.sample_rate (:410, babe_amd/resample.py = torchaudio.functional.resample as published; a 48 kHz file is
accepted) -> normalise to sigma_norm std -> segments + blind restoration + cross-fade (babe_amd/testing/long_file.py) -> write wav
(at exp.sample_rate, like the reference) + filter pickle.
Without --ckpt the network has random weights (useful only to exercise the path).
--denoise runs the denoiser pre-pass first (testing/denoise_and_bwe_tester.py:279-289: file rate -> --denoiser-rate (22050) ->
apply_denoiser on the whole file -> exp.sample_rate, then the BWE on its output).
"""
import argparse
import os
import sys

import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("wav")
    ap.add_argument("out_dir")
    ap.add_argument("--ckpt")
    ap.add_argument("--precision", default="f32", choices=["f32", "bf16x3", "bf16"])
    ap.add_argument("--T", type=int, default=35)
    ap.add_argument("--sample-rate", type=int, default=44100)
    ap.add_argument("--audio-len", type=int, default=368368)
    ap.add_argument("--batch", type=int, default=8, help="segments restored per batch")
    ap.add_argument("--sigma-norm", type=float, default=0.1)
    ap.add_argument("--denoise", action="store_true", help="denoiser pre-pass before the bandwidth extension")
    ap.add_argument("--denoiser-ckpt", help="state_dict of networks.denoiser.MultiStage_denoise")
    ap.add_argument("--denoiser-rate", type=int, default=22050, help="tester.denoiser.sample_rate_denoiser")
    a = ap.parse_args()
    from .config import default_args
    from .diff_params.edm import EDM
    from .io import load_checkpoint, read_audio_file, write_audio_file, write_filter_data
    from .networks.cqtdiff_plus import Unet_CQT_oct_with_attention
    from .testing.blind_bwe_sampler import BlindSampler
    from .testing.long_file import restore_file
    args = default_args(sample_rate=a.sample_rate, audio_len=a.audio_len, T=a.T)
    net = Unet_CQT_oct_with_attention(args, "cuda", precision=a.precision)
    if a.ckpt:
        load_checkpoint(net, a.ckpt)
    from .resample import resample
    y, sr = read_audio_file(a.wav)
    y = y.cuda()
    if a.denoise:
        from .networks.denoiser import MultiStage_denoise
        from .testing.denoise import DenoiserPrepass
        dargs = dict(sample_rate_denoiser=a.denoiser_rate, segment_size=5, stft_win_size=1024, stft_hop_size=256, depth=6,
                     num_tfc=3, num_stages=2, use_SAM=True, use_fencoding=True, f_dim=513)   # blind_bwe_denoise*.yaml `denoiser:`
        dnet = MultiStage_denoise(dargs)
        if a.denoiser_ckpt:
            dnet.load_state_dict(torch.load(a.denoiser_ckpt, map_location="cpu"))
        prepass = DenoiserPrepass(dnet.to("cuda"), dargs, "cuda")
    # one rule for both file-level entry points (long_file.rate_plan): the reference's, quirk included - a file already at the
    # denoiser's rate goes to the model unconverted (denoise_and_bwe_tester.py:279-289)
    from .testing.long_file import rate_plan
    plan = rate_plan(sr, a.sample_rate, a.denoiser_rate if a.denoise else None)
    if a.denoise and sr == a.denoiser_rate and sr != a.sample_rate:
        print(f"warning: the file is at the denoiser's rate ({sr} Hz): as in the reference it reaches the {a.sample_rate} Hz model "
              f"unconverted; resample it first if that is not intended", file=sys.stderr)
    for step in plan:
        if step == "denoise":
            y = prepass.apply_denoiser(y.unsqueeze(0))[0]
        else:
            y = resample(y, step[0], step[1])
    std = float(y.std())
    y = y * (a.sigma_norm / std)
    sampler = BlindSampler(net, EDM(args), args, batch_semantics="per_clip")
    out, filt = restore_file(sampler, y, batch_size=a.batch)
    name = os.path.splitext(os.path.basename(a.wav))[0]
    p = write_audio_file(out * (std / a.sigma_norm), a.sample_rate, name, a.out_dir)
    write_filter_data(filt, a.out_dir, name)
    print(p)


if __name__ == "__main__":
    main()
