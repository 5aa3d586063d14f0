"""One tiny invocation of the hot path on cuda:0, checked against the CPU oracle (used by smoke())."""
import torch


def run():
    from oracle import edm as E
    from oracle import unet as UN
    from oracle.nsgt import CQT_nsgt as OracleCQT
    from oracle.sampler import OracleBlindSampler
    from .config import default_args
    from .diff_params.edm import EDM
    from .networks.cqtdiff_plus import Unet_CQT_oct_with_attention, init_state_dict
    from .testing.blind_bwe_sampler import BlindSampler

    Ns, L, fs = [8, 8, 8, 8, 16, 16, 16], 92092, 22050
    args = default_args(sample_rate=fs, audio_len=L, Ns=Ns, T=2, start_sigma=0.05)
    sd = init_state_dict(Ns, args.network.num_dils, seed=0, gate_scale=1.0)
    net = Unet_CQT_oct_with_attention(args, "cuda:0")
    net.load_state_dict(sd)
    g = torch.Generator().manual_seed(0)
    y = 0.1 * torch.randn(1, L, generator=g)
    noises = [torch.randn(1, L, generator=g) for _ in range(3)]
    smp = BlindSampler(net, EDM(args), args)
    it = iter(noises)
    smp._randn = lambda shape, device: next(it).to(device)
    x, fp = smp.predict_blind_bwe(y.cuda())
    torch.cuda.synchronize()
    # oracle (checker only)
    cqt = OracleCQT(7, 64, "oct", ("kaiser", 1), fs, L)
    cfg = dict(num_octs=7, bins_per_oct=64, num_dils=args.network.num_dils)
    onet = lambda xx, cn: UN.unet_forward(sd, cfg, cqt, xx, cn)
    osmp = OracleBlindSampler(onet, cqt, E.EDMParams(0.063, 1e-4, 1.0, 8, Schurn=10), fs=fs, audio_len=L, T=2,
                              start_sigma=0.05)
    xo, fpo = osmp.predict_blind_bwe(y, noises)
    rms = float((x.cpu() - xo).pow(2).mean().sqrt())
    print(f"smoke: blind BWE step on cuda:0, RMS diff vs CPU oracle = {rms:.3e}, fc = {fp[0].tolist()}")
    assert rms < 1e-3, rms
    return rms
