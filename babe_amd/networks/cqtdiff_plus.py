"""CQTDiff+ denoiser (HIP-backed), drop-in for the reference's
``networks.cqtdiff+.Unet_CQT_oct_with_attention`` (/root/reference/networks/cqtdiff+.py:583-845).

Same constructor ``(args, device)``, same ``state_dict`` key names / shapes so the
reference's checkpoints load with ``load_state_dict(state['ema'])``; ``net(x[B,L], cnoise[B,1]) ->
[B,L]``; ``net.CQTransform.apply_hpf_DC``.  The forward is autograd-transparent w.r.t. ``x`` (a
``torch.autograd.Function`` whose backward is the hand-wired HIP input-VJP), so the reference's own
sampler code - which calls ``torch.autograd.grad`` through the model - runs on it unchanged.
There is no CPU path: device must be a GPU and libbabe_hip.so must be present.

Not implemented (disabled in every blind-BWE config, conf/network/cqtdiff+.yaml:8,23):
frequency encodings (``use_fencoding``) and time-attention layers.
"""
import math
import os

import numpy as np
import torch
import torch.nn as nn

from ..cqt import CQT_nsgt
from .unet_engine import UnetEngine


def param_specs(Ns, num_dils, emb_dim=256, num_octs=7):
    """[(key, shape, init)] for every parameter/buffer of the reference module, init in {'w','gate','ones','rff','buf'}."""
    out = [("embedding.RFF_freq", (1, 32), "rff")]
    for i, (o, k) in enumerate([(128, 64), (256, 128), (emb_dim, 256)]):
        out += [(f"embedding.MLP.{i}.weight", (o, k), "w"), (f"embedding.MLP.{i}.bias", (o,), "zero")]
    out += [("downsamplerT.kernel", (8,), "buf"), ("upsamplerT.kernel", (8,), "buf")]

    def block(p, dim, dim_out, nd, k, after):
        N = dim if after else dim_out
        r = []
        if after and N != dim_out:
            r.append((p + "proj_out.weight", (dim_out, N, 1, 1), "w"))
        if dim != dim_out:
            r.append((p + "res_conv.weight", (dim_out, dim, 1, 1), "w"))
        if dim != N:
            r.append((p + "proj_in.weight", (N, dim, 1, 1), "w"))
        for d in range(nd):
            r += [(p + f"norm.{d}.gamma", (1, N, 1, 1), "ones"),
                  (p + f"affine.{d}.weight", (N, emb_dim), "w"), (p + f"affine.{d}.bias", (N,), "zero"),
                  (p + f"gate.{d}.weight", (N, emb_dim), "gate"), (p + f"gate.{d}.bias", (N,), "zero"),
                  (p + f"H.{d}.weight", (N, N, k[0], k[1]), "w")]
        return r

    for i in range(num_octs):
        din, dout = (Ns[0], Ns[0]) if i == 0 else (Ns[i - 1], Ns[i])
        out += block(f"downs.{i}.0.", 2, din, 1, (1, 1), False)
        out.append((f"downs.{i}.1.weight", (dout, 2, 5, 3), "w"))
        out += block(f"downs.{i}.2.", din, dout, num_dils[i], (5, 3), False)
    out += block("middle.0.0.", Ns[-1], 2, 1, (1, 1), True)
    out += block("middle.0.1.", Ns[-1], Ns[-1], num_dils[-1], (5, 3), False)
    for ii, i in enumerate(range(num_octs - 1, -1, -1)):
        din, dout = (Ns[0] * 2, Ns[0]) if i == 0 else (Ns[i] * 2, Ns[i - 1])
        out += block(f"ups.{ii}.0.", dout, 2, 1, (1, 1), True)
        out += block(f"ups.{ii}.1.", din, dout, num_dils[i], (5, 3), False)
    return out


CUBIC = [-0.01171875, -0.03515625, 0.11328125, 0.43359375, 0.43359375, 0.11328125, -0.03515625, -0.01171875]


def init_state_dict(Ns, num_dils, emb_dim=256, seed=0, gate_scale=1e-7):
    """Random weights with the reference's init rule (kaiming_uniform * sqrt(1/3); gates * 1e-7, cqtdiff+.py:599-600).
    gate_scale=1 gives O(1) gates (an untrained net with 1e-7 gates has numerically dead residual branches)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for key, shape, kind in param_specs(Ns, num_dils, emb_dim):
        if kind in ("w", "gate"):
            fan_in = int(np.prod(shape[1:]))
            w = math.sqrt(3.0 / fan_in) * (torch.rand(shape, generator=g) * 2 - 1)
            sd[key] = w * (math.sqrt(1 / 3) if kind == "w" else gate_scale)
        elif kind == "zero":
            sd[key] = torch.zeros(shape)
        elif kind == "ones":
            sd[key] = torch.ones(shape)
        elif kind == "rff":
            sd[key] = 16 * torch.randn(shape, generator=g)
        else:
            sd[key] = torch.tensor(CUBIC)
    return sd


class _Node(nn.Module):
    """Anonymous container so that parameter paths reproduce the reference's dotted names."""


def _attach(root, key, tensor, is_buffer):
    parts = key.split(".")
    node = root
    for p in parts[:-1]:
        if p not in node._modules:
            node.add_module(p, _Node())
        node = node._modules[p]
    if is_buffer:
        node.register_buffer(parts[-1], tensor)
    else:
        node.register_parameter(parts[-1], nn.Parameter(tensor, requires_grad=False))


class _UnetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, cnoise, net):
        ctx.net = net
        return net.fwd_nograd(x, cnoise)

    @staticmethod
    def backward(ctx, g):
        return ctx.net.vjp(g.contiguous()), None, None


class Unet_CQT_oct_with_attention(nn.Module):
    def __init__(self, args, device, precision=None):
        super().__init__()
        self.args = args
        nw = args.network
        # conv arithmetic: 'f32' (default, the parity path), 'bf16x3', 'bf16' (csrc/conv_bf16.hip); can also be
        # given as args.network.precision
        self.precision = precision or nw.get("precision", "f32")
        if nw.get("use_fencoding", False):
            raise NotImplementedError("use_fencoding=True (disabled in the blind-BWE configs)")
        if any(nw.get("attention_layers", [0])):
            raise NotImplementedError("attention layers (disabled in the blind-BWE configs)")
        if not nw.get("use_norm", True):
            raise NotImplementedError("use_norm=False")
        self.Ns, self.num_dils = list(nw.Ns), list(nw.num_dils)
        self.num_octs, self.bins_per_oct = nw.cqt.num_octs, nw.cqt.bins_per_oct
        assert self.num_octs == 7 and len(self.Ns) == 7
        self.emb_dim = nw.emb_dim
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("babe_amd networks run on the GPU only (no CPU fallback); use device='cuda'")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        win = ("kaiser", nw.cqt.beta) if nw.cqt.window == "kaiser" else nw.cqt.window
        self.CQTransform = CQT_nsgt(self.num_octs, self.bins_per_oct, mode="oct", window=win,
                                    fs=args.exp.sample_rate, audio_len=args.exp.audio_len, device=self.device)
        for key, t in init_state_dict(self.Ns, self.num_dils, self.emb_dim).items():
            _attach(self, key, t.to(self.device), is_buffer=key.endswith(".kernel"))
        self._engine = None
        self.register_load_state_dict_post_hook(lambda m, k: setattr(m, "_engine", None))

    # ---------------------------------------------------------------- engine
    def engine(self):
        if self._engine is None:
            sd = {k: v.detach().to(self.device, torch.float32).contiguous() for k, v in self.state_dict().items()}
            self._engine = UnetEngine(sd, self.Ns, self.num_dils, self.num_octs, self.bins_per_oct, self.precision)
            self._lanes = None
        return self._engine

    # Batch items are independent, so they run on separate HIP streams (one engine state each, shared packed weights):
    # the HBM-bound passes of one item (GroupNorm / GELU / resampling, (1,1) convs) then overlap the MFMA-bound (5,3)
    # convolutions of another, and the 1.75-round grids of the 7x64-bin layers interleave.  BABE_UNET_STREAMS=1 keeps
    # everything on the caller's stream.
    MAX_LANES = int(os.environ.get("BABE_UNET_STREAMS", "2"))

    def _get_lanes(self, B):
        n = min(B, self.MAX_LANES) if self.concurrent_lanes_ok else 1      # (bf16: one stream if BABE_BF16_LANES=0)
        if n <= 1:
            return None
        if getattr(self, "_lanes", None) is None or len(self._lanes) != n:
            eng = self.engine()
            self._lanes = [(torch.cuda.Stream(device=self.device), eng if i == 0 else eng.clone_state()) for i in range(n)]
        return self._lanes

    def _run_lanes(self, B, fn):
        """fn(engine, b0, b1) -> list of tensors for batch rows [b0, b1); rows are dealt to the lanes in contiguous blocks.
        Returns the per-lane results after making the caller's stream wait for every lane."""
        lanes = self._get_lanes(B)
        main = torch.cuda.current_stream(self.device)
        ready = torch.cuda.Event()
        ready.record(main)
        per = -(-B // len(lanes))
        results = []
        for i, (st, eng) in enumerate(lanes):
            b0, b1 = i * per, min(B, (i + 1) * per)
            if b0 >= b1:
                continue
            with torch.cuda.stream(st):
                st.wait_event(ready)
                out = fn(eng, b0, b1)
                for t in out:
                    t.record_stream(main)              # consumed on the caller's stream after the join below
                done = torch.cuda.Event()
                done.record(st)
            results.append((out, done))
        for _, done in results:
            main.wait_event(done)
        return [r for r, _ in results]

    def _apply(self, fn, *a, **k):
        self._engine = None
        return super()._apply(fn, *a, **k)

    # ---------------------------------------------------------------- raw (no autograd) interface
    supports_lanes = True
    # precision='bf16' and clip lanes.  Kernels that contain the packed-fp32 form v_pk_{mul,add,fma}_f32 ... op_sel:[0,1] read the
    # HIGH word of their second source as 0 while another kernel's waves execute bf16 MFMA on the same CU (reduced in round 4:
    # tools/erratum/pk_opsel_min.hip, DESIGN.md "Co-residency finding").  THIS library contains no packed-fp32 instruction
    # (babe_amd/build.py, tests/test_no_packed_fp32.py), but what ELSE runs beside conv_bf16p is outside its control: PyTorch's own
    # kernels (device-side noise: bench.py's noise_device='cuda'), RCCL, a second process on the same GPU (two ranks on one
    # device), a host application's hipcc-default kernels.  So a bf16 network keeps its batch items on ONE stream by default;
    # BABE_BF16_LANES=1 opts in to two clip lanes (measured +13 %, profiles/r04_bench_bf16_two_lanes.json) for a host that knows
    # what else it launches - and even then a live torch.distributed process group or device-side noise falls back to one
    # stream (BlindSampler asks lanes_ok_for()).  The fp32 network has no bf16 MFMA anywhere and always runs two lanes.
    @property
    def concurrent_lanes_ok(self):
        return self.precision != "bf16" or os.environ.get("BABE_BF16_LANES", "0") == "1"

    def lanes_ok_for(self, noise_device="cpu"):
        """concurrent_lanes_ok narrowed by what the CALLER will launch beside the lanes: with precision='bf16' two lanes also
        need host-side noise (torch.randn on the device is an ATen kernel inside the lane loop) and no live process group."""
        if self.precision != "bf16":
            return True
        if not self.concurrent_lanes_ok:
            return False
        import torch.distributed as dist
        return str(noise_device) == "cpu" and not (dist.is_available() and dist.is_initialized())

    def lane_engine(self, lane):
        """Engine state number `lane` (saved activations + scratch of its own over the shared packed weights): a caller that
        pipelines independent clips on its own streams (BlindSampler) passes lane=k to fwd_nograd / vjp, which then run
        entirely on the CALLER's current stream with that state instead of forking streams themselves."""
        eng = self.engine()
        if getattr(self, "_lane_engines", None) is None or self._lane_engines[0] is not eng:
            self._lane_engines = [eng]
        while len(self._lane_engines) <= lane:
            self._lane_engines.append(eng.clone_state())
        return self._lane_engines[lane]

    def fwd_nograd(self, x, cnoise, lane=None):
        """x [B,L], cnoise [B,1] -> [B,L]; keeps what vjp() needs until the next call (of the same lane)."""
        assert x.device == self.device, f"input on {x.device}, network on {self.device}"
        with torch.cuda.device(self.device):       # every launch below goes to THIS device's current stream
            return self._fwd_nograd(x, cnoise, lane)

    def _fwd_nograd(self, x, cnoise, lane=None):
        if lane is not None:
            eng = self.lane_engine(lane)
            x = x.detach().contiguous().float()
            assert x.shape[-1] == self.CQTransform.Ls
            film = eng.embed(cnoise.detach().reshape(-1, 1).contiguous().float())
            return self.CQTransform.bwd_planar(eng.forward(self.CQTransform.fwd_planar(x), film))
        eng = self.engine()
        x = x.detach().contiguous().float()
        assert x.shape[-1] == self.CQTransform.Ls, "input length must equal exp.audio_len (the CQT is built for it)"
        film = eng.embed(cnoise.detach().reshape(-1, 1).contiguous().float())
        co = self.CQTransform.fwd_planar(x)
        B = x.shape[0]
        if self._get_lanes(B) is None:
            outs = eng.forward(co, film)
        else:
            parts = self._run_lanes(B, lambda e, b0, b1: e.forward([c[b0:b1] for c in co], film[b0:b1]))
            outs = [torch.cat([p[j] for p in parts], 0) for j in range(len(co))]
        return self.CQTransform.bwd_planar(outs)

    def vjp(self, g, lane=None):
        """Gradient of <net(x), g> w.r.t. x for the last fwd_nograd call (of the same lane)."""
        assert g.device == self.device, f"gradient on {g.device}, network on {self.device}"
        with torch.cuda.device(self.device):
            return self._vjp(g, lane)

    def _vjp(self, g, lane=None):
        if lane is not None:
            eng = self.lane_engine(lane)
            return self.CQTransform.fwd_adjoint(eng.vjp(self.CQTransform.bwd_adjoint(g.contiguous())))
        eng = self.engine()
        gouts = self.CQTransform.bwd_adjoint(g.contiguous())
        B = g.shape[0]
        if self._get_lanes(B) is None:
            gC = eng.vjp(gouts)
        else:
            parts = self._run_lanes(B, lambda e, b0, b1: e.vjp([c[b0:b1] for c in gouts]))
            gC = [torch.cat([p[j] for p in parts], 0) for j in range(len(gouts))]
        return self.CQTransform.fwd_adjoint(gC)

    # ---------------------------------------------------------------- nn.Module call
    def forward(self, inputs, sigma):
        if torch.is_grad_enabled() and inputs.requires_grad:
            return _UnetFn.apply(inputs, sigma, self)
        return self.fwd_nograd(inputs, sigma)
