"""Denoiser pre-pass network on the babe_hip kernels.

Mirrors /root/reference/networks/denoiser.py: `MultiStage_denoise(unet_args)` (:232-321) with the reference's
state_dict names (293 tensors / 72.59 M parameters in the shipped configuration, conf/tester/blind_bwe_denoise*.yaml
`denoiser:`), `forward(X[B,2,T,F]) -> (pred_stage_2, pred_stage_1)` (or pred_stage_1 for one stage).  Inference only,
like the reference's use (`torch.no_grad`, testing/denoise_and_bwe_tester.py:157).  Every convolution, the SAM gate, the
nearest-neighbour up-sampling merge and the frequency encoding run in csrc/denoiser.hip; concatenations are channel
slices of pre-allocated buffers (no copies).  GPU only: there is no CPU fallback.
"""
import ctypes as C
import math

import torch
import torch.nn as nn

from .._lib import check, lib, ptr, stream

NS = [64, 64, 64, 128, 128, 256, 512]        # denoiser.py:243


class DnConvArgs(C.Structure):
    _fields_ = [("in_", C.c_void_p), ("in_bs", C.c_long), ("in_cs", C.c_long), ("IH", C.c_int), ("IW", C.c_int),
                ("bias", C.c_void_p),
                ("out", C.c_void_p), ("out_bs", C.c_long), ("out_cs", C.c_long), ("out_H", C.c_int), ("out_W", C.c_int),
                ("out_hstep", C.c_int), ("out_h0", C.c_int), ("out_wstep", C.c_int), ("out_w0", C.c_int),
                ("res", C.c_void_p), ("res_bs", C.c_long), ("res_cs", C.c_long),
                ("B", C.c_int), ("Cin", C.c_int), ("Cout", C.c_int), ("OH", C.c_int), ("OW", C.c_int),
                ("KH", C.c_int), ("KW", C.c_int), ("stride", C.c_int), ("pad_t", C.c_int), ("pad_l", C.c_int),
                ("pad_mode", C.c_int), ("act", C.c_int), ("ksplit", C.c_int), ("ws", C.c_void_p)]


_registered = False


def _register():
    global _registered
    if _registered:
        return
    L = lib()
    P, I, Lg = C.c_void_p, C.c_int, C.c_long
    sig = {
        "babe_dn_conv2d": [C.POINTER(DnConvArgs), P, P],
        "babe_dn_pack_weights": [P, P, I, I, I, I, I, I, I, P],
        "babe_dn_upsample_add": [P, Lg, Lg, P, Lg, Lg, I, I, I, I, I, I, I, I, P],
        "babe_dn_sam_gate": [P, P, P, Lg, Lg, P, Lg, Lg, I, I, Lg, P],
        "babe_dn_fill_input": [P, P, P, I, I, I, I, P],
        "babe_dn_stft": [P, Lg, I, P, I, I, I, I, P, P],
        "babe_dn_istft": [P, P, P, Lg, I, I, I, I, I, P, P],
    }
    for n, s in sig.items():
        fn = getattr(L, n)
        fn.argtypes = s
        fn.restype = C.c_int
    L.babe_dn_packed_size.argtypes = [I, I, I, I]
    L.babe_dn_packed_size.restype = Lg
    _registered = True


def param_shapes(cfg):
    """Reference state_dict names -> shapes, in the reference's registration order."""
    depth, n = cfg["depth"], cfg["num_tfc"]
    nin = 12 if cfg["use_fencoding"] else 2
    out = {}
    if cfg["use_fencoding"]:
        out["freq_encoding.fembeddings"] = (cfg["f_dim"], 10)

    def conv(name, co, ci, kh, kw):
        out[name + ".weight"] = (co, ci, kh, kw)
        out[name + ".bias"] = (co,)

    def iblock(pre, n0, width):
        for i in range(n):
            conv(f"{pre}.tfc.H.{i}.0", width, n0 + i * width, 3, 3)
        conv(f"{pre}.conv2d_res", width, n0, 1, 1)

    def encoder(pre, n0):
        for i in range(depth):
            iblock(f"{pre}.eblocks.{i}.i_block", n0 if i == 0 else NS[i], NS[i])
            conv(f"{pre}.eblocks.{i}.conv2d_2.0", NS[i + 1], NS[i], 4, 4)
        iblock(f"{pre}.i_block", NS[depth], NS[depth])

    def decoder(pre):
        for i in range(depth):
            out[f"{pre}.dblocks.{i}.tconv_1.0.weight"] = (NS[i + 1], NS[i], 4, 4)       # [in, out, kh, kw]
            out[f"{pre}.dblocks.{i}.tconv_1.0.bias"] = (NS[i],)
            conv(f"{pre}.dblocks.{i}.projection", NS[i], NS[i + 1], 1, 1)
            iblock(f"{pre}.dblocks.{i}.i_block", 2 * NS[i], NS[i])

    conv("conv2d_1.0", NS[0], nin, 7, 7)
    encoder("encoder_s1", NS[0])
    decoder("decoder_s1")
    conv("finalblock.conv2", 2, NS[0], 3, 3)
    if cfg["num_stages"] > 1:
        conv("sam_1.conv1", NS[0], NS[0], 3, 3)
        conv("sam_1.conv2", 2, NS[0], 3, 3)
        conv("sam_1.conv3", NS[0], 2, 3, 3)
        conv("conv2d_2.0", NS[0], nin, 7, 7)
        encoder("encoder_s2", 2 * NS[0])
        decoder("decoder_s2")
    return out


def freq_embeddings(f_dim):
    """AddFreqEncoding (denoiser.py:141-157)."""
    pi = torch.acos(torch.zeros(1)).item() * 2
    n = torch.arange(start=0, end=f_dim) / (f_dim - 1)
    return torch.stack([torch.cos(pi * n)] + [torch.cos(2 ** k * pi * n) for k in range(1, 10)], -1)


def init_state_dict(cfg, seed=0):
    """Synthetic weights (the pretrained checkpoint is not available here): per-tensor generators, variance 1/fan_in."""
    sd = {}
    for idx, (name, shp) in enumerate(param_shapes(cfg).items()):
        if name == "freq_encoding.fembeddings":
            sd[name] = freq_embeddings(cfg["f_dim"])
            continue
        g = torch.Generator().manual_seed(seed * 100003 + idx)
        if name.endswith(".bias"):
            sd[name] = (torch.rand(shp, generator=g) * 2 - 1) * 0.1
        else:
            fan_in = shp[0] * 4 if ".tconv_1." in name else shp[1] * shp[2] * shp[3]
            sd[name] = (torch.rand(shp, generator=g) * 2 - 1) * math.sqrt(3.0 / fan_in)
    return sd


_ws = {}


def _split_k(a, dev):
    """Planes too small to fill the GPU (deep U-Net levels) split the input channels over several workgroups per tile;
    the partial sums pass through a cached workspace."""
    tiles = -(-a.OH // 4) * -(-a.OW // 64) * -(-a.Cout // 64) * a.B
    chunks = -(-a.Cin // 32) * 4 if (a.KH, a.KW) != (1, 1) else -(-a.Cin // 32)
    ks = min(chunks, max(1, 512 // tiles), 32)
    if ks <= 1 or chunks < 8:
        return
    n = ks * a.B * a.Cout * a.OH * a.OW
    w = _ws.get(dev)
    if w is None or w.numel() < n:
        w = _ws[dev] = torch.empty(max(n, 1 << 22), device=dev)
    a.ksplit, a.ws = ks, ptr(w)


def _planes(t):
    assert t.dim() == 4 and t.dtype == torch.float32 and t.is_cuda
    assert t.stride(3) == 1 and t.stride(2) == t.shape[3], f"planes must be contiguous, got strides {t.stride()}"
    return ptr(t), t.stride(0), t.stride(1)


class _Packed:
    """One convolution's weights in the kernel's layout (+ bias)."""

    def __init__(self, w, b, tconv=False):
        _register()
        L = lib()
        w = w.contiguous()
        self.bias = b.contiguous()
        self.tconv = tconv
        if not tconv:
            self.Cout, self.Cin, self.KH, self.KW = w.shape
            self.w = torch.empty(L.babe_dn_packed_size(self.Cout, self.Cin, self.KH, self.KW), device=w.device)
            check(L.babe_dn_pack_weights(ptr(w), ptr(self.w), self.Cout, self.Cin, self.KH, self.KW, 0, 0, 0, stream()),
                  "dn_pack_weights")
        else:
            self.Cin, self.Cout = w.shape[0], w.shape[1]
            assert tuple(w.shape[2:]) == (4, 4)
            self.KH = self.KW = 2
            self.w = {}
            for ph in (0, 1):
                for pw in (0, 1):
                    d = torch.empty(L.babe_dn_packed_size(self.Cout, self.Cin, 2, 2), device=w.device)
                    check(L.babe_dn_pack_weights(ptr(w), ptr(d), self.Cout, self.Cin, 2, 2, 1, ph, pw, stream()),
                          "dn_pack_weights")
                    self.w[(ph, pw)] = d


def _conv(x, pc, out, *, stride=1, pad=None, reflect=True, act=False, res=None):
    """out = [ELU](conv(x) + bias) [+ res]; x/out/res are [B,C,H,W] views with contiguous planes."""
    a = DnConvArgs()
    B, Cin, IH, IW = x.shape
    assert Cin == pc.Cin and out.shape[1] == pc.Cout, (x.shape, out.shape, pc.Cin, pc.Cout)
    a.in_, a.in_bs, a.in_cs = _planes(x)
    a.IH, a.IW = IH, IW
    a.bias = ptr(pc.bias)
    a.out, a.out_bs, a.out_cs = _planes(out)
    a.out_H, a.out_W = out.shape[2], out.shape[3]
    a.out_hstep = a.out_wstep = 1
    if res is not None:
        assert res.shape == out.shape
        a.res, a.res_bs, a.res_cs = _planes(res)
    pt, pl = pad if pad is not None else ((pc.KH - 1) // 2, (pc.KW - 1) // 2)
    a.B, a.Cin, a.Cout, a.OH, a.OW = B, Cin, pc.Cout, out.shape[2], out.shape[3]
    a.KH, a.KW, a.stride, a.pad_t, a.pad_l = pc.KH, pc.KW, stride, pt, pl
    a.pad_mode, a.act = (1 if reflect else 0), (1 if act else 0)
    _split_k(a, x.device)
    check(lib().babe_dn_conv2d(C.byref(a), ptr(pc.w), stream()), "dn_conv2d")


def _tconv(x, pc, out, crop_h, crop_w):
    """out[b,co,h,w] = ELU(conv_transpose2d(x, 4x4, stride 2) + bias)[h + crop_h, w + crop_w] (denoiser.py:383-388)."""
    B, Cin, IH, IW = x.shape
    for (ph, pw), wq in pc.w.items():
        a = DnConvArgs()
        a.in_, a.in_bs, a.in_cs = _planes(x)
        a.IH, a.IW = IH, IW
        a.bias = ptr(pc.bias)
        a.out, a.out_bs, a.out_cs = _planes(out)
        a.out_H, a.out_W = out.shape[2], out.shape[3]
        a.out_hstep, a.out_h0, a.out_wstep, a.out_w0 = 2, ph - crop_h, 2, pw - crop_w
        a.B, a.Cin, a.Cout, a.OH, a.OW = B, Cin, pc.Cout, IH + 1, IW + 1
        a.KH, a.KW, a.stride, a.pad_t, a.pad_l, a.pad_mode, a.act = 2, 2, 1, 1, 1, 0, 1
        _split_k(a, x.device)
        check(lib().babe_dn_conv2d(C.byref(a), ptr(wq), stream()), "dn_conv2d(tconv)")


class DenoiserEngine:
    """Launch sequence of MultiStage_denoise.forward over packed weights; buffers are cached per input shape."""

    def __init__(self, sd, cfg):
        _register()
        self.cfg = cfg
        self.depth, self.n = cfg["depth"], cfg["num_tfc"]
        self.dev = next(iter(sd.values())).device
        self.pc = {}
        for name in sd:
            if name.endswith(".weight"):
                base = name[:-7]
                self.pc[base] = _Packed(sd[name], sd[base + ".bias"], tconv=".tconv_1." in name)
        self.femb = sd["freq_encoding.fembeddings"].contiguous() if cfg["use_fencoding"] else None
        self._bufs = {}

    # ---- buffers: one dense buffer per I_Block, [B, n0 + (n-1)*width, H, W]; the block input is its tail
    def _alloc(self, B, T, F):
        key = (B, T, F)
        if key in self._bufs:
            return self._bufs[key]
        depth, n = self.depth, self.n
        sizes = [(T, F)]
        for _ in range(depth):
            h, w = sizes[-1]
            sizes.append((h // 2 + 1, w // 2 + 1))               # 4x4 stride-2 conv on a 2-sample reflect pad
        mk = lambda c, hw: torch.empty(B, c, hw[0], hw[1], device=self.dev)
        bufs = {"sizes": sizes, "xin": mk(12 if self.cfg["use_fencoding"] else 2, sizes[0])}
        for s in range(1, self.cfg["num_stages"] + 1):
            n0 = NS[0] if s == 1 else 2 * NS[0]
            st = {"enc": [], "dec": [], "dec_out": [], "low": []}
            for i in range(depth):
                st["enc"].append(mk((n0 if i == 0 else NS[i]) + (n - 1) * NS[i], sizes[i]))
                st["dec"].append(mk(2 * NS[i] + (n - 1) * NS[i], sizes[i]))
                st["dec_out"].append(mk(NS[i], sizes[i]))
                st["low"].append(mk(NS[i], sizes[i + 1]))
            st["bottom"] = mk(NS[depth] + (n - 1) * NS[depth], sizes[depth])
            st["bottom_out"] = mk(NS[depth], sizes[depth])
            bufs[s] = st
        bufs["pred1"] = mk(2, sizes[0])
        bufs["pred2"] = mk(2, sizes[0])
        if self.cfg["num_stages"] > 1:
            bufs["sam_x1"] = mk(NS[0], sizes[0])
            bufs["sam_m"] = mk(NS[0], sizes[0])
        self._bufs[key] = bufs
        return bufs

    def _tail(self, buf, n0):
        return buf[:, buf.shape[1] - n0:]

    def _iblock(self, pre, buf, n0, width, out):
        """I_Block (denoiser.py:342-346) on a dense buffer whose last n0 channels hold the input."""
        n = self.n
        ctot = buf.shape[1]
        x = buf[:, ctot - n0:]
        _conv(x, self.pc[f"{pre}.conv2d_res"], out)                                  # out = proj(input)
        for i in range(n):
            src = buf[:, ctot - n0 - i * width:]
            if i == n - 1:
                _conv(src, self.pc[f"{pre}.tfc.H.{i}.0"], out, act=True, res=out)    # out = ELU(conv) + proj(input)
            else:
                _conv(src, self.pc[f"{pre}.tfc.H.{i}.0"], buf[:, ctot - n0 - (i + 1) * width: ctot - n0 - i * width],
                      act=True)

    def _stage(self, s, bufs, first_n0):
        st, depth, sizes = bufs[s], self.depth, bufs["sizes"]
        enc, dec = f"encoder_s{s}", f"decoder_s{s}"
        for i in range(depth):
            n0 = first_n0 if i == 0 else NS[i]
            bridge = st["dec"][i][:, st["dec"][i].shape[1] - NS[i]:]                 # second half of the D_Block input
            self._iblock(f"{enc}.eblocks.{i}.i_block", st["enc"][i], n0, NS[i], bridge)
            nxt = st["enc"][i + 1] if i + 1 < depth else st["bottom"]
            _conv(bridge, self.pc[f"{enc}.eblocks.{i}.conv2d_2.0"], self._tail(nxt, NS[i + 1]), stride=2, pad=(2, 2), act=True)
        self._iblock(f"{enc}.i_block", st["bottom"], NS[depth], NS[depth], st["bottom_out"])
        x = st["bottom_out"]
        for i in range(depth - 1, -1, -1):
            buf = st["dec"][i]
            ctot = buf.shape[1]
            y = buf[:, ctot - 2 * NS[i]: ctot - NS[i]]                               # first half of the D_Block input
            H, W = sizes[i]
            h, w = x.shape[2], x.shape[3]
            d2h, d2w = (2 * h - H) // 2, (2 * w - W) // 2
            _tconv(x, self.pc[f"{dec}.dblocks.{i}.tconv_1.0"], y, 1 + d2h, 1 + d2w)
            _conv(x, self.pc[f"{dec}.dblocks.{i}.projection"], st["low"][i])         # proj(upsample(x)) == upsample(proj(x))
            p, bs, cs = _planes(y)
            lp, lbs, lcs = _planes(st["low"][i])
            check(lib().babe_dn_upsample_add(p, bs, cs, lp, lbs, lcs, x.shape[0], NS[i], H, W, h, w, d2h, d2w, stream()),
                  "dn_upsample_add")
            self._iblock(f"{dec}.dblocks.{i}.i_block", buf, 2 * NS[i], NS[i], st["dec_out"][i])
            x = st["dec_out"][i]
        return x

    def forward(self, X):
        assert X.dim() == 4 and X.shape[1] == 2 and X.is_cuda, "X must be a device tensor [B, 2, T, F]"
        X = X.contiguous().float()
        B, _, T, F = X.shape
        cfg = self.cfg
        if cfg["use_fencoding"]:
            assert F == cfg["f_dim"], f"F={F} but the frequency encoding was built for f_dim={cfg['f_dim']}"
        bufs = self._alloc(B, T, F)
        L = lib()
        check(L.babe_dn_fill_input(ptr(X), ptr(self.femb), ptr(bufs["xin"]), B, T, F, 10 if cfg["use_fencoding"] else 0,
                                   stream()), "dn_fill_input")
        xin = bufs["xin"]
        _conv(xin, self.pc["conv2d_1.0"], self._tail(bufs[1]["enc"][0], NS[0]), act=True)
        feats1 = self._stage(1, bufs, NS[0])
        if cfg["num_stages"] <= 1:
            _conv(feats1, self.pc["finalblock.conv2"], bufs["pred1"])
            return bufs["pred1"].clone()
        # SAM (denoiser.py:117-132)
        _conv(feats1, self.pc["sam_1.conv1"], bufs["sam_x1"])
        _conv(feats1, self.pc["sam_1.conv2"], bufs["pred1"], res=X)
        _conv(bufs["pred1"], self.pc["sam_1.conv3"], bufs["sam_m"])
        tail2 = self._tail(bufs[2]["enc"][0], 2 * NS[0])
        _conv(xin, self.pc["conv2d_2.0"], tail2[:, :NS[0]], act=True)
        second = tail2[:, NS[0]:]
        if cfg["use_SAM"]:
            fp, fbs, fcs = _planes(feats1)
            op, obs, ocs = _planes(second)
            check(L.babe_dn_sam_gate(ptr(bufs["sam_x1"]), ptr(bufs["sam_m"]), fp, fbs, fcs, op, obs, ocs, B, NS[0],
                                     T * F, stream()), "dn_sam_gate")
        else:
            second.copy_(feats1)
        feats2 = self._stage(2, bufs, 2 * NS[0])
        _conv(feats2, self.pc["finalblock.conv2"], bufs["pred2"])
        return bufs["pred2"].clone(), bufs["pred1"].clone()


class _Node(nn.Module):
    pass


class MultiStage_denoise(nn.Module):
    """Drop-in for networks.denoiser.MultiStage_denoise(unet_args) (reference parameter names; `.to('cuda')` before use)."""

    def __init__(self, unet_args=None, device=None):
        super().__init__()
        g = (lambda k, d=None: unet_args[k] if k in unet_args else d) if isinstance(unet_args, dict) else \
            (lambda k, d=None: getattr(unet_args, k, d))
        self.cfg = dict(depth=int(g("depth")), num_tfc=int(g("num_tfc")), num_stages=int(g("num_stages")),
                        use_SAM=bool(g("use_SAM")), use_fencoding=bool(g("use_fencoding")), f_dim=int(g("f_dim", 513)))
        for flag in ("use_csff", "use_cam", "use_fam", "use_tdf", "use_alttdfs"):
            if g(flag, False):
                raise NotImplementedError(f"denoiser option {flag}=True (unused by the reference network, off in every config)")
        assert 1 <= self.cfg["depth"] <= 6 and self.cfg["num_stages"] in (1, 2)
        for key, t in init_state_dict(self.cfg).items():
            node = self
            parts = key.split(".")
            for p in parts[:-1]:
                if p not in node._modules:
                    node.add_module(p, _Node())
                node = node._modules[p]
            node.register_parameter(parts[-1], nn.Parameter(t, requires_grad=False))
        self._engine = None
        self.register_load_state_dict_post_hook(lambda m, k: setattr(m, "_engine", None))
        if device is not None:
            self.to(device)

    def _apply(self, fn, *a, **k):
        self._engine = None
        return super()._apply(fn, *a, **k)

    def engine(self):
        if self._engine is None:
            sd = {k: v.detach().float().contiguous() for k, v in self.state_dict().items()}
            dev = next(iter(sd.values())).device
            if dev.type != "cuda":
                raise RuntimeError("babe_amd networks run on the GPU only (no CPU fallback); call .to('cuda') first")
            self._engine = DenoiserEngine(sd, self.cfg)
        return self._engine

    def forward(self, inputs):
        return self.engine().forward(inputs)
