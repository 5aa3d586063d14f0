"""ctypes binding of the library-side UNet engine (csrc/unet_engine.hip: babe_unet_plan_* / babe_unet_fwd / babe_unet_vjp).

`CUnet(engine)` describes a `UnetEngine`'s packed weights to the library once (a plan handle); `fwd` / `vjp` are then ONE C call per
direction instead of ~1100 op-level calls from Python.  Results are bit-identical to the Python-sequenced engine
(tests/test_gpu_unet_c.py).  fp32 convs only.  Workspace and outputs are torch allocations (the library never allocates)."""
import ctypes as C

import torch

from .._lib import check, lib, ptr, stream

_I, _P = C.c_int, C.c_void_p


class CPackedConv(C.Structure):
    _fields_ = [("Cout", _I), ("Cin", _I), ("KH", _I), ("KW", _I), ("nt", _I), ("splits", _I), ("fwd", _P), ("bwd", _P),
                ("fwd_wino", _P), ("bwd_wino", _P), ("fwd_wino4", _P), ("bwd_wino4", _P), ("fwd_wino45", _P), ("bwd_wino45", _P),
                ("w_raw", _P), ("fwd_wino85", _P), ("bwd_wino85", _P)]


class CBlock(C.Structure):
    _fields_ = [("N", _I), ("nd", _I), ("k53", _I), ("proj_in", CPackedConv), ("res_conv", CPackedConv), ("proj_out", CPackedConv),
                ("H", CPackedConv * 8), ("gamma", _P * 8), ("film_aff", _I * 8), ("film_gate", _I * 8)]


class CPlanDesc(C.Structure):
    _fields_ = [("nocts", _I), ("bpo", _I), ("Ns", _I * 8), ("init_blk", CBlock * 8), ("main_blk", CBlock * 8), ("up_out", CBlock * 8),
                ("up_blk", CBlock * 8), ("mid_blk", CBlock), ("mid_out", CBlock), ("pyr_conv", CPackedConv * 8)]


_registered = False


def _register():
    global _registered
    if _registered:
        return
    L = lib()
    L.babe_unet_plan_create.argtypes = [C.POINTER(CPlanDesc)]
    L.babe_unet_plan_create.restype = _P
    L.babe_unet_plan_destroy.argtypes = [_P]
    L.babe_unet_plan_destroy.restype = None
    L.babe_unet_state_create.argtypes = []
    L.babe_unet_state_create.restype = _P
    L.babe_unet_state_destroy.argtypes = [_P]
    L.babe_unet_state_destroy.restype = None
    L.babe_unet_workspace_bytes.argtypes = [_P, _I, C.POINTER(_I)]
    L.babe_unet_workspace_bytes.restype = C.c_long
    L.babe_unet_fwd.argtypes = [_P, _P, C.POINTER(_P), _P, C.c_long, _I, C.POINTER(_I), _P, C.c_long, C.POINTER(_P), _P]
    L.babe_unet_fwd.restype = _I
    L.babe_unet_vjp.argtypes = [_P, _P, C.POINTER(_P), C.POINTER(_P), _P]
    L.babe_unet_vjp.restype = _I
    L.babe_conv2d_auto.restype = _I
    _registered = True


def _pc(dst, pc):
    """PackedConv -> CPackedConv (absent layer: Cout stays 0)."""
    if pc is None:
        return
    g = lambda k: ptr(getattr(pc, k, None)) if getattr(pc, k, None) is not None else None
    dst.Cout, dst.Cin, dst.KH, dst.KW, dst.nt, dst.splits = pc.Cout, pc.Cin, pc.KH, pc.KW, pc.nt, pc.splits
    dst.fwd, dst.bwd = g("fwd"), g("bwd")
    for k in ("fwd_wino", "bwd_wino", "fwd_wino4", "bwd_wino4", "fwd_wino45", "bwd_wino45", "w_raw", "fwd_wino85", "bwd_wino85"):
        setattr(dst, k, g(k))


def _blk(dst, b):
    dst.N, dst.nd, dst.k53 = b.N, b.nd, int(b.k53)
    _pc(dst.proj_in, b.proj_in)
    _pc(dst.res_conv, b.res_conv)
    _pc(dst.proj_out, b.proj_out)
    for d in range(b.nd):
        _pc(dst.H[d], b.H[d])
        dst.gamma[d] = ptr(b.gamma[d])
        dst.film_aff[d], dst.film_gate[d] = b.film_off[d]


class CUnet:
    """Plan (shared, immutable) + one state and workspace per engine state (= per clip lane)."""

    def __init__(self, eng, plan=None):
        _register()
        assert eng.precision == "f32", "the library-side engine sequences the fp32 network"
        self.eng = eng
        self.n = eng.nocts
        L = lib()
        if plan is None:
            d = CPlanDesc()
            d.nocts, d.bpo = eng.nocts, eng.bpo
            for i, v in enumerate(eng.Ns):
                d.Ns[i] = v
            for i in range(eng.nocts):
                _blk(d.init_blk[i], eng.init_blk[i])
                _blk(d.main_blk[i], eng.main_blk[i])
                _blk(d.up_out[i], eng.up_out[i])
                _blk(d.up_blk[i], eng.up_blk[i])
                _pc(d.pyr_conv[i], eng.pyr_conv[i])
            _blk(d.mid_blk, eng.mid_blk)
            _blk(d.mid_out, eng.mid_out)
            plan = L.babe_unet_plan_create(C.byref(d))
            if not plan:
                raise RuntimeError("babe_unet_plan_create: " + L.babe_last_error().decode())
            self._owns_plan = True
        else:
            self._owns_plan = False
        self.plan = plan
        self.state = L.babe_unet_state_create()
        self.ws = None
        self.key = None

    def clone(self, eng):
        return CUnet(eng, plan=self.plan)

    def __del__(self):
        try:
            L = lib()
            if getattr(self, "state", None):
                L.babe_unet_state_destroy(self.state)
            if getattr(self, "_owns_plan", False) and getattr(self, "plan", None):
                L.babe_unet_plan_destroy(self.plan)
        except Exception:
            pass

    def fwd(self, C_list, film):
        L = lib()
        n = self.n
        B = C_list[0].shape[0]
        Ts = [int(c.shape[-1]) for c in C_list]
        assert all(c.is_contiguous() and c.dtype == torch.float32 for c in C_list) and film.stride(1) == 1
        T_oct = (_I * n)(*Ts)
        key = (B, tuple(Ts))
        if key != self.key:
            nbytes = L.babe_unet_workspace_bytes(self.plan, B, T_oct)
            if nbytes < 0:
                raise RuntimeError("babe_unet_workspace_bytes: " + L.babe_last_error().decode())
            self.ws = torch.empty(nbytes, device=C_list[0].device, dtype=torch.uint8)
            self.key = key
        outs = [torch.empty_like(c) for c in C_list]
        cin = (_P * n)(*[ptr(c) for c in C_list])
        cout = (_P * n)(*[ptr(o) for o in outs])
        self._keep = (C_list, film)              # inputs are read again by nothing after the call, but keep them until the VJP
        check(L.babe_unet_fwd(self.plan, self.state, cin, ptr(film), film.stride(0), B, T_oct, ptr(self.ws), self.ws.numel(), cout,
                              stream()), "unet_fwd")
        return outs

    def vjp(self, gouts):
        L = lib()
        n = self.n
        assert all(g.is_contiguous() and g.dtype == torch.float32 for g in gouts)
        gC = [torch.empty_like(g) for g in gouts]
        gin = (_P * n)(*[ptr(g) for g in gouts])
        gout = (_P * n)(*[ptr(g) for g in gC])
        check(L.babe_unet_vjp(self.plan, self.state, gin, gout, stream()), "unet_vjp")
        self._keep = None
        return gC
