"""BABE_EVAL_C=1: BlindSampler.evaluate as ONE call into the library (csrc/score_eval.hip, babe_score_eval) - the path a
non-Python host takes: UNet plan + state (csrc/unet_engine.hip), CQT plan (csrc/cqt_plan.hip), the STFT tables and the sampler's
scalars in a babe_eval_desc.  Default configuration only (what predict_blind_bwe / predict_bwe(..., 'fc_A') run with the formal
tester YAMLs); everything else stays with the Python sequencer.  Bit-identical to it: tests/test_gpu_eval_c.py."""
import ctypes as C

import torch

from .._lib import check, lib, ptr, stream
from ..stft import FitCfg

_P, _I, _F, _L = C.c_void_p, C.c_int, C.c_float, C.c_long


class EvalDesc(C.Structure):
    _fields_ = [("unet_plan", _P), ("unet_state", _P), ("cqt_plan", _P), ("L", _I),
                ("rff_freq", _P), ("rff_n", _I), ("emb_W", _P * 3), ("emb_b", _P * 3), ("emb_dim", _I * 4),
                ("film_W", _P), ("film_b", _P), ("film_J", _I),
                ("nfft", _I), ("fs", _F), ("env_inv", _P), ("tw4096", _P), ("K", _I), ("fit", FitCfg),
                ("blind", _I), ("shared", _I), ("hpf", _I), ("xi", _F), ("score_mode", _I), ("audio_len_norm", _F)]


_registered = False


def _register():
    global _registered
    if _registered:
        return
    L = lib()
    L.babe_eval_workspace_bytes.restype = _L
    L.babe_eval_workspace_bytes.argtypes = [C.POINTER(EvalDesc), _I]
    L.babe_score_eval.restype = _I
    L.babe_score_eval.argtypes = [C.POINTER(EvalDesc), _P, _F, _F, _F, _F, _F, _P, _P, _P, _P, _P, _P, _P, _L, _I, _P]
    L.babe_cqt_plan_create.restype = _P
    L.babe_cqt_plan_create.argtypes = [C.c_double, _I, _I, _I, C.c_double]
    _registered = True


def cqt_plan_of(cq):
    """The library-side plan of a CQT_nsgt object: its own (the default), or one made here for an object that sequences the kernels
    itself (BABE_CQT_C=0); None where the library cannot plan the length."""
    _register()
    if getattr(cq, "_plan", None):
        return cq._plan
    if getattr(cq, "_plan_eval", None) is None:
        cq._plan_eval = lib().babe_cqt_plan_create(float(cq.fs), cq.Ls, cq.numocts, cq.binsoct, float(cq.design["beta"])) or 0
    return cq._plan_eval or None


def supported(smp, y, blind):
    """True when this evaluation is the default configuration babe_score_eval sequences."""
    m = smp.model
    return (y is not None and smp.norm == 2 and smp.stft_dist is None and smp.obs_snr is None and not smp.sigma_den
            and smp.ar_mask is None and smp.fir_taps is None and smp.dc is None and getattr(smp, "inpaint_mask", None) is None
            and not (smp._dc_cfg if blind else smp.data_consistency)
            and getattr(m, "precision", None) == "f32" and hasattr(m, "lane_engine") and cqt_plan_of(m.CQTransform) is not None)


class CEval:
    """One lane's descriptor + workspace (the UNet state and the workspace belong to one stream at a time)."""

    def __init__(self, smp, lane):
        _register()
        from ..networks.unet_c import CUnet
        net, st = smp.model, smp._stft
        eng = net.lane_engine(lane or 0)
        root = net.lane_engine(0)
        if getattr(root, "_eval_cunet", None) is None:
            root._eval_cunet = CUnet(root)
        self.cu = root._eval_cunet if eng is root else root._eval_cunet.clone(eng)
        cq = net.CQTransform
        self.cq_plan = cqt_plan_of(cq)
        d = EvalDesc()
        d.unet_plan, d.unet_state, d.cqt_plan, d.L = self.cu.plan, self.cu.state, self.cq_plan, cq.Ls
        d.rff_freq, d.rff_n = ptr(eng.rff_freq), eng.rff_freq.numel()
        d.emb_dim[0] = 2 * d.rff_n
        for i, (W, b) in enumerate(eng.emb_W):
            d.emb_W[i], d.emb_b[i], d.emb_dim[i + 1] = ptr(W), ptr(b), W.shape[0]
        d.film_W, d.film_b, d.film_J = ptr(eng.film_idx.Wcat), ptr(eng.film_idx.bcat), eng.film_idx.J
        d.nfft, d.fs, d.env_inv, d.tw4096 = st.nfft, st.fs, ptr(st.env_inv), ptr(st.tw4096)
        d.fit = smp.fit_cfg
        d.shared = int(smp.batch_semantics == "reference")
        d.hpf = int(bool(smp.args.tester.filter_out_cqt_DC_Nyq))
        d.xi, d.score_mode, d.audio_len_norm = float(smp.xi), int(smp.SCORE_MODE), float(smp.args.exp.audio_len)
        self.desc, self.ws, self.key = d, None, None
        self._keep = (eng, st, cq)

    def __call__(self, smp, x, t, y, specY, filter_params, blind):
        d = self.desc
        B, L = x.shape
        d.K, d.blind = int(filter_params.shape[-1]), int(bool(blind))
        if self.key != B:
            nbytes = lib().babe_eval_workspace_bytes(C.byref(d), B)
            if nbytes < 0:
                check(-1, "eval_workspace_bytes")
            self.ws = torch.empty(nbytes, device=x.device, dtype=torch.uint8)
            self.key = B
        dp = smp.diff_params
        s = torch.as_tensor(t, dtype=torch.float32)
        cskip, cout, cin, cnoise = float(dp.cskip(s)), float(dp.cout(s)), float(dp.cin(s)), float(dp.cnoise(s))
        p = filter_params.clone() if blind else filter_params.contiguous()
        dd, x_den = torch.empty_like(x), torch.empty_like(x)
        nit = torch.empty(p.shape[0], dtype=torch.int32, device=x.device)
        check(lib().babe_score_eval(C.byref(d), ptr(x), float(t), cskip, cout, cin, cnoise, ptr(y), ptr(specY), ptr(p), ptr(dd),
                                    ptr(x_den), ptr(nit), ptr(self.ws), self.ws.numel(), B, stream()), "score_eval")
        if blind:
            smp.last_n_iter = nit
        return dd, x_den, p
