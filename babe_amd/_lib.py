"""ctypes binding of libbabe_hip.so (the C-ABI declared in include/babe_hip.h).

There is no CPU fallback: if the library is missing or a call fails this raises.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get("BABE_HIP_LIB") or os.path.join(_HERE, "libbabe_hip.so")
_lib = None


class BabeHipError(RuntimeError):
    pass


class ConvArgs(C.Structure):
    _fields_ = [
        ("in_", C.c_void_p), ("in_bs", C.c_long), ("in_cs", C.c_long),
        ("in2", C.c_void_p), ("in2_bs", C.c_long), ("in2_cs", C.c_long), ("cin_split", C.c_int),
        ("w_packed", C.c_void_p),
        ("out", C.c_void_p), ("out_bs", C.c_long), ("out_cs", C.c_long),
        ("res", C.c_void_p), ("res_bs", C.c_long), ("res_cs", C.c_long),
        ("in_scale", C.c_void_p), ("oscale", C.c_void_p),
        ("alpha", C.c_float), ("rbeta", C.c_float),
        ("B", C.c_int), ("Cin", C.c_int), ("Cout", C.c_int), ("F", C.c_int), ("T", C.c_int),
        ("KH", C.c_int), ("KW", C.c_int), ("dil", C.c_int),
        # optional reduction fused into the F(4,5) kernels' epilogue (include/babe_hip.h; zero = off)
        ("stat_mode", C.c_int), ("stat_cg", C.c_int), ("stat_x", C.c_void_p), ("stat_scale", C.c_void_p), ("stat_part", C.c_void_p),
    ]


_P, _L, _I, _F = C.c_void_p, C.c_long, C.c_int, C.c_float
_SIGS = {
    "babe_conv2d": [C.POINTER(ConvArgs), _P],
    "babe_conv_pack_weights": [_P, _P, _I, _I, _I, _I, _I, _P],
    "babe_conv2d_nt": [C.POINTER(ConvArgs), _I, _P],
    "babe_conv2d_fewco": [C.POINTER(ConvArgs), _P, _I, _P],
    "babe_conv2d_fewco_supported": [C.POINTER(ConvArgs)],
    "babe_conv_pack_weights_nt": [_P, _P, _I, _I, _I, _I, _I, _I, _P],
    "babe_conv2d_bf16": [C.POINTER(ConvArgs), _P, _I, _P],
    "babe_conv2d_wino": [C.POINTER(ConvArgs), _P, _P],
    "babe_conv2d_wino_supported": [C.POINTER(ConvArgs)],
    "babe_conv_pack_weights_wino": [_P, _P, _I, _I, _I, _I, _I, _P],
    "babe_conv2d_wino4": [C.POINTER(ConvArgs), _P, _P],
    "babe_conv2d_wino4_supported": [C.POINTER(ConvArgs)],
    "babe_conv_pack_weights_wino4": [_P, _P, _I, _I, _I, _I, _I, _P],
    "babe_conv2d_wino45": [C.POINTER(ConvArgs), _P, _P],
    "babe_conv2d_wino45_supported": [C.POINTER(ConvArgs)],
    "babe_conv2d_wino45_preferred": [C.POINTER(ConvArgs)],
    "babe_conv2d_wino85": [C.POINTER(ConvArgs), _P, _P],
    "babe_conv2d_wino85_supported": [C.POINTER(ConvArgs)],
    "babe_conv2d_wino85_preferred": [C.POINTER(ConvArgs)],
    "babe_conv2d_wino85_stat_slots": [C.POINTER(ConvArgs)],
    "babe_conv_pack_weights_wino45": [_P, _P, _I, _I, _I, _I, _I, _P],
    "babe_conv_pack_weights_bf16": [_P, _P, _I, _I, _I, _I, _I, _I, _P],
    "babe_gn_partial": [_P, _P, _I, _I, _L, _I, _P],
    "babe_scale_gelu_fin": [_P, _P, _P, _P, _L, _P, _P, _P, _I, _I, _I, _L, _I, _F, _P],
    "babe_gn_stats": [_P, _P, _P, _P, _P, _L, _P, _P, _I, _I, _I, _L, _I, _F, _P],
    "babe_gn_finalize": [_P, _P, _P, _L, _P, _P, _I, _I, _I, _L, _I, _F, _P],
    "babe_scale_gelu": [_P, _P, _P, _I, _I, _L, _P],
    "babe_units_size": [_I, _I, _I],
    "babe_scale_gelu_units": [_P, _P, _P, _I, _I, _I, _I, _P],
    "babe_conv2d_bf16_units_supported": [C.POINTER(ConvArgs)],
    "babe_conv2d_bf16_units": [C.POINTER(ConvArgs), _P, _P],
    "babe_gn_bwd_partial": [_P, _P, _P, _P, _I, _I, _I, _L, _I, _P],
    "babe_gn_bwd_apply": [_P, _P, _P, _P, _P, _P, _P, _F, _I, _I, _I, _L, _I, _F, _P],
    "babe_gn_bwd_apply_merge": [_P, _P, _P, _P, _P, _P, _P, _F, _I, _I, _I, _L, _I, _F, _P, _P, _F, _F],
    "babe_resample": [_P, _L, _L, _P, _L, _L, _I, _I, _I, _I, _I, _F, _F, _P],
    "babe_resample_res": [_P, _L, _L, _P, _L, _L, _P, _L, _L, _I, _I, _I, _I, _I, _F, _F, _P],
    "babe_axpby4d": [_P, _L, _L, _P, _L, _L, _I, _I, _I, _I, _F, _F, _P],
    "babe_axpby2_4d": [_P, _L, _L, _P, _L, _L, _P, _L, _L, _I, _I, _I, _I, _F, _F, _P],
    "babe_linear": [_P, _P, _P, _P, _I, _I, _I, _I, _P],
    "babe_rff": [_P, _P, _P, _I, _I, _P],
}


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise BabeHipError(
                f"{_LIB_PATH} not found: build it with `python -m babe_amd.build` (there is no CPU fallback)")
        L = C.CDLL(_LIB_PATH)
        L.babe_last_error.restype = C.c_char_p
        L.babe_version.restype = C.c_char_p
        L.babe_conv_packed_size.restype = C.c_long
        L.babe_conv_packed_size.argtypes = [_I, _I, _I, _I, _I]
        L.babe_conv_packed_size_wino.restype = C.c_long
        L.babe_conv_packed_size_wino.argtypes = [_I, _I, _I, _I]
        L.babe_conv_packed_size_wino4.restype = C.c_long
        L.babe_conv_packed_size_wino4.argtypes = [_I, _I, _I, _I]
        L.babe_conv_packed_size_wino45.restype = C.c_long
        L.babe_conv_packed_size_wino45.argtypes = [_I, _I, _I]
        L.babe_conv_packed_size_wino85.restype = C.c_long
        L.babe_conv_packed_size_wino85.argtypes = [_I, _I, _I]
        L.babe_conv_pack_weights_wino85.argtypes = [_P, _P, _I, _I, _I, _I, _I, _P]
        L.babe_conv_packed_size_bf16.restype = C.c_long
        L.babe_conv_packed_size_bf16.argtypes = [_I, _I, _I, _I, _I, _I]
        for name, sig in _SIGS.items():
            fn = getattr(L, name)
            fn.argtypes = sig
            fn.restype = C.c_int
        L.babe_units_size.restype = C.c_long
        L.babe_prof_nslots.restype = C.c_int
        L.babe_prof_slot_name.restype = C.c_char_p
        L.babe_prof_slot_name.argtypes = [_I]
        L.babe_prof_enable.argtypes = [_I]
        L.babe_prof_conv_slot.argtypes = [_I]
        L.babe_prof_read.argtypes = [_P, _P, _P, _P, _P]
        L.babe_prof_dispatch_counts.argtypes = [_P, _I]
        L.babe_prof_timeline.restype = C.c_long
        L.babe_prof_timeline.argtypes = [_P, _P, _P, _P, _P, C.c_long]
        L.babe_prof_pending.restype = C.c_long
        _lib = L
    return _lib


def prof_slot_names():
    L = lib()
    return [L.babe_prof_slot_name(i).decode() for i in range(L.babe_prof_nslots())]


_prof_on = False

def prof_enable(on):
    """Measurement hook (include/babe_hip.h): HIP-event timing of every launch, tallied per kernel slot."""
    global _prof_on
    _prof_on = bool(on)
    lib().babe_prof_enable(int(bool(on)))


def prof_enabled():
    """True while the measurement hook is on.  Asks the LIBRARY (babe_prof_enable(-1) = query): another binding may have
    switched it on."""
    return bool(_prof_on or (lib().babe_prof_enable(-1) == 1))


def prof_read():
    """{slot name: dict(ms, bytes, flops, exec_flops, launches)} since the last read (waits for the GPU); resets."""
    L = lib()
    n = L.babe_prof_nslots()
    D, Lg = (C.c_double * n), (C.c_long * n)
    ms, by, fl, ex, nl = D(), D(), D(), D(), Lg()
    check(L.babe_prof_read(ms, by, fl, ex, nl), "prof_read")
    return {name: dict(ms=ms[i], bytes=by[i], flops=fl[i], exec_flops=ex[i], launches=nl[i])
            for i, name in enumerate(prof_slot_names())}


def prof_timeline():
    """Per-launch records pending since the last prof_read() - call BEFORE it: dict of numpy arrays t0_ms, t1_ms (relative to
    the first record), slot (index into prof_slot_names()), lane (stream index), flops.  Waits for the GPU."""
    import numpy as np
    L = lib()
    n = int(L.babe_prof_pending())
    t0, t1, fl = np.zeros(n), np.zeros(n), np.zeros(n)
    sl, ln = np.zeros(n, np.int32), np.zeros(n, np.int32)
    as_p = lambda a: a.ctypes.data_as(C.c_void_p)
    m = L.babe_prof_timeline(as_p(t0), as_p(t1), as_p(sl), as_p(ln), as_p(fl), n)
    if m < 0:
        raise BabeHipError(f"prof_timeline failed ({m}): {L.babe_last_error().decode()}")
    return dict(t0_ms=t0[:m], t1_ms=t1[:m], slot=sl[:m], lane=ln[:m], flops=fl[:m])


def dispatch_counts(reset=False):
    """Always-on launch counters per slot: which kernel each conv call really took (wino4 / wino2 / direct / bf16)."""
    L = lib()
    n = L.babe_prof_nslots()
    cnt = (C.c_long * n)()
    L.babe_prof_dispatch_counts(cnt, int(reset))
    return {name: cnt[i] for i, name in enumerate(prof_slot_names())}


def check(rc, what=""):
    if rc != 0:
        raise BabeHipError(f"{what} failed ({rc}): {lib().babe_last_error().decode()}")


def ptr(t):
    if t is None:
        return None
    assert t.is_cuda, "babe_amd ops need device tensors (no CPU fallback)"
    return t.data_ptr()


def stream(t=None):
    """HIP stream the next launch goes to: the current stream OF THE DEVICE the operands live on (`t`: a tensor or a
    torch.device).  The library never calls hipSetDevice; launching device-1 pointers on device 0's stream faults, so
    multi-device hosts must either pass `t` or run under torch.cuda.device(...) as the network/sampler entry points do."""
    if t is None:
        return torch.cuda.current_stream().cuda_stream
    return torch.cuda.current_stream(t.device if torch.is_tensor(t) else t).cuda_stream
