"""ctypes binding of libbabe_hip.so (the C-ABI declared in include/babe_hip.h).

There is no CPU fallback: if the library is missing or a call fails this raises.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libbabe_hip.so")
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
    ]


_P, _L, _I, _F = C.c_void_p, C.c_long, C.c_int, C.c_float
_SIGS = {
    "babe_conv2d": [C.POINTER(ConvArgs), _P],
    "babe_conv_pack_weights": [_P, _P, _I, _I, _I, _I, _I, _P],
    "babe_conv2d_bf16": [C.POINTER(ConvArgs), _P, _I, _P],
    "babe_conv2d_wino": [C.POINTER(ConvArgs), _P, _P],
    "babe_conv2d_wino_supported": [C.POINTER(ConvArgs)],
    "babe_conv_pack_weights_wino": [_P, _P, _I, _I, _I, _I, _I, _P],
    "babe_conv2d_wino4": [C.POINTER(ConvArgs), _P, _P],
    "babe_conv2d_wino4_supported": [C.POINTER(ConvArgs)],
    "babe_conv_pack_weights_wino4": [_P, _P, _I, _I, _I, _I, _I, _P],
    "babe_conv_pack_weights_bf16": [_P, _P, _I, _I, _I, _I, _I, _I, _P],
    "babe_gn_partial": [_P, _P, _I, _I, _L, _I, _P],
    "babe_gn_finalize": [_P, _P, _P, _L, _P, _P, _I, _I, _I, _L, _I, _F, _P],
    "babe_scale_gelu": [_P, _P, _P, _I, _I, _L, _P],
    "babe_gn_bwd_partial": [_P, _P, _P, _P, _I, _I, _I, _L, _I, _P],
    "babe_gn_bwd_apply": [_P, _P, _P, _P, _P, _P, _P, _F, _I, _I, _I, _L, _I, _F, _P],
    "babe_resample": [_P, _L, _L, _P, _L, _L, _I, _I, _I, _I, _I, _F, _F, _P],
    "babe_axpby4d": [_P, _L, _L, _P, _L, _L, _I, _I, _I, _I, _F, _F, _P],
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
        L.babe_conv_packed_size_bf16.restype = C.c_long
        L.babe_conv_packed_size_bf16.argtypes = [_I, _I, _I, _I, _I, _I]
        for name, sig in _SIGS.items():
            fn = getattr(L, name)
            fn.argtypes = sig
            fn.restype = C.c_int
        _extra_sigs(L)
        _lib = L
    return _lib


def _extra_sigs(L):
    """Signatures of the CQT / STFT / sampler entry points (registered if present)."""
    from . import _sigs_extra
    _sigs_extra.register(L)


def check(rc, what=""):
    if rc != 0:
        raise BabeHipError(f"{what} failed ({rc}): {lib().babe_last_error().decode()}")


def ptr(t):
    if t is None:
        return None
    assert t.is_cuda, "babe_amd ops need device tensors (no CPU fallback)"
    return t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream
