"""ctypes signatures for the CQT / STFT / sampler entry points of libbabe_hip.so."""
import ctypes as C

_P, _L, _I, _F, _D = C.c_void_p, C.c_long, C.c_int, C.c_float, C.c_double

SIGS = {}


def register(L):
    for name, sig in SIGS.items():
        if hasattr(L, name):
            fn = getattr(L, name)
            fn.argtypes = sig
            fn.restype = C.c_int
