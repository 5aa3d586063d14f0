"""The C-ABI library loads without a GPU and exports every symbol include/babe_hip.h declares."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_exports_match_header():
    import __graft_entry__ as ge
    ge.build()
    hdr = open(os.path.join(ROOT, "include", "babe_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b(babe_[a-z0-9_]+)\s*\(", hdr))
    assert len(names) >= 10
    L = ctypes.CDLL(os.path.join(ROOT, "babe_amd", "libbabe_hip.so"))
    missing = [n for n in sorted(names) if not hasattr(L, n)]
    assert not missing, missing
    L.babe_version.restype = ctypes.c_char_p
    assert b"gfx950" in L.babe_version()
