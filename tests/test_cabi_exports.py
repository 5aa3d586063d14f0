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


def test_unet_plan_structs_have_the_headers_layout(tmp_path):
    """babe_amd/networks/unet_c.py mirrors babe_packed_conv / babe_unet_block / babe_unet_plan_desc with ctypes: sizes and a few
    field offsets must be what a C compiler gives the header (a silent mismatch would hand the library garbage pointers)."""
    import shutil
    import subprocess
    import pytest
    if shutil.which("gcc") is None:
        pytest.skip("no C compiler")
    from babe_amd.networks import unet_c as uc
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "%s"\nint main(void){printf("%%zu %%zu %%zu %%zu %%zu %%zu %%zu\\n",'
                   ' sizeof(babe_packed_conv), sizeof(babe_unet_block), sizeof(babe_unet_plan_desc), offsetof(babe_packed_conv, w_raw),'
                   ' offsetof(babe_unet_block, gamma), offsetof(babe_unet_block, film_gate), offsetof(babe_unet_plan_desc, pyr_conv));return 0;}\n'
                   % os.path.join(ROOT, "include", "babe_hip.h"))
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-o", str(exe), str(src)], check=True)
    want = [int(v) for v in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    got = [ctypes.sizeof(uc.CPackedConv), ctypes.sizeof(uc.CBlock), ctypes.sizeof(uc.CPlanDesc), uc.CPackedConv.w_raw.offset,
           uc.CBlock.gamma.offset, uc.CBlock.film_gate.offset, uc.CPlanDesc.pyr_conv.offset]
    assert got == want, (got, want)
