"""The HIP library must not contain packed-fp32 instructions (v_pk_fma_f32, v_pk_mul_f32, v_pk_add_f32, v_pk_mov_b32):
kernels that contain them return wrong sums when they run on a second stream beside the bf16 conv (DESIGN.md 8,
babe_amd/build.py, profiles/r03_coresidency_probe.txt).  Disassembles the device code of the built library - no GPU needed."""
import glob
import os
import shutil
import subprocess

import pytest

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


@pytest.mark.skipif(not os.path.exists(OBJDUMP), reason="llvm-objdump not available")
def test_library_has_no_packed_fp32_instructions(tmp_path):
    from babe_amd.build import COMMON_FLAGS, build
    assert "-fno-slp-vectorize" in COMMON_FLAGS and "-packed-fp32-ops" in COMMON_FLAGS
    so = build(verbose=False)
    local = tmp_path / "libbabe_hip.so"
    shutil.copy(so, local)
    subprocess.run([OBJDUMP, "--offloading", str(local)], check=True, capture_output=True)     # extracts the code objects
    bundles = [p for p in glob.glob(str(local) + ".*") if "amdgcn" in p]
    assert bundles, "no device code objects found in the library"
    packed, mfma = [], 0
    for b in bundles:
        asm = subprocess.run([OBJDUMP, "-d", b], check=True, capture_output=True, text=True).stdout
        mfma += asm.count("v_mfma_")
        packed += [l.strip() for l in asm.splitlines() if "v_pk_" in l and "_f32" in l or "v_pk_mov_b32" in l]
    assert mfma > 1000, "disassembly found no MFMA instructions: the check did not see the kernels"
    assert not packed, f"{len(packed)} packed-fp32 instructions in the library, e.g. {packed[:3]}"
