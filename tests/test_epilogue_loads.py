"""Round 6 regression guard, on the disassembly of the built library (no GPU needed): the epilogues of the kernels that own their CU
(one or two workgroups per CU, nothing else resident to cover a stall) must not wait out one memory latency per output element.
hipcc turns `cond ? load : 0` / `if (valid) { load; use; }` per element into one branch per load with its own `s_waitcnt vmcnt(0)`;
the F(4,5) conv (+3.3 % on the whole job when fixed), conv11p's output scales, the bf16 conv (+2.3 %) and the denoiser's conv (+7 %)
all had it.  Counted per kernel: (a) a run of one or two loads followed (counted waits aside) by `s_waitcnt vmcnt(0)`,
(b) `s_waitcnt vmcnt(0)` directly followed by a store."""
import glob
import os
import re
import shutil
import subprocess

import pytest

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
FAMILIES = ("conv_wino85s_kernel", "conv_wino85_kernel", "conv11p_kernel", "conv_bf16p_kernel", "dn_conv_kernel")


def _events(lines):
    ev = []
    for l in lines:
        t = l.split("//")[0].strip()
        op = t.split()[0] if t else ""
        if op.startswith(("global_load", "buffer_load", "flat_load")) and "lds" not in t:
            ev.append("L")
        elif op.startswith(("global_store", "buffer_store", "flat_store")):
            ev.append("S")
        elif op == "s_waitcnt" and "vmcnt(0)" in t:
            ev.append("W")
        elif op == "s_waitcnt" and "vmcnt" in t:
            ev.append("w")
    return "".join(ev)


@pytest.mark.skipif(not os.path.exists(OBJDUMP), reason="llvm-objdump not available")
def test_owning_kernels_do_not_wait_per_loaded_element(tmp_path):
    from babe_amd.build import build
    so = build(verbose=False)
    local = tmp_path / "libbabe_hip.so"
    shutil.copy(so, local)
    subprocess.run([OBJDUMP, "--offloading", str(local)], check=True, capture_output=True)
    bundles = [p for p in glob.glob(str(local) + ".*") if "amdgcn" in p]
    assert bundles, "no device code objects found in the library"
    seen, bad = 0, []
    for b in bundles:
        asm = subprocess.run([OBJDUMP, "-d", "--demangle", b], check=True, capture_output=True, text=True).stdout
        for chunk in asm.split("\n\n"):
            lines = chunk.strip().splitlines()
            m = re.match(r"^[0-9a-f]+ <(.*)>:$", lines[0]) if lines else None
            if not m or not any(f + "<" in m.group(1) for f in FAMILIES):
                continue
            seen += 1
            ev = _events(lines[1:])
            lone = len(re.findall(r"(?<!L)L{1,2}w*W", ev))
            ws = len(re.findall(r"WS", ev))
            if lone > 3 or ws > 8:
                bad.append((m.group(1)[:80], lone, ws))
    assert seen >= 30, f"only {seen} kernels of the guarded families found: the check did not see the library"
    assert not bad, f"kernels with per-element load waits (name, lone load + vmcnt(0), vmcnt(0) + store): {bad[:5]}"
