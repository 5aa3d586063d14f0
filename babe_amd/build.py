"""Build libbabe_hip.so (gfx950) in-tree with hipcc.  Cross-compiles without a GPU."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libbabe_hip.so")
SOURCES = ["misc.hip", "conv.hip", "conv11p.hip", "conv_fewco.hip", "conv_bf16.hip", "conv_bf16p.hip", "conv_wino.hip", "conv_wino4.hip", "conv_wino4p.hip", "conv_wino45.hip", "norm.hip", "resample.hip", "resample_sinc.hip", "cqt.hip", "fft_mixed.hip", "unet_engine.hip", "stft.hip", "sampler.hip", "denoiser.hip"]


# Every source is built WITHOUT packed-fp32 instructions: no SLP vectoriser (-fno-slp-vectorize) and the target feature
# packed-fp32-ops switched off (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 / v_pk_mov_b32 are never selected, also not for
# explicit float2 / float4 vector arithmetic).  Two reasons, both measured in round 3:
#  * speed of the fp32-MFMA kernels: next to fp32 MFMAs every vector instruction costs matrix-pipe time
#    (tools/mfma_valu_coexec.hip: a v_pk_fma_f32 costs 1.6x a v_fma_f32, a v_mov as much as a v_fma), and the vectoriser pays
#    for its packed arithmetic with register moves: conv_wino45 207 instead of 245 vector instructions per two K-slabs;
#  * CORRECTNESS next to the bf16 conv: kernels that contain packed-fp32 instructions (conv_fewco, the FFT's twiddle /
#    transpose, epilogues with float4 arithmetic, ...) return slightly wrong sums - one of a thread's four outputs, half of
#    the lanes, errors the size of a few product terms - when they run on a second stream beside conv_bf16p
#    (v_mfma_f32_32x32x16_bf16): 25-30 of 30 runs in tools/coresidency_probe.py, about one two-lane bf16 sampler run in four.
#    Alone they are exact.  Without the vectoriser only: 0 of 30 in the probe, 2 of 12 sampler runs still differ (explicit
#    vector arithmetic still becomes v_pk_*); with packed-fp32-ops off as well - not one v_pk_ instruction in the library -
#    0 of 16 two-lane bf16 sampler runs (tools/bf16_lanes_soak.py) and 0 of 30 in every pairing of the probe.  Neither an LDS
#    canary nor hand-written v_pk_* chains beside the bf16 conv reproduce it (tools/lds_canary.hip), so the trigger is
#    narrower than "any packed instruction" - but with none in the library the question does not arise.
#    Whole-job throughput is unchanged (f32 1.794 vs 1.797, same box); bf16 runs its clips on two lanes again: 3.18 -> 3.68.
# (The host pass prints "'-packed-fp32-ops' is not a recognized feature for this target (ignoring feature)": filtered below.)
COMMON_FLAGS = ["-fno-slp-vectorize", "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
EXTRA_FLAGS = {}


def _compile_cmd(src):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    return [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", os.path.join(CSRC, src), "-o",
            os.path.join(HERE, "build", src + ".o"), "-Wno-unused-result"] + COMMON_FLAGS + EXTRA_FLAGS.get(src, [])


def needs_build():
    if not os.path.exists(OUT):
        return True
    for src in SOURCES:                                # flags changed since the objects were built?
        stamp = os.path.join(HERE, "build", src + ".o.cmd")
        if os.path.isdir(os.path.join(HERE, "build")) and os.path.exists(os.path.join(HERE, "build", src + ".o")):
            cmd = " ".join(_compile_cmd(src))
            if not os.path.exists(stamp) or open(stamp).read() != cmd:
                return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(os.path.dirname(HERE), "include", "babe_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    """Compile what is out of date.  Safe to call from several processes at once (one rank per GPU calls it): a file
    lock serialises them and the late comers find the library up to date."""
    if not force and not needs_build():
        return OUT
    import fcntl
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    with open(os.path.join(HERE, "build", ".lock"), "w") as lk:
        fcntl.flock(lk, fcntl.LOCK_EX)
        try:
            if not force and not needs_build():
                return OUT
            return _build_locked(force, verbose)
        finally:
            fcntl.flock(lk, fcntl.LOCK_UN)


def _build_locked(force, verbose):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    stamps = []
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    hdr_mtime = max([os.path.getmtime(os.path.join(CSRC, f)) for f in os.listdir(CSRC) if f.endswith(".h")] +
                    [os.path.getmtime(os.path.join(os.path.dirname(HERE), "include", "babe_hip.h"))])
    for src in SOURCES:
        p = os.path.join(CSRC, src)
        if not os.path.exists(p):
            raise FileNotFoundError(f"{p}: listed in babe_amd/build.py SOURCES but missing")
        o = os.path.join(HERE, "build", src + ".o")
        objs.append(o)
        cmd = _compile_cmd(src)
        # every object depends on its source, on every header in csrc/, on the public header AND on its compile command (the
        # flags are a correctness contract - no packed-fp32 instructions -: an object built under other flags is rebuilt)
        stamp = o + ".cmd"
        cmd_txt = " ".join(cmd)
        same_cmd = os.path.exists(stamp) and open(stamp).read() == cmd_txt
        if (not force) and same_cmd and os.path.exists(o) and os.path.getmtime(o) > max([os.path.getmtime(p), hdr_mtime]):
            continue
        stamps.append((stamp, cmd_txt))
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    failed = False
    for src, pr in procs:
        out, _ = pr.communicate()
        if pr.returncode != 0:
            failed = True
            print(f"--- {src} FAILED\n{out}", file=sys.stderr)
        elif verbose:
            out = "\n".join(l for l in out.splitlines() if "packed-fp32-ops" not in l)
            if out.strip():
                print(out)
    if failed:
        raise RuntimeError("hipcc failed")
    for stamp, txt in stamps:
        with open(stamp, "w") as fh:
            fh.write(txt)
    tmp = OUT + f".tmp{os.getpid()}"
    # -z defs: an undefined symbol (e.g. a kernel stub the host pass silently dropped) fails the link, not the first call
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,-z,defs", "-o", tmp] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    os.replace(tmp, OUT)               # atomic: a concurrent dlopen never sees a half-written library
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
