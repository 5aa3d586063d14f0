"""Build libbabe_hip.so (gfx950) in-tree with hipcc.  Cross-compiles without a GPU."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libbabe_hip.so")
SOURCES = ["misc.hip", "conv.hip", "conv11p.hip", "conv_fewco.hip", "conv_bf16.hip", "conv_bf16p.hip", "conv_wino.hip", "conv_wino4.hip", "conv_wino4p.hip", "conv_wino45.hip", "conv_wino85.hip", "norm.hip", "resample.hip", "resample_sinc.hip", "cqt.hip", "cqt_plan.hip", "fft_mixed.hip", "unet_engine.hip", "score_eval.hip", "stft.hip", "sampler.hip", "denoiser.hip"]


# Every source is built WITHOUT packed-fp32 instructions: no SLP vectoriser (-fno-slp-vectorize) and the target feature
# packed-fp32-ops switched off (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 / v_pk_mov_b32 are never selected, also not for
# explicit float2 / float4 vector arithmetic).  Two reasons:
#  * speed of the fp32-MFMA kernels: next to fp32 MFMAs every vector instruction costs matrix-pipe time
#    (tools/mfma_valu_coexec.hip: a v_pk_fma_f32 costs 1.6x a v_fma_f32, a v_mov as much as a v_fma), and the vectoriser pays
#    for its packed arithmetic with register moves: conv_wino45 207 instead of 245 vector instructions per two K-slabs;
#  * CORRECTNESS next to the bf16 conv.  Round 3 saw kernels built with hipcc's defaults (conv_fewco, the FFT's twiddle /
#    transpose, epilogues with float4 arithmetic) return wrong sums when they ran on a second stream beside conv_bf16p.  Round 4
#    reduced it to ONE instruction form and a 60-line reproducer with no library in it (tools/erratum/pk_opsel_min.hip,
#    profiles/r04_coresidency_repro.txt): v_pk_{mul,add,fma}_f32 with op_sel:[0,1] - the low half of the result reads the HIGH
#    word of src1 - computes as if that word were 0 while waves of ANOTHER kernel execute bf16 MFMA on the same CU; plain packed
#    forms, op_sel:[1,0] and op_sel_hi:[1,0] were never wrong.  hipcc emits exactly that selector when it vectorises
#    `w[k] * float4`.  With no packed-fp32 instruction in the library the question does not arise for its own kernels
#    (tests/test_no_packed_fp32.py disassembles the library and keeps it that way); kernels of OTHER code running beside the
#    bf16 conv are the reason a bf16 network keeps one stream by default (networks/cqtdiff_plus.py).
#    Whole-job throughput is unchanged by the flags (f32 1.794 vs 1.797 audio-sec/s, same box).
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
