"""Does conv_bf16p write into LDS that belongs to another workgroup?  Runs an LDS canary kernel (tools/erratum/lds_canary.hip, compiled
here with hipcc) on one stream beside bf16 / fp32 conv launches on another and prints what changed in the canaries' LDS."""
import ctypes as C, math, os, subprocess, sys, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from babe_amd import ops
from babe_amd._lib import stream
so = "/tmp/lds_canary.so"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", os.path.join(R, "tools", "lds_canary.hip"), "-o", so])
lib = C.CDLL(so)
lib.lds_canary.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_long, C.c_void_p]
lib.valu_canary.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
lib.pk_canary.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
def mkconv(prec, Cin, Cout, F, T, kh, dil, B=2):
    kw = 3 if kh == 5 else 1
    ww = torch.randn(Cout, Cin, kh, kw, device="cuda") / math.sqrt(Cin * kh * kw); pc = ops.PackedConv(ww, prec)
    xx = torch.randn(B, Cin, F, T, device="cuda"); out = torch.empty(B, Cout, F, T, device="cuda")
    return lambda: ops.conv2d(xx, pc, out, dil=dil)
partners = {"bf16p fwd 256ch": mkconv("bf16", 256, 256, 448, 64, 5, 2), "bf16p fwd 64ch": mkconv("bf16", 64, 64, 64, 4096, 5, 1),
            "f32 wide 256ch": mkconv("f32", 256, 256, 448, 64, 5, 2), "none": lambda: None}
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
words_list = [int(v) for v in os.environ.get("WORDS", "8192").split(",")]
for words in words_list:
    for pn, pf in partners.items():
        out = torch.zeros(4 + 4 * 1024, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        for i in range(10):
            with torch.cuda.stream(sB): pf()
            with torch.cuda.stream(sA):
                rc = lib.lds_canary(out.data_ptr(), 512, 256, words, 200000, stream())
                assert rc == 0, rc
            with torch.cuda.stream(sB): pf()
        torch.cuda.synchronize()
        o = out.cpu()
        n = int(o[0])
        print(f"canary {words * 4} B LDS beside {pn:18s}: {n} changed dwords", flush=True)
        for k in range(min(n, 12)):
            wg, idx, exp, got = (int(v) & 0xffffffff for v in o[4 + 4 * k: 8 + 4 * k])
            print(f"    wg {wg} dword {idx}: expected {exp:08x} found {got:08x}")

# VALU canary: two identical FMA chains per lane must agree bit for bit
for pn, pf in partners.items():
    out = torch.zeros(4 + 4 * 1024, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    for i in range(10):
        with torch.cuda.stream(sB): pf()
        with torch.cuda.stream(sA):
            rc = lib.valu_canary(out.data_ptr(), 1024, 256, 2000, stream())
            assert rc == 0, rc
        with torch.cuda.stream(sB): pf()
    torch.cuda.synchronize()
    o = out.cpu()
    n = int(o[0])
    print(f"VALU canary beside {pn:18s}: {n} mismatching registers", flush=True)
    for k in range(min(n, 12)):
        wg, th, reg, x = (int(v) & 0xffffffff for v in o[4 + 4 * k: 8 + 4 * k])
        print(f"    wg {wg} thread {th} (lane {th & 63}) register {reg}: xor {x:08x}")

# packed-fp32 canary
for pn, pf in partners.items():
    out = torch.zeros(4 + 4 * 1024, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    for i in range(10):
        with torch.cuda.stream(sB): pf()
        with torch.cuda.stream(sA):
            rc = lib.pk_canary(out.data_ptr(), 1024, 256, 3000, stream())
            assert rc == 0, rc
        with torch.cuda.stream(sB): pf()
    torch.cuda.synchronize()
    o = out.cpu()
    n = int(o[0])
    print(f"packed-fp32 canary beside {pn:18s}: {n} mismatching register pairs", flush=True)
    for k in range(min(n, 16)):
        wg, th, reg, x = (int(v) & 0xffffffff for v in o[4 + 4 * k: 8 + 4 * k])
        print(f"    wg {wg} thread {th} (lane {th & 63}) pair {reg & 0xff} halves {'x' if reg & 0x100 else ''}{'y' if reg & 0x200 else ''}: xor {x:08x}")
