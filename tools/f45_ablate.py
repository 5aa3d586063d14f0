"""Timing ablations of the F(4,5) x F(4,3) kernel: tools/abl_out/libbabe_abl_<k>.so are the library with conv_wino85.hip built
-DW85_ABL=<k> (1 no transform arithmetic, 2 no row loads, 4 no weight DMA, 8 no operand reads, 16 no MFMA; results are wrong with
any bit set).  Each variant runs in its own process (BABE_HIP_LIB); prints microseconds per launch for three layer shapes."""
import math, os, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = [("enc3", 1, 128, 256, 512, 4), ("enc5", 1, 256, 384, 128, 8), ("enc6", 2, 256, 448, 64, 8)]
if os.environ.get("F45_SHORT"):                         # few super-slabs: mostly the fixed cost per workgroup
    SHAPES = [("Cin16", 1, 16, 256, 512, 4), ("Cin32", 1, 32, 256, 512, 4), ("Cin128", 1, 128, 256, 512, 4)]
if len(sys.argv) > 1 and sys.argv[1] == "child":
    os.environ["BABE_CONV_F45"] = "1"
    sys.path.insert(0, R)
    import torch
    from babe_amd import ops
    res = []
    for name, B, C, Fq, T, dil in SHAPES:
        g = torch.Generator().manual_seed(1)
        x = torch.randn(B, C, Fq, T, generator=g).cuda()
        Co = 128 if os.environ.get("F45_SHORT") else C
        pc = ops.PackedConv((torch.randn(Co, C, 5, 3, generator=g) / math.sqrt(C * 15)).cuda())
        out = torch.empty(B, Co, Fq, T, device="cuda")
        fn = lambda: ops.conv2d(x, pc, out, dil=dil, force_f45=True)
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 20 * 1e3)
    print(" ".join(f"{r:8.1f}" for r in res))
    sys.exit(0)
names = {0: "full kernel", 256: "weights always from the first half-slot (L1 hits)", 1: "no transform arithmetic", 2: "no row loads", 4: "no weight DMA", 8: "no operand reads", 16: "no MFMA",
         3: "no loads, no transform", 7: "no loads / transform / DMA", 15: "MFMA only", 512: "no epilogue", 1024: "no pass carry",
         527: "MFMA only, no epilogue", 1551: "MFMA only, no epilogue, no carry", 2048: "epilogue without stores",
         4096: "epilogue stores raw accumulators"}
print("variant".ljust(30) + " ".join(f"{s[0]:>8s}" for s in SHAPES) + "   (us per launch)")
for k in [int(a) for a in sys.argv[1:]] or sorted(names, key=lambda v: (bin(v).count("1"), v)):
    lib = os.path.join(R, "tools", "abl_out", f"libbabe_abl_{k}.so")
    if not os.path.exists(lib):
        continue
    env = dict(os.environ, BABE_HIP_LIB=lib)
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, capture_output=True, text=True, timeout=300)
    print(f"{k:2d} {names.get(k, ''):27s}" + (r.stdout.strip() or r.stderr[-300:]), flush=True)
