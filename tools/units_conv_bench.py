"""Per-shape throughput of the bf16 units conv (forward (5,3) layers under precision='bf16')."""
import sys, os, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from babe_amd import ops
from babe_amd._lib import lib
B = int(os.environ.get("B", "2"))
shapes = [("enc1", 96, 128, 2048, 2), ("enc2", 96, 192, 1024, 4), ("enc3", 128, 256, 512, 8), ("enc4", 128, 320, 256, 4),
          ("enc5", 256, 384, 128, 8), ("enc6", 256, 448, 64, 16), ("dec5", 128, 384, 128, 4)]
for name, N, F, T, dil in shapes:
    x = torch.randn(B, N, F, T, device="cuda")
    sc = torch.rand(B, N, device="cuda") + 0.5
    w = torch.randn(N, N, 5, 3, device="cuda") / math.sqrt(N * 15)
    pc = ops.PackedConv(w, "bf16")
    au = torch.empty(B * lib().babe_units_size(N, F, T) * 8, dtype=torch.int16, device="cuda")
    ops.scale_gelu_units(x, sc, au)
    out = torch.empty_like(x)
    for _ in range(2): ops.conv2d_units(au, pc, out, N, dil=dil)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): ops.conv2d_units(au, pc, out, N, dil=dil)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    fl = 2.0 * B * N * N * 15 * F * T
    print(f"{name} N={N} F={F} T={T}: {ms*1e3:7.1f} us {fl/ms/1e9:7.1f} TF/s")
