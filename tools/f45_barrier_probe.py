"""Where a workgroup of the 12-wave F(4,5) kernel waits: per-super-slab barrier time of the first multiplying wave and of the first
transform wave against their total time (s_memtime, 100 MHz ticks on gfx950 - only ratios are used).  Needs a library with conv_wino85.hip
built -DW85_ABL=16384 -DW85_FLAGS=0 (tools/ab/variant_build.sh probe conv_wino85 -DW85_ABL=16384 -DW85_FLAGS=0: the probe brackets the
barrier of the barrier form; the default form since round 6 has none), selected with BABE_HIP_LIB."""
import ctypes as C
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from babe_amd import ops
from babe_amd._lib import ConvArgs, check, lib, ptr, stream

SHAPES = [("enc3", 128, 256, 512, 4), ("enc4", 128, 320, 256, 2), ("enc5", 256, 384, 128, 8), ("enc6", 256, 448, 64, 8), ("enc1", 96, 128, 2048, 2),
          ("enc0", 64, 64, 4096, 1)]
for name, Cc, Fq, T, dil in SHAPES:
    g = torch.Generator().manual_seed(1)
    x = torch.randn(1, Cc, Fq, T, generator=g).cuda()
    res = torch.randn(1, Cc, Fq, T, generator=g).cuda()
    gate = torch.randn(1, Cc, generator=g).cuda()
    pc = ops.PackedConv((torch.randn(Cc, Cc, 5, 3, generator=g) / math.sqrt(Cc * 15)).cuda())
    out = torch.empty_like(x)
    a = ConvArgs()
    a.in_, a.in_bs, a.in_cs = ptr(x), Cc * Fq * T, Fq * T
    a.in2, a.cin_split = None, Cc
    a.out, a.out_bs, a.out_cs = ptr(out), Cc * Fq * T, Fq * T
    a.res, a.res_bs, a.res_cs = ptr(res), Cc * Fq * T, Fq * T
    a.oscale = ptr(gate)
    a.alpha, a.rbeta = 0.7, 0.7
    a.B, a.Cin, a.Cout, a.F, a.T, a.KH, a.KW, a.dil = 1, Cc, Cc, Fq, T, 5, 3, dil
    ntile = ((T + 63) // 64) * dil * ((((Fq + dil - 1) // dil) + 3) // 4) * max(1, Cc // 128 if Cc % 128 == 0 else 1)
    dbg = torch.zeros(4 * (ntile + 64) + 8 * (ntile + 64), device="cuda", dtype=torch.float64)
    a.stat_mode, a.stat_cg, a.stat_part = 99, 4, ptr(dbg)
    assert lib().babe_conv2d_wino85_supported(C.byref(a))
    for _ in range(3):
        check(lib().babe_conv2d_wino85(C.byref(a), ptr(pc.fwd_wino85), stream()), "conv2d_wino85")
    torch.cuda.synchronize()
    per_wave = dbg[4 * (ntile + 64):].view(-1, 8).cpu()
    per_wave = per_wave[per_wave[:, 0] > 0].mean(0)
    d = dbg[:4 * (ntile + 64)].view(-1, 4).cpu()
    d = d[d[:, 1] > 0]
    mw, mt, tw, tt = (float(d[:, i].mean()) for i in range(4))
    tvm = float(((d[:, 2] - d[:, 2].floor()) * 1e9).mean())           # ticks waiting for rows (fraction of entry 2)
    tw = float(d[:, 2].floor().mean())
    print(f"{name} C={Cc:3d} F={Fq} T={T} dil={dil}: {len(d)} workgroups | multiplying wave: barrier wait {mw / mt:5.1%} of its main loop "
          f"({mt:8.0f} ticks) | transform wave: barrier wait {tw / tt:5.1%}, waiting for its rows {tvm / tt:5.1%} of its life ({tt:8.0f} ticks)\n      barrier share of multiplying waves 0..7: " + " ".join(f"{float(v):.2f}" for v in per_wave))
