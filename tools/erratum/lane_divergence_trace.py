import sys, os, math, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import tests.test_gpu_unet_full as T
from babe_amd import ops
torch.manual_seed(0)
L = int(os.environ.get("LEN", "92092"))
net = T.full_net(L, "bf16")
logs = {}
def rec(name, t, extra=""):
    key = torch.cuda.current_stream().cuda_stream
    tt = t if t.dtype != torch.int16 else t.view(torch.int16).float()
    logs.setdefault(key, []).append((name + extra, tuple(t.shape), tt.double().abs().sum() + tt.double().sum() * 1e-3))
def wrap(name, outidx=None, outkw=None):
    f = getattr(ops, name)
    def g(*a, **k):
        if name == "conv2d":
            rec("conv2d-input", a[0], f" shape={tuple(a[0].shape)} contiguous={a[0].is_contiguous()}")
        r = f(*a, **k)
        t = r
        if isinstance(r, tuple): t = r[1]
        desc = ""
        if name == "conv2d":
            pc = a[1]; desc = f" k={pc.KH}x{pc.KW} Cin={pc.Cin} Cout={pc.Cout} T={a[0].shape[-1]} F={a[0].shape[-2]} tr={k.get('transpose', False)} splits={pc.splits}"
        rec(name, t, desc)
        return r
    setattr(ops, name, g)
for n in ("conv2d", "conv2d_units", "scale_gelu_units", "scale_gelu", "gn_scale", "gn_bwd", "resample", "axpby"):
    wrap(n)
x = 0.1 * torch.randn(2, L, device="cuda"); cn = torch.full((2, 1), -0.4, device="cuda"); g = torch.randn(2, L, device="cuda")
s = [torch.cuda.Stream(), torch.cuda.Stream()]
def run_solo():
    logs.clear()
    with torch.cuda.stream(s[0]):
        net.fwd_nograd(x, cn, lane=0); net.vjp(g, lane=0)
    torch.cuda.synchronize()
    return [(n, sh, float(c)) for (n, sh, c) in logs[s[0].cuda_stream]]
ref = run_solo(); ref2 = run_solo()
print("solo reproducible:", ref == ref2, len(ref), "ops", flush=True)
found = {}
for it in range(10):
    logs.clear()
    for k in range(2):
        with torch.cuda.stream(s[k]): net.fwd_nograd(x, cn, lane=k)
    for k in range(2):
        with torch.cuda.stream(s[k]): net.vjp(g, lane=k)
    torch.cuda.synchronize()
    for k in range(2):
        cur = [(n, sh, float(c)) for (n, sh, c) in logs[s[k].cuda_stream]]
        for i, (a, b) in enumerate(zip(cur, ref)):
            if a != b:
                print(f"iter {it} lane {k}: first mismatch at op #{i}: {a[0]} shape {a[1]}  (checksum {a[2]!r} vs {b[2]!r}); previous op: {cur[i-1][0]}", flush=True)
                found[a[0].split()[0]] = found.get(a[0].split()[0], 0) + 1
                break
print("first-mismatch op histogram:", found)
