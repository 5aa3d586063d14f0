"""Soak test for bf16 networks on TWO clip lanes (the configuration that corrupted about one run in four before the library
was built without the SLP vectoriser): the benchmark-width network, four clips = two clips twice, N runs; duplicates must be
bit-identical and every clip within the bf16 bar of the fp32 reference golden."""
import os, sys, torch
os.environ.setdefault("BABE_BF16_LANES", "1")          # the opt-in this soak is about
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import test_gpu_unet_full as tf
s = tf.load("sampler_full_46046.npz")
L, T = int(s["L"]), int(s["T"])
net = tf.full_net(L, "bf16")
assert net.concurrent_lanes_ok
N = int(os.environ.get("N", "12"))
bad = 0
for run in range(N):
    smp = tf._full_sampler(net, s)
    y = torch.cat([s["y0"], s["y1"], s["y0"], s["y1"]], 0).cuda()
    noises = [torch.cat([s["noises0"][i:i + 1], s["noises1"][i:i + 1]] * 2, 0) for i in range(T + 1)]
    it = iter(noises)
    smp._randn = lambda shape, device: next(it).to(device)
    x, fp = smp.predict_blind_bwe(y)
    torch.cuda.synchronize()
    used = smp._use_lanes(4, y, False, fp.reshape(4, 2, -1))
    e = [tf.rms_err(x[b:b + 1], s[f"x{b % 2}"]) for b in range(4)]
    same = bool(torch.equal(x[0], x[2]) and torch.equal(x[1], x[3]))
    bad += int(not same or max(e) > 5e-3)
    print(f"run {run}: lanes {used}, duplicates bit-identical {same}, RMS err vs fp32 golden {[f'{v:.2e}' for v in e]}", flush=True)
print(f"{bad} of {N} runs bad")
