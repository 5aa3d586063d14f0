import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import tests.test_gpu_unet_full as T
s = T.load("sampler_full_46046.npz")
L, TT = int(s["L"]), int(s["T"])
prec = sys.argv[1]
net = T.full_net(L, prec)
for lanes in (1, 2, 2, 2, 1):
    smp = T._full_sampler(net, s)
    smp.LANES = lanes
    y = torch.cat([s["y0"], s["y1"], s["y0"], s["y1"]], 0).cuda()
    noises = [torch.cat([s["noises0"][i:i + 1], s["noises1"][i:i + 1]] * 2, 0) for i in range(TT + 1)]
    it = iter(noises)
    smp._randn = lambda shape, device: next(it).to(device)
    x, fp = smp.predict_blind_bwe(y)
    torch.cuda.synchronize()
    print(prec, "lanes", lanes, ["%.2e" % T.rms_err(x[b:b + 1], s[f"x{b % 2}"]) for b in range(4)], flush=True)
