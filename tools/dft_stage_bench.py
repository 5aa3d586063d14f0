"""Duration of each dense-DFT stage of the length-L real FFT (four-step), per call."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from babe_amd import ops
from babe_amd.cqt import RealFFT, _register_sigs
from babe_amd._lib import lib, check, ptr, stream, dispatch_counts
B = int(os.environ.get("B", "1"))
dev = torch.device("cuda", 0)
_register_sigs()
f = RealFFT(368368, dev)
N1, N2, K2 = f.N1, f.N2, f.K2
print("N1", N1, "N2", N2, "K2", K2)
x = torch.randn(B, 368368, device=dev)
W1, W3 = (f.W1s, f.W3s) if B < 4 else (f.W1b, f.W3b)
A = torch.empty(B, 2 * N1, 1, N2, device=dev); At = torch.randn(B, 2 * N2, 1, N1, device=dev)
out = torch.empty(B, 2, f.KX, device=dev); spec = torch.randn(B, 2, f.KX, device=dev); xo = torch.empty(B, 368368, device=dev)
calls = {
    "W1   stage over N2 positions     ": lambda: ops.conv2d(x.reshape(B, N1, 1, N2), W1, A),
    "W3   stage over N1 positions   ": lambda: ops.conv2d(At, W3, out.view(B, 2 * K2, 1, N1)),
    "W3^T stage over N1 positions   ": lambda: ops.conv2d(spec.view(B, 2 * K2, 1, N1), W3, At, transpose=True),
    "W1^T stage over N2 positions  ": lambda: ops.conv2d(A, W1, xo.view(B, N1, 1, N2), transpose=True),
    "twiddle_transpose": lambda: check(lib().babe_fft_twiddle_transpose(ptr(A), ptr(At), ptr(f.tw), B, N1, N2, 0, stream()), "tw"),
}
for name, fn in calls.items():
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"B={B} {name:32s} {e0.elapsed_time(e1)/20*1e3:7.1f} us")
