"""Device timing of the zero-phase FIR noise filter kernels via the torch profiler (dev tool)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from golf_amd import functional as GF

B, T, F, n_mag, hop = int(sys.argv[1]) if len(sys.argv) > 1 else 32, 48000, 200, 256, 240
torch.manual_seed(0)
ex = torch.randn(B, T, device="cuda", requires_grad=True)
lm = (torch.randn(B, F, n_mag, device="cuda") * 0.5 - 2).requires_grad_(True)
win = torch.hann_window(2 * (n_mag - 1), device="cuda")
gy = torch.randn(B, 199 * hop, device="cuda")
for _ in range(3):
    y = GF.zero_phase_fir_filter(ex, lm, win, hop); y.backward(gy)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for _ in range(10):
        y = GF.zero_phase_fir_filter(ex, lm, win, hop); y.backward(gy)
    torch.cuda.synchronize()
for e in sorted(prof.key_averages(), key=lambda e: -e.device_time_total):
    if e.device_time_total > 0:
        print(f"{e.device_time_total / e.count:9.1f} us x{e.count:3d}  {e.key[:110]}")
