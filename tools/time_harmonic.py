"""Device timing of the harmonic oscillator bank at the DDSP shape (dev tool)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from golf_amd import functional as GF
from golf_amd.synthetic import make_inputs
B, H = 32, 155
inp = make_inputs(B=B, device="cuda")
amp = (torch.rand(B, 201, H, device="cuda") / torch.arange(1, H + 1, device="cuda")).requires_grad_(True)
gy = torch.randn(B, 48000, device="cuda")
f = lambda: GF.harmonic_osc(inp["phase"], H, 1, amp, 240).backward(gy)
for _ in range(3): f()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA]) as p:
    for _ in range(10): f()
    torch.cuda.synchronize()
for e in sorted(p.key_averages(), key=lambda e: -e.device_time_total):
    if "golf" in e.key: print(f"{e.device_time_total/e.count:8.1f} us x{e.count}", e.key[:70])
