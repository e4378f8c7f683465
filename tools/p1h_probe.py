"""dev probe: fp64 transition kernel duration (training path)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from golf_amd import functional as GF
from golf_amd.synthetic import make_inputs
inp = make_inputs(B=int(sys.argv[1]) if len(sys.argv) > 1 else 32, device="cuda")
noise, gain, a = inp["noise"], inp["gain"], inp["a"]
f = lambda: GF.ltv_allpole_ss(noise, gain, a, 240, fast_inference=False)
for _ in range(3): f()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA]) as p:
    for _ in range(10): f()
    torch.cuda.synchronize()
for e in p.key_averages():
    if "p1h" in e.key or "transpose" in e.key:
        print(f"KT={os.environ.get('GOLF_P1H_KT','default')} {e.device_time_total/e.count:8.1f} us {e.key[:60]}")
