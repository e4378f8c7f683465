"""Quick device timing of the LPC-ss path (dev tool; bench.py is the contract)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from golf_amd import functional as GF
from golf_amd.synthetic import make_inputs

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
inp = make_inputs(B=B, device="cuda")
ex, gain, a = inp["noise"], inp["gain"], inp["a"]
for _ in range(3):
    y = GF.ltv_allpole_ss(ex, gain, a, 240)
torch.cuda.synchronize()
n = 20
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(n):
    y = GF.ltv_allpole_ss(ex, gain, a, 240)
e1.record(); torch.cuda.synchronize()
print(f"B={B} fwd {e0.elapsed_time(e1)/n*1000:.1f} us/iter")
exg, gg, ag = ex.clone().requires_grad_(True), gain.clone().requires_grad_(True), a.clone().requires_grad_(True)
gy = torch.randn(B, 47761, device="cuda")
for _ in range(3):
    y = GF.ltv_allpole_ss(exg, gg, ag, 240); y.backward(gy)
torch.cuda.synchronize()
e0.record()
for _ in range(n):
    y = GF.ltv_allpole_ss(exg, gg, ag, 240); y.backward(gy)
e1.record(); torch.cuda.synchronize()
print(f"B={B} fwd+bwd {e0.elapsed_time(e1)/n*1000:.1f} us/iter")
