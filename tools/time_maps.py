"""dev: the fp32 transition kernel alone (launch_maps via ltv_allpole_prepare(maps_only=True)), HIP-event time per call."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from golf_amd import functional as GF
from golf_amd.synthetic import make_inputs

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
inp = make_inputs(B=B, device="cuda", seed=2434)
a, hop = inp["a"], inp["hop"]
T = (a.shape[1] - 1) * hop + 1
run = lambda: GF.ltv_allpole_prepare(a, hop, T, fast=True, maps_only=True)
for _ in range(20):
    run()
torch.cuda.synchronize()
ts = []
for rep in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        run()
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) * 1e3 / 50)
print("B=%d transition kernel alone (back to back): %s us per call" % (B, [round(t, 2) for t in ts]))
