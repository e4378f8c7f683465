"""Kernels of one module-level decoder call (golf-precise.yaml) driven from encoder logits: names, counts, device time."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ["bench.py"]
import bench
from golf_amd.synthetic import make_inputs
dev = torch.device("cuda", 0)
inp = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in make_inputs(B=32, device="cpu", with_noise_filter=True).items()}
osc, ss, ff = bench.build_modules(dev)
step, _, _ = bench.make_step("golf-ss-decoder-logits", inp, osc, ss, ff)
for _ in range(3): step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for _ in range(10): step()
    torch.cuda.synchronize()
evs = [e for e in prof.key_averages() if e.device_time_total > 0]
print("kernels per call: %.1f, device us per call: %.1f" % (sum(e.count for e in evs) / 10, sum(e.device_time_total for e in evs) / 10))
for e in sorted(evs, key=lambda e: -e.device_time_total):
    print("%7.1f us x%4.1f  %s" % (e.device_time_total / 10, e.count / 10, e.key[:110]))
