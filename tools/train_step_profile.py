#!/usr/bin/env python
"""Where one config-5 optimisation step spends its device time (bench.py --workload golf-ss-train-step): device kernels
grouped into the HIP decoder (golf::), MIOpen convolutions / batch-norm / pooling, the LSTM, rocFFT (spectrograms of
encoder and loss) and the rest (elementwise, optimiser).  Prints JSON; run on the GPU box."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from golf_amd.synthetic import make_inputs  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda", 0)
inp = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in make_inputs(B=B, device="cpu").items()}
osc, ss, ff = bench.build_modules(dev)
step, samples, t_out = bench.make_step("golf-ss-train-step", inp, osc, ss, ff)
for _ in range(3):
    step()
torch.cuda.synchronize()
n = 5
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA]) as prof:
    for _ in range(n):
        step()
    torch.cuda.synchronize()
groups, top = {}, {}
for ev in prof.key_averages():
    t = getattr(ev, "device_time_total", 0) or getattr(ev, "cuda_time_total", 0)
    if t <= 0:
        continue
    name = ev.key
    low = name.lower()
    if "golf::" in name:
        g = "hip_decoder"
    elif any(s in low for s in ("fft", "bluestein", "transpose_kernel", "real2complex", "complex2real", "stockham")):
        g = "rocfft"
    elif any(s in low for s in ("lstm", "rnn")):
        g = "lstm"
    elif any(s in low for s in ("conv", "igemm", "gemm", "cijk", "winograd", "batchnorm", "batch_norm", "pool", "miopen",
                                "naive", "im2col", "col2im")):
        g = "conv_gemm_bn_pool"
    else:
        g = "elementwise_optimizer_other"
    groups[g] = groups.get(g, 0.0) + t / n
    top[name[:90]] = top.get(name[:90], 0.0) + t / n
total = sum(groups.values())
print(json.dumps({"batch": B, "device_us_per_step": round(total, 1),
                  "groups_us": {k: round(v, 1) for k, v in sorted(groups.items(), key=lambda kv: -kv[1])},
                  "top_kernels_us": {k: round(v, 1) for k, v in sorted(top.items(), key=lambda kv: -kv[1])[:25]}},
                 indent=1))
