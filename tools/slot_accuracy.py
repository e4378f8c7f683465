"""Accuracy of the sample-wise filter on the inputs of bench.py's four in-flight slots (seeds 2434 .. 2437), per
utterance, for the default path and the sequential (serial) kernels against the float64 oracle (GPU)."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import golf_oracle as O
from golf_amd import functional as GF
from golf_amd.synthetic import make_inputs
for seed in range(2434, 2438):
    inp = make_inputs(B=32, seed=seed)
    ex, gain, a, hop = inp["noise"].numpy(), inp["gain"].numpy(), inp["a"].numpy(), inp["hop"]
    ref = O.ltv_allpole_ss_forward(ex, gain, a, hop)
    sc = np.abs(ref).max(1)
    d = lambda x: torch.as_tensor(x).cuda()
    out = {}
    for name, kw in (("default", {}), ("flat", dict(mode="flat-scan")), ("serial", dict(mode="serial")), ("fp64-phi", dict(fast_inference=False))):
        y = GF.ltv_allpole_ss(d(ex), d(gain), d(a), hop, **kw).cpu().numpy()
        out[name] = np.abs(y - ref).max(1) / sc
    worst = np.argsort(out["default"])[-3:]
    print(f"seed {seed}: default max {out['default'].max():.2e} flat {out['flat'].max():.2e} serial {out['serial'].max():.2e} "
          f"fp64-phi {out['fp64-phi'].max():.2e} | worst rows {worst} default {out['default'][worst]} serial {out['serial'][worst]}")
