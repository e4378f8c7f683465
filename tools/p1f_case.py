"""dev probe: one composition (argv[1] = slow|fast), 10 iterations, for rocprofv3 counter passes"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from golf_amd import functional as GF
from golf_amd.synthetic import make_inputs
inp = make_inputs(B=32, device="cuda")
noise, gain, a = inp["noise"], inp["gain"], inp["a"]
mode = sys.argv[1]
for _ in range(12):
    if mode == "slow":
        src = noise[:, :47760] + noise[:, :47760]
    else:
        src = noise[:, :47760].contiguous()
    y = GF.ltv_allpole_ss(src, gain, a, 240)
torch.cuda.synchronize()
