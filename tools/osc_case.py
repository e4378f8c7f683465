"""dev: the oscillator forward alone, for rocprofv3 counter passes (B from argv, default 256)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from golf_amd import functional as GF
from golf_amd.synthetic import make_inputs
from golf_amd.synth import DownsampledIndexedGlottalFlowTable
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
inp = make_inputs(B=32, device="cuda")
rep = (B + 31) // 32
phase, wsel = inp["phase"].repeat(rep, 1)[:B].contiguous(), inp["wsel"].repeat(rep, 1)[:B].contiguous()
osc = DownsampledIndexedGlottalFlowTable(hop_rate=10, in_channels=64, oversampling=4, equal_energy=True, lf_v2=True, points=2048).cuda()
for _ in range(6):
    o = GF.glottal_osc(phase, wsel, osc.table, osc.decimater.taps, 1, inp["w_hop"], 4, True)
torch.cuda.synchronize()
