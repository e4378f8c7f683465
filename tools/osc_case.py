"""dev: the oscillator forward alone, for rocprofv3 counter passes"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from golf_amd import functional as GF
from golf_amd.synthetic import make_inputs
from golf_amd.synth import DownsampledIndexedGlottalFlowTable
inp = make_inputs(B=32, device="cuda")
osc = DownsampledIndexedGlottalFlowTable(hop_rate=10, in_channels=64, oversampling=4, equal_energy=True, lf_v2=True, points=2048).cuda()
for _ in range(12):
    o = GF.glottal_osc(inp["phase"], inp["wsel"], osc.table, osc.decimater.taps, 1, inp["w_hop"], 4, True)
torch.cuda.synchronize()
