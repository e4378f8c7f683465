"""dev: timeline of golf_source_transitions_f32's one launch (oscillator workgroups beside the transition-map waves) -- needs a
-DSRCMAPS_TIMING build of glottal_osc.hip (tools/build_variant.sh srct glottal_osc.hip -DSRCMAPS_TIMING; GOLF_HIP_LIBRARY=...).
Entry / exit of every workgroup on the 100 MHz s_memrealtime clock, the CU it ran on (HW_ID + XCC_ID): how many transition
workgroups share a CU, when the oscillator's rounds start and end, when the transition waves end.   python tools/srcmaps_timeline.py [B]"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from golf_amd import _lib, functional as GF
from golf_amd.synth import DownsampledIndexedGlottalFlowTable
from golf_amd.synthetic import make_inputs

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
inp = make_inputs(B=B, device="cuda")
osc = DownsampledIndexedGlottalFlowTable(hop_rate=10, in_channels=64, oversampling=4, equal_energy=True, lf_v2=True, points=2048).cuda()
run = lambda: GF.source_filter_ss(inp["phase"], inp["wsel"], osc.table, osc.decimater.taps, 1, inp["w_hop"], 4, True, inp["gain"], inp["a"], 240, add=inp["noise"])
for _ in range(5):
    run()
torch.cuda.synchronize()
lib = _lib.load()
n = 4 * 2048
buf = (ctypes.c_ulonglong * n)()
assert lib.golf_debug_srcmaps_rt(buf, n) == 0
r = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 4).astype(np.int64)
nblk_f = (B * 199 + 39) // 40      # 10 chunks per wave, 4 waves per workgroup
nw = nblk_f + B * 24
r = r[:nw]
t0 = r[:, 0].min()
ent, ext = (r[:, 0] - t0) / 100.0, (r[:, 1] - t0) / 100.0
hw = r[:, 2]
cu = ((hw >> 32) & 15) * 1000 + ((hw >> 13) & 7) * 100 + ((hw >> 12) & 1) * 16 + ((hw >> 8) & 15)   # xcc, se, sh, cu
maps, oscw = slice(0, nblk_f), slice(nblk_f, nw)
print(f"transition workgroups: {nblk_f}, on {len(set(cu[maps]))} distinct CUs; entry {ent[maps].min():.2f} .. {ent[maps].max():.2f} us, exit {ext[maps].min():.1f} .. {ext[maps].max():.1f} us (mean {ext[maps].mean():.1f})")
cnt = np.bincount(np.unique(cu[maps], return_counts=True)[1])
print("  transition workgroups per CU -> CUs:", {k: int(v) for k, v in enumerate(cnt) if v})
print(f"oscillator workgroups: {nw - nblk_f}; entry quartiles {np.percentile(ent[oscw], [0, 25, 50, 75, 100]).round(1)} us, exit {np.percentile(ext[oscw], [0, 25, 50, 75, 100]).round(1)} us, lifetime mean {(ext[oscw] - ent[oscw]).mean():.2f} us")
shared = np.isin(cu[oscw], list(set(cu[maps])))
print(f"  oscillator workgroups on a CU that holds transition waves: {int(shared.sum())}; lifetime there {(ext[oscw] - ent[oscw])[shared].mean():.2f} us, elsewhere {(ext[oscw] - ent[oscw])[~shared].mean():.2f} us")
both = np.unique(cu[oscw][shared], return_counts=True)[1]
print(f"  kernel ends at {max(ext.max(), 0):.1f} us")
