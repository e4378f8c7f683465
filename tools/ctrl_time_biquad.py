"""Cost of the ISMIR'23 ("coef" biquad) control transform of one LPC filter on the GPU: logits (B, F, 22) -> 11 biquads ->
direct form, eager and as a hipGraph.  python tools/ctrl_time_biquad.py"""
import sys, time
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from golf_amd.audiotensor import AudioTensor
from golf_amd.filters import LTVMinimumPhaseFilter

m = LTVMinimumPhaseFilter(window="hanning", window_length=480, centred=False, lpc_order=22, lpc_parameterisation="coef",
                          max_abs_value=0.99).cuda()
(split, trs) = m.ctrl(lambda s_, t_: (s_, t_))((), ())
lg = AudioTensor(torch.randn(32, 200, device="cuda") * 0.1, 240)
lo = AudioTensor(torch.randn(32, 200, 22, device="cuda") * 0.3, 240)
f = lambda: trs[0](lg, lo)
for _ in range(5): f()
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(50): f()
torch.cuda.synchronize(); print("eager: %.1f us" % ((time.perf_counter() - t) / 50 * 1e6))
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): f()
torch.cuda.current_stream().wait_stream(s)
with torch.cuda.graph(g): o = f()
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(50): g.replay()
torch.cuda.synchronize(); print("graph: %.1f us" % ((time.perf_counter() - t) / 50 * 1e6))
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for _ in range(10): f()
    torch.cuda.synchronize()
evs = [e for e in prof.key_averages() if e.device_time_total > 0]
print("kernels per call: %.1f, device us per call: %.1f" % (sum(e.count for e in evs) / 10, sum(e.device_time_total for e in evs) / 10))
