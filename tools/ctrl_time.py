import os, torch, time, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from golf_amd.synthetic import make_decoder
from golf_amd.audiotensor import AudioTensor
dec = make_decoder().cuda()
split, trsfms, keys = dec.split_sizes_and_trsfms
B, F = 32, 200
h = torch.randn(B, F, 343, device='cuda') * 0.3
flat = [s for g in split for s in g]
def ctrl():
    pieces = [AudioTensor(t.squeeze(2) if t.shape[2] == 1 else t, 240) for t in torch.split(h, flat, dim=2)]
    out, i = [], 0
    for g, f in zip(split, trsfms):
        out.append(f(*pieces[i:i+len(g)])); i += len(g)
    return out
for _ in range(5): ctrl()
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(50): ctrl()
torch.cuda.synchronize(); print("eager ctrl transforms: %.1f us" % ((time.perf_counter()-t)/50*1e6))
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): ctrl()
torch.cuda.current_stream().wait_stream(s)
with torch.cuda.graph(g): o = ctrl()
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(50): g.replay()
torch.cuda.synchronize(); print("graph ctrl transforms: %.1f us" % ((time.perf_counter()-t)/50*1e6))
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for _ in range(10): ctrl()
    torch.cuda.synchronize()
n = sum(e.count for e in prof.key_averages()); tt = sum(e.device_time_total for e in prof.key_averages())
print("kernels per call:", n/10, "device us per call:", tt/10)
for e in sorted(prof.key_averages(), key=lambda e: -e.device_time_total)[:16]:
    print("%7.1f us x%4.1f  %s" % (e.device_time_total / 10, e.count / 10, e.key[:100]))
