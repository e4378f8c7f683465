"""dev probe: host issue time per step, and hipGraph replay of the whole GOLF-ss step on S streams."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from golf_amd.synthetic import make_inputs

dev = torch.device("cuda", 0)
inp = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in make_inputs(B=32).items()}
osc, ss, ff = bench.build_modules(dev)
step, samples, t_out = bench.make_step("golf-ss-synth", inp, osc, ss, ff)
for _ in range(10): step()
torch.cuda.synchronize()
n = 200
t0 = time.perf_counter()
for _ in range(n): step()
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"eager 1 stream: host issue {t_issue/n*1e6:.1f} us/step, total {t_all/n*1e6:.1f} us/step")

def make_graph():
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3): step()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        y = step()
    return g, y

for S in (1, 2, 3, 4):
    graphs = [make_graph() for _ in range(S)]
    streams = [torch.cuda.Stream() for _ in range(S)]
    torch.cuda.synchronize()
    ref = step()
    graphs[0][0].replay(); torch.cuda.synchronize()
    assert torch.equal(graphs[0][1], ref), "graph replay differs from eager"
    for k in range(20):
        with torch.cuda.stream(streams[k % S]): graphs[k % S][0].replay()
    torch.cuda.synchronize()
    n = 300
    t0 = time.perf_counter()
    for k in range(n):
        with torch.cuda.stream(streams[k % S]): graphs[k % S][0].replay()
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"graphs S={S}: host issue {t_issue/n*1e6:.1f} us/step, total {t_all/n*1e6:.1f} us/step -> {samples/(t_all/n)/1e9:.2f} G samples/s")
