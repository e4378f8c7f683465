"""How much of the pipelined step is host-side issue?  Replays the bench's 4 slot graphs round-robin and reports the time
the host needs to ISSUE n replays (no sync) next to the time until the device has finished them."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from golf_amd.synthetic import make_inputs

dev = torch.device("cuda:0")
osc, ss, ff = bench.build_modules(dev)
S = int(sys.argv[1]) if len(sys.argv) > 1 else 4
graphs, keep = [], []   # keep: the step closures own the inputs the graphs read
SEEDS = [int(x) for x in os.environ.get("PROBE_SEEDS", "").split(",") if x] or [2434 + k for k in range(S)]
for k in range(S):
    inp = make_inputs(B=32, device=dev, seed=SEEDS[k])
    fn, _, _ = bench.make_step("golf-ss-synth", inp, osc, ss, ff, fast=True, mode="auto")
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        y = fn()
    graphs.append(g)
    keep.append((fn, inp, y))
streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
for rep in range(3):
    for n in (200, 800):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            with torch.cuda.stream(streams[i % S]):
                graphs[i % S].replay()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"S={S} n={n}: issue {1e6 * (t1 - t0) / n:.1f} us/replay, done {1e6 * (t2 - t0) / n:.1f} us/step", flush=True)
