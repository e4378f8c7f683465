"""Determinism soak: the captured decoder step replayed many times on 4 streams must reproduce its first output bit for
bit every time (races between the independent waves of the fused kernels, stale workspace reuse, stream aliasing would
show up here).  python tools/soak.py [replays] [latency]
("latency": the pipeline keeps the lone-batch chain of the filter, whose chunk passes are ONE launch with waves that wait for each
other -- four of those in flight at once is the placement its look-back rule has to survive.)"""
import sys
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
latency = len(sys.argv) > 2 and sys.argv[2] == "latency"
sys.argv, n = [sys.argv[0]], int(sys.argv[1]) if len(sys.argv) > 1 else 2000
import bench
from golf_amd.pipeline import ReplayPipeline
from golf_amd.synthetic import make_inputs

dev = torch.device("cuda", 0)
inp = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in make_inputs(B=32, device="cpu", with_noise_filter=True).items()}
osc, ss, ff = bench.build_modules(dev)
bad = 0
for workload in ("golf-ss-synth", "golf-ss-decoder", "golf-ff-synth", "golf-ss-train"):
    step, _, _ = bench.make_step(workload, inp, osc, ss, ff)
    pipe = ReplayPipeline(lambda _: step(), lambda: {"x": inp["phase"]}, n_slots=4, throughput=False if latency else None)
    ref = step().clone()
    mism = 0
    for i in range(n):
        s = pipe.submit()
        if i % 50 == 49:                      # check a slot now and then without serialising the pipeline
            s.stream.synchronize()
            mism += int(not torch.equal(s.output, ref))
    pipe.synchronize()
    mism += sum(int(not torch.equal(s.output, ref)) for s in pipe.slots)
    print(f"{workload}: {n} replays, {mism} mismatching outputs")
    bad += mism
sys.exit(1 if bad else 0)
