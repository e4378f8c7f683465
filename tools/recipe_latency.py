"""Latency of one batch alone (graph replay, one stream) for consecutive recipe seeds, with the filter's conditioning
words beside it: what a hot / tier-3 utterance costs its batch.   python tools/recipe_latency.py [n_seeds] [B]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from golf_amd.synthetic import make_inputs

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
dev = torch.device("cuda:0")
osc, ss, ff = bench.build_modules(dev)
for k in range(n):
    inp = make_inputs(B=B, device=dev, seed=2434 + k)
    fn, _, _ = bench.make_step("golf-ss-synth", inp, osc, ss, ff, fast=True, mode="auto")
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        y = fn()
    torch.cuda.synchronize()
    us = min(bench.event_time_us(g.replay, n=50) for _ in range(3))
    c = bench.conditioning_of("golf-ss-synth", inp, dev)
    print(f"seed {2434 + k}: {us:7.1f} us  hot {c['hot_utterances']} tier3 {c['tier3_utterances']} max|Phi| {c['max_phi']:.1f}", flush=True)
