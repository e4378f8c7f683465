"""dev probe: deterministic 2-graph software pipeline of the GOLF-ss step on S streams."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from golf_amd import functional as GF
from golf_amd.synthetic import make_inputs

dev = torch.device("cuda", 0)
inp = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in make_inputs(B=32).items()}
osc, ss, ff = bench.build_modules(dev)
step, samples, t_out = bench.make_step("golf-ss-synth", inp, osc, ss, ff)
phase, wsel, w_hop, noise, gain, a, hop = (inp[k] for k in ("phase", "wsel", "w_hop", "noise", "gain", "a", "hop"))
taps, table = osc.decimater.taps, osc.table
T = 47761

def g1():
    return GF.ltv_allpole_prepare(a, hop, T)

def g2(prep):
    o = GF.glottal_osc(phase, wsel, table, taps, 1, w_hop, 4, True)
    return GF.ltv_allpole_ss(o + noise[:, : o.shape[1]], gain, a, hop, prep)

ref = step(); torch.cuda.synchronize()

def capture():
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2): g2(g1())
    torch.cuda.current_stream().wait_stream(s)
    G1 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(G1):
        prep = g1()
    G2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(G2):
        y = g2(prep)
    return G1, G2, y, prep

for S in (2, 3, 4):
    slots = [capture() for _ in range(S)]
    streams = [torch.cuda.Stream() for _ in range(S)]
    torch.cuda.synchronize()
    slots[0][0].replay(); slots[0][1].replay(); torch.cuda.synchronize()
    assert torch.equal(slots[0][2], ref)
    def run(n):
        prev = None
        for k in range(n):
            i = k % S
            with torch.cuda.stream(streams[i]):
                if prev is not None:
                    streams[i].wait_event(prev)   # P1h kernels never overlap each other
                slots[i][0].replay()
                prev = torch.cuda.Event(); prev.record(streams[i])
                slots[i][1].replay()
    run(4 * S); torch.cuda.synchronize()
    n = 400
    t0 = time.perf_counter(); run(n); ti = time.perf_counter() - t0
    torch.cuda.synchronize(); ta = time.perf_counter() - t0
    print(f"2-graph pipeline S={S}: host issue {ti/n*1e6:.1f} us/step, total {ta/n*1e6:.1f} us/step -> {samples/(ta/n)/1e9:.2f} G samples/s")
