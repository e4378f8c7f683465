"""dev: phase timing of osc_fused_kernel's workgroups (build with -DOSCF_TIMING: tools/build_variant.sh osct glottal_osc.hip
-DOSCF_TIMING; GOLF_HIP_LIBRARY=.../libgolf_osct.so).  s_memtime stamps of thread 0: 0 entry, 1 global loads consumed + rows
written to LDS (issue side), 2 after the first barrier (scan totals), 3 after the second barrier, 4 render done (this wave),
5 after the third barrier, 6 MFMA chain done, 7 stores issued.  One batch alone and with 4 batches in flight."""
import ctypes, os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from golf_amd import _lib, functional as GF
from golf_amd.synthetic import make_inputs
from golf_amd.synth import DownsampledIndexedGlottalFlowTable

lib = _lib.load()

cdll = ctypes.CDLL(__import__("os").environ["GOLF_HIP_LIBRARY"])
cdll.golf_debug_oscf_stamps.restype = ctypes.c_int
B = 32
osc = DownsampledIndexedGlottalFlowTable(hop_rate=10, in_channels=64, oversampling=4, equal_energy=True, lf_v2=True, points=2048).cuda()
inps = [make_inputs(B=B, device="cuda", seed=2434 + i) for i in range(4)]
run = lambda i: GF.glottal_osc(inps[i]["phase"], inps[i]["wsel"], osc.table, osc.decimater.taps, 1, inps[i]["w_hop"], 4, True, add=inps[i]["noise"])
n = int(os.environ.get('OSC_PHASES_WGS', 24 * B))   # workgroups that leave stamps (persistent grid: CUs x workgroups per CU)
import os as _os
print("library:", _os.environ.get("GOLF_HIP_LIBRARY"), "GOLF_OSCF_OLD =", _os.environ.get("GOLF_OSCF_OLD"))
def dump(tag):
    torch.cuda.synchronize()
    buf = np.zeros(8 * n, dtype=np.uint64)
    rc = cdll.golf_debug_oscf_stamps(buf.ctypes.data_as(ctypes.POINTER(ctypes.c_ulonglong)), 8 * n)
    assert rc == 0, rc
    raw = buf.reshape(n, 8)
    xcc = (raw[:, 0] >> np.uint64(60)).astype(np.int64)
    st = (raw & np.uint64(0x0fffffffffffffff)).astype(np.int64)
    rel = st - st[:, :1]
    ok = [i for i in range(8) if (st[:, i] > 0).all()]
    print(tag, "stamps relative to entry, mean ticks:", {i: int(rel[:, i].mean()) for i in ok},
          " lifetime mean %.0f p90 %.0f" % (rel[:, 7].mean(), np.percentile(rel[:, 7], 90)))
    if hasattr(cdll, "golf_debug_oscf_rt"):
        rt = np.zeros(2 * n, dtype=np.uint64)
        assert cdll.golf_debug_oscf_rt(rt.ctypes.data_as(ctypes.POINTER(ctypes.c_ulonglong)), 2 * n) == 0
        rt = rt.reshape(n, 2).astype(np.int64)
        t0 = rt[:, 0].min()
        pc = [0, 10, 25, 50, 75, 90, 100]
        print("    real-time counter (10 ns ticks -> us): kernel first entry -> last exit %.2f us; entries p%s: %s; exits: %s; by XCC first entry: %s" % (
            (rt[:, 1].max() - t0) / 100, pc, (np.percentile(rt[:, 0] - t0, pc) / 100).round(2), (np.percentile(rt[:, 1] - t0, pc) / 100).round(2),
            {int(x): round(float((rt[xcc == x, 0].min() - t0) / 100), 2) for x in sorted(set(xcc.tolist()))}))
for _ in range(3):
    run(0)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record(); run(0); e1.record(); torch.cuda.synchronize()
print("one call alone (totals + fused kernel), HIP events: %.1f us" % (e0.elapsed_time(e1) * 1e3))
dump("alone     ")
streams = [torch.cuda.Stream() for _ in range(4)]
torch.cuda.synchronize()
for rep in range(6):
    for i, s in enumerate(streams):
        with torch.cuda.stream(s):
            run(i)
dump("4 streams ")
