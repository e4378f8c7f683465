"""dev: phase timing of the two-level chunk passes (build with -DFWDQ2_TIMING: tools/build_variant.sh fqt lpc_ss.hip
-DFWDQ2_TIMING; GOLF_HIP_LIBRARY=.../libgolf_fqt.so).  s_memtime stamps (100 MHz ticks) of lane 0 of every (utterance, group)
wave: 0 entry, 1 fold over the earlier groups done, 2 prologue done, 3 states written / added, 4 chunk recursion done,
5 epilogue done.  Printed per group index (mean over the batch), refinement pass and final pass, one batch alone.
(The pair of launches is what a caller with batches in flight gets -- THROUGHPUT_MODE; one batch alone otherwise takes the
merged kernel: tools/fwdq2m_phases.py.)"""
import ctypes, os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from golf_amd import _lib, functional as GF
from golf_amd.synthetic import make_inputs

GF.THROUGHPUT_MODE = True
lib = _lib.load()
cdll = ctypes.CDLL(os.environ["GOLF_HIP_LIBRARY"])
cdll.golf_debug_fwdq2_stamps.restype = ctypes.c_int
B = 32
inp = make_inputs(B=B, device="cuda", seed=2434)
T = (inp["a"].shape[1] - 1) * 240 + 1
ex = torch.randn(B, T, device="cuda")
run = lambda: GF.ltv_allpole_ss(ex, inp["gain"], inp["a"], 240)
for _ in range(5):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(); e1.record(); torch.cuda.synchronize()
print("filter alone, HIP events: %.1f us" % (e0.elapsed_time(e1) * 1e3))
n = 2 * 64 * 64 * 8
buf = np.zeros(n, dtype=np.uint64)
assert cdll.golf_debug_fwdq2_stamps(buf.ctypes.data_as(ctypes.POINTER(ctypes.c_ulonglong)), n) == 0
st = buf.reshape(2, 64, 64, 8).astype(np.int64)
NG = int((st[0, 0, :, 0] > 0).sum())
for m, name in ((0, "refinement pass"), (1, "final pass")):
    s = st[m, :B, :NG, :6]
    print(name, "groups", NG)
    print("   g    fold    own    s1/add  body   epilogue   total   (k ticks of s_memtime, mean over utterances)")
    for g in range(NG):
        d = np.diff(s[:, g, :], axis=1).mean(0) / 1000.0
        print("  %2d  %5.2f  %5.2f  %5.2f  %6.2f  %6.2f   %6.2f" % (g, d[0], d[1], d[2], d[3], d[4], (s[:, g, 5] - s[:, g, 0]).mean() / 1000.0))
