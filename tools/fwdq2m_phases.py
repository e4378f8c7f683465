"""dev: phase timeline of the merged chunk pass (lpc_fwdq2m_kernel; build with -DFWDQ2_TIMING: tools/build_variant.sh fqt lpc_ss.hip
-DFWDQ2_TIMING; GOLF_HIP_LIBRARY=.../libgolf_fqt.so).  s_memrealtime stamps (100 MHz) of lane 0 of every (utterance, group) wave:
0 entry, 1 first prologue done, 2 chunks re-run (defects), 3 group response, 4 released + flagged, 5 predecessors seen,
6 second prologue done, 7 chunks written.  Printed per group (mean over the batch) in us, entry relative to the earliest entry."""
import ctypes, os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from golf_amd import _lib, functional as GF
from golf_amd.synthetic import make_inputs

lib = _lib.load()
cdll = ctypes.CDLL(os.environ["GOLF_HIP_LIBRARY"])
cdll.golf_debug_fwdq2_stamps.restype = ctypes.c_int
B = 32
inp = make_inputs(B=B, device="cuda", seed=int(os.environ.get("PHASES_SEED", "2434")))   # 2476 / 2488: a tier-3 utterance
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
st = buf.reshape(2, 64, 64, 8).astype(np.int64)[0]
NG = int((st[0, :, 0] > 0).sum())
s = st[:B, :NG, :] / 100.0
t0 = s[:, :, 0].min()
print("groups", NG, " kernel span %.1f us" % (s[:, :, 7].max() - t0))
print("   g   entry  prologueA  bodyA  response  release   wait  prologueB  bodyB   exit")
t3 = [b for b in range(B) if st[b, 0, 1] == 0]   # tier-3 utterances have no phase A: stamps 0, 6, 7 only
cold = [b for b in range(B) if b not in t3]
for b in t3:
    print("  tier-3 utterance %d: per group  wait + fp64 prologue / chunks / exit (us):" % b,
          " ".join("%.1f/%.1f/%.1f" % (s[b, g, 6] - s[b, g, 0], s[b, g, 7] - s[b, g, 6], s[b, g, 7] - t0) for g in range(NG)))
s = s[cold]
for g in range(NG):
    d = np.diff(s[:, g, :], axis=1).mean(0)
    print("  %2d  %6.2f   %6.2f  %6.2f   %6.2f  %6.2f  %6.2f   %6.2f  %6.2f  %6.2f" % ((g, s[:, g, 0].mean() - t0) + tuple(d) + (s[:, g, 7].mean() - t0,)))
