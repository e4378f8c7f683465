"""dev: phase timing of zp_gemm_kernel<0>'s workgroups (build with -DZPG_TIMING: tools/build_variant.sh zpt noise_fir.hip -DZPG_TIMING;
GOLF_HIP_LIBRARY=.../libgolf_zpt.so).  s_memtime stamps of thread 0: 0 entry, 1 first A chunk staged (after the barrier), 2 K loop
done, 3 accumulators in LDS (after the barrier), 4 stores issued.  G = 6400 frames, n_mag = 256 (the decoder bench) or argv[1]."""
import ctypes, os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from golf_amd import _lib, functional as GF

lib = _lib.load()
cdll = ctypes.CDLL(os.environ["GOLF_HIP_LIBRARY"])
cdll.golf_debug_zpg_stamps.restype = ctypes.c_int
n_mag = int(sys.argv[1]) if len(sys.argv) > 1 else 256
G = 6400
log_mag = (torch.randn(32, 200, n_mag, device="cuda") * 0.3 - 3).contiguous()
window = torch.hann_window(2 * (n_mag - 1), device="cuda")
run = lambda: GF.zero_phase_fir_kernels(log_mag, window)
for _ in range(5):
    k = run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(); e1.record(); torch.cuda.synchronize()
print("one call, HIP events: %.1f us; output %s" % (e0.elapsed_time(e1) * 1e3, tuple(k.shape)))
nwg = (G + 31) // 32 * ((n_mag + 127) // 128)
buf = np.zeros(8 * nwg, dtype=np.uint64)
assert cdll.golf_debug_zpg_stamps(buf.ctypes.data_as(ctypes.POINTER(ctypes.c_ulonglong)), 8 * nwg) == 0
st = buf.reshape(nwg, 8).astype(np.int64)[:, :5]
d = np.diff(st, axis=1)
print("workgroups", nwg, "per-phase mean k ticks (stage first chunk, K loop, acc -> LDS, stores):", np.round(d.mean(0) / 1e3, 2),
      " workgroup lifetime mean %.2f k, p90 %.2f k; kernel span %.2f k ticks; entry spread p50 / p90 / max: %s k" % (
          (st[:, 4] - st[:, 0]).mean() / 1e3, np.percentile(st[:, 4] - st[:, 0], 90) / 1e3, (st[:, 4].max() - st[:, 0].min()) / 1e3,
          np.round(np.percentile(st[:, 0] - st[:, 0].min(), [50, 90, 100]) / 1e3, 2)))
