"""dev: soak test of the sample-wise filter's conditioning tiers over random shapes and coefficient harshness -- two-level scan
(default), flat scan and the serial kernels against the float64 oracle (C restatement), forward and -- every third case --
backward; prints one line per case and the number of failures.  python tools/fuzz_tiers.py [cases] [seed]
The cases and the check are tests/test_gpu_lpc_ss.py's (soak_cases / soak_case): the suite runs the first 40 of two seeds."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import test_gpu_lpc_ss as T

if os.environ.get("FUZZ_THROUGHPUT"):   # the launch chain of a caller with batches in flight (pair of thin chunk passes)
    from golf_amd import functional as _GF
    _GF.THROUGHPUT_MODE = True
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 7
bad = 0
for case, params, gy in T.soak_cases(seed, n):
    if os.environ.get("FUZZ_ONLY") and case != int(os.environ["FUZZ_ONLY"]):
        continue
    B, F, M, hop, sigma, _ = params
    line = f"case {case:3d} B{B} F{F} M{M} hop{hop} sigma{sigma}:"
    for label, ratio, flag in T.soak_case(params, gy):
        bad += int(flag)
        line += (f" good {ratio}" if label == "good" else f" | {label} {ratio:.2f}{' <-- FAIL' if flag else ''}")
    print(line, flush=True)
print("failures:", bad)
