"""dev: soak test of the sample-wise filter's conditioning tiers over random shapes and coefficient harshness -- two-level scan
(default), flat scan and the serial kernels against the float64 oracle (C restatement), forward and -- every third case --
backward; prints one line per case and the worst ratios.  python tools/fuzz_tiers.py [cases] [seed]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import test_gpu_lpc_ss as T

if os.environ.get("FUZZ_THROUGHPUT"):   # the launch chain of a caller with batches in flight (pair of thin chunk passes)
    from golf_amd import functional as _GF
    _GF.THROUGHPUT_MODE = True
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
bad = 0
for case in range(n):
    B = int(rng.integers(1, 14))
    M = int(rng.choice([6, 12, 16, 20, 22]))
    hop = int(rng.choice([240, 240, 240, 120, 480]))
    F = int(rng.integers(50, 230)) if hop != 480 else int(rng.integers(30, 120))
    sigma = float(rng.choice([0.3, 0.7, 1.0, 1.3]))
    inner = int(rng.integers(1 << 30))
    if os.environ.get("FUZZ_ONLY") and case != int(os.environ["FUZZ_ONLY"]):
        if case % 3 == 0:
            rng.normal(0, 1, (B, (F - 1) * hop + 1))   # (keeps the stream of the backward cases' gy aligned -- approximately: only for replays)
        continue
    ex, gain, a = T.harsh_case(B, F, M, hop, sigma, inner)
    ref = T.oracle_rows(ex, gain, a, hop)
    ok = np.isfinite(ref).all(1) & (np.abs(ref).max(1) < 1e12)
    scale = np.abs(ref).max(1) + 1e-300
    e_ser = np.abs(T.run_mode(ex, gain, a, hop, "serial") - ref).max(1) / scale
    good = ok & (e_ser < 0.05)
    line = f"case {case:3d} B{B} F{F} M{M} hop{hop} sigma{sigma}: good {int(good.sum())}/{B}"
    for mode in (None, "flat-scan"):
        y, st = T.run_status(ex, gain, a, hop, fast=True, mode=mode)
        e = np.abs(y - ref).max(1) / scale
        ratio = float((e[good] / (3 * e_ser[good] + 1e-4)).max()) if good.any() else 0.0
        flag = st["nonfinite"] != (not np.isfinite(y).all()) or ratio > 1.0 or st["fixup_timeout"]
        bad += int(flag)
        line += f" | {mode or 'two-level'}: hot {st['hot_utterances']} t3 {st['tier3_utterances']} ratio {ratio:.2f}{' <-- FAIL' if flag else ''}"
        if os.environ.get("FUZZ_ONLY"):
            print(mode, "max_phi %.3g" % st["max_phi"], "rows e / e_ser:", [(int(i), float("%.2e" % e[i]), float("%.2e" % e_ser[i])) for i in np.nonzero(good)[0]])
    if case % 3 == 0 and good.any():   # backward: two-level and flat adjoint scans against the oracle's closed-form backward
        from oracle import golf_oracle as O

        gy = (rng.normal(0, 1, ref.shape) / scale[:, None]).astype(np.float32)
        gy[~ok] = 0
        ng = int(good.sum())
        want = O.ltv_allpole_ss_backward(gy[good], ex[good], gain[good], a[good], hop)
        ser = T.run_mode(ex, gain, a, hop, "serial", gy)

        def gerr(r, w):
            w = w.reshape(ng, -1)
            r = r[good].reshape(ng, -1)[:, : w.shape[1]]
            return np.abs(r - w).max(1) / (np.abs(w).max(1) + 1e-30)

        for mode in (None, "flat-scan"):
            res = T.run_mode(ex, gain, a, hop, mode, gy)
            worst = 0.0
            for k in (1, 2, 3):
                e_c, e_s = gerr(res[k], want[k - 1]), gerr(ser[k], want[k - 1])
                worst = max(worst, float((e_c / (3 * e_s + 2e-4)).max()))
                if not np.isfinite(res[k][ok]).all():
                    worst = float("inf")
            flag = not (worst <= 1.0)
            bad += int(flag)
            line += f" | bwd {mode or 'two-level'} {worst:.2f}{' <-- FAIL' if flag else ''}"
    print(line, flush=True)
print("failures:", bad)
