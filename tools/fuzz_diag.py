"""dev: per-row view of one case of tools/fuzz_tiers.py -- error of the chunked scans (two-level, flat) and of the serial kernels
against the float64 oracle, each row's tier and largest map entry (the row run alone), forward and backward.
python tools/fuzz_diag.py seed case [bwd]      (threshold knobs: GOLF_SS_PHI_GUARD / _GUARD2 / _GUARD3 / GOLF_SS_GROUP_LOG2)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import test_gpu_lpc_ss as T

seed, case = int(sys.argv[1]), int(sys.argv[2])
(B, F, M, hop, sigma, inner), gy = T.fuzz_case(seed, case, with_gy=len(sys.argv) > 3)
print(f"seed {seed} case {case}: B{B} F{F} M{M} hop{hop} sigma{sigma} inner {inner}")
ex, gain, a = T.harsh_case(B, F, M, hop, sigma, inner)
ref = T.oracle_rows(ex, gain, a, hop)
ok = np.isfinite(ref).all(1) & (np.abs(ref).max(1) < 1e12)
scale = np.abs(ref).max(1) + 1e-300
e_ser = np.abs(T.run_mode(ex, gain, a, hop, "serial") - ref).max(1) / scale
good = ok & (e_ser < 0.05)
rows = {}
for mode in (None, "flat-scan"):
    y, st = T.run_status(ex, gain, a, hop, fast=True, mode=mode)
    rows[mode] = np.abs(y - ref).max(1) / scale
for b in range(B):
    _, st = T.run_status(ex[b:b + 1], gain[b:b + 1], a[b:b + 1], hop, fast=True)
    tier = 3 if st["tier3_utterances"] else (2 if st["hot_utterances"] else 1)
    r2, rf = (rows[m][b] / (3 * e_ser[b] + 1e-4) for m in (None, "flat-scan"))
    print(f"  row {b:2d} good {int(good[b])} tier {tier} max_phi {st['max_phi']:9.3g} e_ser {e_ser[b]:.2e} two-level {rows[None][b]:.2e} ({r2:.2f}) flat {rows['flat-scan'][b]:.2e} ({rf:.2f})")
if gy is not None:
    from oracle import golf_oracle as O
    gy = (gy / scale[:, None]).astype(np.float32)
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
        for k, name in ((1, "g_ex"), (2, "g_gain"), (3, "g_a")):
            e_c, e_s = gerr(res[k], want[k - 1]), gerr(ser[k], want[k - 1])
            print(f"  bwd {mode or 'two-level'} {name}: ratio per good row", np.round(e_c / (3 * e_s + 2e-4), 2), "e_ser", np.array2string(e_s, precision=1))
