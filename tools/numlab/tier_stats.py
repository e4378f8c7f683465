"""dev (CPU): per-utterance conditioning statistics of the chunk maps -- largest entry, chunks beyond 10 / 16 / 20 / 30, largest
16-chunk group sum of log2(max entry) -- for rows of tools/fuzz_tiers.py cases and for utterances of the benchmark recipe.
python tools/numlab/tier_stats.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import lab

def stats(a, hop, T, L=240):
    NC = -(-T // L); NP = NC - 1
    Phi = lab.phi_all(np.ascontiguousarray(a, np.float32), NP, L, hop, 64)
    mx = np.abs(Phi).reshape(NP, -1).max(1)
    lg = np.log2(np.maximum(mx, 1.0))
    NG = -(-NP // 16)
    gs = np.array([lg[g * 16:(g + 1) * 16].sum() for g in range(NG)])
    return dict(max=mx.max(), n10=int((mx > 10).sum()), n16=int((mx > 16).sum()), n20=int((mx > 20).sum()), n30=int((mx > 30).sum()),
                gmax=gs.max(), tot=lg.sum(), NP=NP)

def show(tag, s):
    print(f"{tag}: max {s['max']:8.3g} n>10 {s['n10']:3d} n>16 {s['n16']:3d} n>20 {s['n20']:3d} n>30 {s['n30']:3d} of {s['NP']:3d}  gmax {s['gmax']:6.1f} tot {s['tot']:7.1f}")

if __name__ == "__main__":
    import torch
    what = sys.argv[1] if len(sys.argv) > 1 else "fail"
    if what == "fail":
        import importlib
        sys.argv = sys.argv[:1]
        import test_gpu_lpc_ss as T
        for seed, case, rows in ((31, 11, [12, 1, 8, 9]), (31, 115, [3, 4, 9]), (909, 55, [4, 6]), (606, 90, [3]), (808, 57, [0])):
            (B, F, M, hop, sigma, inner), _ = T.fuzz_case(seed, case)
            ex, gain, a = T.harsh_case(B, F, M, hop, sigma, inner)
            for r in rows:
                show(f"seed {seed} case {case} row {r}", stats(a[r], hop, ex.shape[1]))
    else:
        from golf_amd.synthetic import make_inputs
        n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
        for k in range(n):
            inp = make_inputs(B=32, seed=2434 + k)
            a = inp["a"].numpy()
            for b in range(32):
                s = stats(a[b], 240, 47761)
                if s["max"] > 8:
                    show(f"recipe seed {2434 + k} row {b}", s)
