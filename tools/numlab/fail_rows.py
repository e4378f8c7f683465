"""dev (CPU): the failing rows of the round-5 soak through the numerics lab's emulation of the chunked algorithm -- which knob
(thresholds, a second sweep, every map from fp64 trajectories) moves the error, and how it sits against the sequential fp32 recursion."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import lab, lab2
import test_gpu_lpc_ss as T
f32 = np.float32
for seed, case, rows in ((31, 11, [12]), (31, 115, [3]), (909, 55, [4]), (808, 57, [0])):
    (B, F, M, hop, sigma, inner), _ = T.fuzz_case(seed, case)
    ex, gain, a = T.harsh_case(B, F, M, hop, sigma, inner)
    Tn = ex.shape[1]
    for r in rows:
        ar, gr, er = a[r].copy(), gain[r].copy(), ex[r].copy()
        NP = -(-Tn // 240) - 1
        P32 = lab.phi_all(ar, NP, 240, hop, 32); P64 = lab.phi_all(ar, NP, 240, hop, 64)
        ref = np.zeros(Tn); lab.seq64(er, gr, ar, 0, Tn, hop, y=ref); sc = np.abs(ref).max()
        ys = np.zeros(Tn, f32); lab.seq32(er, gr, ar, 0, Tn, hop, y=ys)
        es = np.abs(ys - ref).max() / sc
        out = [f"seed {seed} case {case} row {r}: seq32 {es:.2e} |"]
        for name, kw in {"2L 30/10 d1": dict(thr=30, thr2=10, sweeps=1, two_level=True),
                         "flat 30/10 d1": dict(thr=30, thr2=10, sweeps=1, two_level=False),
                         "2L 30/10 d2": dict(thr=30, thr2=10, sweeps=2, two_level=True),
                         "flat 30/10 d2": dict(thr=30, thr2=10, sweeps=2, two_level=False),
                         "2L 20/8 d1": dict(thr=20, thr2=8, sweeps=1, two_level=True),
                         "flat 20/8 d1": dict(thr=20, thr2=8, sweeps=1, two_level=False),
                         "2L all64 d1": dict(thr=-1, sweeps=1, two_level=True),
                         "flat all64 d1": dict(thr=-1, sweeps=1, two_level=False),
                         "flat all64 d2": dict(thr=-1, sweeps=2, two_level=False)}.items():
            y, nh, _ = lab2.solve(er, gr, ar, Tn, hop, 240, Phi32=P32, Phi64=P64, **kw)
            e = np.abs(y - ref).max() / sc
            out.append(f"{name} {e:.1e} ({e / (3 * es + 1e-4):.2f}) [{nh}] |")
        print(" ".join(out), flush=True)
