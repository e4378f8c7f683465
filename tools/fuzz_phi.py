"""Which utterances break the time-chunked algorithm?  For fuzz cases (as tools/fuzz_lpc.py) print, per utterance, the
largest transition-matrix entry over all chunks (float64) next to the error of the chunked path and of the sequential
fp32 recursion.  python tools/fuzz_phi.py [cases] [seed]"""
import sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import golf_oracle as O
from golf_amd import functional as GF

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
dev = lambda x: torch.as_tensor(x).cuda()


def max_phi(a_row, F, M, hop, T, L):
    """max |entry| of the chunk transition matrices of one utterance (float64 homogeneous recursion, vectorised over the
    M unit start vectors)."""
    worst = 0.0
    H = None
    for t in range(T):
        if t % L == 0:
            if H is not None:
                worst = max(worst, float(np.abs(H).max()))
            H = np.eye(M)
        f = min(t // hop, F - 2)
        w = (t - f * hop) / hop
        at = a_row[f] * (1 - w) + a_row[f + 1] * w
        H = np.vstack([-(at @ H), H[:-1]])
    return worst


rows = []
for case in range(n_cases):
    M = int(rng.choice([8, 14, 22]))
    hop = int(rng.choice([40, 80, 120, 240]))
    L = hop * (240 // hop)
    T = int(rng.integers(20, 60)) * L - int(rng.integers(0, L))
    F = -(-(T - 1) // hop) + 1
    B = 4
    sigma = float(rng.choice([0.5, 0.8, 1.0, 1.2]))
    walk = float(rng.choice([0.003, 0.01, 0.03])) * (hop / 240) ** 0.5
    logits = rng.normal(0, sigma, (B, 1, M)) + np.cumsum(rng.normal(0, walk, (B, F, M)), 1)
    a64 = O.rc2lpc(np.tanh(logits))
    a = a64.astype(np.float32)
    gain = np.exp(-3 + np.cumsum(rng.normal(0, 0.05, (B, F)), 1)).astype(np.float32)
    ex = rng.normal(0, 1, (B, T)).astype(np.float32)
    ref = O.ltv_allpole_ss_forward(ex, gain, a, hop)
    if not np.isfinite(ref).all():
        continue
    scale = np.abs(ref).max(1) + 1e-30
    ys = GF.ltv_allpole_ss(dev(ex), dev(gain), dev(a), hop, mode="serial").cpu().numpy()
    yc = GF.ltv_allpole_ss(dev(ex), dev(gain), dev(a), hop, mode="flat-scan").cpu().numpy()
    es = np.abs(ys - ref).max(1) / scale
    ec = np.abs(yc - ref).max(1) / scale
    for b in range(B):
        mp = max_phi(a64[b], F, M, hop, T, L)
        rows.append((mp, es[b], ec[b], np.abs(ref[b]).max()))
        print(f"case {case} row {b}: M{M} hop{hop} sigma{sigma}  max|Phi| {mp:9.2e}  |y|max {np.abs(ref[b]).max():8.1e}  serial {es[b]:.1e}  chunked {ec[b]:.1e}", flush=True)
rows.sort()
print("---- sorted by max|Phi|: (max|Phi|, serial, chunked)")
for r in rows:
    print(f"{r[0]:9.2e} {r[1]:.1e} {r[2]:.1e}")
