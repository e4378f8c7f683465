"""Differential fuzz of the sample-wise filter (GPU): random shapes and random -- also badly conditioned -- coefficient
tracks through every algorithm of the library (serial / chunked two-level / chunked flat; fp32 transitions + refinement
and fp64 transitions) against the float64 oracle.  A case is reported when a path is worse than 1e-4 AND worse than 25 x the
sequential fp32 recursion (the reference's arithmetic) on some utterance.   python tools/fuzz_lpc.py [cases] [seed]"""
import sys, os, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import golf_oracle as O
from golf_amd import functional as GF

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
rings = [(8, (2, 4, 6)), (16, (8, 12, 14)), (24, (8, 16, 20, 22)), (32, (16, 22)), (40, (22,))]
dev = lambda x: torch.as_tensor(x).cuda()
bad = 0
t_start = time.time()
for case in range(n_cases):
    W, orders = rings[rng.integers(len(rings))]
    M = int(orders[rng.integers(len(orders))]) - int(rng.integers(0, 2))           # also orders below the instance's
    M = max(M, 1)
    hop = int(W * rng.integers(1, 11)) if W < 24 else int(rng.choice([W, 2 * W, 5 * W, 10 * W]))
    if hop > 480:
        hop = W * 10 if W * 10 <= 480 else W
    L = hop * (240 // hop) if hop < 240 else hop
    chunks = int(rng.integers(2, 160))
    T = max(2, chunks * L - int(rng.integers(0, L)))
    F = -(-(T - 1) // hop) + 1 + int(rng.integers(0, 2))
    F = max(F, 2)
    B = int(rng.choice([1, 2, 3, 7]))
    sigma = float(rng.choice([0.3, 0.6, 1.0, 1.4]))                                   # 1.4: reflection coefficients near +-1
    walk = float(rng.choice([0.003, 0.01, 0.03])) * (hop / 240) ** 0.5
    logits = rng.normal(0, sigma, (B, 1, M)) + np.cumsum(rng.normal(0, walk, (B, F, M)), 1)
    a = O.rc2lpc(np.tanh(logits)).astype(np.float32)
    gain = np.exp(-3 + np.cumsum(rng.normal(0, 0.05, (B, F)), 1)).astype(np.float32)
    Tx = T + int(rng.integers(0, 3)) * 7
    ex = rng.normal(0, 1, (B, Tx)).astype(np.float32)
    ref = O.ltv_allpole_ss_forward(ex, gain, a, hop)
    if not np.isfinite(ref).all() or np.abs(ref).max() > 1e12:
        continue                                                                      # unstable track: nothing to compare
    scale = np.abs(ref).max(1) + 1e-30
    res = {}
    for name, kw in (("serial", dict(mode="serial")), ("default", dict()), ("flat", dict(mode="flat-scan")),
                     ("fp64-phi", dict(fast_inference=False)), ("chunked", dict(mode="chunked"))):
        try:
            y = GF.ltv_allpole_ss(dev(ex), dev(gain), dev(a), hop, **kw).cpu().numpy()
        except Exception as e:   # noqa: BLE001
            res[name] = "ERR " + str(e)[:80]
            continue
        res[name] = np.abs(y - ref).max(1) / scale if y.shape == ref.shape else "SHAPE %s vs %s" % (y.shape, ref.shape)
    seq = res["serial"] if isinstance(res["serial"], np.ndarray) else None
    line = f"case {case}: B{B} F{F} M{M} hop{hop} T{T} sigma{sigma} |"
    flag = False
    for name, e in res.items():
        if isinstance(e, str):
            line += f" {name}: {e};"
            flag = flag or not ("no kernel" in e or "serial path" in e)
            continue
        base = seq if seq is not None else np.zeros_like(e)
        worst = float((e - np.maximum(1e-4, 25 * base)).max())
        line += f" {name} {e.max():.1e}"
        if worst > 0:
            flag = True
            line += "(!)"
    if flag:
        bad += 1
        print("SUSPECT", line, flush=True)
    elif case % 10 == 0:
        print(line, flush=True)
print(f"{n_cases} cases, {bad} suspect, {time.time() - t_start:.0f} s")
