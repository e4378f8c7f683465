"""Numerics lab, part 2: the algorithm as planned for round 3 -- delta-form refinement, flat or two-level first pass,
per-chunk fp64 fix-up of the transition matrices above a threshold."""
import sys, os
import numpy as np
import lab

f32 = np.float32


def matvec32(Mx, s):
    return (Mx @ s).astype(f32)


def solve(ex, gain, a, T, hop, L, thr, sweeps, two_level, Phi32=None, Phi64=None, per_chunk=True, ret=False, thr2=None):
    F, M = a.shape
    NC = -(-T // L); NP = NC - 1
    if Phi32 is None: Phi32 = lab.phi_all(a, NP, L, hop, 32)
    if Phi64 is None: Phi64 = lab.phi_all(a, NP, L, hop, 64)
    mx = np.abs(Phi32).reshape(NP, -1).max(1)
    mx = np.where(np.isfinite(mx), mx, np.inf)
    if per_chunk:
        hot = mx > thr
        if thr2 is not None and hot.any():   # an utterance with a hot chunk: its other chunks are hot from thr2 on
            hot = mx > thr2
    else:
        hot = np.full(NP, mx.max() > thr)
    Phi = np.where(hot[:, None, None], Phi64, Phi32).astype(f32)
    z = np.zeros((NP, M), f32)
    for c in range(NP):
        z[c] = lab.seq32(ex, gain, a, c * L, (c + 1) * L, hop)
    GS = 16
    NG = -(-NP // GS)
    if two_level:
        comp = []
        for g in range(NG):
            P = np.eye(M)
            for c in range(g * GS, min((g + 1) * GS, NP)):
                P = Phi[c].astype(np.float64) @ P
            comp.append(P.astype(f32))

    def first_pass(inp):
        """states used to start every chunk (S[c], c = 0..NP)"""
        S = np.zeros((NC, M), f32)
        if not two_level:
            s = np.zeros(M, f32)
            for c in range(NP):
                s = (matvec32(Phi[c], s) + inp[c]).astype(f32)
                S[c + 1] = s
            return S
        # group responses from zero state
        V = np.zeros((NG, M), f32)
        for g in range(NG):
            s = np.zeros(M, f32)
            for c in range(g * GS, min((g + 1) * GS, NP)):
                s = (matvec32(Phi[c], s) + inp[c]).astype(f32)
            V[g] = s
        t = np.zeros(M, f32)
        for g in range(NG + 1):
            c0 = g * GS
            if c0 > NP: break
            s = t.copy()
            S[c0] = s
            for c in range(c0, min(c0 + GS, NP)):
                s = (matvec32(Phi[c], s) + inp[c]).astype(f32)
                if c + 1 < c0 + GS: S[c + 1] = s      # the next group's first state comes from the fold
            if g < NG: t = (matvec32(comp[g], t) + V[g]).astype(f32)
        return S
    S = first_pass(z)
    for _ in range(sweeps):
        d = np.zeros((NP, M), f32)
        for c in range(NP):
            E = lab.seq32(ex, gain, a, c * L, (c + 1) * L, hop, s0=S[c])
            d[c] = E - S[c + 1]
        S = (S + first_pass(d)).astype(f32)
    if ret: return S
    y = np.zeros(T, f32)
    for c in range(NC):
        lab.seq32(ex, gain, a, c * L, min((c + 1) * L, T), hop, s0=S[c], y=y)
    return y, int(hot.sum()), mx.max()


def gen(rng, B, F, M, sigma, walk):
    from oracle import golf_oracle as O
    logits = rng.normal(0, sigma, (B, 1, M)) + np.cumsum(rng.normal(0, walk, (B, F, M)), 1)
    a = O.rc2lpc(np.tanh(logits)).astype(f32)
    gain = np.exp(-3 + np.cumsum(rng.normal(0, 0.05, (B, F)), 1)).astype(f32)
    return a, gain


if __name__ == "__main__":
    sigma = float(sys.argv[1]) if len(sys.argv) > 1 else 0.7
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    minphi = float(sys.argv[4]) if len(sys.argv) > 4 else 30
    rng = np.random.default_rng(seed)
    T, hop, M = 47761, 240, 22
    F = 200
    a, gain = gen(rng, n, F, M, sigma, 0.02)
    ex = rng.normal(0, 1, (n, T)).astype(f32)
    for r in range(n):
        ar, gr, er = a[r].copy(), gain[r].copy(), ex[r].copy()
        NP = 199
        P32 = lab.phi_all(ar, NP, 240, hop, 32); P64 = lab.phi_all(ar, NP, 240, hop, 64)
        mp = np.abs(P64).max()
        if not (mp > minphi): continue
        ref = np.zeros(T); lab.seq64(er, gr, ar, 0, T, hop, y=ref); sc = np.abs(ref).max()
        if not np.isfinite(sc) or sc > 1e12:
            print(f"row {r} maxphi {mp:.1e} unstable (|y| {sc:.1e})"); continue
        ys = np.zeros(T, f32); lab.seq32(er, gr, ar, 0, T, hop, y=ys)
        out = [f"row {r:3d} maxphi {mp:8.1e} |y| {sc:7.1e} seq32 {np.abs(ys-ref).max()/sc:.1e} |"]
        for name, kw in {"2L thr30 d1": dict(thr=30, sweeps=1, two_level=True),
                         "2L thr30 d2": dict(thr=30, sweeps=2, two_level=True),
                         "flat thr30 d1": dict(thr=30, sweeps=1, two_level=False),
                         "2L utt30 d1": dict(thr=30, sweeps=1, two_level=True, per_chunk=False),
                         "2L thr10 d1": dict(thr=10, sweeps=1, two_level=True),
                         "2L all64 d1": dict(thr=-1, sweeps=1, two_level=True),
                         "2L all64 d0": dict(thr=-1, sweeps=0, two_level=True)}.items():
            y, nh, _ = solve(er, gr, ar, T, hop, 240, Phi32=P32, Phi64=P64, **kw)
            out.append(f"{name} {np.abs(y-ref).max()/sc:.1e}[{nh}]")
        print(" ".join(out), flush=True)
