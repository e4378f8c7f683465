"""Numerics lab (CPU emulation of the chunked GOLF-ss filter's arithmetic).  See lab.c.
   python tools/numlab/lab.py [B] [seed] [rows...]"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
_here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(_here, "liblab.so"))
fp = ctypes.POINTER(ctypes.c_float)
dp = ctypes.POINTER(ctypes.c_double)
P = lambda x, t: x.ctypes.data_as(t) if x is not None else None


def seq32(ex, gain, a, t0, t1, hop, s0=None, y=None):
    F, M = a.shape
    send = np.zeros(M, np.float32)
    lib.seq_f32(P(ex, fp), P(gain, fp), P(a, fp), P(y, fp), t0, t1, F, M, hop, P(s0, fp), P(send, fp))
    return send


def seq64(ex, gain, a, t0, t1, hop, s0=None, y=None):
    F, M = a.shape
    send = np.zeros(M, np.float64)
    lib.seq_f64(P(ex, fp), P(gain, fp), P(a, fp), P(y, dp), t0, t1, F, M, hop, P(s0, dp), P(send, dp))
    return send


def phi_all(a, NP, L, hop, prec):
    F, M = a.shape
    Phi = np.zeros((NP, M, M), np.float64)
    lib.phi_all(P(a, fp), P(Phi, dp), NP, L, F, M, hop, prec)
    return Phi


def chunked(ex, gain, a, T, hop, L, phi, scan_dtype, refine, Phi_cache=None):
    """phi: 'f32' | 'f64r' (fp64 trajectories rounded to fp32) | 'f64' (doubles).  Returns y (float32)."""
    F, M = a.shape
    NC = -(-T // L)
    NP = NC - 1
    key = 32 if phi == 'f32' else 64
    Phi = Phi_cache[key] if Phi_cache is not None and key in Phi_cache else phi_all(a, NP, L, hop, key)
    if Phi_cache is not None:
        Phi_cache[key] = Phi
    if phi in ('f32', 'f64r'):
        Phi = Phi.astype(np.float32)
    Phi = Phi.astype(scan_dtype)
    z = np.zeros((NP, M), np.float32)
    for c in range(NP):
        z[c] = seq32(ex, gain, a, c * L, (c + 1) * L, hop)

    def scan(inp):
        S = np.zeros((NC, M), scan_dtype)
        s = np.zeros(M, scan_dtype)
        for c in range(NP):
            s = (Phi[c] @ s + inp[c].astype(scan_dtype)).astype(scan_dtype)
            S[c + 1] = s
        return S
    S = scan(z)
    if refine:
        z2 = np.zeros((NP, M), np.float32)
        for c in range(NP):
            E = seq32(ex, gain, a, c * L, (c + 1) * L, hop, s0=S[c].astype(np.float32))
            z2[c] = E + z[c] - S[c + 1].astype(np.float32)
        S = scan(z2)
    y = np.zeros(T, np.float32)
    for c in range(NC):
        seq32(ex, gain, a, c * L, min((c + 1) * L, T), hop, s0=S[c].astype(np.float32), y=y)
    return y


def study(rowdata, T, hop, L, variants):
    ex, gain, a = rowdata
    ref = np.zeros(T, np.float64)
    seq64(ex, gain, a, 0, T, hop, y=ref)
    sc = np.abs(ref).max() + 1e-300
    ys = np.zeros(T, np.float32)
    seq32(ex, gain, a, 0, T, hop, y=ys)
    out = {"seq32": np.abs(ys - ref).max() / sc}
    cache = {}
    for name, (phi, sd, rf) in variants.items():
        y = chunked(ex, gain, a, T, hop, L, phi, sd, rf, cache)
        out[name] = np.abs(y.astype(np.float64) - ref).max() / sc
    out["maxphi"] = np.abs(cache[64]).max()
    out["ymax"] = sc
    return out


VARIANTS = {
    "F(f32+ref)": ('f32', np.float32, True),
    "f32 noref": ('f32', np.float32, False),
    "T(f64r,s32)": ('f64r', np.float32, False),
    "T+ref": ('f64r', np.float32, True),
    "D(f64,s64)": ('f64', np.float64, False),
}

if __name__ == "__main__":
    import torch
    from golf_amd.synthetic import make_inputs
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 2434
    rows = [int(r) for r in sys.argv[3:]] or list(range(B))
    inp = make_inputs(B=B, seed=seed)
    T, hop = 47761, 240
    a = inp["a"].numpy(); gain = inp["gain"].numpy(); ex = inp["noise"].numpy()
    for r in rows:
        o = study((np.ascontiguousarray(ex[r]), np.ascontiguousarray(gain[r]), np.ascontiguousarray(a[r])), T, hop, 240, VARIANTS)
        print(f"row {r:3d} max|Phi| {o['maxphi']:9.2e} |y| {o['ymax']:8.1e} seq32 {o['seq32']:.1e} | " +
              " ".join(f"{k} {o[k]:.1e}" for k in VARIANTS), flush=True)


def chunked_delta(ex, gain, a, T, hop, L, phi, sweeps, Phi_cache=None, scan_dtype=np.float32, ret_states=False):
    """delta-form refinement: S <- S + delta, delta_{c+1} = Phi delta_c + (F(S_c) - S_{c+1}); `sweeps` iterations."""
    F, M = a.shape
    NC = -(-T // L); NP = NC - 1
    key = 32 if phi == 'f32' else 64
    Phi = Phi_cache[key] if Phi_cache is not None and key in Phi_cache else phi_all(a, NP, L, hop, key)
    if Phi_cache is not None:
        Phi_cache[key] = Phi
    Phi = Phi.astype(np.float32).astype(scan_dtype)
    z = np.zeros((NP, M), np.float32)
    for c in range(NP):
        z[c] = seq32(ex, gain, a, c * L, (c + 1) * L, hop)

    def scan(inp):
        S = np.zeros((NC, M), scan_dtype); s = np.zeros(M, scan_dtype)
        for c in range(NP):
            s = (Phi[c] @ s + inp[c].astype(scan_dtype)).astype(scan_dtype)
            S[c + 1] = s
        return S
    S = scan(z).astype(np.float32)
    for _ in range(sweeps):
        d = np.zeros((NP, M), np.float32)
        for c in range(NP):
            E = seq32(ex, gain, a, c * L, (c + 1) * L, hop, s0=S[c])
            d[c] = E - S[c + 1]
        S = (S + scan(d).astype(np.float32)).astype(np.float32)
    if ret_states:
        return S
    y = np.zeros(T, np.float32)
    for c in range(NC):
        seq32(ex, gain, a, c * L, min((c + 1) * L, T), hop, s0=S[c], y=y)
    return y
