"""CPU oracle for the GOLF hot path (TEST INFRASTRUCTURE — NOT PRODUCT CODE).

Plain numpy float64 restatements of the reference algorithm for the path named by
BASELINE.json ``north_star``: time-varying LPC synthesis filter + glottal-flow source.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module; the product package ``golf_amd`` never does.

Pinning status
--------------
* The reference's *glue* (padding, unfold, window, OLA, normaliser, truncation, table blend,
  grid construction, ctrl transforms) is pinned: ``oracle/make_fixtures.py`` imports the
  reference's own ``models/*.py`` in the build container and commits its outputs under
  ``tests/golden/``; ``tests/test_oracle_golden.py`` checks every function here against them.
* The third-party arithmetic the reference calls is NOT vendored in /root/reference and not
  installable here: ``torchlpc.sample_wise_lpc`` (unpinned, requirements.txt:19),
  ``torchaudio.functional.lfilter`` (>=2.0.0, requirements.txt:5), ``kazane.Decimate``
  (unpinned, requirements.txt:9).  Their published difference equations are restated below
  (``sample_wise_lpc``, ``lfilter_allpole``); scipy.signal.lfilter is used as an independent
  witness for the LTI case.  The reference repo holds no golden vector for them
   => for those three kernels: **parity unpinned** by reference tests.
  ``kazane.Decimate`` taps are unknown => the decimator takes its taps as an input.

Every function cites the reference file:line it follows (paths relative to /root/reference).
"""
from __future__ import annotations

import math
import numpy as np

__all__ = [
    "linear_upsample",
    "sample_wise_lpc",
    "ltv_allpole_ss_forward",
    "ltv_allpole_ss_backward",
    "lfilter_allpole",
    "hann_window_periodic",
    "lti_frames_ola_forward",
    "lti_frames_ola_backward",
    "ltv_inverse_filter",
    "ltv_inverse_backward",
    "rc2lpc",
    "logits2biquads",
    "biquads2lpc",
    "lf_table_v2",
    "lf_pulse_v1",
    "build_glottal_table",
    "wavetable_generate",
    "indexed_glottal_forward",
    "decimate_fir",
    "default_decimation_taps",
    "source_filter_ss",
    "zero_phase_fir_kernels",
    "ltv_fir_frames_forward",
    "ltv_fir_frames_backward",
    "ltv_fir_precise_forward",
    "ltv_fir_precise_backward",
    "lti_acoustic_filter_forward",
    "lti_acoustic_filter_backward",
    "golf_ss_decoder",
    "biquad_frames_ola_forward",
    "harmonic_oscillator_forward",
    "harmonic_oscillator_backward_amp",
    "wavetable_generate_backward",
    "decimate_fir_adjoint",
    "indexed_glottal_backward",
    "weighted_glottal_forward",
    "weighted_glottal_backward",
    "pulse_train",
    "noise_band_forward",
    "noise_band_backward",
]


# --------------------------------------------------------------------------------------
# a-3  frame -> sample linear upsampling
# --------------------------------------------------------------------------------------
def linear_upsample(z: np.ndarray, hop: int, axis: int = 1) -> np.ndarray:
    """models/utils.py:538-544 (F.interpolate linear, align_corners=True) as used by
    AudioTensor.reduce_hop_length (models/utils.py:171-191).

    out[n] = z[f]*(1-w) + z[f+1]*w, f = n // hop, w = (n % hop)/hop, length (F-1)*hop+1.
    """
    z = np.asarray(z, dtype=np.float64)
    z = np.moveaxis(z, axis, -1)
    F = z.shape[-1]
    if hop == 1 or F == 1:
        return np.moveaxis(z.copy(), -1, axis)
    n = np.arange((F - 1) * hop + 1)
    f = np.minimum(n // hop, F - 2)
    w = (n - f * hop) / hop
    out = z[..., f] * (1.0 - w) + z[..., f + 1] * w
    return np.moveaxis(out, -1, axis)


# --------------------------------------------------------------------------------------
# a-1  sample-wise LTV all-pole (torchlpc.sample_wise_lpc restated) + fused module forward
# --------------------------------------------------------------------------------------
def sample_wise_lpc(x: np.ndarray, A: np.ndarray, zi: np.ndarray | None = None) -> np.ndarray:
    """torchlpc.sample_wise_lpc (third-party, call site models/filters.py:112):
    y[b,t] = x[b,t] - sum_{i=0}^{M-1} A[b,t,i] * y[b,t-1-i], zero initial state.
    x (B,T), A (B,T,M).  zi (B,M) optional: zi[:,i] = y[-1-i].
    """
    x = np.asarray(x, dtype=np.float64)
    A = np.asarray(A, dtype=np.float64)
    B, T = x.shape
    M = A.shape[2]
    ypad = np.zeros((B, T + M), dtype=np.float64)
    if zi is not None:
        ypad[:, :M] = np.asarray(zi, dtype=np.float64)[:, ::-1]
    for t in range(T):
        # window ypad[:, t:t+M] = y[t-M .. t-1]; reversed => y[t-1-i]
        hist = ypad[:, t : t + M][:, ::-1]
        ypad[:, t + M] = x[:, t] - np.einsum("bi,bi->b", A[:, t, :], hist)
    return ypad[:, M:]


def ltv_allpole_ss_forward(ex, gain, a, hop: int) -> np.ndarray:
    """LTVMinimumPhaseFilterPrecise.forward, models/filters.py:99-113.

    ex (B,Tx) hop 1, gain (B,F) hop ``hop``, a (B,F,M) hop ``hop``.
    x = ex * up(gain) (AudioTensor broadcasting truncates to the shorter, utils.py:230-232),
    A = up(a)[:, :len(x)], y = sample_wise_lpc(x, A).  Output length min(Tx, (F-1)*hop+1).
    """
    ex = np.asarray(ex, dtype=np.float64)
    G = linear_upsample(gain, hop, axis=1)
    T = min(ex.shape[1], G.shape[1])
    x = ex[:, :T] * G[:, :T]
    A = linear_upsample(a, hop, axis=1)[:, :T]
    return sample_wise_lpc(x, A)


def _upsample_adjoint(v: np.ndarray, hop: int, F: int) -> np.ndarray:
    """Adjoint of linear_upsample along axis 1: v (B,T,...) -> (B,F,...)."""
    B, T = v.shape[:2]
    out = np.zeros((B, F) + v.shape[2:], dtype=np.float64)
    if hop == 1 or F == 1:
        out[:, :T] = v
        return out
    n = np.arange(T)
    f = np.minimum(n // hop, F - 2)
    w = (n - f * hop) / hop
    wshape = (1, T) + (1,) * (v.ndim - 2)
    np.add.at(out, (slice(None), f), v * (1.0 - w).reshape(wshape))
    np.add.at(out, (slice(None), f + 1), v * w.reshape(wshape))
    return out


def ltv_allpole_ss_backward(gy, ex, gain, a, hop: int):
    """Closed-form gradient of ltv_allpole_ss_forward (what torchlpc's
    autograd Function + autograd through F.interpolate compute in the reference).

    g[t] = gy[t] - sum_i A[t+1+i, i] * g[t+1+i]          (reverse-time recursion)
    d/d ex[t]   = g[t] * G[t]
    d/d gain[f] = up^T(g * ex)[f]
    d/d a[f,i]  = up^T(-g[t] * y[t-1-i])[f,i]
    Returns (g_ex (B,Tx), g_gain (B,F), g_a (B,F,M)); g_ex is zero beyond the output length.
    """
    gy = np.asarray(gy, dtype=np.float64)
    ex = np.asarray(ex, dtype=np.float64)
    a = np.asarray(a, dtype=np.float64)
    B, F, M = a.shape
    G = linear_upsample(gain, hop, axis=1)
    T = min(ex.shape[1], G.shape[1])
    G = G[:, :T]
    A = linear_upsample(a, hop, axis=1)[:, :T]
    x = ex[:, :T] * G
    y = sample_wise_lpc(x, A)
    gpad = np.zeros((B, T + M), dtype=np.float64)  # gpad[:, t] = g[t], zeros beyond T
    for t in range(T - 1, -1, -1):
        acc = gy[:, t].copy()
        for i in range(M):
            tt = t + 1 + i
            if tt < T:
                acc -= A[:, tt, i] * gpad[:, tt]
        gpad[:, t] = acc
    g = gpad[:, :T]
    g_ex = np.zeros_like(ex)
    g_ex[:, :T] = g * G
    g_gain = _upsample_adjoint(g * ex[:, :T], hop, F)
    ypad = np.concatenate([np.zeros((B, M)), y], axis=1)  # ypad[:, M+t] = y[t]
    gA = np.empty((B, T, M), dtype=np.float64)
    for i in range(M):
        gA[:, :, i] = -g * ypad[:, M - 1 - i : M - 1 - i + T]
    g_a = _upsample_adjoint(gA, hop, F)
    return g_ex, g_gain, g_a


# --------------------------------------------------------------------------------------
# a-4  frame-wise LTI all-pole + windowed overlap-add (GOLF-ff)
# --------------------------------------------------------------------------------------
def lfilter_allpole(x: np.ndarray, a: np.ndarray) -> np.ndarray:
    """torchaudio.functional.lfilter(x, [1,a], [1,0..], clamp=False) per row, as called by
    lpc_synthesis (models/lpc.py:11-16) with gains == 1:
    y[r,t] = x[r,t] - sum_i a[r,i] * y[r,t-1-i], zero state.  x (R,W), a (R,M)."""
    x = np.asarray(x, dtype=np.float64)
    a = np.asarray(a, dtype=np.float64)
    R, W = x.shape
    M = a.shape[1]
    ypad = np.zeros((R, W + M), dtype=np.float64)
    for t in range(W):
        hist = ypad[:, t : t + M][:, ::-1]
        ypad[:, t + M] = x[:, t] - np.einsum("ri,ri->r", a, hist)
    return ypad[:, M:]


def hann_window_periodic(W: int) -> np.ndarray:
    """torch.hann_window(W) (periodic=True default) — get_window_fn("hanning"),
    models/utils.py:417-419."""
    n = np.arange(W)
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * n / W)


def lti_frames_ola_forward(ex, gain, a, hop: int, window: np.ndarray, centred: bool = True):
    """LTVMinimumPhaseFilter.forward, models/filters.py:131-184.

    frames of zero-padded x = ex*up(gain); per-frame LTI all-pole from zero state
    (lpc_synthesis, models/lpc.py:11-16); windowed OLA through conv_transpose1d with a
    diag(window) kernel, stride hop, padding W//2, normalised by the OLA of ones
    (filters.py:169-180).  centred=False drops the first hop//2 input samples and
    reflect-pads the output (filters.py:147,181-182).
    Returns (y, norm)."""
    ex = np.asarray(ex, dtype=np.float64)
    a = np.asarray(a, dtype=np.float64)
    window = np.asarray(window, dtype=np.float64)
    W = window.shape[0]
    assert W >= 2 * hop
    pad = W // 2
    G = linear_upsample(gain, hop, axis=1)
    e = ex if centred else ex[:, hop // 2 :]
    T = min(e.shape[1], G.shape[1])
    x = e[:, :T] * G[:, :T]
    B = x.shape[0]
    xp = np.concatenate([np.zeros((B, pad)), x, np.zeros((B, pad))], axis=1)
    nfr = (xp.shape[1] - W) // hop + 1
    assert nfr <= a.shape[1]
    frames = np.stack([xp[:, f * hop : f * hop + W] for f in range(nfr)], axis=1)  # (B,nfr,W)
    filt = lfilter_allpole(frames.reshape(B * nfr, W), a[:, :nfr].reshape(B * nfr, -1)).reshape(
        B, nfr, W
    )
    # conv_transpose1d(stride=hop, padding=pad): full length (nfr-1)*hop + W, then trim pad each side
    full = (nfr - 1) * hop + W
    acc = np.zeros((B, full))
    norm = np.zeros(full)
    for f in range(nfr):
        acc[:, f * hop : f * hop + W] += filt[:, f] * window
        norm[f * hop : f * hop + W] += window
    acc = acc[:, pad : full - pad]
    norm = norm[pad : full - pad]
    y = acc / norm
    if not centred:
        k = hop // 2
        y = np.concatenate([y[:, 1 : k + 1][:, ::-1], y], axis=1)  # F.pad reflect (left)
    return y, norm


def lti_frames_ola_backward(gy, ex, gain, a, hop: int, window: np.ndarray, centred: bool = True):
    """Closed-form gradients of lti_frames_ola_forward (what autograd computes in the reference through the
    reflect pad, conv_transpose1d, lfilter, unfold, zero pad and the gain product; pinned by tests/golden/g15):
      g_q = gy / norm;  per frame g_yf[k] = window[k] * g_q[f*hop + k - pad]
      u_f[k] = g_yf[k] - sum_i a_f[i] * u_f[k+1+i]                 (the all-pole recursion run backwards in time)
      g_a[f,i] = -sum_k u_f[k] * y_f[k-1-i];   g_x = overlap-add of the u_f;   g_ex = g_x*G;   g_gain = up^T(g_x*ex)."""
    gy = np.asarray(gy, dtype=np.float64)
    ex = np.asarray(ex, dtype=np.float64)
    gain = np.asarray(gain, dtype=np.float64)
    a = np.asarray(a, dtype=np.float64)
    window = np.asarray(window, dtype=np.float64)
    W = window.shape[0]
    pad = W // 2
    B, F, M = a.shape
    G = linear_upsample(gain, hop, axis=1)
    off = 0 if centred else hop // 2
    e = ex[:, off:]
    T = min(e.shape[1], G.shape[1])
    x = e[:, :T] * G[:, :T]
    xp = np.concatenate([np.zeros((B, pad)), x, np.zeros((B, pad))], axis=1)
    nfr = (xp.shape[1] - W) // hop + 1
    frames = np.stack([xp[:, f * hop : f * hop + W] for f in range(nfr)], axis=1)
    filt = lfilter_allpole(frames.reshape(B * nfr, W), a[:, :nfr].reshape(B * nfr, -1)).reshape(B, nfr, W)
    full = (nfr - 1) * hop + W
    Ty = full - 2 * pad
    norm = np.zeros(full)
    for f in range(nfr):
        norm[f * hop : f * hop + W] += window
    if centred:
        g_y = gy.copy()
    else:  # adjoint of the left reflect pad: y_out[j] = y[k-j] for j < k, y_out[k+n] = y[n]
        k = hop // 2
        g_y = gy[:, k:].copy()
        for j in range(k):
            g_y[:, k - j] += gy[:, j]
    g_full = np.zeros((B, full))
    g_full[:, pad : pad + Ty] = g_y / norm[pad : pad + Ty]
    g_a = np.zeros_like(a)
    g_xp = np.zeros_like(xp)
    for f in range(nfr):
        g_yf = g_full[:, f * hop : f * hop + W] * window
        u = np.zeros((B, W + M))
        for k in range(W - 1, -1, -1):
            u[:, k] = g_yf[:, k] - np.einsum("bi,bi->b", a[:, f], u[:, k + 1 : k + 1 + M])
        u = u[:, :W]
        ypad = np.concatenate([np.zeros((B, M)), filt[:, f]], axis=1)        # ypad[M+k] = y_f[k]
        for i in range(M):
            g_a[:, f, i] = -np.einsum("bk,bk->b", u, ypad[:, M - 1 - i : M - 1 - i + W])
        g_xp[:, f * hop : f * hop + W] += u
    g_x = g_xp[:, pad : pad + T]
    g_ex = np.zeros_like(ex)
    g_ex[:, off : off + T] = g_x * G[:, :T]
    gG = np.zeros_like(G)
    gG[:, :T] = g_x * e[:, :T]
    g_gain = _upsample_adjoint(gG, hop, F)
    return g_ex, g_gain, g_a


# --------------------------------------------------------------------------------------
# a-5  inverse (analysis) filter
# --------------------------------------------------------------------------------------
def ltv_inverse_filter(y, a, hop: int) -> np.ndarray:
    """LTVMinimumPhaseFilter.reverse, models/filters.py:186-195 + fir_filt utils.py:433-441:
    e[t] = y[t] + sum_i A[t,i] * y[t-1-i] with sample-rate A = up(a)."""
    y = np.asarray(y, dtype=np.float64)
    A = linear_upsample(a, hop, axis=1)
    T = min(y.shape[1], A.shape[1])
    y = y[:, :T]
    A = A[:, :T]
    B, _, M = A.shape
    ypad = np.concatenate([np.zeros((B, M)), y], axis=1)
    e = y.copy()
    for i in range(M):
        e += A[:, :, i] * ypad[:, M - 1 - i : M - 1 - i + T]
    return e


def ltv_inverse_backward(g_e, y, a, hop: int):
    """Closed-form gradients of ltv_inverse_filter w.r.t. y and a (autograd through fir_filt's unfold + matmul and
    F.interpolate in the reference; pinned by tests/golden/g17)."""
    g_e = np.asarray(g_e, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    a = np.asarray(a, dtype=np.float64)
    B, F, M = a.shape
    A = linear_upsample(a, hop, axis=1)
    T = min(y.shape[1], A.shape[1])
    A = A[:, :T]
    ypad = np.concatenate([np.zeros((B, M)), y[:, :T]], axis=1)
    g_y = np.zeros_like(y)
    g_y[:, :T] = g_e
    gA = np.zeros((B, (F - 1) * hop + 1, M))
    for i in range(M):
        prod = A[:, :, i] * g_e                     # contributes to y[t-1-i]
        g_y[:, : T - 1 - i] += prod[:, 1 + i :] if T - 1 - i > 0 else 0
        gA[:, :T, i] = g_e * ypad[:, M - 1 - i : M - 1 - i + T]
    return g_y, _upsample_adjoint(gA, hop, F)


# --------------------------------------------------------------------------------------
# a-2  control transforms
# --------------------------------------------------------------------------------------
def rc2lpc(rc: np.ndarray) -> np.ndarray:
    """models/utils.py:581-593 — Levinson step-up; returns a_1..a_M."""
    rc = np.asarray(rc, dtype=np.float64)
    order = rc.shape[-1]
    if order == 1:
        return rc
    cur = np.concatenate([np.ones_like(rc[..., :1]), rc[..., :1]], axis=-1)
    for n in range(1, order):
        prev = np.concatenate([cur, np.zeros_like(rc[..., :1])], axis=-1)
        cur = prev + rc[..., n : n + 1] * prev[..., ::-1]
    return cur[..., 1:]


def logits2biquads(logits: np.ndarray, rep_type: str, max_abs_pole: float = 0.99) -> np.ndarray:
    """models/utils.py:487-525 get_logits2biquads; logits (...,2) -> (...,3) = [1,a1,a2]."""
    l = np.asarray(logits, dtype=np.float64)
    assert l.shape[-1] == 2
    sig = lambda v: 1.0 / (1.0 + np.exp(-v))
    if rep_type == "coef":
        a1 = np.tanh(l[..., 0]) * max_abs_pole * 2
        a1a = np.abs(a1)
        a2 = 0.5 * ((2 - a1a) * np.tanh(l[..., 1]) * max_abs_pole + a1a)
    elif rep_type == "conj":
        mag = sig(l[..., 0]) * max_abs_pole
        cos = np.tanh(l[..., 1])
        a1 = -2 * mag * cos
        a2 = mag**2
    elif rep_type == "real":
        z1 = np.tanh(l[..., 0]) * max_abs_pole
        z2 = np.tanh(l[..., 1]) * max_abs_pole
        a1 = -z1 - z2
        a2 = z1 * z2
    else:
        raise ValueError(rep_type)
    return np.stack([np.ones_like(a1), a1, a2], axis=-1)


def biquads2lpc(biquads: np.ndarray) -> np.ndarray:
    """models/utils.py:480-484 (+coeff_product :444-460): multiply K second-order sections
    (...,K,3) out to direct form and drop the leading 1 -> (...,2K)."""
    bq = np.asarray(biquads, dtype=np.float64)
    lead = bq.shape[:-2]
    K = bq.shape[-2]
    flat = bq.reshape(-1, K, 3)
    out = np.empty((flat.shape[0], 2 * K + 1))
    for r in range(flat.shape[0]):
        p = np.array([1.0])
        for k in range(K):
            p = np.convolve(p, flat[r, k])
        out[r] = p
    return out.reshape(*lead, 2 * K + 1)[..., 1:]


# --------------------------------------------------------------------------------------
# a-7  LF glottal-flow derivative tables
# --------------------------------------------------------------------------------------
def lf_table_v2(Rd: np.ndarray, points: int = 1024) -> np.ndarray:
    """get_transformed_lf_v2, models/utils.py:363-400 (closed-form LF from R_d)."""
    Rd = np.asarray(Rd, dtype=np.float64).reshape(-1, 1)
    Ra = -0.01 + 0.048 * Rd
    Rk = 0.224 + 0.118 * Rd
    Rg = (Rk / 4) * (0.5 + 1.2 * Rk) / (0.11 * Rd - Ra * (0.5 + 1.2 * Rk))
    Ta = Ra
    Tp = 1 / (2 * Rg)
    Te = Tp + Tp * Rk
    epsilon = 1 / Ta
    shift = np.exp(-epsilon * (1 - Te))
    delta = 1 - shift
    rhs = (1 / epsilon) * (shift - 1) + (1 - Te) * shift
    rhs = rhs / delta
    lower = -(Te - Tp) / 2 + rhs
    upper = -lower
    omega = np.pi / Tp
    s = np.sin(omega * Te)
    y = -np.pi * s * upper / (Tp * 2)
    z = np.log(y)
    alpha = z / (Tp / 2 - Te)
    EO = -1 / (s * np.exp(alpha * Te))
    # torch.linspace(0,1,points+1)[:-1] is computed in float32 in the reference
    t = np.linspace(0.0, 1.0, points + 1)[None, :-1]
    before = EO * np.exp(alpha * t) * np.sin(omega * t)
    after = (-np.exp(-epsilon * (t - Te)) + shift) / delta
    out = np.where(t < Te, before, after)
    return out.squeeze() if out.shape[0] == 1 else out


def lf_pulse_v1(R_d: float = 0.3, T_0: float = 5.0, n_iter_eps: int = 5, n_iter_a: int = 100,
                points: int = 1000) -> np.ndarray:
    """get_transformed_lf, models/utils.py:308-360 (Newton iterations for eps and a)."""
    R_ap = 0.048 * R_d - 0.01
    R_kp = 0.118 * R_d + 0.224
    R_gp = 0.25 * R_kp * (0.5 + 1.2 * R_kp) / (0.11 * R_d - R_ap * (0.5 + 1.2 * R_kp))
    T_a = R_ap * T_0
    T_p = 0.5 * T_0 / R_gp
    T_e = T_p * (R_kp + 1)
    T_b = T_0 - T_e
    omega_g = math.pi / T_p
    E_e = 1
    a = 1
    eps = 1
    for _ in range(n_iter_eps):
        f_eps = eps * T_a + math.expm1(-eps * T_b)
        f_eps_grad = T_a - T_b * math.exp(-eps * T_b)
        eps = abs(eps - f_eps / f_eps_grad)
    E_0 = 0.0
    for _ in range(n_iter_a):
        E_0 = -E_e * math.exp(-a * T_e) / math.sin(omega_g * T_e)
        A_o = E_0 * math.exp(a * T_e) / math.sqrt(omega_g**2 + a**2) * math.sin(
            omega_g * T_e - math.atan(omega_g / a)
        ) + E_0 * omega_g / (omega_g**2 + a**2)
        A_r = -E_e / (eps**2 * T_a) * (1 - math.exp(-eps * T_b) * (1 + eps * T_b))
        f_a = A_o + A_r
        f_a_grad = (1 - 2 * a * A_r / E_e) * math.sin(omega_g * T_e) - omega_g * T_e * math.exp(
            -a * T_e
        )
        a = a - f_a / f_a_grad
    t = np.linspace(0.0, T_0, points + 1)[:-1]
    before_t = t[t < T_e]
    after_t = t[t >= T_e]
    before = E_0 * np.exp(a * before_t) * np.sin(omega_g * before_t)
    after = -E_e / eps / T_a * (np.exp(-eps * (after_t - T_e)) - math.exp(-eps * T_b))
    return np.concatenate([before, after])


def build_glottal_table(table_size=100, table_type="derivative", normalize_method="constant_power",
                        align_peak=True, min_R_d=0.3, max_R_d=2.7, lf_v2=False, **lf_kwargs):
    """GlottalFlowTable.__init__, models/synth.py:59-120.  Returns (R_d_values, table)."""
    Rd = np.exp(np.linspace(math.log(min_R_d), math.log(max_R_d), table_size))
    if lf_v2:
        table = lf_table_v2(Rd, **lf_kwargs)
    else:
        table = np.stack([lf_pulse_v1(R_d=float(r), **lf_kwargs) for r in Rd])
    table = np.array(table, dtype=np.float64)
    if table_type == "flow":
        table = np.cumsum(table, axis=1)
    elif table_type != "derivative":
        raise ValueError(table_type)
    if align_peak:
        peak = table.argmin(axis=1) if table_type == "derivative" else table.argmax(axis=1)
        target = int(peak.max())
        for i in range(table.shape[0]):
            table[i] = np.roll(table[i], target - int(peak[i]))
    if normalize_method == "constant_power":
        table = table / np.linalg.norm(table, axis=1, keepdims=True) * math.sqrt(table.shape[1])
    elif normalize_method == "peak":
        if table_type == "flow":
            table = table / table.max(axis=1, keepdims=True)
    elif normalize_method is not None:
        raise ValueError(normalize_method)
    return Rd, table


# --------------------------------------------------------------------------------------
# a-9 / a-8  wavetable lookup and the indexed glottal oscillator
# --------------------------------------------------------------------------------------
def wavetable_generate(wrapped_phase: np.ndarray, tables: np.ndarray, hop_t: int) -> np.ndarray:
    """GlottalFlowTable.generate, models/synth.py:124-177 (grid_sample bilinear,
    align_corners=True) — SURVEY.md App. A-4.
    wrapped_phase (B,N) in [0,1); tables (B,K,L) at hop ``hop_t``."""
    ph = np.asarray(wrapped_phase, dtype=np.float64)
    tb = np.asarray(tables, dtype=np.float64)
    B, N = ph.shape
    L = tb.shape[2]
    blocks = (N + hop_t - 1) // hop_t
    if tb.shape[1] < blocks + 1:
        reps = blocks + 1 - tb.shape[1]
        tb = np.concatenate([tb, np.repeat(tb[:, -1:], reps, axis=1)], axis=1)
    else:
        tb = tb[:, : blocks + 1]
    tb = np.concatenate([tb, tb[:, :, :1]], axis=2)  # wrap column
    c = ph * L
    c0 = np.floor(c).astype(np.int64)
    c0 = np.clip(c0, 0, L - 1)
    cf = c - c0
    r = np.arange(N) / hop_t
    r0 = np.minimum(np.floor(r).astype(np.int64), blocks - 1) if blocks > 0 else np.zeros(N, np.int64)
    rf = r - r0
    bi = np.arange(B)[:, None]
    r0b = np.broadcast_to(r0[None, :], (B, N))
    top = tb[bi, r0b, c0] * (1 - cf) + tb[bi, r0b, c0 + 1] * cf
    bot = tb[bi, r0b + 1, c0] * (1 - cf) + tb[bi, r0b + 1, c0 + 1] * cf
    return top * (1 - rf[None, :]) + bot * rf[None, :]


def indexed_glottal_forward(phase, phase_hop: int, weight, weight_hop: int, table,
                            oversampling: int = 1, equal_energy: bool = False,
                            phase_offset=None, decim_taps=None):
    """IndexedGlottalFlowTable.forward, models/synth.py:213-263.

    phase (B,Tp) per-sample phase increment (cycles/sample) at hop ``phase_hop``;
    weight (B,Fw) table_select_weight in [0,1] at hop ``weight_hop``; table (n_tab,L).
    Returns dict(instant_phase, pre (the signal handed to the decimator), out).
    ``out`` = pre when oversampling == 1, else decimate_fir(pre, decim_taps, oversampling)
    (None if no taps given: kazane's taps are unknown, parity is asserted on ``pre``)."""
    phase = np.asarray(phase, dtype=np.float64)
    weight = np.asarray(weight, dtype=np.float64)
    table = np.asarray(table, dtype=np.float64)
    n_tab, L = table.shape
    idx_raw = weight * (n_tab - 1)
    i0 = np.clip(np.trunc(idx_raw).astype(np.int64), 0, n_tab - 2)
    p = (idx_raw - i0)[..., None]
    interp_tables = table[i0] * (1 - p) + table[i0 + 1] * p  # (B,Fw,L)
    hop_t = weight_hop
    if oversampling > 1:
        hop_t = weight_hop * oversampling
        phase = phase / oversampling
        phase_hop = phase_hop * oversampling
    up = linear_upsample(phase, phase_hop, axis=1)
    inst = np.cumsum(up, axis=1)
    if phase_offset is not None:
        inst = inst + phase_offset
    wrapped = inst % 1.0
    y = wavetable_generate(wrapped, interp_tables, hop_t)
    if equal_energy:
        y = y / np.sqrt(up)
    out = y
    if oversampling > 1:
        out = decimate_fir(y, decim_taps, oversampling) if decim_taps is not None else None
    return {"instant_phase": inst, "pre": y, "out": out}


def wavetable_generate_backward(g_y, wrapped_phase, tables, hop_t: int):
    """Adjoint of wavetable_generate (what autograd computes through F.grid_sample, models/synth.py:167-176):
    returns (g_phase (B,N) = d/d wrapped_phase, g_tables (B,K,L)).  The bilinear lookup is linear in the table and
    piecewise linear in the phase: d v/d phase = L * [(1-rf)(T[r0,c0+1]-T[r0,c0]) + rf(T[r0+1,c0+1]-T[r0+1,c0])];
    replicated rows (synth.py:141-146) credit the last real row, the wrap column credits column 0."""
    g = np.asarray(g_y, dtype=np.float64)
    ph = np.asarray(wrapped_phase, dtype=np.float64)
    tb0 = np.asarray(tables, dtype=np.float64)
    B, N = ph.shape
    K, L = tb0.shape[1], tb0.shape[2]
    blocks = (N + hop_t - 1) // hop_t
    rows = np.minimum(np.arange(blocks + 1), K - 1)            # padded-image row -> real row (replicate / truncate)
    tb = np.concatenate([tb0[:, rows], tb0[:, rows][:, :, :1]], axis=2)
    c = ph * L
    c0 = np.clip(np.floor(c).astype(np.int64), 0, L - 1)
    cf = c - c0
    r = np.arange(N) / hop_t
    r0 = np.minimum(np.floor(r).astype(np.int64), blocks - 1)
    rf = (r - r0)[None, :]
    bi = np.arange(B)[:, None]
    r0b = np.broadcast_to(r0[None, :], (B, N))
    dtop = tb[bi, r0b, c0 + 1] - tb[bi, r0b, c0]
    dbot = tb[bi, r0b + 1, c0 + 1] - tb[bi, r0b + 1, c0]
    g_phase = g * L * (dtop * (1 - rf) + dbot * rf)
    g_tb = np.zeros((B, K, L), dtype=np.float64)
    bb = np.broadcast_to(bi, (B, N))
    c1 = (c0 + 1) % L
    for rr, wr in ((r0b, 1 - rf), (r0b + 1, rf)):
        real = rows[rr]
        np.add.at(g_tb, (bb, real, c0), g * wr * (1 - cf))
        np.add.at(g_tb, (bb, real, c1), g * wr * cf)
    return g_phase, g_tb


def decimate_fir_adjoint(g_out, taps, q: int, N: int) -> np.ndarray:
    """Adjoint of decimate_fir: g_x[m*q + k - half] += taps[k] * g_out[m]."""
    g_out = np.asarray(g_out, dtype=np.float64)
    taps = np.asarray(taps, dtype=np.float64)
    K = taps.shape[0]
    half = (K - 1) // 2
    B, n_out = g_out.shape
    gp = np.zeros((B, half + N + half + q))
    for k in range(K):
        gp[:, k : k + (n_out - 1) * q + 1 : q] += taps[k] * g_out
    return gp[:, half : half + N]


def indexed_glottal_backward(g_out, phase, phase_hop: int, weight, weight_hop: int, table, oversampling: int = 1,
                             equal_energy: bool = False, phase_offset=None, decim_taps=None, g_is_pre: bool = False):
    """Closed-form gradients of indexed_glottal_forward w.r.t. everything the reference differentiates
    (models/synth.py:213-263 under autograd): table_select_weight, phase, phase_offset, table.
    Returns dict(g_weight (B,Fw), g_phase (B,Tp), g_phase_offset (B,N) or None, g_table (n_tab,L))."""
    phase = np.asarray(phase, dtype=np.float64)
    weight = np.asarray(weight, dtype=np.float64)
    table = np.asarray(table, dtype=np.float64)
    n_tab, L = table.shape
    Fw = weight.shape[1]
    idx_raw = weight * (n_tab - 1)
    i0 = np.clip(np.trunc(idx_raw).astype(np.int64), 0, n_tab - 2)
    p = (idx_raw - i0)[..., None]
    interp_tables = table[i0] * (1 - p) + table[i0 + 1] * p
    hop_t, ph, ph_hop = weight_hop, phase, phase_hop
    if oversampling > 1:
        hop_t, ph, ph_hop = weight_hop * oversampling, phase / oversampling, phase_hop * oversampling
    up = linear_upsample(ph, ph_hop, axis=1)
    inst = np.cumsum(up, axis=1)
    if phase_offset is not None:
        inst = inst + np.asarray(phase_offset, dtype=np.float64)
    wrapped = inst % 1.0
    N = up.shape[1]
    g_y = np.asarray(g_out, dtype=np.float64)
    if oversampling > 1 and not g_is_pre:      # g_is_pre: the gradient is given on the oversampled signal `pre`
        g_y = decimate_fir_adjoint(g_y, decim_taps, oversampling, N)
    g_up = np.zeros_like(up)
    if equal_energy:
        v = wavetable_generate(wrapped, interp_tables, hop_t)
        g_up += g_y * v * (-0.5) * up ** -1.5
        g_v = g_y / np.sqrt(up)
    else:
        g_v = g_y
    g_inst, g_T = wavetable_generate_backward(g_v, wrapped, interp_tables, hop_t)
    g_up += np.cumsum(g_inst[:, ::-1], axis=1)[:, ::-1]           # adjoint of the cumulative sum
    g_phase = _upsample_adjoint(g_up, ph_hop, phase.shape[1]) / (oversampling if oversampling > 1 else 1)
    g_p = np.einsum("bkl,bkl->bk", g_T, table[i0 + 1] - table[i0])
    g_table = np.zeros_like(table)
    np.add.at(g_table, i0, g_T * (1 - p))
    np.add.at(g_table, i0 + 1, g_T * p)
    return {"g_weight": g_p * (n_tab - 1), "g_phase": g_phase,
            "g_phase_offset": g_inst if phase_offset is not None else None, "g_table": g_table}


def weighted_glottal_forward(phase, phase_hop: int, weights, weight_hop: int, table, phase_offset=None):
    """WeightedGlottalFlowTable.forward, models/synth.py:275-294: tables = weights (B,Fw,n_tab) @ table (n_tab,L),
    looked up at the wrapped running phase (no oversampling, no equal-energy scaling)."""
    tables = np.asarray(weights, dtype=np.float64) @ np.asarray(table, dtype=np.float64)
    up = linear_upsample(np.asarray(phase, dtype=np.float64), phase_hop, axis=1)
    inst = np.cumsum(up, axis=1)
    if phase_offset is not None:
        inst = inst + np.asarray(phase_offset, dtype=np.float64)
    return wavetable_generate(inst % 1.0, tables, weight_hop)


def weighted_glottal_backward(g_out, phase, phase_hop: int, weights, weight_hop: int, table):
    """Gradients of weighted_glottal_forward: dict(g_phase, g_weights, g_table)."""
    weights = np.asarray(weights, dtype=np.float64)
    table = np.asarray(table, dtype=np.float64)
    phase = np.asarray(phase, dtype=np.float64)
    up = linear_upsample(phase, phase_hop, axis=1)
    wrapped = np.cumsum(up, axis=1) % 1.0
    g_inst, g_T = wavetable_generate_backward(g_out, wrapped, weights @ table, weight_hop)
    g_up = np.cumsum(g_inst[:, ::-1], axis=1)[:, ::-1]
    return {"g_phase": _upsample_adjoint(g_up, phase_hop, phase.shape[1]), "g_weights": g_T @ table.T,
            "g_table": np.einsum("bkn,bkl->nl", weights, g_T)}


def pulse_train(phase, phase_hop: int, phase_offset=None) -> np.ndarray:
    """PulseTrain.forward, models/synth.py:507-523: rsqrt(increment) where the wrapped running phase steps down."""
    up = linear_upsample(np.asarray(phase, dtype=np.float64), phase_hop, axis=1)
    inst = np.cumsum(up, axis=1)
    if phase_offset is not None:
        off = np.asarray(phase_offset, dtype=np.float64)
        n = min(inst.shape[1], off.shape[1])
        inst, up = inst[:, :n] + off[:, :n], up[:, :n]
    wrapped = inst % 1.0
    out = np.zeros_like(up)
    hit = (wrapped[:, 1:] - wrapped[:, :-1]) < 0
    out[:, 1:][hit] = 1.0 / np.sqrt(up[:, 1:][hit])
    return out


def noise_band_forward(noise_bands, offsets, log_gain, hop: int, T: int) -> np.ndarray:
    """NoiseBand.forward, models/noise.py:114-124, with the random start offsets given:
    out[b,t] = sum_k noise_bands[k, (t + offsets[b,k]) % L] * up(exp(log_gain))[b,t,k], length min(T, (F-1)*hop+1)."""
    nb = np.asarray(noise_bands, dtype=np.float64)
    off = np.asarray(offsets, dtype=np.int64)
    G = linear_upsample(np.exp(np.asarray(log_gain, dtype=np.float64)), hop, axis=1)      # (B, T', K)
    n = min(T, G.shape[1])
    K, L = nb.shape
    idx = (np.arange(n)[None, None, :] + off[:, :, None]) % L                             # (B, K, n)
    noise = nb[np.arange(K)[None, :, None], idx]                                          # (B, K, n)
    return np.einsum("bkt,btk->bt", noise, G[:, :n])


def noise_band_backward(g_out, noise_bands, offsets, log_gain, hop: int) -> np.ndarray:
    """d/d log_gain of noise_band_forward: up^T(g * noise) * exp(log_gain)."""
    nb = np.asarray(noise_bands, dtype=np.float64)
    off = np.asarray(offsets, dtype=np.int64)
    lg = np.asarray(log_gain, dtype=np.float64)
    g = np.asarray(g_out, dtype=np.float64)
    n = g.shape[1]
    K, L = nb.shape
    idx = (np.arange(n)[None, None, :] + off[:, :, None]) % L
    noise = nb[np.arange(K)[None, :, None], idx]
    gG = g[:, :, None] * np.transpose(noise, (0, 2, 1))                                  # (B, n, K)
    return _upsample_adjoint(gG, hop, lg.shape[1]) * np.exp(lg)


def default_decimation_taps(q: int, zeros: int = 16, rolloff: float = 0.945) -> np.ndarray:
    """Own windowed-sinc design standing in for kazane.Decimate(q) (third-party, absent,
    unpinned — SURVEY.md §8c): K = 2*zeros*q+1 taps, cutoff rolloff/(2q), Hann window,
    unity DC gain.  Parity for the decimator is unpinned; the taps are an *input* of both
    the oracle and the HIP kernel so kazane's real kernel can be dropped in."""
    n = np.arange(-zeros * q, zeros * q + 1, dtype=np.float64)
    h = np.sinc(n * rolloff / q) * (0.5 + 0.5 * np.cos(np.pi * n / (zeros * q + 1)))
    return h / h.sum()


def decimate_fir(x: np.ndarray, taps: np.ndarray, q: int) -> np.ndarray:
    """Strided FIR: out[m] = sum_k taps[k] * x[m*q + k - (K-1)/2], zero padding,
    out length (N-1)//q + 1  (conv1d stride q, padding (K-1)/2; correlation, as F.conv1d)."""
    x = np.asarray(x, dtype=np.float64)
    taps = np.asarray(taps, dtype=np.float64)
    K = taps.shape[0]
    assert K % 2 == 1
    half = (K - 1) // 2
    B, N = x.shape
    n_out = (N - 1) // q + 1
    xp = np.concatenate([np.zeros((B, half)), x, np.zeros((B, half + q))], axis=1)
    out = np.zeros((B, n_out))
    for k in range(K):
        out += taps[k] * xp[:, k : k + (n_out - 1) * q + 1 : q]
    return out


# --------------------------------------------------------------------------------------
# a-13  composition (SourceFilterSynth with PassThrough noise/room filters)
# --------------------------------------------------------------------------------------
def source_filter_ss(phase, phase_hop, weight, weight_hop, table, noise, gain, a, hop,
                     oversampling=1, equal_energy=False, decim_taps=None):
    """models/sf.py:35-64 with noise_filter = room_filter = PassThrough,
    subtract_harmonics=False, voicing=None, injected noise:
    src = osc + noise ; y = LTVMinimumPhaseFilterPrecise(src, gain, a)."""
    osc = indexed_glottal_forward(phase, phase_hop, weight, weight_hop, table, oversampling,
                                  equal_energy, None, decim_taps)["out"]
    T = min(osc.shape[1], np.asarray(noise).shape[1])
    src = osc[:, :T] + np.asarray(noise, dtype=np.float64)[:, :T]
    return src, ltv_allpole_ss_forward(src, gain, a, hop)


# --------------------------------------------------------------------------------------
# f-1  zero-phase FIR noise filter
# --------------------------------------------------------------------------------------
def zero_phase_fir_kernels(log_mag: np.ndarray, window: np.ndarray) -> np.ndarray:
    """models/filters.py:294-306 (get_zero_phase_fir + windowing):
    kernel = fftshift(irfft(exp(log_mag))) * window, n_fft = 2*(n_mag-1) taps."""
    mag = np.exp(np.asarray(log_mag, dtype=np.float64))
    fir = np.fft.fftshift(np.fft.irfft(mag, axis=-1), axes=-1)
    return fir * np.asarray(window, dtype=np.float64)


def _fir_frames_geometry(T: int, F: int, N: int, hop: int):
    pad = (N - 1) // 2                       # filters.py:357
    span = N + hop - 1                       # unfold size, filters.py:362
    if T + 2 * pad < span:
        raise ValueError("excitation shorter than one frame span")
    nfr = min((T + 2 * pad - span) // hop + 1, F)   # filters.py:368-369 mutual truncation
    return pad, span, nfr


def ltv_fir_frames_forward(ex, kernel, hop: int) -> np.ndarray:
    """models/filters.py:340-384 LTVZeroPhaseFIRFilter.forward after the kernel is built: zero-pad by
    (N-1)//2 both sides, frames of N+hop-1 samples every hop, one cross-correlation (F.conv1d, groups =
    B*nfr) per frame with that frame's kernel -> hop outputs per frame, concatenated: (B, nfr*hop)."""
    ex = np.asarray(ex, dtype=np.float64)
    kernel = np.asarray(kernel, dtype=np.float64)
    B, T = ex.shape
    _, F, N = kernel.shape
    pad, span, nfr = _fir_frames_geometry(T, F, N, hop)
    xp = np.pad(ex, ((0, 0), (pad, pad)))
    y = np.zeros((B, nfr * hop))
    for f in range(nfr):
        seg = xp[:, f * hop : f * hop + span]
        win = np.lib.stride_tricks.sliding_window_view(seg, N, axis=1)       # (B, hop, N)
        y[:, f * hop : (f + 1) * hop] = np.einsum("bnk,bk->bn", win, kernel[:, f])
    return y


def ltv_fir_frames_backward(gy, ex, log_mag, window, hop: int):
    """Closed-form gradients of zero_phase_fir_kernels + ltv_fir_frames_forward (what autograd computes in
    the reference through conv1d, the window product, fftshift, irfft and exp; pinned by tests/golden/g13)."""
    gy = np.asarray(gy, dtype=np.float64)
    ex = np.asarray(ex, dtype=np.float64)
    log_mag = np.asarray(log_mag, dtype=np.float64)
    window = np.asarray(window, dtype=np.float64)
    B, T = ex.shape
    _, F, n_mag = log_mag.shape
    N = 2 * (n_mag - 1)
    kernel = zero_phase_fir_kernels(log_mag, window)
    pad, span, nfr = _fir_frames_geometry(T, F, N, hop)
    xp = np.pad(ex, ((0, 0), (pad, pad)))
    g_xp = np.zeros_like(xp)
    g_kernel = np.zeros_like(kernel)
    for f in range(nfr):
        g = gy[:, f * hop : (f + 1) * hop]                                    # (B, hop)
        seg = xp[:, f * hop : f * hop + span]
        win = np.lib.stride_tricks.sliding_window_view(seg, N, axis=1)       # (B, hop, N)
        g_kernel[:, f] = np.einsum("bn,bnk->bk", g, win)
        for n in range(hop):
            g_xp[:, f * hop + n : f * hop + n + N] += g[:, n : n + 1] * kernel[:, f]
    g_ex = g_xp[:, pad : pad + T]
    return g_ex, _zero_phase_kernel_adjoint(g_kernel, log_mag, window)


def _zero_phase_kernel_adjoint(g_kernel, log_mag, window):
    """d/d log_mag of zero_phase_fir_kernels: adjoint of window * fftshift * irfft, times exp(log_mag)."""
    n_mag = log_mag.shape[-1]
    N = 2 * (n_mag - 1)
    g_fir = np.fft.ifftshift(g_kernel * window, axes=-1)                      # adjoint of fftshift (N even: same roll)
    # irfft: fir[n] = (1/N) * sum_k c_k mag[k] cos(2 pi k n / N), c_0 = c_{N/2} = 1, else 2
    n = np.arange(N)
    k = np.arange(n_mag)
    c = np.full(n_mag, 2.0)
    c[0] = c[-1] = 1.0
    basis = c[:, None] * np.cos(2 * np.pi * k[:, None] * n[None, :] / N) / N  # (n_mag, N)
    return (g_fir @ basis.T) * np.exp(log_mag)


def ltv_fir_precise_forward(ex, kernel, hop: int) -> np.ndarray:
    """models/filters.py:308-337 LTVZeroPhaseFIRFilterPrecise.forward after the kernel is built: the (B,F,N) kernels
    are linearly upsampled to sample rate (reduce_hop_length), the excitation is padded by (N-1)//2 left and
    N-1-(N-1)//2 right and unfolded sample by sample; y[t] = <pad(ex)[t:t+N], K[t]>, length min(T, (F-1)*hop+1)."""
    ex = np.asarray(ex, dtype=np.float64)
    kernel = np.asarray(kernel, dtype=np.float64)
    B, T = ex.shape
    _, F, N = kernel.shape
    K = linear_upsample(kernel, hop, axis=1)                       # (B, (F-1)*hop+1, N)
    Tout = min(T, K.shape[1])
    pl = (N - 1) // 2
    xp = np.pad(ex, ((0, 0), (pl, N - 1 - pl)))
    win = np.lib.stride_tricks.sliding_window_view(xp, N, axis=1)  # (B, T, N)
    return np.einsum("btk,btk->bt", win[:, :Tout], K[:, :Tout])


def ltv_fir_precise_backward(gy, ex, log_mag, window, hop: int):
    gy = np.asarray(gy, dtype=np.float64)
    ex = np.asarray(ex, dtype=np.float64)
    log_mag = np.asarray(log_mag, dtype=np.float64)
    window = np.asarray(window, dtype=np.float64)
    B, T = ex.shape
    F = log_mag.shape[1]
    kernel = zero_phase_fir_kernels(log_mag, window)
    N = kernel.shape[-1]
    K = linear_upsample(kernel, hop, axis=1)
    Tout = min(T, K.shape[1])
    pl = (N - 1) // 2
    xp = np.pad(ex, ((0, 0), (pl, N - 1 - pl)))
    win = np.lib.stride_tricks.sliding_window_view(xp, N, axis=1)
    gK = np.zeros_like(K)
    gK[:, :Tout] = gy[:, :Tout, None] * win[:, :Tout]
    g_kernel = _upsample_adjoint(gK, hop, F)
    g_xp = np.zeros_like(xp)
    for t in range(Tout):
        g_xp[:, t : t + N] += gy[:, t : t + 1] * K[:, t]
    return g_xp[:, pl : pl + T], _zero_phase_kernel_adjoint(g_kernel, log_mag, window)


# --------------------------------------------------------------------------------------
# f-2  room filter
# --------------------------------------------------------------------------------------
def lti_acoustic_filter_forward(ex, kernel) -> np.ndarray:
    """models/filters.py:426-449 LTIAcousticFilter.forward: y = ex + conv1d(pad(ex[:, :-1], (K, 0)), kernel),
    K = len(kernel) = length-1, i.e. y[t] = ex[t] + sum_k kernel[k]*ex[t-(K-k)]  (strictly causal tail)."""
    ex = np.asarray(ex, dtype=np.float64)
    kernel = np.asarray(kernel, dtype=np.float64)
    K = kernel.shape[0]
    T = ex.shape[1]
    zp = np.pad(ex[:, :-1], ((0, 0), (K, 0)))
    y = ex.copy()
    for k in range(K):
        y += kernel[k] * zp[:, k : k + T]
    return y


def lti_acoustic_filter_backward(gy, ex, kernel):
    gy = np.asarray(gy, dtype=np.float64)
    ex = np.asarray(ex, dtype=np.float64)
    kernel = np.asarray(kernel, dtype=np.float64)
    K = kernel.shape[0]
    T = ex.shape[1]
    zp = np.pad(ex[:, :-1], ((0, 0), (K, 0)))
    g_kernel = np.array([(gy * zp[:, k : k + T]).sum() for k in range(K)])
    g_zp = np.zeros_like(zp)
    for k in range(K):
        g_zp[:, k : k + T] += kernel[k] * gy
    g_ex = gy.copy()
    g_ex[:, :-1] += g_zp[:, K:]
    return g_ex, g_kernel


def golf_ss_decoder(phase, phase_hop, weight, weight_hop, table, noise, log_mag, fir_window, gain, a, hop,
                    room_kernel=None, oversampling=1, equal_energy=False, decim_taps=None):
    """models/sf.py:35-64 as configured by cfg/ae/decoder/golf-precise.yaml: src = osc + noise_filter(noise);
    y = room_filter(end_filter(src)); binary ops truncate to the shorter operand (utils.py:230-232)."""
    osc = indexed_glottal_forward(phase, phase_hop, weight, weight_hop, table, oversampling,
                                  equal_energy, None, decim_taps)["out"]
    nz = np.asarray(noise, dtype=np.float64)[:, : osc.shape[1]]
    fn = ltv_fir_frames_forward(nz, zero_phase_fir_kernels(log_mag, fir_window), hop)
    T = min(osc.shape[1], fn.shape[1])
    src = osc[:, :T] + fn[:, :T]
    y = ltv_allpole_ss_forward(src, gain, a, hop)
    if room_kernel is not None:
        y = lti_acoustic_filter_forward(y, room_kernel)
    return src, y


# --------------------------------------------------------------------------------------
# a-11  harmonic oscillator bank (DDSP / NHV / WORLD / MLSA / SawSing / PULF baselines)
# --------------------------------------------------------------------------------------
def _harmonic_terms(phase, phase_hop, n_harm, n_out, phase_offset=None, po_hop=1, initial_phase=None, deriv=False):
    up = linear_upsample(np.asarray(phase, dtype=np.float64), phase_hop, axis=1)[:, :n_out]   # cycles / sample
    h = np.arange(1, n_harm + 1, dtype=np.float64)
    inst = np.cumsum(up, axis=1)[:, :, None] * h                         # cumsum(h * up) = h * cumsum(up)
    if phase_offset is not None:                                         # synth.py:429-432: + up(offset) * h
        po = linear_upsample(np.asarray(phase_offset, dtype=np.float64), po_hop, axis=1)[:, :n_out]
        inst = inst + po[:, :, None] * h
    if initial_phase is not None:                                        # synth.py:434-435: + initial_phase[b, h]
        inst = inst + np.asarray(initial_phase, dtype=np.float64)[:, None, :]
    mask = (up[:, :, None] * h) < 0.5                                    # anti-aliasing: synth.py:440
    if deriv:                                                            # d sin(2 pi (h Phi + ..)) / d Phi
        return 2.0 * np.pi * h * np.cos(2.0 * np.pi * inst), mask
    return np.sin(2.0 * np.pi * inst), mask


def harmonic_oscillator_forward(phase, phase_hop: int, amplitudes, amp_hop: int, phase_offset=None, po_hop: int = 1,
                                initial_phase=None) -> np.ndarray:
    """HarmonicOscillator.forward, models/synth.py:403-446 (optional phase_offset (B,Fo) at po_hop, initial_phase (B,H)):
    harmonic h runs at h * up(phase) cycles per sample, phase = inclusive cumsum, amplitudes (B,Fa,H) at ``amp_hop``
    are linearly upsampled, zeroed where h * up(phase) >= 0.5, out[t] = sum_h sin(2 pi phase_h[t]) * amp[t,h];
    length = min((Tp-1)*phase_hop+1, (Fa-1)*amp_hop+1)  (mixed-hop truncation, utils.py:230-232)."""
    amplitudes = np.asarray(amplitudes, dtype=np.float64)
    A = linear_upsample(amplitudes, amp_hop, axis=1)
    N = (np.asarray(phase).shape[1] - 1) * phase_hop + 1 if phase_hop > 1 else np.asarray(phase).shape[1]
    n_out = min(N, A.shape[1])
    if phase_offset is not None:
        n_out = min(n_out, (np.asarray(phase_offset).shape[1] - 1) * po_hop + 1 if po_hop > 1 else np.asarray(phase_offset).shape[1])
    sines, mask = _harmonic_terms(phase, phase_hop, amplitudes.shape[-1], n_out, phase_offset, po_hop, initial_phase)
    return np.einsum("bth,bth->bt", sines * mask, A[:, :n_out])


def harmonic_oscillator_backward_offset(gy, phase, phase_hop: int, amplitudes, amp_hop: int, phase_offset, po_hop: int,
                                        initial_phase=None) -> np.ndarray:
    """d/d phase_offset of harmonic_oscillator_forward: up^T( gy * sum_h amp * mask * 2 pi h cos(2 pi phase_h) )."""
    amplitudes = np.asarray(amplitudes, dtype=np.float64)
    gy = np.asarray(gy, dtype=np.float64)
    n_out = gy.shape[1]
    A = linear_upsample(amplitudes, amp_hop, axis=1)[:, :n_out]
    dcos, mask = _harmonic_terms(phase, phase_hop, amplitudes.shape[-1], n_out, phase_offset, po_hop, initial_phase, deriv=True)
    d = np.einsum("bth,bth->bt", dcos * mask, A)
    return _upsample_adjoint(gy * d, po_hop, np.asarray(phase_offset).shape[1])


def harmonic_oscillator_backward_initial_phase(gy, phase, phase_hop: int, amplitudes, amp_hop: int, phase_offset=None,
                                               po_hop: int = 1, initial_phase=None) -> np.ndarray:
    """d/d initial_phase[b,h] of harmonic_oscillator_forward (autograd through models/synth.py:434-446):
    sum_t gy[t] * amp[t,h] * mask * 2 pi cos(2 pi phase_h[t])."""
    amplitudes = np.asarray(amplitudes, dtype=np.float64)
    gy = np.asarray(gy, dtype=np.float64)
    n_out = gy.shape[1]
    H = amplitudes.shape[-1]
    A = linear_upsample(amplitudes, amp_hop, axis=1)[:, :n_out]
    dcos, mask = _harmonic_terms(phase, phase_hop, H, n_out, phase_offset, po_hop, initial_phase, deriv=True)
    dcos = dcos / np.arange(1, H + 1, dtype=np.float64)          # d / d initial_phase: no factor h
    return np.einsum("bt,bth->bh", gy, dcos * mask * A)


def harmonic_oscillator_backward_amp(gy, phase, phase_hop: int, amplitudes_shape, amp_hop: int) -> np.ndarray:
    """d/d amplitudes of harmonic_oscillator_forward (the phase is data)."""
    gy = np.asarray(gy, dtype=np.float64)
    B, Fa, H = amplitudes_shape
    n_out = gy.shape[1]
    sines, mask = _harmonic_terms(phase, phase_hop, H, n_out)
    gA = np.zeros((B, (Fa - 1) * amp_hop + 1 if amp_hop > 1 else Fa, H))
    gA[:, :n_out] = gy[:, :, None] * sines * mask
    return _upsample_adjoint(gA, amp_hop, Fa)


# --------------------------------------------------------------------------------------
# a-6  frame-wise all-pole synthesis as a CASCADE of second-order sections
# --------------------------------------------------------------------------------------
def biquad_frames_ola_forward(ex, gain, biquads, hop: int, window: np.ndarray, pad: int | None = None,
                              frame_gain: bool = True):
    """BatchSecondOrderLPCSynth.forward, models/lpc.py:94-131: zero-pad by (W-hop)//2, unfold into frames of W every
    hop, scale frame f by gain[b,f], run it through the K sections 1/(a0 + a1 z^-1 + a2 z^-2) one after the other
    (torchaudio lfilter with b = [1,0,0], zero state), windowed overlap-add normalised by the overlap-add of the window.
    ``frame_gain=False`` / ``pad=W//2`` give the conventions of LTVMinimumPhaseFilter (gain interpolated to sample rate
    before framing) for the same cascade."""
    ex = np.asarray(ex, dtype=np.float64)
    gain = np.asarray(gain, dtype=np.float64)
    bq = np.asarray(biquads, dtype=np.float64)
    window = np.asarray(window, dtype=np.float64)
    W = window.shape[0]
    pad = (W - hop) // 2 if pad is None else pad
    B = ex.shape[0]
    if frame_gain:
        x = ex
    else:
        G = linear_upsample(gain, hop, axis=1)
        T = min(ex.shape[1], G.shape[1])
        x = ex[:, :T] * G[:, :T]
    xp = np.pad(x, ((0, 0), (pad, pad)))
    nfr = (xp.shape[1] - W) // hop + 1
    assert nfr <= bq.shape[1]
    full = (nfr - 1) * hop + W
    acc = np.zeros((B, full))
    norm = np.zeros(full)
    K = bq.shape[2]
    for f in range(nfr):
        v = xp[:, f * hop : f * hop + W].copy()
        if frame_gain:
            v *= gain[:, f : f + 1]
        for k in range(K):
            a0, a1, a2 = bq[:, f, k, 0], bq[:, f, k, 1], bq[:, f, k, 2]
            y = np.zeros((B, W + 2))
            for n in range(W):
                y[:, n + 2] = (v[:, n] - a1 * y[:, n + 1] - a2 * y[:, n]) / a0
            v = y[:, 2:]
        acc[:, f * hop : f * hop + W] += v * window
        norm[f * hop : f * hop + W] += window
    return acc[:, pad : full - pad] / norm[pad : full - pad]


def biquad_frames_ola_backward(gy, ex, gain, biquads, hop: int, window: np.ndarray, pad: int | None = None,
                               frame_gain: bool = True):
    """Closed-form backward of biquad_frames_ola_forward (the reference differentiates through its K lfilter calls,
    models/lpc.py:115-118; pinned by the reference's own autograd gradients in tests/golden/g26).
    Per frame: u_K = window * gy/norm;  u_{k-1}[n] = (u_k[n] - a1 u_{k-1}[n+1] - a2 u_{k-1}[n+2]) / a0 (section k run
    backwards in time);  d/d a_i of section k = -sum_n u_{k-1}[n] y_k[n-i] with y_k the section's output;  u_0 is the
    gradient w.r.t. the (gain-scaled) input frame.  Returns (g_ex, g_gain, g_biquads)."""
    ex = np.asarray(ex, dtype=np.float64)
    gain = np.asarray(gain, dtype=np.float64)
    bq = np.asarray(biquads, dtype=np.float64)
    window = np.asarray(window, dtype=np.float64)
    gy = np.asarray(gy, dtype=np.float64)
    W = window.shape[0]
    pad = (W - hop) // 2 if pad is None else pad
    B, F = gain.shape
    if frame_gain:
        x = ex
    else:
        G = linear_upsample(gain, hop, axis=1)
        T = min(ex.shape[1], G.shape[1])
        x = ex[:, :T] * G[:, :T]
    xp = np.pad(x, ((0, 0), (pad, pad)))
    nfr = (xp.shape[1] - W) // hop + 1
    full = (nfr - 1) * hop + W
    K = bq.shape[2]
    norm = np.zeros(full)
    for f in range(nfr):
        norm[f * hop: f * hop + W] += window
    gq_full = np.zeros((B, full))
    gq_full[:, pad: full - pad] = gy / norm[pad: full - pad]
    g_xp = np.zeros_like(xp)
    g_gain = np.zeros_like(gain)
    g_bq = np.zeros_like(bq)
    for f in range(nfr):
        v = xp[:, f * hop: f * hop + W].copy()
        if frame_gain:
            v *= gain[:, f: f + 1]
        ys = []
        for k in range(K):   # forward, keeping every section's output (two leading zeros: y[-1], y[-2])
            a0, a1, a2 = bq[:, f, k, 0], bq[:, f, k, 1], bq[:, f, k, 2]
            y = np.zeros((B, W + 2))
            for n in range(W):
                y[:, n + 2] = (v[:, n] - a1 * y[:, n + 1] - a2 * y[:, n]) / a0
            ys.append(y)
            v = y[:, 2:]
        u = gq_full[:, f * hop: f * hop + W] * window
        for k in range(K - 1, -1, -1):
            a0, a1, a2 = bq[:, f, k, 0], bq[:, f, k, 1], bq[:, f, k, 2]
            un = np.zeros((B, W + 2))     # un[:, n] = u_{k-1}[n], two trailing zeros
            for n in range(W - 1, -1, -1):
                un[:, n] = (u[:, n] - a1 * un[:, n + 1] - a2 * un[:, n + 2]) / a0
            y = ys[k]
            for i in range(3):
                g_bq[:, f, k, i] = -(un[:, :W] * y[:, 2 - i: 2 - i + W]).sum(1)
            u = un[:, :W]
        frame = xp[:, f * hop: f * hop + W]
        if frame_gain:
            g_gain[:, f] = (u * frame).sum(1)
            g_xp[:, f * hop: f * hop + W] += u * gain[:, f: f + 1]
        else:
            g_xp[:, f * hop: f * hop + W] += u
    g_x = g_xp[:, pad: xp.shape[1] - pad]
    g_ex = np.zeros_like(ex)
    if frame_gain:
        g_ex[:, : g_x.shape[1]] = g_x
    else:
        T = g_x.shape[1]
        g_ex[:, :T] = g_x * G[:, :T]
        g_gain = _upsample_adjoint(g_x * ex[:, :T], hop, F)
    return g_ex, g_gain, g_bq
