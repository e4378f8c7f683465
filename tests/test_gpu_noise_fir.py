"""GPU parity of the zero-phase FIR noise filter (golf_zero_phase_fir_* + golf_ltv_fir_frames_*; reference
LTVZeroPhaseFIRFilter, models/filters.py:286-384): golden vectors produced by the reference itself (g13), the
float64 oracle at larger sizes incl. the BASELINE shape, gradients, and size-independent properties."""
import numpy as np
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-4


def dev(x, grad=False):
    return torch.as_tensor(np.asarray(x), dtype=torch.float32).cuda().requires_grad_(grad)


def check(y, ref, what, tol=TOL):
    y = np.asarray(y)
    emax, el2 = rel_err(y, ref)
    print(f"{what}: rel-max {emax:.3e} rel-l2 {el2:.3e}")
    assert np.isfinite(y).all()
    assert emax <= tol and el2 <= tol, (what, emax, el2)


def np_window(name, N):
    fn = {"hanning": torch.hann_window, "hamming": torch.hamming_window}[name]
    return fn(N, dtype=torch.float64).numpy()


def run_module(ex, log_mag, hop, window="hanning", gy=None):
    from golf_amd.audiotensor import AudioTensor
    from golf_amd.filters import LTVZeroPhaseFIRFilter

    m = LTVZeroPhaseFIRFilter(window=window, conv_method="direct", n_mag=log_mag.shape[-1]).cuda()
    x, lm = dev(ex, gy is not None), dev(log_mag, gy is not None)
    y = m(AudioTensor(x), AudioTensor(lm, hop))
    assert y.hop_length == 1
    yt = y.as_tensor()
    if gy is None:
        torch.cuda.synchronize()
        return yt.detach().cpu().numpy()
    (yt * dev(gy)).sum().backward()
    torch.cuda.synchronize()
    return yt.detach().cpu().numpy(), x.grad.cpu().numpy(), lm.grad.cpu().numpy()


@pytest.mark.parametrize("tag,window", [("a", "hanning"), ("b", "hanning"), ("c", "hanning"), ("h", "hamming")])
def test_golden_g13(golden, tag, window):
    from golf_amd import functional as GF

    g = golden("g13_zero_phase_fir")
    ex, lm, hop = g[f"{tag}_ex"], g[f"{tag}_log_mag"], int(g[f"{tag}_hop"])
    N = 2 * (lm.shape[-1] - 1)
    k = GF.zero_phase_fir_kernels(dev(lm), dev(np_window(window, N))).cpu().numpy()
    check(k, g[f"{tag}_kernel"], f"g13{tag} kernel", 2e-6)
    y, gx, glm = run_module(ex, lm, hop, window, gy=g[f"{tag}_gy"])
    check(y, g[f"{tag}_y"], f"g13{tag} y", 1e-5)
    check(gx, g[f"{tag}_g_ex"], f"g13{tag} g_ex", 1e-5)
    check(glm, g[f"{tag}_g_log_mag"], f"g13{tag} g_log_mag", 1e-5)


def case(B, T, F, n_mag, seed):
    rng = np.random.default_rng(seed)
    ex = rng.normal(0, 1, (B, T)).astype(np.float32)
    # smooth spectral envelope + frame-to-frame drift, 40 dB of dynamic range
    base = np.cumsum(rng.normal(0, 0.25, (B, 1, n_mag)), axis=-1)
    drift = np.cumsum(rng.normal(0, 0.05, (B, F, n_mag)), axis=1)
    lm = (base + drift - 2.0).clip(-6, 3).astype(np.float32)
    return ex, lm


@pytest.mark.parametrize("B,T,F,n_mag,hop", [(3, 2000, 9, 65, 240), (2, 1500, 20, 33, 64), (2, 999, 3, 256, 240),
                                             (1, 3000, 5, 129, 600), (2, 700, 30, 17, 24), (5, 480, 2, 9, 240)])
def test_fwd_bwd_vs_oracle(B, T, F, n_mag, hop):
    from oracle import golf_oracle as O

    ex, lm = case(B, T, F, n_mag, seed=T + n_mag)
    N = 2 * (n_mag - 1)
    win = np_window("hanning", N)
    ref = O.ltv_fir_frames_forward(ex, O.zero_phase_fir_kernels(lm, win), hop)
    gy = np.random.default_rng(7).normal(0, 1, ref.shape).astype(np.float32)
    y, gx, glm = run_module(ex, lm, hop, gy=gy)
    assert y.shape == ref.shape
    check(y, ref, f"fir fwd B{B} T{T} F{F} n_mag{n_mag} hop{hop}", 1e-5)
    rgx, rglm = O.ltv_fir_frames_backward(gy, ex, lm, win, hop)
    check(gx, rgx, "g_ex", 2e-5)
    check(glm, rglm, "g_log_mag", 2e-5)


def test_full_size_config():
    """The noise branch of BASELINE configs[1]: B=32, 2 s @ 24 kHz, F=200 frames, n_mag=256 (510 taps), hop 240."""
    from oracle import golf_oracle as O

    B, T, F, n_mag, hop = 32, 48000, 200, 256, 240
    ex, lm = case(B, T, F, n_mag, seed=2434)
    win = np_window("hanning", 510)
    gy = np.random.default_rng(11).normal(0, 1, (B, 199 * hop)).astype(np.float32)
    y, gx, glm = run_module(ex, lm, hop, gy=gy)
    assert y.shape == (B, 47760)
    nb = 4  # oracle on a slice of the batch (utterances are independent)
    ref = O.ltv_fir_frames_forward(ex[:nb], O.zero_phase_fir_kernels(lm[:nb], win), hop)
    check(y[:nb], ref, "full-size fwd", 1e-5)
    rgx, rglm = O.ltv_fir_frames_backward(gy[:nb], ex[:nb], lm[:nb], win, hop)
    check(gx[:nb], rgx, "full-size g_ex", 2e-5)
    check(glm[:nb], rglm, "full-size g_log_mag", 2e-5)
    # size-independent properties over the whole batch
    assert np.all(glm[:, 199:] == 0), "the 200th frame is never used (unfold yields 199 frames)"
    y2 = run_module(2.0 * ex, lm, hop)
    np.testing.assert_allclose(y2, 2.0 * y, rtol=0, atol=1e-5 * np.abs(y).max())  # linear in the excitation
    y3 = run_module(ex, lm + np.log(3.0).astype(np.float32), hop)
    check(y3, 3.0 * y.astype(np.float64), "homogeneous in the magnitude", 1e-5)


def test_flat_spectrum_is_a_delayed_delta():
    """log_mag = 0 -> irfft = unit impulse at n = 0 -> after fftshift tap N/2 -> y[t] = window[N/2] * ex[t + N/2 - P]."""
    B, T, F, n_mag, hop = 2, 1000, 5, 33, 240
    N, P = 64, 31
    ex = np.random.default_rng(0).normal(0, 1, (B, T)).astype(np.float32)
    y = run_module(ex, np.zeros((B, F, n_mag), np.float32), hop)
    w = float(np_window("hanning", N)[N // 2])
    ref = w * ex[:, N // 2 - P: N // 2 - P + y.shape[1]]
    np.testing.assert_allclose(y, ref, rtol=0, atol=2e-6)


def test_errors():
    from golf_amd import _lib
    from golf_amd import functional as GF

    ex = torch.zeros(2, 100, device="cuda")
    lm = torch.zeros(2, 4, 129, device="cuda")
    with pytest.raises(_lib.GolfError):  # shorter than one frame span
        GF.zero_phase_fir_filter(ex, lm, torch.hann_window(256).cuda(), 240)
    with pytest.raises(_lib.GolfError):  # window / n_mag mismatch
        GF.zero_phase_fir_filter(torch.zeros(2, 1000, device="cuda"), lm, torch.hann_window(100).cuda(), 240)
    with pytest.raises(_lib.GolfError):  # CPU tensors: there is no CPU path
        GF.zero_phase_fir_filter(torch.zeros(2, 1000), lm.cpu(), torch.hann_window(256), 240)


def test_generic_frames_match_torch_conv():
    """golf_ltv_fir_frames_fwd_f32 with arbitrary (asymmetric) per-frame kernels vs a plain torch fp64 grouped
    cross-correlation on the GPU tensors' CPU copies."""
    from golf_amd import functional as GF

    rng = np.random.default_rng(5)
    B, T, F, N, hop = 3, 900, 7, 50, 120
    ex = rng.normal(0, 1, (B, T))
    kern = rng.normal(0, 1, (B, F, N))
    y = GF.ltv_fir_frames(dev(ex), dev(kern), hop).cpu().numpy()
    P = (N - 1) // 2
    xp = np.pad(ex, ((0, 0), (P, P)))
    nfr = y.shape[1] // hop
    ref = np.zeros_like(y, dtype=np.float64)
    for f in range(nfr):
        for n in range(hop):
            ref[:, f * hop + n] = (xp[:, f * hop + n: f * hop + n + N] * kern[:, f]).sum(-1)
    check(y, ref, "generic frames", 1e-5)


@pytest.mark.parametrize("N,KS", [(250, 256), (252, 256), (242, 256), (500, 512), (50, 64)])
def test_kernel_gradient_writes_every_padding_tap(N, KS):
    """the kernel-gradient launch took its pass count from N, so with N just under a multiple of 252 and rows padded
    beyond it (N = 250 / 252 with 256-tap rows, 500 with 512) the taps [npass * 252, KS) of g_kern were never written and the
    caller's torch.empty memory came back as a gradient.  Through the C ABI with a NaN-poisoned gradient buffer: every tap below N
    equals the float64 correlation, every padding tap is exactly 0."""
    from golf_amd import _lib

    lib = _lib.load()
    rng = np.random.default_rng(N)
    B, F, hop = 2, 3, 120
    T = F * hop + 40
    nfr = lib.golf_ltv_fir_frames_length(T, F, N, hop) // hop      # frames the excitation reaches (the last kernel rows may go unused)
    assert 1 <= nfr <= F
    ex = rng.normal(0, 1, (B, T))
    kern = np.zeros((B * F, KS))
    kern[:, :N] = rng.normal(0, 1, (B * F, N))
    gy = rng.normal(0, 1, (B, nfr * hop))
    ex_d, kern_d, gy_d = dev(ex), dev(kern), dev(gy)
    g_kern = torch.full((B * F, KS), float("nan"), dtype=torch.float32, device="cuda")
    _lib.check(lib.golf_ltv_fir_frames_bwd_f32(gy_d.data_ptr(), gy_d.stride(0), ex_d.data_ptr(), ex_d.stride(0), kern_d.data_ptr(), KS,
                                               None, 0, g_kern.data_ptr(), B, T, F, N, hop, 0, _lib.stream_ptr()),
               "golf_ltv_fir_frames_bwd_f32")
    torch.cuda.synchronize()
    got = g_kern.cpu().numpy().reshape(B, F, KS)
    assert np.isfinite(got).all(), "unwritten taps: %s" % np.argwhere(~np.isfinite(got))[:4]
    assert np.all(got[:, :, N:] == 0.0)
    P = (N - 1) // 2
    xp = np.pad(ex, ((0, 0), (P, N)))
    ref = np.zeros((B, F, N))
    for f in range(nfr):
        for n in range(hop):
            ref[:, f] += gy[:, f * hop + n, None] * xp[:, f * hop + n: f * hop + n + N]
    check(got[:, :nfr, :N], ref[:, :nfr], f"g_kern N={N} KS={KS} ({nfr} of {F} frames used)", 2e-5)
    assert np.all(got[:, nfr:] == 0.0)


def run_precise(ex, log_mag, hop, gy=None):
    from golf_amd.audiotensor import AudioTensor
    from golf_amd.filters import LTVZeroPhaseFIRFilterPrecise

    m = LTVZeroPhaseFIRFilterPrecise(window="hanning", n_mag=log_mag.shape[-1]).cuda()
    x, lm = dev(ex, gy is not None), dev(log_mag, gy is not None)
    yt = m(AudioTensor(x), AudioTensor(lm, hop)).as_tensor()
    if gy is None:
        return yt.detach().cpu().numpy()
    (yt * dev(gy)).sum().backward()
    torch.cuda.synchronize()
    return yt.detach().cpu().numpy(), x.grad.cpu().numpy(), lm.grad.cpu().numpy()


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_golden_g16_precise(golden, tag):
    """LTVZeroPhaseFIRFilterPrecise (kernels interpolated to sample rate) against the reference's own run."""
    g = golden("g16_zero_phase_fir_precise")
    y, gx, glm = run_precise(g[f"{tag}_ex"], g[f"{tag}_log_mag"], int(g[f"{tag}_hop"]), gy=g[f"{tag}_gy"])
    check(y, g[f"{tag}_y"], f"g16{tag} y", 1e-5)
    check(gx, g[f"{tag}_g_ex"], f"g16{tag} g_ex", 1e-5)
    check(glm, g[f"{tag}_g_log_mag"], f"g16{tag} g_log_mag", 1e-5)


@pytest.mark.parametrize("B,T,F,n_mag,hop", [(2, 2500, 11, 65, 240), (2, 1000, 6, 256, 240), (1, 700, 30, 17, 24)])
def test_precise_vs_oracle(B, T, F, n_mag, hop):
    from oracle import golf_oracle as O

    ex, lm = case(B, T, F, n_mag, seed=T)
    win = np_window("hanning", 2 * (n_mag - 1))
    ref = O.ltv_fir_precise_forward(ex, O.zero_phase_fir_kernels(lm, win), hop)
    gy = np.random.default_rng(9).normal(0, 1, ref.shape).astype(np.float32)
    y, gx, glm = run_precise(ex, lm, hop, gy=gy)
    assert y.shape == ref.shape
    check(y, ref, "precise fwd", 1e-5)
    rgx, rglm = O.ltv_fir_precise_backward(gy, ex, lm, win, hop)
    check(gx, rgx, "precise g_ex", 2e-5)
    check(glm, rglm, "precise g_log_mag", 2e-5)


def test_random_shape_sweep():
    """30 random shapes of the zero-phase FIR noise filter (bins, hop, frames, excitation length, batch): forward and
    both gradients against the float64 oracle."""
    from oracle import golf_oracle as O

    rng = np.random.default_rng(404)
    worst = 0.0
    done = 0
    for trial in range(40):
        n_mag = int(rng.choice([5, 9, 17, 33, 65, 129, 256, 257]))
        hop = int(rng.choice([4, 8, 24, 60, 240, 256]))
        B = int(rng.integers(1, 4))
        F = int(rng.integers(2, 25))
        N = 2 * (n_mag - 1)
        T = int(rng.integers(max(N, hop) + 1, (F + 2) * hop + N))
        ex, lm = case(B, T, F, n_mag, seed=1000 + trial)
        win = np_window("hanning", N)
        try:
            ref = O.ltv_fir_frames_forward(ex, O.zero_phase_fir_kernels(lm, win), hop)
        except (AssertionError, ValueError):
            continue                                  # a shape the reference's unfold rejects
        if ref.shape[1] == 0:
            continue
        gy = rng.normal(0, 1, ref.shape).astype(np.float32)
        y, gx, glm = run_module(ex, lm, hop, gy=gy)
        assert y.shape == ref.shape, (trial, y.shape, ref.shape)
        rgx, rglm = O.ltv_fir_frames_backward(gy, ex, lm, win, hop)
        for what, got, want, tol in (("y", y, ref, 1e-5), ("g_ex", gx, rgx, 3e-5), ("g_log_mag", glm, rglm, 3e-5)):
            emax, el2 = rel_err(got, want)
            worst = max(worst, emax)
            assert emax <= tol and el2 <= tol, (trial, what, B, T, F, n_mag, hop, emax, el2)
        done += 1
    print("random FIR sweep:", done, "shapes, worst rel-max", worst)
    assert done >= 20
