"""GPU parity of the frame-wise LTI all-pole + OLA filter (golf_lti_frames_ola_fwd_f32, GOLF-ff)."""
import numpy as np
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-4


def dev(x):
    return torch.as_tensor(np.asarray(x), dtype=torch.float32).cuda()


def check(y, ref, what, tol=TOL):
    emax, el2 = rel_err(y, ref)
    print(f"{what}: rel-max {emax:.3e} rel-l2 {el2:.3e}")
    assert np.isfinite(y).all()
    assert emax <= tol and el2 <= tol, (what, emax, el2)


def run_module(ex, gain, a, hop, W, centred=True, M=None):
    from golf_amd.audiotensor import AudioTensor
    from golf_amd.filters import LTVMinimumPhaseFilter

    m = LTVMinimumPhaseFilter(window="hanning", window_length=W, centred=centred, lpc_order=M or a.shape[-1]).cuda()
    y = m(AudioTensor(dev(ex)), AudioTensor(dev(gain), hop), AudioTensor(dev(a), hop))
    torch.cuda.synchronize()
    assert y.hop_length == 1
    return y.as_tensor().cpu().numpy()


def test_golden_g7(golden):
    g = golden("g7_framewise")
    hop, W = int(g["hop"]), int(g["W"])
    check(run_module(g["ex"], g["gain"], g["a"], hop, W, True), g["y_centred"], "g7 centred")
    check(run_module(g["ex"], g["gain"], g["a"], hop, W, False), g["y_uncentred"], "g7 uncentred")
    check(run_module(g["ex"], g["gain"], g["a"], hop, 16, True), g["y_w16"], "g7 W=2*hop")


@pytest.mark.parametrize("B,F,M,hop,W,Tx", [(3, 12, 22, 240, 960, 2880), (2, 12, 22, 240, 960, 2500),
                                            (2, 20, 12, 24, 96, 480), (2, 9, 6, 16, 50, 129), (1, 30, 26, 120, 480, 3600)])
def test_ff_vs_oracle(B, F, M, hop, W, Tx):
    from oracle import golf_oracle as O
    from test_gpu_lpc_ss import smooth_case

    ex, gain, a = smooth_case(B, F, M, hop, Tx=Tx, seed=F + M)
    win = torch.hann_window(W).double().numpy()
    ref, _ = O.lti_frames_ola_forward(ex, gain, a, hop, win, centred=True)
    y = run_module(ex, gain, a, hop, W, True)
    assert y.shape == ref.shape
    check(y, ref, f"ff B{B} F{F} M{M} hop{hop} W{W}")


def test_ff_full_size():
    """BASELINE configs[1]: GOLF-ff, B=32, 2 s @ 24 kHz, W=960, hop 240, M=22, forward only."""
    from golf_amd.synthetic import make_inputs
    from oracle import golf_oracle as O

    inp = make_inputs(B=32)
    ex, gain, a = inp["noise"].numpy(), inp["gain"].numpy(), inp["a"].numpy()
    y = run_module(ex, gain, a, 240, 960, True)
    assert y.shape == (32, 47760)
    ref, norm = O.lti_frames_ola_forward(ex, gain, a, 240, torch.hann_window(960).double().numpy())
    check(y, ref, "ff full-size (all 32 utterances vs oracle)")
    for b in range(32):   # every row on its own scale
        emax, el2 = rel_err(y[b], ref[b])
        assert emax <= TOL and el2 <= TOL, (b, emax, el2)
    # size-independent property: zero coefficients => identity on x*gain (OLA of hann frames is a partition of unity)
    y0 = run_module(ex, gain, 0 * a, 240, 960, True)
    G = O.linear_upsample(gain, 240)[:, :47760]
    check(y0, ex[:, :47760] * G, "ff with a=0 is x*gain")


def run_module_grad(ex, gain, a, hop, W, gy, centred=True):
    from golf_amd.audiotensor import AudioTensor
    from golf_amd.filters import LTVMinimumPhaseFilter

    m = LTVMinimumPhaseFilter(window="hanning", window_length=W, centred=centred, lpc_order=a.shape[-1]).cuda()
    x, g, aa = (dev(v).requires_grad_(True) for v in (ex, gain, a))
    y = m(AudioTensor(x), AudioTensor(g, hop), AudioTensor(aa, hop)).as_tensor()
    (y * dev(gy)).sum().backward()
    torch.cuda.synchronize()
    return [t.detach().cpu().numpy() for t in (y, x.grad, g.grad, aa.grad)]


@pytest.mark.parametrize("tag", ["c", "u", "s"])
def test_golden_g15_grads(golden, tag):
    """Gradients captured from the reference's own glue (autograd through its unfold / lfilter / conv_transpose1d)."""
    g = golden("g15_ff_grads")
    hop, W, centred = int(g[f"{tag}_hop"]), int(g[f"{tag}_W"]), bool(g[f"{tag}_centred"])
    y, gx, gg, ga = run_module_grad(g[f"{tag}_ex"], g[f"{tag}_gain"], g[f"{tag}_a"], hop, W, g[f"{tag}_gy"], centred)
    check(y, g[f"{tag}_y"], f"g15{tag} y")
    check(gx, g[f"{tag}_g_ex"], f"g15{tag} g_ex")
    check(gg, g[f"{tag}_g_gain"], f"g15{tag} g_gain")
    check(ga, g[f"{tag}_g_a"], f"g15{tag} g_a")


@pytest.mark.parametrize("B,F,M,hop,W,Tx", [(3, 12, 22, 240, 960, 2641), (2, 12, 22, 240, 960, 2500),
                                            (2, 20, 12, 24, 96, 480), (1, 30, 26, 120, 480, 3481)])
def test_ff_grads_vs_oracle(B, F, M, hop, W, Tx):
    from oracle import golf_oracle as O
    from test_gpu_lpc_ss import smooth_case

    ex, gain, a = smooth_case(B, F, M, hop, Tx=Tx, seed=F + M)
    win = torch.hann_window(W).double().numpy()
    ref, _ = O.lti_frames_ola_forward(ex, gain, a, hop, win, centred=True)
    gy = np.random.default_rng(3).normal(0, 1, ref.shape).astype(np.float32)
    y, gx, gg, ga = run_module_grad(ex, gain, a, hop, W, gy)
    rgx, rgg, rga = O.lti_frames_ola_backward(gy, ex, gain, a, hop, win, centred=True)
    check(y, ref, "y")
    check(gx, rgx, "g_ex")
    check(gg, rgg, "g_gain")
    check(ga, rga, "g_a")


def test_ff_full_size_grads():
    """GOLF-ff training shape (B=32, 2 s, W=960, hop 240, M=22): gradients vs the oracle on a slice of the batch."""
    from golf_amd.synthetic import make_inputs
    from oracle import golf_oracle as O

    inp = make_inputs(B=32)
    ex, gain, a = (inp[k].numpy() for k in ("noise", "gain", "a"))
    win = torch.hann_window(960).double().numpy()
    gy = np.random.default_rng(5).normal(0, 1, (32, 47760)).astype(np.float32)
    y, gx, gg, ga = run_module_grad(ex, gain, a, 240, 960, gy)
    nb = 2
    rgx, rgg, rga = O.lti_frames_ola_backward(gy[:nb], ex[:nb], gain[:nb], a[:nb], 240, win, centred=True)
    check(gx[:nb], rgx, "full-size g_ex")
    check(gg[:nb], rgg, "full-size g_gain")
    check(ga[:nb], rga, "full-size g_a")
    assert np.all(gx[:, 47761:] == 0) and np.isfinite(gx).all() and np.isfinite(ga).all()


@pytest.mark.parametrize("tag", ["a", "b"])
def test_biquad_cascade_golden_g19(golden, tag):
    """BatchSecondOrderLPCSynth (cascade of second-order sections, models/lpc.py:94-131) against the reference's run."""
    from golf_amd.lpc import BatchSecondOrderLPCSynth

    g = golden("g19_biquad_cascade")
    hop, W = int(g[f"{tag}_hop"]), int(g[f"{tag}_W"])
    m = BatchSecondOrderLPCSynth(hop_length=hop, window_size=W, window="hanning").cuda()
    y = m(dev(g[f"{tag}_ex"]), dev(g[f"{tag}_gain"]), dev(g[f"{tag}_biquads"])).cpu().numpy()
    check(y, g[f"{tag}_y"], f"g19{tag} cascade", 2e-5)


def test_biquad_cascade_equals_direct_form_full_size():
    """GOLF-ff shape (B=32, W=960, hop 240, 11 sections = order 22): the systolic cascade against (i) the float64
    oracle cascade on a batch slice and (ii) the direct-form kernel on the multiplied-out coefficients."""
    from golf_amd import functional as GF
    from golf_amd.synthetic import make_inputs
    from golf_amd.utils import biquads2lpc, get_logits2biquads
    from oracle import golf_oracle as O

    B, F, K, hop, W = 32, 200, 11, 240, 960
    inp = make_inputs(B=B)
    g = torch.Generator().manual_seed(3)
    logits = 0.5 * torch.randn(B, 1, K, 2, generator=g) + torch.cumsum(0.02 * torch.randn(B, F, K, 2, generator=g), 1)
    bq = get_logits2biquads("coef", 0.97)(logits)
    ex, gain = inp["noise"], inp["gain"]
    win = torch.hann_window(W)
    y = GF.biquad_frames_ola(ex.cuda(), gain.cuda(), bq.cuda(), win.cuda(), hop, pad=W // 2, frame_gain=False)
    y = y.cpu().numpy()
    nb = 2
    ref = O.biquad_frames_ola_forward(ex[:nb].numpy(), gain[:nb].numpy(), bq[:nb].numpy(), hop,
                                      win.double().numpy(), pad=W // 2, frame_gain=False)
    check(y[:nb], ref, "cascade vs oracle cascade")
    # the direct form of the same filter (coefficients multiplied out in float64): mathematically identical for LTI
    # frames, numerically not -- an order-22 direct form with poles at 0.97 loses ~3 digits in fp32,
    # the cascade does not.  So: both oracles agree, the cascade kernel matches its oracle at 1e-4, and the direct-form
    # kernel is only as close as fp32 direct form allows.
    lpc = biquads2lpc(bq.double())
    ref_d, _ = O.lti_frames_ola_forward(ex[:nb].numpy(), gain[:nb].numpy(), lpc[:nb].numpy(), hop,
                                        win.double().numpy(), centred=True)
    check(ref, ref_d, "oracle cascade vs oracle direct form", 1e-9)
    yd = GF.lti_frames_ola(ex.cuda(), gain.cuda(), lpc.float().cuda(), win.cuda(), hop).cpu().numpy()
    emax, el2 = rel_err(yd[:nb], ref_d)
    print(f"direct-form kernel on the same filter: rel-max {emax:.3e} rel-l2 {el2:.3e}")
    assert emax < 5e-2


def test_random_shape_sweep():
    """30 random frame-wise shapes (hop, window 2..5 hops and not a multiple of the hop, order, centred or not, ragged
    excitation) against the float64 oracle, forward and — on every third — gradients."""
    from golf_amd.audiotensor import AudioTensor
    from golf_amd.filters import LTVMinimumPhaseFilter
    from oracle import golf_oracle as O
    from test_gpu_lpc_ss import smooth_case

    rng = np.random.default_rng(31)
    worst = 0.0
    for case in range(30):
        hop = int(rng.choice([8, 16, 24, 40, 64, 120, 240]))
        W = int(2 * hop + 2 * rng.integers(0, 3 * hop // 2 + 1))          # even, >= 2*hop
        M = int(rng.integers(1, min(30, hop - 2) + 1))   # the kernels need a ring width above M that fits in one hop
        B = int(rng.integers(1, 4))
        centred = bool(rng.integers(0, 2))
        F = int(rng.integers(3, max(4, 1500 // hop)))
        Tx = int(rng.integers((F - 2) * hop, (F - 1) * hop + 1))
        ex, gain, a = smooth_case(B, F, M, hop, Tx=Tx, seed=100 + case)
        win = torch.hann_window(W).double().numpy()
        try:
            ref, _ = O.lti_frames_ola_forward(ex, gain, a, hop, win, centred=centred)
        except AssertionError:
            continue                      # a shape the reference rejects (more frames than parameters)
        y = run_module(ex, gain, a, hop, W, centred)
        assert y.shape == ref.shape, (case, y.shape, ref.shape)
        emax, el2 = rel_err(y, ref)
        worst = max(worst, emax)
        assert emax <= TOL and el2 <= TOL, (case, B, F, M, hop, W, Tx, centred, emax, el2)
        if case % 3 == 0:
            m = LTVMinimumPhaseFilter(window="hanning", window_length=W, centred=centred, lpc_order=M).cuda()
            t = [dev(v).requires_grad_(True) for v in (ex, gain, a)]
            gy = rng.normal(0, 1, ref.shape).astype(np.float32)
            out = m(AudioTensor(t[0]), AudioTensor(t[1], hop), AudioTensor(t[2], hop)).as_tensor()
            from golf_amd._lib import GolfError

            try:
                (out * dev(gy)).sum().backward()
            except GolfError as e:
                # documented limit of the backward (the forward has none): the window must be a multiple of the ring
                # width the kernel picked; it says so instead of computing something else
                assert "multiple of the ring width" in str(e), e
                continue
            refs = O.lti_frames_ola_backward(gy, ex, gain, a, hop, win, centred=centred)
            for name, got, want in zip(("g_ex", "g_gain", "g_a"), t, refs):
                emax, el2 = rel_err(got.grad.cpu().numpy(), want)
                assert emax <= 2e-4 and el2 <= 2e-4, (case, name, B, F, M, hop, W, Tx, centred, emax, el2)
    print("random ff sweep worst forward rel-max", worst)


def test_ff_ill_conditioned_rows():
    """The utterances that break an fp32 product chain in the sample-wise filter (filters at the edge of stability,
    tests/test_gpu_lpc_ss.py::test_ill_conditioned_rows) through the frame-wise filter: every frame starts from a zero
    state and runs 960 samples, so the error growth is bounded by the frame length: the hard row lands at 1.05e-4 (any
    fp32 direct-form recursion over 960 samples of a resonance at radius 0.9999 does), every other row within 1e-4."""
    from oracle import golf_oracle as O

    rng = np.random.default_rng(40)
    B, F, M, hop, W = 48, 200, 22, 240, 960
    logits = rng.normal(0, 0.5, (B, 1, M)) + np.cumsum(rng.normal(0, 0.02, (B, F, M)), 1)
    a = O.rc2lpc(np.tanh(logits)).astype(np.float32)[:32]
    gain = np.exp(-3 + np.cumsum(rng.normal(0, 0.05, (B, F)), 1)).astype(np.float32)[:32]
    ex = rng.normal(0, 1, (B, (F - 1) * hop + 1)).astype(np.float32)[:32]
    win = torch.hann_window(W).double().numpy()
    ref, _ = O.lti_frames_ola_forward(ex, gain, a, hop, win, centred=True)
    y = run_module(ex, gain, a, hop, W)
    err = np.abs(y - ref).max(1) / np.abs(ref).max(1)
    print("worst rows", np.argsort(err)[-3:], np.sort(err)[-3:])
    assert err.max() <= 3e-4 and np.sort(err)[-2] <= 1e-4, (int(err.argmax()), float(err.max()))


def _bq_grads(ex, gain, bq, win, gy, hop, pad=None, frame_gain=True):
    from golf_amd import functional as GF

    t = [dev(v).requires_grad_(True) for v in (ex, gain, bq)]
    y = GF.biquad_frames_ola(t[0], t[1], t[2], dev(win), hop, pad=pad, frame_gain=frame_gain)
    (y * dev(gy)).sum().backward()
    torch.cuda.synchronize()
    return (y.detach().cpu().numpy(),) + tuple(v.grad.cpu().numpy() for v in t)


@pytest.mark.parametrize("tag", ["a", "b"])
def test_biquad_cascade_grads_golden_g26(golden, tag):
    """Backward of the cascaded-biquad synthesiser (row a-6) against the reference's own autograd gradients."""
    g = golden("g26_biquad_cascade_grads")
    hop, W = int(g[f"{tag}_hop"]), int(g[f"{tag}_W"])
    win = torch.hann_window(W).numpy()
    y, g_ex, g_gain, g_bq = _bq_grads(g[f"{tag}_ex"], g[f"{tag}_gain"], g[f"{tag}_biquads"], win, g[f"{tag}_gy"], hop)
    check(y, g[f"{tag}_y"], f"g26{tag} y", 2e-5)
    check(g_ex, g[f"{tag}_g_ex"], f"g26{tag} g_ex", 5e-5)
    check(g_gain, g[f"{tag}_g_gain"], f"g26{tag} g_gain", 5e-5)
    check(g_bq, g[f"{tag}_g_biquads"], f"g26{tag} g_biquads", 5e-5)


@pytest.mark.parametrize("frame_gain", [True, False])
def test_biquad_cascade_grads_config_size(frame_gain):
    """The same at the GOLF-ff frame shape (11 sections = order 22, window 960, hop 240), both gain conventions, against
    the float64 closed form; and through the drop-in module (BatchSecondOrderLPCSynth is differentiable like the
    reference's)."""
    from golf_amd.utils import get_logits2biquads
    from oracle import golf_oracle as O

    rng = np.random.default_rng(11)
    B, F, K, hop, W = 2, 12, 11, 240, 960
    logits = torch.from_numpy(rng.normal(0, 0.7, (B, F, K, 2)).astype(np.float32))
    bq = get_logits2biquads("coef", 0.95)(logits).numpy().astype(np.float32)
    gain = np.exp(rng.normal(-1, 0.3, (B, F))).astype(np.float32)
    pad = (W - hop) // 2 if frame_gain else W // 2
    Tx = (F - 1) * hop + W - 2 * pad if frame_gain else (F - 1) * hop + 1
    ex = rng.normal(0, 1, (B, Tx)).astype(np.float32)
    win = torch.hann_window(W).numpy()
    ref = O.biquad_frames_ola_forward(ex, gain, bq, hop, win, pad=pad, frame_gain=frame_gain)
    gy = (rng.normal(0, 1, ref.shape) / np.abs(ref).max()).astype(np.float32)
    r_ex, r_gain, r_bq = O.biquad_frames_ola_backward(gy, ex, gain, bq, hop, win, pad=pad, frame_gain=frame_gain)
    y, g_ex, g_gain, g_bq = _bq_grads(ex, gain, bq, win, gy, hop, pad=pad, frame_gain=frame_gain)
    check(y, ref, "cascade y")
    check(g_ex, r_ex, "cascade g_ex")
    check(g_gain, r_gain, "cascade g_gain")
    check(g_bq, r_bq, "cascade g_biquads")
    if frame_gain:
        from golf_amd.lpc import BatchSecondOrderLPCSynth

        m = BatchSecondOrderLPCSynth(hop_length=hop, window_size=W, window="hanning").cuda()
        t = [dev(v).requires_grad_(True) for v in (ex, gain, bq)]
        (m(*t) * dev(gy)).sum().backward()
        check(t[2].grad.cpu().numpy(), r_bq, "module g_biquads")


def test_ff_backward_writes_the_whole_excitation_gradient():
    """An excitation longer than (F-1)*hop + 1 reaches no output beyond that sample: the backward writes those zeros itself
    (round 4: the host used to allocate g_ex with torch.zeros, a full-size fill in front of every backward).  The gradient
    buffer comes from the caching allocator: a NaN-filled block of its size is released right before, three times."""
    from golf_amd import functional as GF
    from test_gpu_lpc_ss import smooth_case

    B, F, M, hop, W = 3, 12, 22, 240, 960
    Tx = (F - 1) * hop + 1 + 239
    ex, gain, a = smooth_case(B, F, M, hop, Tx=Tx, seed=5)
    win = torch.hann_window(W, device="cuda")
    for _ in range(3):
        x, g, aa = (dev(v).requires_grad_(True) for v in (ex, gain, a))
        y = GF.lti_frames_ola(x, g, aa, win, hop)
        poison = torch.full((B, Tx), float("nan"), device="cuda")
        del poison
        y.sum().backward()
        torch.cuda.synchronize()
        assert torch.isfinite(x.grad).all()
        assert (x.grad[:, (F - 1) * hop + 1:] == 0).all() and x.grad[:, : (F - 1) * hop + 1].abs().sum() > 0
