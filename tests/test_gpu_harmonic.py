"""GPU parity of the harmonic oscillator bank (golf_harmonic_osc_*; reference models/synth.py:403-547): golden vectors
from the reference's own float32 run (g18), the float64 oracle at the size the baselines use, gradients."""
import numpy as np
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu


def dev(x, grad=False):
    return torch.as_tensor(np.asarray(x), dtype=torch.float32).cuda().requires_grad_(grad)


def check(y, ref, what, tol):
    y = np.asarray(y)
    emax, el2 = rel_err(y, ref)
    print(f"{what}: rel-max {emax:.3e} rel-l2 {el2:.3e}")
    assert np.isfinite(y).all()
    assert emax <= tol and el2 <= tol, (what, emax, el2)


@pytest.mark.parametrize("tag", ["t", "r", "q"])
def test_golden_g18(golden, tag):
    from golf_amd.audiotensor import AudioTensor as AT
    from golf_amd import synth as S

    g = golden("g18_harmonic_oscillators")
    ph, ah = int(g[f"{tag}_phase_hop"]), int(g[f"{tag}_amp_hop"])
    phase, amp = dev(g[f"{tag}_phase"]), dev(g[f"{tag}_amp"], True)
    H = amp.shape[-1]
    y = S.HarmonicOscillator().cuda()(AT(phase, ph), AT(amp, ah))
    assert y.hop_length == 1
    yt = y.as_tensor()
    (yt * dev(g[f"{tag}_gy"])).sum().backward()
    # the reference itself runs this in float32 (cumsum, sin): 2e-5 is its own accuracy here
    check(yt.detach().cpu().numpy(), g[f"{tag}_y"], f"g18{tag} harmonic", 3e-5)
    check(amp.grad.cpu().numpy(), g[f"{tag}_g_amp"], f"g18{tag} g_amp", 3e-5)
    add = S.AdditiveSynthesizer(num_harmonics=H).cuda()(AT(phase, ph), AT(amp.detach(), ah)).as_tensor()
    check(add.cpu().numpy(), g[f"{tag}_additive"], f"g18{tag} additive", 3e-5)
    saw = S.SawToothOscillator(num_harmonics=H).cuda()(AT(phase, ph)).as_tensor()
    check(saw.cpu().numpy(), g[f"{tag}_saw"], f"g18{tag} sawtooth", 3e-5)
    pulse = S.AdditivePulseTrain(num_harmonics=H).cuda()(AT(phase, ph)).as_tensor()
    check(pulse.cpu().numpy(), g[f"{tag}_pulse"], f"g18{tag} pulse train", 3e-5)


def test_ctrl_transforms(golden):
    from golf_amd.audiotensor import AudioTensor as AT
    from golf_amd import synth as S

    g = golden("g18_harmonic_oscillators")
    H = g["ctrl_logits"].shape[-1]
    m = S.AdditiveSynthesizer(num_harmonics=H)
    (split, trsfm) = m.ctrl(lambda s_, t_: (s_, t_))((), ())
    assert tuple(split[0]) == tuple(g["ctrl_split"])
    (amp,) = trsfm[0](AT(torch.as_tensor(g["ctrl_log_gain"]), 16), AT(torch.as_tensor(g["ctrl_logits"]), 16))
    np.testing.assert_allclose(amp.as_tensor().numpy(), g["ctrl_amp"], rtol=1e-6, atol=1e-7)
    v1 = S.V1AdditiveSynthesizer(num_harmonics=H)
    (_, t1) = v1.ctrl(lambda s_, t_: (s_, t_))((), ())
    (a1,) = t1[0](AT(torch.as_tensor(g["ctrl_log_gain"]), 16), AT(torch.as_tensor(g["ctrl_logits"]), 16))
    sg = torch.sigmoid(torch.as_tensor(g["ctrl_logits"]))
    ref = torch.exp(torch.as_tensor(g["ctrl_log_gain"]))[..., None] * sg / sg.sum(-1, keepdim=True)
    np.testing.assert_allclose(a1.as_tensor().numpy(), ref.numpy(), rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("B,Tp,ph,Fa,ah,H", [(4, 48000, 1, 201, 240, 155), (2, 401, 120, 201, 240, 150),
                                             (3, 5000, 1, 40, 128, 31), (2, 3000, 1, 1, 1, 5)])
def test_vs_oracle(B, Tp, ph, Fa, ah, H):
    """DDSP-style shapes: 2 s @ 24 kHz, 155 harmonics, amplitudes at hop 240; f0 from 80 to 1000 Hz so that the
    Nyquist mask cuts the upper harmonics of the high voices."""
    from golf_amd import functional as GF
    from oracle import golf_oracle as O

    rng = np.random.default_rng(H + Tp)
    f0 = rng.uniform(80, 1000, (B, 1)) * (1 + 0.03 * np.sin(np.linspace(0, 20, Tp))[None])
    phase = (f0 / 24000).astype(np.float32)
    amp = (rng.uniform(0, 1, (B, Fa, H)) / np.arange(1, H + 1)).astype(np.float32)
    ref = O.harmonic_oscillator_forward(phase, ph, amp, ah)
    gy = rng.normal(0, 1, ref.shape).astype(np.float32)
    a = dev(amp, True)
    y = GF.harmonic_osc(dev(phase), H, phase_hop=ph, amp=a, amp_hop=ah)
    assert tuple(y.shape) == ref.shape
    (y * dev(gy)).sum().backward()
    check(y.detach().cpu().numpy(), ref, f"harmonic fwd B{B} Tp{Tp} H{H}", 2e-5)
    rga = O.harmonic_oscillator_backward_amp(gy, phase, ph, amp.shape, ah)
    check(a.grad.cpu().numpy(), rga, "harmonic g_amp", 2e-5)


def test_random_shape_sweep():
    """25 random harmonic-bank shapes (harmonic count incl. non-multiples of the 32-harmonic anchor block, phase and
    amplitude hops, very high voices that the Nyquist mask cuts to a few harmonics): forward + amplitude gradient."""
    from golf_amd import functional as GF
    from oracle import golf_oracle as O

    rng = np.random.default_rng(99)
    worst = 0.0
    for trial in range(25):
        H = int(rng.choice([1, 2, 7, 31, 32, 33, 64, 100, 155]))
        ph = int(rng.choice([1, 1, 2, 5, 60]))
        ah = int(rng.choice([k for k in (8, 16, 64, 120, 240) if k % ph == 0 or ph == 1] or [ph * 4]))
        if ah % ph:
            ah = ph * 4
        B = int(rng.integers(1, 4))
        Tp = int(rng.integers(3, 3000 // ph + 4))
        n_out = (Tp - 1) * ph + 1
        Fa = n_out // ah + 2
        f0 = rng.uniform(60, 4000, (B, 1)) * (1 + 0.02 * np.sin(np.linspace(0, 9, Tp))[None])
        phase = (f0 / 24000).astype(np.float32)
        amp = (rng.uniform(0, 1, (B, Fa, H)) / np.arange(1, H + 1)).astype(np.float32)
        ref = O.harmonic_oscillator_forward(phase, ph, amp, ah)
        gy = rng.normal(0, 1, ref.shape).astype(np.float32)
        a = dev(amp, True)
        y = GF.harmonic_osc(dev(phase), H, ph, a, ah)
        assert tuple(y.shape) == ref.shape, (trial, y.shape, ref.shape)
        (y * dev(gy)).sum().backward()
        emax, el2 = rel_err(y.detach().cpu().numpy(), ref)
        # samples whose (float32) harmonic frequency sits exactly at Nyquist may flip the mask: tolerate by max-norm
        assert emax <= 2e-4 and el2 <= 2e-4, (trial, B, Tp, ph, ah, H, emax, el2)
        worst = max(worst, emax)
        gref = O.harmonic_oscillator_backward_amp(gy, phase, ph, amp.shape, ah)
        gmax, gl2 = rel_err(a.grad.cpu().numpy(), gref)
        assert gmax <= 3e-4 and gl2 <= 3e-4, (trial, "g_amp", B, Tp, ph, ah, H, gmax, gl2)
    print("random harmonic sweep worst forward rel-max", worst)


@pytest.mark.parametrize("phase_hop,amp_hop", [(1, 8), (4, 8)])
def test_phase_gradient_vs_finite_differences(phase_hop, amp_hop):
    """d out / d phase (HarmonicPlusNoiseSynth with non-detached voicing and AdditiveSynthesizer with a
    differentiable f0 used to raise): the derivative bank + reverse cumulative sum + transposed upsampling against central
    differences of the float64 oracle.  Harmonics are kept clear of Nyquist, where the mask makes the output
    discontinuous in the phase (a measure-zero set the reference's autograd ignores as well)."""
    from golf_amd import functional as GF
    from oracle import golf_oracle as O

    rng = np.random.default_rng(phase_hop)
    B, Tp, H = 2, 24, 6
    phase = rng.uniform(0.01, 0.03, (B, Tp))
    N = (Tp - 1) * phase_hop + 1
    Fa = (N - 1) // amp_hop + 2
    amp = rng.uniform(0.1, 1.0, (B, Fa, H))
    pt, at = dev(phase, True), dev(amp, True)
    y = GF.harmonic_osc(pt, H, phase_hop=phase_hop, amp=at, amp_hop=amp_hop)
    gy = rng.normal(0, 1, tuple(y.shape))
    (y * dev(gy)).sum().backward()
    torch.cuda.synchronize()
    f = lambda p: (O.harmonic_oscillator_forward(p, phase_hop, amp, amp_hop) * gy).sum()
    ref = np.zeros_like(phase)
    eps = 1e-7
    for i in np.ndindex(*phase.shape):
        pp, pm = phase.copy(), phase.copy()
        pp[i] += eps
        pm[i] -= eps
        ref[i] = (f(pp) - f(pm)) / (2 * eps)
    check(pt.grad.cpu().numpy(), ref, f"harmonic d/d phase (hop {phase_hop})", 2e-3)
    check(at.grad.cpu().numpy(), O.harmonic_oscillator_backward_amp(gy, phase, phase_hop, amp.shape, amp_hop),
          "harmonic d/d amp", 1e-4)


def test_additive_synth_trains_through_f0():
    """AdditiveSynthesizer derives its per-sample scale from the phase: with a differentiable phase both the phase and the
    scale paths carry gradient (the reference: autograd through rsqrt(0.5 / phase) and the cumsum)."""
    from golf_amd.audiotensor import AudioTensor as AT
    from golf_amd import synth as S

    torch.manual_seed(0)
    B, T, H = 2, 960, 12
    phase = (torch.rand(B, T, device="cuda") * 0.01 + 0.005).requires_grad_(True)
    amp = torch.rand(B, 5, H, device="cuda").requires_grad_(True)
    y = S.AdditiveSynthesizer(num_harmonics=H).cuda()(AT(phase), AT(amp, 240)).as_tensor()
    y.square().mean().backward()
    assert torch.isfinite(phase.grad).all() and float(phase.grad.abs().max()) > 0
    assert torch.isfinite(amp.grad).all()
    # the scale path alone: compare with autograd through the explicit product out = S * rsqrt(0.5 / phase)
    p2 = phase.detach().clone().requires_grad_(True)
    from golf_amd import functional as GF
    S_ = GF.harmonic_osc(phase.detach(), H, 1, amp.detach(), 240)
    n = S_.shape[1]
    (S_ * torch.rsqrt(0.5 / p2[:, :n])).square().mean().backward()
    ts = torch.rsqrt(0.5 / phase.detach()).requires_grad_(True)
    y3 = GF.harmonic_osc(phase.detach(), H, 1, amp.detach(), 240, ts, 1)
    y3.square().mean().backward()
    ref = ts.grad[:, :n] * (0.5 * (0.5 / phase.detach()[:, :n]) ** -1.5 * 0.5 / phase.detach()[:, :n] ** 2)
    emax, el2 = rel_err(ref.cpu().numpy(), p2.grad[:, :n].cpu().numpy())
    assert emax < 1e-4 and el2 < 1e-4, (emax, el2)


@pytest.mark.parametrize("tag", ["t", "r"])
def test_harmonic_phase_terms_golden_g27(golden, tag):
    """initial_phase (B,H) and phase_offset (an AudioTensor at its own hop) of HarmonicOscillator.forward,
    models/synth.py:429-435: the drop-in module against the reference's values and its autograd gradients w.r.t. the
    amplitudes, the offset and the initial phase."""
    from golf_amd.audiotensor import AudioTensor
    from golf_amd.synth import HarmonicOscillator

    g = golden("g27_harmonic_phase_terms")
    dev = lambda x: torch.as_tensor(np.asarray(x), dtype=torch.float32).cuda()
    amp = dev(g[f"{tag}_amp"]).requires_grad_(True)
    off = dev(g[f"{tag}_offset"]).requires_grad_(True)
    ip = dev(g[f"{tag}_initial_phase"]).requires_grad_(True)
    osc = HarmonicOscillator().cuda()
    y = osc(AudioTensor(dev(g[f"{tag}_phase"]), int(g[f"{tag}_phase_hop"])), AudioTensor(amp, int(g[f"{tag}_amp_hop"])),
            initial_phase=ip, phase_offset=AudioTensor(off, int(g[f"{tag}_offset_hop"]))).as_tensor()
    (y * dev(g[f"{tag}_gy"])).sum().backward()
    torch.cuda.synchronize()
    for name, got, want in (("y", y.detach(), g[f"{tag}_y"]), ("g_amp", amp.grad, g[f"{tag}_g_amp"]),
                            ("g_offset", off.grad, g[f"{tag}_g_offset"]),
                            ("g_initial_phase", ip.grad, g[f"{tag}_g_initial_phase"])):
        got = got.cpu().numpy()
        err = np.abs(got - want).max() / np.abs(want).max()
        print(f"g27{tag} {name}: rel-max {err:.2e}")
        assert got.shape == want.shape and err <= 1e-4, (name, err)


def test_initial_phase_gradient_without_amplitude_track():
    """SawToothOscillator (constant 1/h scale, no amplitude track) with a trainable initial_phase: the gradient comes from the
    amplitude-gradient kernel run a quarter cycle ahead over two frames whose hat functions sum to one; checked against the
    float64 oracle's closed form with amplitudes = 1/h at every frame."""
    from golf_amd import functional as GF
    from oracle import golf_oracle as O

    rng = np.random.default_rng(31)
    B, T, H = 2, 1201, 12
    dev = lambda x: torch.as_tensor(np.asarray(x), dtype=torch.float32).cuda()
    phase = (rng.uniform(150, 500, (B, 1)) * (1 + 0.05 * np.sin(np.arange(T) / 90.0)) / 24000).astype(np.float32)
    hs = (1.0 / np.arange(1, H + 1)).astype(np.float32)
    ip = dev(rng.uniform(-1, 1, (B, H))).requires_grad_(True)
    gy = rng.normal(0, 1, (B, T)).astype(np.float32)
    y = GF.harmonic_osc(dev(phase), H, 1, hscale=dev(hs), initial_phase=ip)
    (y * dev(gy)).sum().backward()
    amp = np.broadcast_to(hs, (B, 2, H)).astype(np.float64)
    ipn = ip.detach().cpu().numpy()
    ref_y = O.harmonic_oscillator_forward(phase, 1, amp, T - 1, None, 1, ipn)
    ref_g = O.harmonic_oscillator_backward_initial_phase(gy, phase, 1, amp, T - 1, None, 1, ipn)
    ey = np.abs(y.detach().cpu().numpy() - ref_y).max() / np.abs(ref_y).max()
    eg = np.abs(ip.grad.cpu().numpy() - ref_g).max() / np.abs(ref_g).max()
    print(f"saw-tooth bank with initial_phase: y {ey:.2e}, g_initial_phase {eg:.2e}")
    assert ey <= 1e-4 and eg <= 1e-4, (ey, eg)


def test_harmonic_phase_terms_full_size():
    """The same at the DDSP shape (155 harmonics, 2 s) against the float64 oracle; an integer phase offset changes nothing."""
    from golf_amd import functional as GF
    from oracle import golf_oracle as O

    rng = np.random.default_rng(8)
    B, T, H, hop = 2, 9601, 155, 240
    dev = lambda x: torch.as_tensor(np.asarray(x), dtype=torch.float32).cuda()
    phase = (rng.uniform(100, 400, (B, 1)) * (1 + 0.02 * np.sin(np.arange(T) / 700.0)) / 24000).astype(np.float32)
    amp = rng.uniform(0, 1, (B, (T - 1) // hop + 1, H)).astype(np.float32)
    off = rng.uniform(-2, 2, (B, (T - 1) // hop + 1)).astype(np.float32)
    ip = rng.uniform(0, 1, (B, H)).astype(np.float32)
    ref = O.harmonic_oscillator_forward(phase, 1, amp, hop, off, hop, ip)
    y = GF.harmonic_osc(dev(phase), H, 1, dev(amp), hop, phase_offset=dev(off), po_hop=hop, initial_phase=dev(ip))
    err = np.abs(y.cpu().numpy() - ref).max() / np.abs(ref).max()
    print("harmonic bank with phase terms vs oracle:", err)
    assert err <= 1e-4
    y0 = GF.harmonic_osc(dev(phase), H, 1, dev(amp), hop)
    y1 = GF.harmonic_osc(dev(phase), H, 1, dev(amp), hop, phase_offset=dev(np.full_like(off, 3.0)), po_hop=hop)
    assert (y0 - y1).abs().max() <= 1e-5 * y0.abs().max()
