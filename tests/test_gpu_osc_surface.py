"""GPU parity of the oscillator surface beyond the fused GOLF path (SURVEY §8a-8 gradients, a-11 / f-4 oscillators):
gradients w.r.t. phase / phase_offset / a trainable table of IndexedGlottalFlowTable, WeightedGlottalFlowTable,
WrappedPhaseDownsampledIndexedGlottalFlowTable, PulseTrain -- against the reference's own autograd run (golden g23) and
the float64 oracle at larger sizes."""
import numpy as np
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu


def dev(x, grad=False):
    return torch.as_tensor(np.asarray(x), dtype=torch.float32).cuda().requires_grad_(grad)


def close(got, ref, what, tol):
    emax, el2 = rel_err(np.asarray(got), np.asarray(ref))
    print(f"{what}: rel-max {emax:.2e} rel-l2 {el2:.2e}")
    assert emax <= tol and el2 <= tol, (what, emax, el2)


def _indexed(table, os_, trainable):
    from golf_amd.synth import IndexedGlottalFlowTable

    m = IndexedGlottalFlowTable(table_size=table.shape[0], lf_v2=True, points=table.shape[1], oversampling=os_,
                                equal_energy=True, trainable=trainable)
    with torch.no_grad():
        m.table.copy_(torch.as_tensor(table))
    return m.cuda()


@pytest.mark.parametrize("name,os_", [("ix1", 1), ("ix4", 4)])
def test_indexed_gradients_golden_g23(golden, name, os_):
    """trainable table + differentiable phase (+ phase_offset): the module's general path against the reference."""
    from golf_amd.audiotensor import AudioTensor

    g = golden("g23_oscillator_surface")
    m = _indexed(g[name + "_table"], os_, trainable=True)
    phase, w = dev(g[name + "_phase"], True), dev(g[name + "_w"], True)
    off = dev(g[name + "_off"], True) if name == "ix1" else None
    out, pre = m(AudioTensor(phase), AudioTensor(w, 16), None if off is None else AudioTensor(off), return_pre=True)
    sig = pre if os_ > 1 else out.as_tensor()
    close(sig.detach().cpu(), g[name + "_sig"], f"g23 {name} signal", 5e-5)
    (sig * dev(g[name + "_gy"])).sum().backward()
    torch.cuda.synchronize()
    close(phase.grad.cpu(), g[name + "_g_phase"], f"g23 {name} d/d phase", 3e-4)
    close(w.grad.cpu(), g[name + "_g_w"], f"g23 {name} d/d weight", 3e-4)
    close(m.table.grad.cpu(), g[name + "_g_table"], f"g23 {name} d/d table", 3e-4)
    if off is not None:
        close(off.grad.cpu(), g[name + "_g_off"], f"g23 {name} d/d phase_offset", 3e-4)


def test_weighted_and_wrapped_phase_golden_g23(golden):
    from golf_amd.audiotensor import AudioTensor
    from golf_amd.synth import WeightedGlottalFlowTable, WrappedPhaseDownsampledIndexedGlottalFlowTable

    g = golden("g23_oscillator_surface")
    m = WeightedGlottalFlowTable(table_size=7, lf_v2=True, points=16, trainable=True)
    with torch.no_grad():
        m.table.copy_(torch.as_tensor(g["wt_table"]))
    m = m.cuda()
    split, trs, _ = m.ctrl(lambda s_, t_: (s_, t_, None))((), ())
    assert tuple(split[0]) == tuple(int(v) for v in g["wt_split"])
    phase, logits = dev(g["wt_phase"], True), dev(g["wt_logits"], True)
    (wsm,) = trs[0](AudioTensor(logits, 16))
    close(wsm.as_tensor().detach().cpu(), g["wt_w"], "softmax weights", 1e-6)
    y = m(AudioTensor(phase), wsm).as_tensor()
    close(y.detach().cpu(), g["wt_out"], "weighted out", 5e-5)
    (y * dev(g["wt_gy"])).sum().backward()
    close(phase.grad.cpu(), g["wt_g_phase"], "weighted d/d phase", 3e-4)
    close(logits.grad.cpu(), g["wt_g_logits"], "weighted d/d logits", 3e-4)
    close(m.table.grad.cpu(), g["wt_g_table"], "weighted d/d table", 3e-4)
    with torch.no_grad():
        y8 = m(AudioTensor(dev(g["wt_phase8"]), 8), AudioTensor(dev(g["wt_w8"]), 32)).as_tensor()
    close(y8.cpu(), g["wt_out8"], "weighted out (phase hop 8, replicated frames)", 5e-5)

    wpm = WrappedPhaseDownsampledIndexedGlottalFlowTable(hop_rate=2, in_channels=3, table_size=7, lf_v2=True, points=16)
    assert sorted(wpm.state_dict().keys()) == [str(k) for k in g["wp_state_keys"]]
    with torch.no_grad():
        wpm.table.copy_(torch.as_tensor(g["wp_table"]))
    wpm = wpm.cuda()
    wp, w = dev(g["wp_phase"], True), dev(g["wp_w"], True)
    y = wpm(AudioTensor(wp), AudioTensor(w, 16)).as_tensor()
    close(y.detach().cpu(), g["wp_out"], "wrapped-phase out", 5e-5)
    (y * dev(g["wp_gy"])).sum().backward()
    close(wp.grad.cpu(), g["wp_g_phase"], "wrapped-phase d/d phase", 3e-4)
    close(w.grad.cpu(), g["wp_g_w"], "wrapped-phase d/d weight", 3e-4)


def test_pulse_train_golden_g23(golden):
    from golf_amd.audiotensor import AudioTensor
    from golf_amd.synth import PulseTrain

    g = golden("g23_oscillator_surface")
    pt = PulseTrain().cuda()
    y = pt(AudioTensor(dev(g["pt_phase"]))).as_tensor().cpu().numpy()
    np.testing.assert_allclose(y, g["pt_out"], rtol=2e-6, atol=1e-7)
    y8 = pt(AudioTensor(dev(g["pt_phase8"]), 8), AudioTensor(dev(g["pt_off8"]))).as_tensor().cpu().numpy()
    np.testing.assert_allclose(y8, g["pt_out8"], rtol=2e-6, atol=1e-7)
    # gradient reaches the impulse heights only: d rsqrt(p)/dp at the wrap instants
    p = dev(g["pt_phase"], True)
    pt(AudioTensor(p)).as_tensor().sum().backward()
    ref = np.where(g["pt_out"] > 0, -0.5 * g["pt_phase"].astype(np.float64) ** -1.5, 0.0)
    np.testing.assert_allclose(p.grad.cpu().numpy(), ref, rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("os_,Tp,w_hop", [(1, 3000, 240), (4, 2500, 600)])
def test_indexed_general_path_vs_oracle(os_, Tp, w_hop):
    """Larger shapes, decimation included: forward of the general path equals the fused kernel's, every gradient equals
    the float64 closed form (oracle.indexed_glottal_backward, itself pinned by g23 and finite differences)."""
    from golf_amd import functional as GF
    from golf_amd.synth import IndexedGlottalFlowTable
    from oracle import golf_oracle as O

    rng = np.random.default_rng(Tp)
    B = 3
    f0 = rng.uniform(90, 380, (B, 1)) * (1 + 0.03 * np.sin(2 * np.pi * 5.5 * np.arange(Tp) / 24000))
    phase = (f0 / 24000).astype(np.float32)
    w = rng.uniform(0.02, 0.98, (B, (Tp - 1) // w_hop + 2)).astype(np.float32)
    m = IndexedGlottalFlowTable(table_size=20, lf_v2=True, points=256, oversampling=os_, equal_energy=True)
    table = m.table.numpy().copy()
    taps = m.decimater.taps.numpy() if os_ > 1 else None
    pt, wt, tt = dev(phase, True), dev(w, True), dev(table, True)
    tp = None if taps is None else dev(taps)
    y = GF.wavetable_osc(pt, 1, GF.blend_tables(tt, wt), w_hop, os_, True, None, tp)
    y_fused = GF.glottal_osc(dev(phase), dev(w), dev(table), tp, 1, w_hop, os_, True)
    ref = O.indexed_glottal_forward(phase, 1, w, w_hop, table, os_, True, None, taps)["out"]
    close(y.detach().cpu(), ref, "general path forward", 1e-4)
    close(y.detach().cpu(), y_fused.cpu(), "general vs fused forward", 1e-4)
    gy = rng.normal(0, 1, ref.shape).astype(np.float32)
    (y * dev(gy)).sum().backward()
    torch.cuda.synchronize()
    bw = O.indexed_glottal_backward(gy, phase, 1, w, w_hop, table, os_, True, None, taps)
    close(wt.grad.cpu(), bw["g_weight"], "d/d weight", 2e-4)
    close(tt.grad.cpu(), bw["g_table"], "d/d table", 2e-4)
    close(pt.grad.cpu(), bw["g_phase"], "d/d phase", 5e-4)
    # the fused kernel's own weight gradient agrees with the general path's
    wf = dev(w, True)
    (GF.glottal_osc(dev(phase), wf, dev(table), tp, 1, w_hop, os_, True) * dev(gy)).sum().backward()
    close(wf.grad.cpu(), bw["g_weight"], "fused d/d weight", 2e-4)


def test_decoder_trains_through_predicted_f0():
    """VoiceAutoEncoder(train_with_true_f0=False, detach_f0=False) and GlottalFlowTable(trainable=True) used
    to raise on the first backward.  One step of the GOLF-ss source-filter decoder with a differentiable phase and a
    trainable table: finite gradients reach both."""
    from golf_amd.audiotensor import AudioTensor
    from golf_amd.ctrl import PassThrough
    from golf_amd.filters import LTVMinimumPhaseFilterPrecise
    from golf_amd.noise import StandardNormalNoise
    from golf_amd.sf import SourceFilterSynth
    from golf_amd.synth import DownsampledIndexedGlottalFlowTable
    from golf_amd.synthetic import make_inputs

    torch.manual_seed(0)
    dec = SourceFilterSynth(
        harm_oscillator=DownsampledIndexedGlottalFlowTable(hop_rate=10, in_channels=8, oversampling=4, equal_energy=True,
                                                           lf_v2=True, points=2048, trainable=True),
        noise_generator=StandardNormalNoise(), noise_filter=PassThrough(),
        end_filter=LTVMinimumPhaseFilterPrecise(lpc_order=22), room_filter=None, subtract_harmonics=False).cuda()
    inp = make_inputs(B=4, T=9600, device="cuda")
    phase = inp["phase"].clone().requires_grad_(True)
    w = inp["wsel"][:, :5].clone().requires_grad_(True)
    y = dec(phase=AudioTensor(phase), harm_oscillator_params=(AudioTensor(w, 2400),), noise_generator_params=(),
            noise_filter_params=(), end_filter_params=(AudioTensor(inp["gain"], 240), AudioTensor(inp["a"], 240)))
    y.as_tensor().square().mean().backward()
    torch.cuda.synchronize()
    for name, t in (("phase", phase.grad), ("weight", w.grad), ("table", dec.harm_oscillator.table.grad)):
        assert t is not None and torch.isfinite(t).all() and float(t.abs().max()) > 0, name
