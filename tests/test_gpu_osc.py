"""GPU parity of the glottal wavetable oscillator (golf_glottal_osc_fwd_f32 / _bwd_wsel_f32)."""
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


def osc(phase, phase_hop, w, w_hop, table, os_, eq, taps=None, return_pre=False):
    from golf_amd import functional as GF

    t = None if taps is None else dev(taps)
    res = GF.glottal_osc(dev(phase), dev(w), dev(table), t, phase_hop, w_hop, os_, eq, return_pre=return_pre)
    torch.cuda.synchronize()
    if return_pre:
        return res[0].cpu().numpy(), (None if res[1] is None else res[1].cpu().numpy())
    return res.cpu().numpy()


@pytest.mark.parametrize("name,os_,eq", [("os1", 1, False), ("os1_eq", 1, True), ("os4_eq", 4, True)])
def test_golden_g6(golden, name, os_, eq):
    from oracle import golf_oracle as O

    g = golden("g6_oscillator")
    tb = g[name + "_table"]
    taps = O.default_decimation_taps(4) if os_ > 1 else None
    for sfx, ph_hop, w_hop in (("", 1, 16), ("2", 8, 32)):
        out, pre = osc(g[name + "_phase" + sfx], ph_hop, g[name + "_w" + sfx], w_hop, tb, os_, eq, taps, True)
        if os_ == 1:
            # golden produced by the reference in float32: tolerance covers ITS rounding
            check(out, g[name + "_out" + sfx], f"g6 {name}{sfx} out", 5e-5)
        else:
            check(pre, g[name + "_pre" + sfx], f"g6 {name}{sfx} pre-decimation", 5e-5)
            ref = O.decimate_fir(g[name + "_pre" + sfx], taps, os_)
            assert out.shape == ref.shape
            check(out, ref, f"g6 {name}{sfx} decimated", 5e-5)


@pytest.mark.parametrize("B,Tp,ph_hop,w_hop,os_,eq", [(3, 2000, 1, 240, 4, True), (2, 1201, 1, 400, 1, False),
                                                      (2, 41, 120, 2400, 4, True), (1, 500, 1, 100, 2, True),
                                                      (2, 300, 1, 64, 1, True)])
def test_osc_vs_oracle(B, Tp, ph_hop, w_hop, os_, eq):
    from golf_amd.synth import IndexedGlottalFlowTable
    from oracle import golf_oracle as O

    rng = np.random.default_rng(Tp)
    f0 = rng.uniform(80, 400, (B, 1)) * (1 + 0.03 * np.sin(2 * np.pi * 5.5 * np.arange(Tp) * ph_hop / 24000))
    phase = (f0 / 24000).astype(np.float32)
    Fw = (Tp - 1) * ph_hop // w_hop + 2
    w = rng.uniform(0, 1, (B, Fw)).astype(np.float32)
    m = IndexedGlottalFlowTable(table_size=100, lf_v2=True, points=2048, oversampling=os_, equal_energy=eq)
    table = m.table.numpy()
    taps = m.decimater.taps.numpy() if os_ > 1 else None
    ref = O.indexed_glottal_forward(phase, ph_hop, w, w_hop, table, os_, eq, decim_taps=taps)
    out, pre = osc(phase, ph_hop, w, w_hop, table, os_, eq, taps, True)
    if os_ > 1:
        check(pre, ref["pre"], "pre")
    assert out.shape == ref["out"].shape
    check(out, ref["out"], f"osc B{B} Tp{Tp} hop{ph_hop} os{os_}")


def test_osc_full_size_module():
    """BASELINE shape: B=32, phase (32,48000) hop 1, weight (32,21) hop 2400, table 100x2048, 4x oversampling."""
    from golf_amd.audiotensor import AudioTensor
    from golf_amd.synth import DownsampledIndexedGlottalFlowTable
    from golf_amd.synthetic import make_inputs
    from oracle import golf_oracle as O

    inp = make_inputs(B=32)
    m = DownsampledIndexedGlottalFlowTable(hop_rate=10, in_channels=64, oversampling=4, equal_energy=True,
                                           lf_v2=True, points=2048).cuda()
    y, pre = m(AudioTensor(inp["phase"].cuda()), AudioTensor(inp["wsel"].cuda(), inp["w_hop"]), return_pre=True)
    torch.cuda.synchronize()
    assert y.shape == (32, 48000) and pre.shape == (32, 191997) and y.hop_length == 1
    ref = O.indexed_glottal_forward(inp["phase"].numpy(), 1, inp["wsel"].numpy(), inp["w_hop"],
                                    m.table.cpu().numpy(), 4, True, decim_taps=m.decimater.taps.cpu().numpy())
    if pre is not None:
        check(pre.cpu().numpy(), ref["pre"], "full-size pre (all 32)")
    yh = y.as_tensor().cpu().numpy()
    check(yh, ref["out"], "full-size out (all 32)")
    for b in range(32):
        emax, el2 = rel_err(yh[b], ref["out"][b])
        assert emax <= TOL and el2 <= TOL, (b, emax, el2)


def test_osc_full_length_closer_to_f64_than_reference_fp32_order():
    """north_star: "within 1e-4 of the reference output".  The reference accumulates the oversampled phase with an
    fp32 cumsum over 192 k samples (models/synth.py:250-251); on a 2 s clip that alone moves ITS output ~1e-3 (max-norm)
    away from exact arithmetic: the LF derivative table has a near-discontinuity at glottal closure, so a phase error
    of 1e-5 cycles flips isolated samples.  The HIP oscillator accumulates phase exactly (Q0.64 fixed point).  This test
    makes that a measured fact: on the BASELINE shape the HIP output is within 1e-4 of the float64 oracle, the
    reference's own fp32 op sequence (oracle/cpu_baseline.py::oscillator_reference_ops, same taps) is NOT, and the HIP
    output is the closer of the two in both norms -- i.e. the deviation from the reference's fp32 path is the
    reference's rounding, not ours."""
    from golf_amd import functional as GF
    from golf_amd.synth import DownsampledIndexedGlottalFlowTable
    from golf_amd.synthetic import make_inputs
    from oracle import golf_oracle as O
    from oracle.cpu_baseline import oscillator_reference_ops

    inp = make_inputs(B=32)
    m = DownsampledIndexedGlottalFlowTable(hop_rate=10, in_channels=64, oversampling=4, equal_energy=True,
                                           lf_v2=True, points=2048)
    ref64 = O.indexed_glottal_forward(inp["phase"].numpy(), 1, inp["wsel"].numpy(), inp["w_hop"], m.table.numpy(), 4,
                                      True, decim_taps=m.decimater.taps.numpy())["out"]
    ref32 = oscillator_reference_ops(inp["phase"], inp["wsel"], inp["w_hop"], m.table, m.decimater.kernel, 4,
                                     True).numpy()
    hip = GF.glottal_osc(inp["phase"].cuda(), inp["wsel"].cuda(), m.table.cuda(), m.decimater.taps.cuda(), 1,
                         inp["w_hop"], 4, True).cpu().numpy()
    assert hip.shape == ref64.shape == ref32.shape == (32, 48000)
    e_hip, e_ref = rel_err(hip, ref64), rel_err(ref32, ref64)
    e_hip_ref = rel_err(hip, ref32)
    print(f"HIP vs f64 {e_hip[0]:.2e}/{e_hip[1]:.2e}   reference-fp32-order vs f64 {e_ref[0]:.2e}/{e_ref[1]:.2e}   "
          f"HIP vs reference-fp32-order {e_hip_ref[0]:.2e}/{e_hip_ref[1]:.2e}")
    assert e_hip[0] <= TOL and e_hip[1] <= TOL
    assert e_hip[0] < e_ref[0] and e_hip[1] < e_ref[1]
    assert e_ref[0] > TOL            # the reference's fp32 cumsum is what breaks 1e-4 (max-norm), documented in DESIGN.md
    # and HIP-vs-reference is explained by the reference's own error (triangle inequality, 10 % slack)
    assert e_hip_ref[0] <= 1.1 * (e_ref[0] + e_hip[0])


def test_b256_source_every_utterance():
    """BASELINE configs[3]: 256 utterances of the benchmark recipe, 32 per rank.  Round 3 checked every row of that batch
    through the FILTER only (tests/test_gpu_lpc_ss.py::test_b256_benchmark_inputs_every_utterance); this is the oscillator
    on the same 256 utterances -- the whole batch in one call (the shape a single rank would see at B = 256) and as the
    eight 32-utterance shards the ranks really run -- every row against the float64 oracle."""
    from golf_amd import functional as GF
    from golf_amd.synth import DownsampledIndexedGlottalFlowTable
    from golf_amd.synthetic import make_inputs
    from oracle import golf_oracle as O

    inp = make_inputs(B=256, seed=2434)
    m = DownsampledIndexedGlottalFlowTable(hop_rate=10, in_channels=64, oversampling=4, equal_energy=True,
                                           lf_v2=True, points=2048)
    ph, ws = inp["phase"].numpy(), inp["wsel"].numpy()
    ref = np.concatenate([O.indexed_glottal_forward(ph[lo:lo + 32], 1, ws[lo:lo + 32], inp["w_hop"], m.table.numpy(), 4,
                                                    True, decim_taps=m.decimater.taps.numpy())["out"]
                          for lo in range(0, 256, 32)])
    table, taps = m.table.cuda(), m.decimater.taps.cuda()
    whole = GF.glottal_osc(inp["phase"].cuda(), inp["wsel"].cuda(), table, taps, 1, inp["w_hop"], 4, True).cpu().numpy()
    assert whole.shape == ref.shape == (256, 48000)
    scale = np.abs(ref).max(1)
    e = np.abs(whole - ref).max(1) / scale
    print("B=256 oscillator: worst row", int(e.argmax()), float(e.max()))
    assert np.all(e <= TOL), (np.nonzero(e > TOL)[0], e.max())
    for lo in range(0, 256, 32):
        sh = GF.glottal_osc(inp["phase"][lo:lo + 32].cuda(), inp["wsel"][lo:lo + 32].cuda(), table, taps, 1, inp["w_hop"], 4,
                            True).cpu().numpy()
        esh = np.abs(sh - ref[lo:lo + 32]).max(1) / scale[lo:lo + 32]
        assert np.all(esh <= TOL), (lo, np.nonzero(esh > TOL)[0], esh.max())
        assert np.array_equal(sh, whole[lo:lo + 32]), lo   # a shard's rows do not depend on the batch they are launched in


@pytest.mark.parametrize("os_,eq", [(1, False), (4, True)])
def test_osc_backward_wsel(os_, eq):
    """d out / d table_select_weight against central differences of the float64 oracle
    (out is piecewise linear in w, so the difference quotient is exact inside a table cell)."""
    from golf_amd import functional as GF
    from golf_amd.synth import IndexedGlottalFlowTable
    from oracle import golf_oracle as O

    rng = np.random.default_rng(7)
    B, Tp, w_hop = 2, 600, 128
    phase = (rng.uniform(100, 300, (B, 1)) / 24000 * np.ones((1, Tp))).astype(np.float32)
    Fw = 6
    w = rng.uniform(0.05, 0.95, (B, Fw)).astype(np.float32)
    m = IndexedGlottalFlowTable(table_size=20, lf_v2=True, points=256, oversampling=os_, equal_energy=eq)
    table = m.table.numpy()
    taps = m.decimater.taps.numpy() if os_ > 1 else None
    wt = dev(w).requires_grad_(True)
    out = GF.glottal_osc(dev(phase), wt, dev(table), None if taps is None else dev(taps), 1, w_hop, os_, eq)
    gy = rng.normal(0, 1, tuple(out.shape)).astype(np.float32)
    (out * dev(gy)).sum().backward()
    torch.cuda.synchronize()
    got = wt.grad.cpu().numpy()
    ref = np.zeros_like(w, dtype=np.float64)
    eps = 1e-4
    f = lambda ww: (O.indexed_glottal_forward(phase, 1, ww, w_hop, table, os_, eq, decim_taps=taps)["out"] * gy).sum()
    for b in range(B):
        for k in range(Fw):
            wp, wm = w.astype(np.float64).copy(), w.astype(np.float64).copy()
            wp[b, k] += eps
            wm[b, k] -= eps
            ref[b, k] = (f(wp) - f(wm)) / (2 * eps)
    check(got, ref, f"g_wsel os{os_}", 2e-3)


@pytest.mark.parametrize("Tadd_delta", [0, -37, 11])
def test_fused_addend(Tadd_delta):
    """``glottal_osc(..., add=x)`` (harm_osc + noise fused into the decimator's epilogue, models/sf.py:53) is bit-identical
    to the separate addition on the common length, leaves the oscillator alone beyond it, and passes gradients through to
    both the addend and table_select_weight."""
    from golf_amd import functional as GF
    from golf_amd.synth import IndexedGlottalFlowTable

    rng = np.random.default_rng(11)
    B, Tp, w_hop = 3, 2500, 240
    phase = dev((rng.uniform(100, 300, (B, 1)) / 24000 * np.ones((1, Tp))).astype(np.float32))
    w = dev(rng.uniform(0.05, 0.95, (B, 12)).astype(np.float32))
    m = IndexedGlottalFlowTable(table_size=20, lf_v2=True, points=256, oversampling=4, equal_energy=True).cuda()
    plain = GF.glottal_osc(phase, w, m.table, m.decimater.taps, 1, w_hop, 4, True)
    Tout = plain.shape[1]
    Tadd = Tout + Tadd_delta
    x = dev(rng.normal(0, 1, (B, Tadd)).astype(np.float32))
    n = min(Tout, Tadd)
    # a non-contiguous addend view (row stride > Tadd) goes through as is
    xw = torch.zeros(B, Tadd + 5, device="cuda")
    xw[:, :Tadd] = x
    for xa in (x, xw[:, :Tadd]):
        wg, xg = w.clone().requires_grad_(True), xa.detach().clone().requires_grad_(True)
        fused = GF.glottal_osc(phase, wg, m.table, m.decimater.taps, 1, w_hop, 4, True, add=xg)
        assert fused.shape == plain.shape
        assert torch.equal(fused[:, :n], plain[:, :n] + x[:, :n])
        assert torch.equal(fused[:, n:], plain[:, n:])
        gy = dev(rng.normal(0, 1, (B, Tout)).astype(np.float32))
        (fused * gy).sum().backward()
        w2 = w.clone().requires_grad_(True)
        (GF.glottal_osc(phase, w2, m.table, m.decimater.taps, 1, w_hop, 4, True) * gy).sum().backward()
        assert torch.equal(wg.grad, w2.grad)
        ref = torch.zeros(B, Tadd, device="cuda")
        ref[:, :n] = gy[:, :n]
        assert torch.equal(xg.grad, ref)


def test_fused_addend_module_truncates_like_audiotensor_addition():
    from golf_amd.audiotensor import AudioTensor
    from golf_amd.synth import IndexedGlottalFlowTable

    m = IndexedGlottalFlowTable(table_size=20, lf_v2=True, points=256, oversampling=4, equal_energy=True).cuda()
    phase = AudioTensor(torch.full((2, 1000), 0.01, device="cuda"))
    w = AudioTensor(torch.full((2, 5), 0.4, device="cuda"), 240)
    nz = AudioTensor(torch.randn(2, 900, device="cuda"))
    sep = m(phase, w) + nz
    fused = m(phase, w, add=nz)
    assert fused.hop_length == 1 and fused.shape == sep.shape == (2, 900)
    assert torch.equal(fused.as_tensor(), sep.as_tensor())


def test_random_configuration_sweep():
    """30 random oscillator configurations — oversampling 1..4, phase hops incl. non-powers of two, table lengths that
    are and are not powers of two, table counts, ragged control grids — i.e. every render / decimate variant the
    launcher can pick, against the float64 oracle."""
    from golf_amd.synth import Decimate
    from oracle import golf_oracle as O

    rng = np.random.default_rng(77)
    worst = 0.0
    for case in range(30):
        os_ = int(rng.choice([1, 2, 3, 4]))
        ph_hop = int(rng.choice([1, 1, 1, 2, 3, 5, 16, 120]))
        Tp = int(rng.integers(3, 2500 // ph_hop + 4))
        L = int(rng.choice([64, 100, 256, 2048]))
        n_tab = int(rng.integers(2, 12))
        w_hop = int(rng.choice([ph_hop * k for k in (1, 2, 7, 20, 200)]))
        B = int(rng.integers(1, 4))
        eq = bool(rng.integers(0, 2))
        f0 = rng.uniform(80, 400, (B, 1)) * (1 + 0.03 * np.sin(2 * np.pi * 5.5 * np.arange(Tp) * ph_hop / 24000))
        phase = (f0 / 24000).astype(np.float32)
        Fw = (Tp - 1) * ph_hop // w_hop + 2
        w = rng.uniform(0, 1, (B, Fw)).astype(np.float32)
        table = rng.normal(0, 1, (n_tab, L)).astype(np.float32)
        taps = Decimate(os_).taps.numpy() if os_ > 1 else None
        ref = O.indexed_glottal_forward(phase, ph_hop, w, w_hop, table, os_, eq, decim_taps=taps)
        out = osc(phase, ph_hop, w, w_hop, table, os_, eq, taps)
        assert out.shape == ref["out"].shape, (case, out.shape, ref["out"].shape)
        emax, el2 = rel_err(out, ref["out"])
        worst = max(worst, emax)
        assert emax <= 1e-4 and el2 <= 1e-4, (case, B, Tp, ph_hop, w_hop, os_, L, n_tab, eq, emax, el2)
    print("random oscillator sweep worst rel-max", worst)


def test_long_input_beyond_256_tiles():
    """12.5 s with a per-sample phase = 300 000 coarse phase samples = 293 scan tiles: more than the 256-entry LDS tile
    prefix of the render / harmonic kernels holds, so the prefix is folded into the per-sample array by two extra passes.
    Forward (wavetable and harmonic oscillators) against the float64 oracle, and the table-selection gradient against the
    same computation split in two halves is not possible (phase accumulates) — so against a float64 finite difference."""
    from golf_amd import functional as GF
    from golf_amd.synth import IndexedGlottalFlowTable
    from oracle import golf_oracle as O

    rng = np.random.default_rng(5)
    B, Tp, w_hop = 1, 300_000, 2400
    t = np.arange(Tp) / 24000
    phase = ((170 + 40 * np.sin(2 * np.pi * 0.3 * t)) / 24000).astype(np.float32)[None]
    w = rng.uniform(0, 1, (B, Tp // w_hop + 2)).astype(np.float32)
    m = IndexedGlottalFlowTable(table_size=100, lf_v2=True, points=2048, oversampling=4, equal_energy=True)
    table, taps = m.table.numpy(), m.decimater.taps.numpy()
    ref = O.indexed_glottal_forward(phase, 1, w, w_hop, table, 4, True, decim_taps=taps)["out"]
    out = osc(phase, 1, w, w_hop, table, 4, True, taps)
    assert out.shape == ref.shape == (1, Tp)
    check(out, ref, "long wavetable oscillator (293 tiles)")
    check(out[:, -48000:], ref[:, -48000:], "  its last 2 s (phase accumulated over 12.5 s)")
    # harmonic bank on the same phase track
    amps = rng.uniform(0, 1, (B, Tp // 240 + 1, 8)).astype(np.float32)
    href = O.harmonic_oscillator_forward(phase, 1, amps, 240)
    hout = GF.harmonic_osc(dev(phase), 8, 1, dev(amps), 240).cpu().numpy()
    assert hout.shape == href.shape
    check(hout, href, "long harmonic oscillator", 2e-4)
    # gradient w.r.t. the table selection on the long input: finite difference of the oracle on two control points
    wt = dev(w).requires_grad_(True)
    gy = rng.normal(0, 1, ref.shape).astype(np.float32)
    (GF.glottal_osc(dev(phase), wt, dev(table), dev(taps), 1, w_hop, 4, True) * dev(gy)).sum().backward()
    got = wt.grad.cpu().numpy()
    f = lambda ww: (O.indexed_glottal_forward(phase, 1, ww, w_hop, table, 4, True, decim_taps=taps)["out"] * gy).sum()
    for k in (3, 120):
        wp, wm = w.astype(np.float64).copy(), w.astype(np.float64).copy()
        wp[0, k] += 1e-4
        wm[0, k] -= 1e-4
        fd = (f(wp) - f(wm)) / 2e-4
        assert abs(got[0, k] - fd) <= 5e-3 * max(1.0, abs(fd)), (k, got[0, k], fd)


@pytest.mark.parametrize("Tp,w_hop,eq", [(48000, 2400, True), (48000, 2400, False), (1, 2400, True), (5, 2400, True),
                                         (2047, 2400, True), (2048, 1200, True), (2049, 2400, True),
                                         (5000, 1200, False), (6000, 800, True), (9999, 4000, True)])
def test_fused_kernel_matches_three_kernel_path(Tp, w_hop, eq):
    """The fused oscillator kernel (in-block phase scan -> render into LDS -> polyphase decimation on the matrix pipe, no
    HBM round trip of the oversampled signal) against the three-kernel path (taken when `pre` is requested).  Both walk
    the same exact Q0.64 phases; the fused one linearises rsqrt over 4 fine samples, interpolates along the control frame
    first and sums the taps in MFMA order, so the values agree to a few 1e-7 -- with and without the fused addend, ragged
    tile counts, control hops that need 3 and 4 staged table rows (and one, w_hop 800, that falls back)."""
    from golf_amd import functional as GF
    from golf_amd.synth import IndexedGlottalFlowTable

    rng = np.random.default_rng(Tp + w_hop)
    B = 3
    f0 = rng.uniform(80, 400, (B, 1)) * (1 + 0.03 * np.sin(2 * np.pi * 5.5 * np.arange(Tp) / 24000 + rng.uniform(0, 6, (B, 1))))
    if Tp > 4000:
        f0[0, 1000:1003] = [50.0, 180.0, 420.0]       # f0 jumps (voicing boundaries): the exact-rsqrt branch
    phase = dev((f0 / 24000).astype(np.float32))
    Fw = (Tp - 1) // w_hop + 2
    w = dev(rng.uniform(0, 1, (B, Fw)).astype(np.float32))
    m = IndexedGlottalFlowTable(table_size=100, lf_v2=True, points=2048, oversampling=4, equal_energy=eq)
    table, taps = m.table.cuda(), m.decimater.taps.cuda()
    add = dev(rng.normal(0, 1, (B, Tp)).astype(np.float32))
    for a in (None, add):
        fused = GF.glottal_osc(phase, w, table, taps, 1, w_hop, 4, eq, add=a)
        unfused, pre = GF.glottal_osc(phase, w, table, taps, 1, w_hop, 4, eq, return_pre=True, add=a)
        torch.cuda.synchronize()
        assert fused.shape == unfused.shape == (B, Tp)
        assert torch.isfinite(fused).all()
        scale = float((unfused - (0 if a is None else a)).abs().max()) + 1e-30
        err = float((fused - unfused).abs().max()) / scale
        print(f"Tp={Tp} w_hop={w_hop} eq={eq} add={a is not None}: fused vs three-kernel rel-max {err:.2e}")
        assert err <= 5e-6, (Tp, w_hop, eq, a is not None, err)


def test_fused_kernel_needs_an_aligned_table_and_falls_back_otherwise():
    """osc_fused2 fetches the table rows as 16-byte words: a table whose storage starts off a 16-byte boundary (a view into a
    larger buffer) takes the three-kernel path instead -- same result to the few 1e-7 the two paths differ by, no fault."""
    from golf_amd import functional as GF
    from golf_amd.synth import IndexedGlottalFlowTable

    rng = np.random.default_rng(3)
    B, Tp, w_hop = 2, 6000, 2400
    f0 = rng.uniform(80, 400, (B, 1)) * np.ones((1, Tp))
    phase = dev((f0 / 24000).astype(np.float32))
    w = dev(rng.uniform(0, 1, (B, 4)).astype(np.float32))
    m = IndexedGlottalFlowTable(table_size=50, lf_v2=True, points=1024, oversampling=4, equal_energy=True)
    table, taps = m.table.cuda().contiguous(), m.decimater.taps.cuda()
    buf = torch.zeros(table.numel() + 1, device="cuda")
    off = buf[1:].view_as(table)                    # same values, storage offset 4 bytes
    off.copy_(table)
    assert table.data_ptr() % 16 == 0 and off.data_ptr() % 16 == 4
    a = GF.glottal_osc(phase, w, table, taps, 1, w_hop, 4, True)
    b = GF.glottal_osc(phase, w, off, taps, 1, w_hop, 4, True)
    torch.cuda.synchronize()
    err = float((a - b).abs().max() / a.abs().max())
    print(f"aligned (fused) vs misaligned (three-kernel) table: rel-max {err:.2e}")
    assert torch.isfinite(b).all() and err <= 5e-6


def test_fused_kernel_other_tap_counts():
    """193 taps (zeros = 24, kazane's default design length at q = 4): the 16-step Toeplitz instance; 257 taps exceed it
    and take the three-kernel path -- either way equal to the float64 oracle."""
    from golf_amd import functional as GF
    from golf_amd.synth import Decimate, IndexedGlottalFlowTable
    from oracle import golf_oracle as O

    rng = np.random.default_rng(11)
    B, Tp, w_hop = 2, 5000, 2400
    f0 = rng.uniform(80, 400, (B, 1)) * np.ones((1, Tp))
    phase = (f0 / 24000).astype(np.float32)
    w = rng.uniform(0, 1, (B, 4)).astype(np.float32)
    m = IndexedGlottalFlowTable(table_size=50, lf_v2=True, points=1024, oversampling=4, equal_energy=True)
    for zeros in (24, 32, 4):
        taps = Decimate(4, zeros=zeros).taps
        out = GF.glottal_osc(dev(phase), dev(w), m.table.cuda(), taps.cuda(), 1, w_hop, 4, True).cpu().numpy()
        ref = O.indexed_glottal_forward(phase, 1, w, w_hop, m.table.numpy(), 4, True, decim_taps=taps.numpy())["out"]
        check(out, ref, f"fused osc, {taps.numel()} taps")


@pytest.mark.parametrize("Tp,w_hop,eq", [(10000, 2400, True), (6145, 2400, False), (47761, 2400, True), (2049, 960, True),
                                         (4000, 4800, True)])
def test_fused_backward_wsel_vs_oracle(Tp, w_hop, eq):
    """The fused backward w.r.t. table_select_weight (round 3: gradient tile -> transposed polyphase FIR on the matrix pipe
    -> table-difference lookups -> per-row sums; osc_fused_bwd_kernel) on the GOLF configuration -- ragged lengths, tiles that
    touch one, two and three control frames, f0 jumps, replicate-padded frames -- against the float64 closed-form adjoint."""
    from golf_amd import functional as GF
    from golf_amd.synth import IndexedGlottalFlowTable
    from oracle import golf_oracle as O

    rng = np.random.default_rng(Tp)
    B = 3
    t = np.arange(Tp) / 24000
    f0 = 120 + 60 * np.sin(2 * np.pi * (0.7 + 0.3 * np.arange(B))[:, None] * t[None])
    f0[:, Tp // 3: Tp // 3 + 5] *= 1.5                            # an f0 jump (voicing boundary)
    phase = (f0 / 24000).astype(np.float32)
    Fw = (Tp - 1) // w_hop + 1                                      # one frame short of full coverage: padding is used
    w = rng.uniform(0.02, 0.98, (B, Fw)).astype(np.float32)
    m = IndexedGlottalFlowTable(table_size=100, lf_v2=True, points=2048, oversampling=4, equal_energy=eq)
    table, taps = m.table.numpy(), m.decimater.taps.numpy()
    wt = dev(w).requires_grad_(True)
    out = GF.glottal_osc(dev(phase), wt, dev(table), dev(taps), 1, w_hop, 4, eq)
    gy = rng.normal(0, 1, tuple(out.shape)).astype(np.float32)
    (out * dev(gy)).sum().backward()
    torch.cuda.synchronize()
    ref = O.indexed_glottal_backward(gy, phase, 1, w, w_hop, table, 4, eq, decim_taps=taps)["g_weight"]
    check(wt.grad.cpu().numpy(), ref, f"fused g_wsel Tp{Tp} w_hop{w_hop} eq{eq}", 2e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("Tp,eq", [(47761, True), (6145, False)])
def test_backward_reuses_the_forward_totals_only_when_told(Tp, eq):
    """ABI 5, GOLF_OSC_WS_KEPT: the autograd node hands the backward the workspace its forward filled, and the backward skips
    its own tile-totals launch.  Without the flag nothing is assumed about the workspace: called on a poisoned one, the C entry
    recomputes the totals and the tap fragments itself.  Both give the same gradient bit for bit."""
    from golf_amd import _lib, functional as GF
    from golf_amd.synth import IndexedGlottalFlowTable

    rng = np.random.default_rng(Tp)
    B, w_hop = 3, 2400
    phase = dev((rng.uniform(80, 400, (B, 1)) / 24000 * np.ones((1, Tp))).astype(np.float32))
    Fw = (Tp - 1) // w_hop + 2
    m = IndexedGlottalFlowTable(table_size=100, lf_v2=True, points=2048, oversampling=4, equal_energy=eq)
    table, taps = dev(m.table.numpy()), dev(m.decimater.taps.numpy())
    wt = dev(rng.uniform(0.02, 0.98, (B, Fw)).astype(np.float32)).requires_grad_(True)
    out = GF.glottal_osc(phase, wt, table, taps, 1, w_hop, 4, eq)
    gy = dev(rng.normal(0, 1, tuple(out.shape)).astype(np.float32))
    (out * gy).sum().backward()
    lib = _lib.load()
    n_tab, L = table.shape
    ws = torch.full((lib.golf_glottal_osc_workspace_bytes(B, Tp, 1, Fw, w_hop, L, 4),), 0xFF, dtype=torch.uint8, device="cuda")
    g_w = torch.full_like(wt, float("nan")).detach()
    rc = lib.golf_glottal_osc_bwd_wsel_f32(gy.data_ptr(), gy.stride(0), phase.data_ptr(), phase.stride(0), Tp, 1,
                                           wt.data_ptr(), Fw, w_hop, table.data_ptr(), n_tab, L, 4, int(eq), taps.data_ptr(),
                                           taps.numel(), g_w.data_ptr(), B, gy.shape[1], ws.data_ptr(), ws.numel(),
                                           _lib.stream_ptr(), None)
    assert rc == 0
    torch.cuda.synchronize()
    assert torch.isfinite(g_w).all()
    assert torch.equal(g_w, wt.grad)


# ---------------------------------------------------------------------------------------------
# round 6: the phase scan inside the kernel (decoupled look-back over tagged entries in the workspace; lpc.. glottal_osc.hip OscLook)
# and the taps' fragments prepared once (ABI 6).  Reference: the cumsum of models/synth.py:250-251.
# ---------------------------------------------------------------------------------------------
def _osc_raw(lib, phase, wt, table, taps, w_hop, eq, ws, frags=None, add=None):
    from golf_amd import _lib

    B, Tp = phase.shape
    n_tab, L = table.shape
    out = torch.full((B, Tp), float("nan"), dtype=torch.float32, device="cuda")
    rc = lib.golf_glottal_osc_fwd_f32(phase.data_ptr(), phase.stride(0), Tp, 1, wt.data_ptr(), wt.shape[1], w_hop,
                                      table.data_ptr(), n_tab, L, 4, int(eq), taps.data_ptr(), taps.numel(), None,
                                      out.data_ptr(), out.stride(0), B, Tp, ws.data_ptr(), ws.numel(), _lib.stream_ptr(),
                                      _lib.ptr(add), 0 if add is None else add.stride(0), 0 if add is None else add.shape[1],
                                      _lib.ptr(frags))
    assert rc == 0, lib.golf_last_error()
    return out


@pytest.mark.parametrize("Tp", [48000, 70001, 2048, 2049, 300])
def test_single_pass_scan_on_a_reused_workspace(Tp):
    """The look-back's entries live in the caller's workspace and are never cleared: a launch must not take an earlier launch's
    entries for its own.  One workspace through six launches with changing phases (every fill a fresh allocation could hold
    first: zeros, 0xFF, 0x7F, random bytes), each compared bit for bit with the same call on a workspace of its own, and the
    first against the float64 oracle.  70 001 samples = 35 tiles: the second round of polls; 2048 / 2049: the tile edge."""
    from golf_amd import _lib
    from golf_amd.synth import IndexedGlottalFlowTable
    from oracle import golf_oracle as O

    lib = _lib.load()
    rng = np.random.default_rng(Tp)
    B, w_hop, eq = 5, 2400, True
    Fw = (Tp - 1) // w_hop + 2
    m = IndexedGlottalFlowTable(table_size=100, lf_v2=True, points=2048, oversampling=4, equal_energy=eq)
    table, taps = dev(m.table.numpy()), dev(m.decimater.taps.numpy())
    nbytes = lib.golf_glottal_osc_workspace_bytes(B, Tp, 1, Fw, w_hop, table.shape[1], 4)

    def batch(k):
        f0 = rng.uniform(80, 400, (B, 1)) * (1 + 0.03 * np.sin(2 * np.pi * 5.5 * np.arange(Tp) / 24000 + k))
        return dev((f0 / 24000).astype(np.float32)), dev(rng.uniform(0.02, 0.98, (B, Fw)).astype(np.float32))

    fills = [lambda: torch.zeros(nbytes, dtype=torch.uint8, device="cuda"),
             lambda: torch.full((nbytes,), 0xFF, dtype=torch.uint8, device="cuda"),
             lambda: torch.full((nbytes,), 0x7F, dtype=torch.uint8, device="cuda"),
             lambda: torch.randint(0, 256, (nbytes,), dtype=torch.uint8, device="cuda")]
    for fill in fills:
        ws = fill()
        for k in range(6):
            phase, wt = batch(k)
            got = _osc_raw(lib, phase, wt, table, taps, w_hop, eq, ws)
            want = _osc_raw(lib, phase, wt, table, taps, w_hop, eq, fill())
            torch.cuda.synchronize()
            assert torch.isfinite(got).all()
            assert torch.equal(got, want), (k, float((got - want).abs().max()))
    phase, wt = batch(0)
    got = _osc_raw(lib, phase, wt, table, taps, w_hop, eq, ws).cpu().numpy()
    ref = O.indexed_glottal_forward(phase.cpu().numpy(), 1, wt.cpu().numpy(), w_hop, table.cpu().numpy(), 4, eq,
                                    decim_taps=taps.cpu().numpy())["out"]
    check(got, ref, f"single-pass scan Tp {Tp}")


def test_prepared_tap_fragments_equal_the_inline_layout():
    """ABI 6: the taps' Toeplitz fragments prepared once (golf_glottal_osc_tap_fragments_f32) or laid out inside the call (tap_frags
    NULL): the same output bit for bit, forward and -- through the autograd node, which hands the backward the forward's
    fragments -- the same gradient as a backward that recomputes everything on a poisoned workspace."""
    from golf_amd import _lib, functional as GF
    from golf_amd.synth import IndexedGlottalFlowTable

    lib = _lib.load()
    rng = np.random.default_rng(3)
    B, Tp, w_hop, eq = 3, 9000, 2400, True
    Fw = (Tp - 1) // w_hop + 2
    m = IndexedGlottalFlowTable(table_size=100, lf_v2=True, points=2048, oversampling=4, equal_energy=eq)
    table, taps = dev(m.table.numpy()), dev(m.decimater.taps.numpy())
    phase = dev((rng.uniform(80, 400, (B, 1)) / 24000 * np.ones((1, Tp))).astype(np.float32))
    wt = dev(rng.uniform(0.02, 0.98, (B, Fw)).astype(np.float32))
    add = dev(rng.normal(0, 1, (B, Tp - 7)).astype(np.float32))
    n = lib.golf_glottal_osc_tap_fragments_bytes(taps.numel(), 4)
    assert n == 2 * 4 * 16 * 64 * 4 and lib.golf_glottal_osc_tap_fragments_bytes(taps.numel(), 2) == 0
    frags = torch.full((n,), 0xFF, dtype=torch.uint8, device="cuda")
    assert lib.golf_glottal_osc_tap_fragments_f32(taps.data_ptr(), taps.numel(), 4, frags.data_ptr(), n, _lib.stream_ptr()) == 0
    assert lib.golf_glottal_osc_tap_fragments_f32(taps.data_ptr(), taps.numel(), 4, frags.data_ptr(), n - 1, _lib.stream_ptr()) == -1
    nbytes = lib.golf_glottal_osc_workspace_bytes(B, Tp, 1, Fw, w_hop, table.shape[1], 4)
    mk = lambda: torch.full((nbytes,), 0xFF, dtype=torch.uint8, device="cuda")
    a = _osc_raw(lib, phase, wt, table, taps, w_hop, eq, mk(), frags=frags, add=add)
    b = _osc_raw(lib, phase, wt, table, taps, w_hop, eq, mk(), frags=None, add=add)
    torch.cuda.synchronize()
    assert torch.isfinite(a).all() and torch.equal(a, b)
    assert GF.osc_tap_fragments(taps, 4) is GF.osc_tap_fragments(taps, 4)          # cached per tap set
    k = 4 * 12 * 64 * 4    # (129 taps: 12 K-steps per branch are laid out, the rest of either block is never read)
    half = frags.numel() // 2
    cached = GF.osc_tap_fragments(taps, 4)
    assert torch.equal(cached[:k], frags[:k]) and torch.equal(cached[half:half + k], frags[half:half + k])
    y = GF.glottal_osc(phase, wt, table, taps, 1, w_hop, 4, eq, add=add)
    assert torch.equal(y, a)


def test_throughput_flag_two_launches_are_bit_identical(monkeypatch):
    """GOLF_OSC_THROUGHPUT (ABI 6): with batches in flight the phase scan runs as a launch of its own in front of the fused kernel
    (round 5's form) instead of inside it.  Integer phase arithmetic either way: the same bits, B = 32 full size."""
    from golf_amd import functional as GF
    from golf_amd.synth import DownsampledIndexedGlottalFlowTable
    from golf_amd.synthetic import make_inputs

    inp = make_inputs(B=32, device="cuda", seed=2436)
    osc = DownsampledIndexedGlottalFlowTable(hop_rate=10, in_channels=64, oversampling=4, equal_energy=True, lf_v2=True, points=2048).cuda()
    run = lambda: GF.glottal_osc(inp["phase"], inp["wsel"], osc.table, osc.decimater.taps, 1, inp["w_hop"], 4, True, add=inp["noise"])
    monkeypatch.setattr(GF, "THROUGHPUT_MODE", False)
    a = run()
    monkeypatch.setattr(GF, "THROUGHPUT_MODE", True)
    b = run()
    torch.cuda.synchronize()
    assert torch.isfinite(a).all() and torch.equal(a, b)
