"""CPU: host-side logic of the drop-in modules (AudioTensor, control transforms, tables, protocol)
against the golden vectors captured from the reference's own code."""
import hashlib

import numpy as np
import pytest
import torch

from conftest import rel_err
from golf_amd import utils as U
from golf_amd.audiotensor import AudioTensor


def close(x, ref, tol=1e-6):
    x = x.detach().numpy() if isinstance(x, torch.Tensor) else np.asarray(x)
    assert rel_err(x, ref)[0] <= tol


def test_rc2lpc_biquads(golden):
    g = golden("g1_rc2lpc")
    close(U.rc2lpc(torch.from_numpy(g["rc"])), g["lpc"], 1e-12)
    close(U.rc2lpc(torch.from_numpy(g["rc1"])), g["lpc1"], 1e-12)
    g = golden("g2_biquads")
    lg = torch.from_numpy(g["logits"])
    for t in ("coef", "conj", "real"):
        bq = U.get_logits2biquads(t)(lg)
        close(bq, g["bq_" + t], 1e-12)
        close(U.biquads2lpc(bq), g["lpc_" + t], 1e-11)
    close(U.get_logits2biquads("coef", 0.9)(lg), g["bq_coef_09"], 1e-12)
    with pytest.raises(ValueError):
        U.get_logits2biquads("nope")


def test_biquads_offline_tool_semantics(golden):
    """biquads.py:13-58 re-derives (log_gain, biquads (...,11,3)) from a (1+22)-channel logit slice with
    get_logits2biquads('coef') (default max_abs_pole 0.99)."""
    g = golden("g2_biquads")
    lg = torch.from_numpy(g["logits"])  # (2,3,11,2)
    sl = torch.cat([torch.zeros(2, 3, 1, dtype=lg.dtype), lg.reshape(2, 3, 22)], -1)
    log_gain, bl = sl[..., 0], sl[..., 1:].reshape(2, 3, -1, 2)
    bq = U.get_logits2biquads("coef")(bl)
    assert bq.shape == (2, 3, 11, 3) and log_gain.shape == (2, 3)
    close(bq, g["bq_coef"], 1e-12)


def test_lf_tables(golden):
    from golf_amd.synth import IndexedGlottalFlowTable

    g = golden("g3_lf_tables")
    m = IndexedGlottalFlowTable(table_size=100, lf_v2=True, points=2048)
    close(m.R_d_values, g["Rd"], 1e-7)
    close(m.table[[0, 1, 49, 98, 99]], g["table_v2_rows"], 1e-5)
    assert (m.table.argmin(1).numpy() == g["table_v2_argmin"]).all()
    close(m.table.double().sum(0), g["table_v2_colsum"], 1e-5)
    if hashlib.sha256(m.table.numpy().tobytes()).digest() != bytes(g["table_v2_sha256"]):
        pytest.skip("table not bit-identical on this host's libm (values checked above)")
    m1 = IndexedGlottalFlowTable(table_size=100, lf_v2=False, T_0=5.0, n_iter_eps=5, n_iter_a=100, points=2048)
    close(m1.table[[0, 1, 49, 98, 99]], g["table_v1_rows"], 5e-5)
    assert (m1.table.argmin(1).numpy() == g["table_v1_argmin"]).all()
    mf = IndexedGlottalFlowTable(table_size=12, table_type="flow", normalize_method="peak", lf_v2=True, points=256)
    close(mf.table, g["table_flow"], 1e-5)
    mn = IndexedGlottalFlowTable(table_size=12, normalize_method=None, align_peak=False, lf_v2=False, points=256)
    close(mn.table, g["table_none"], 5e-5)
    with pytest.raises(ValueError):
        IndexedGlottalFlowTable(table_type="nope")


def test_audiotensor_semantics(golden):
    g = golden("g4_upsample")
    z = torch.from_numpy(g["z"])
    x = AudioTensor(z, hop_length=10)
    assert x.shape == (2, 100) and x.hop_length == 10 and x.steps == 100
    x1 = x.reduce_hop_length()
    assert isinstance(x1, AudioTensor) and x1.shape == (2, 991) and x1.hop_length == 1  # test_time_tensor.py:18-22
    close(x1.as_tensor(), g["up10"], 1e-12)
    x2 = x.reduce_hop_length(5)
    assert x2.shape == (2, 496) and x2.hop_length == 2  # test_time_tensor.py:24-28
    close(x2.as_tensor(), g["up5"], 1e-12)
    x3 = x1 + x2 * x
    assert x3.shape == (2, 991) and x3.hop_length == 1  # test_time_tensor.py:30-31
    close(x3.as_tensor(), g["mixed"], 1e-12)
    a3 = AudioTensor(torch.from_numpy(g["z3"]), 4).reduce_hop_length()
    close(a3.as_tensor(), g["up3"], 1e-12)
    ex = AudioTensor(torch.from_numpy(g["ex"]))
    close((ex * AudioTensor(torch.from_numpy(g["z3"][..., 0]), 4)).as_tensor(), g["ex_times_g"], 1e-12)
    # misc contract
    assert AudioTensor(torch.zeros(5)).hop_length == 9223372036854775807
    assert x[:, :7].hop_length == 10 and x.unfold(4, 2).hop_length == 20 and x.unfold(4, 2).shape == (2, 49, 4)
    assert x.increase_hop_length(5).shape == (2, 20) and x.increase_hop_length(5).hop_length == 50
    assert x.set_hop_length(2).shape == (2, 496) and x.set_hop_length(20).shape == (2, 50)
    assert x.truncate(10).shape == (2, 10) and x.truncate(1000) is x
    s = torch.sigmoid(x)
    assert isinstance(s, AudioTensor) and s.hop_length == 10
    assert not isinstance(torch.sum(x), AudioTensor)
    w = torch.where(x > 0, x, 0 * x)
    assert isinstance(w, AudioTensor) and w.shape == (2, 100)
    with pytest.raises(NotImplementedError):
        torch.cat([x, x], 1)
    with pytest.raises(AssertionError):
        torch.exp(x) if False else torch.maximum(x, x1)  # mismatching hops outside the aligned op set


def test_ctrl_protocol(golden):
    from golf_amd.ctrl import PassThrough
    from golf_amd.filters import LTVMinimumPhaseFilter, LTVMinimumPhaseFilterPrecise
    from golf_amd.noise import StandardNormalNoise
    from golf_amd.sf import HarmonicPlusNoiseSynth, SourceFilterSynth
    from golf_amd.synth import DownsampledIndexedGlottalFlowTable

    g = golden("g12_ctrl_protocol")

    from golf_amd.filters import LTIAcousticFilter, LTVZeroPhaseFIRFilter

    def Stub256():  # the noise filter of every GOLF config (the class, not a stand-in, since round 1's widening)
        return LTVZeroPhaseFIRFilter(window="hanning", n_mag=256)

    def decoder(end_filter):
        return SourceFilterSynth(
            harm_oscillator=DownsampledIndexedGlottalFlowTable(hop_rate=10, in_channels=64, oversampling=4,
                                                               equal_energy=True, lf_v2=True, points=64),
            noise_generator=StandardNormalNoise(), noise_filter=Stub256(), end_filter=end_filter,
            room_filter=LTIAcousticFilter(length=128, conv_method="direct"), subtract_harmonics=False)

    for name, ef in (("ss", LTVMinimumPhaseFilterPrecise(lpc_order=22)),
                     ("ff", LTVMinimumPhaseFilter(window="hanning", window_length=960, lpc_order=22)),
                     ("ss_coef", LTVMinimumPhaseFilterPrecise(lpc_order=22, lpc_parameterisation="coef",
                                                              max_abs_value=0.99))):
        split_sizes, trsfms, keys = decoder(ef).split_sizes_and_trsfms
        assert [",".join(map(str, s)) for s in split_sizes] == list(g[name + "_split_sizes"])
        assert list(keys) == list(g[name + "_keys"])
        assert sum(sum(s) for s in split_sizes) == 343  # encoder out_channels
        assert len(trsfms) == 5
    # the oscillator/filter modules own exactly the reference's checkpoint keys
    dec = decoder(LTVMinimumPhaseFilterPrecise(lpc_order=22))
    assert sorted(dec.state_dict().keys()) == sorted(g["ss_state_dict_keys"])  # the WHOLE decoder, key for key
    assert not [k for k in dec.state_dict() if k.startswith("end_filter")]
    # downsampler ctrl with the reference's weights reproduces its table_select_weight
    osc = dec.harm_oscillator
    osc.model.load_state_dict({k[len("ds_model."):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("ds_model.")})
    (w,) = osc.ctrl(lambda s, t: (s, t))((), ())[1][0](AudioTensor(torch.from_numpy(g["ds_h"]), 240))
    assert w.hop_length == int(g["ds_w_hop"]) == 2400 and w.shape == (2, 21)
    close(w.as_tensor(), g["ds_w"], 1e-6)
    # GOLF-v1 assembly order (models/hpn.py:20-29)
    hpn = HarmonicPlusNoiseSynth(harm_oscillator=dec.harm_oscillator, noise_generator=StandardNormalNoise(),
                                 harm_filter=LTVMinimumPhaseFilterPrecise(lpc_order=22), noise_filter=Stub256(),
                                 end_filter=PassThrough())
    ss, _, keys = hpn.split_sizes_and_trsfms
    assert ss == ((64,), (), (1, 22), (256,), ()) and keys[2] == "harm_filter_params"


def test_filter_ctrl_transform(golden):
    from golf_amd.filters import LTVMinimumPhaseFilterPrecise, convert2samplewise

    g = golden("g8_samplewise")
    m = LTVMinimumPhaseFilterPrecise(lpc_order=22)
    (split,), (trsfm,) = m.ctrl(lambda s, t: (s, t))((), ())
    assert split == (1, 22)
    gain, a = trsfm(AudioTensor(torch.from_numpy(g["ctrl_log_gain"]), 240), AudioTensor(torch.from_numpy(g["logits2"]), 240))
    assert gain.hop_length == a.hop_length == 240
    close(gain.as_tensor(), g["ctrl_gain"], 1e-12)
    close(a.as_tensor(), g["ctrl_a"], 1e-11)
    with pytest.raises(ValueError):
        LTVMinimumPhaseFilterPrecise(lpc_order=4, lpc_parameterisation="nope")
    cfg = {"end_filter": {"class_path": "golf_amd.filters.LTVMinimumPhaseFilter",
                          "init_args": {"window": "hanning", "window_length": 960, "lpc_order": 22}}}
    out = convert2samplewise(cfg)
    assert out["end_filter"]["class_path"].endswith("LTVMinimumPhaseFilterPrecise")
    assert out["end_filter"]["init_args"] == {"lpc_order": 22}


def test_noise_sources():
    from golf_amd.noise import SignFlipNoise, StandardNormalNoise, UniformNoise

    torch.manual_seed(0)
    ref = AudioTensor(torch.zeros(4, 20000))
    n = StandardNormalNoise()(ref)
    assert isinstance(n, AudioTensor) and n.shape == (4, 20000)
    assert abs(n.as_tensor().mean()) < 0.02 and abs(n.as_tensor().std() - 1) < 0.02
    u = UniformNoise()(ref).as_tensor()
    assert abs(u.mean()) < 0.03 and abs(u.std() - 1) < 0.02 and u.abs().max() <= 3 ** 0.5
    s = SignFlipNoise()(ref).as_tensor()
    assert (s.abs() == 1).all() and (s[:, 0::2] == -s[:, 1::2]).all() and (s[:, :-2] == s[:, 2:]).all()


def test_decimator_design():
    from golf_amd.synth import Decimate
    from oracle import golf_oracle as O

    d = Decimate(4)
    assert d.kernel.shape == (1, 1, 129)
    np.testing.assert_allclose(d.taps.numpy(), O.default_decimation_taps(4), atol=1e-7)


def test_remaining_oscillator_surface_cpu():
    """Constructor / control-protocol / checkpoint-key surface of the table oscillators outside the GOLF configs
    (reference models/synth.py:266-294, 343-400, 507-523) -- no kernels involved."""
    import torch
    from golf_amd.synth import (DownsampledWeightedGlottalFlowTable, PulseTrain, WeightedGlottalFlowTable,
                                WrappedPhaseDownsampledIndexedGlottalFlowTable)

    w = WeightedGlottalFlowTable(table_size=9, lf_v2=True, points=32)
    split, trs, _ = w.ctrl(lambda s_, t_: (s_, t_, None))((), ())
    assert split == ((9,),)
    d = DownsampledWeightedGlottalFlowTable(hop_rate=10, in_channels=6, table_size=9, lf_v2=True, points=32)
    split, trs, _ = d.ctrl(lambda s_, t_: (s_, t_, None))((), ())
    assert split == ((6,),)
    assert sorted(d.state_dict()) == ["R_d_values", "model.1.bias", "model.1.weight", "model.3.bias", "model.3.weight",
                                      "table"]
    assert d.model[3].out_channels == 9
    from golf_amd.audiotensor import AudioTensor
    (ws,) = trs[0](AudioTensor(torch.randn(2, 40, 6), 240))
    assert ws.shape == (2, 5, 9) and ws.hop_length == 2400
    assert torch.allclose(ws.as_tensor().sum(-1), torch.ones(2, 5), atol=1e-6)
    t = WeightedGlottalFlowTable(table_size=4, lf_v2=True, points=32, trainable=True)
    assert isinstance(t.table, torch.nn.Parameter)
    assert sorted(WrappedPhaseDownsampledIndexedGlottalFlowTable(hop_rate=2, in_channels=3, table_size=5, lf_v2=True,
                                                                 points=32).state_dict()) == \
        ["R_d_values", "model.1.bias", "model.1.weight", "model.3.bias", "model.3.weight", "table"]
    assert PulseTrain().state_dict() == {}


def test_noise_band_bank_matches_reference_g24(golden):
    """NoiseBand's init-time design (scipy kaiserord / firwin, random-phase loop synthesis) under the reference's seed:
    same band centres, same loopable noise periods, same control split and checkpoint keys."""
    import torch
    from golf_amd.noise import NoiseBand

    g = golden("g24_noiseband_ckpt_biquads")
    torch.manual_seed(7)
    nb = NoiseBand(n_filters=12, fs=24000, attenuation=50, normalize_noise_bands=True)
    np.testing.assert_allclose(nb.band_centers.numpy(), g["nb_band_centers"], rtol=1e-6)
    assert nb.noise_bands.shape == g["nb_noise_bands"].shape
    np.testing.assert_allclose(nb.noise_bands.numpy(), g["nb_noise_bands"], rtol=1e-5, atol=1e-6)
    split, trs = nb.ctrl(lambda s_, t_: (s_, t_))((), ())
    assert tuple(split[0]) == tuple(int(v) for v in g["nb_split"])
    assert sorted(nb.state_dict()) == [str(k) for k in g["nb_state_keys"]]


def test_checkpoint_head_remap_g24(golden):
    """ISMIR'23 -> Interspeech'24 encoder-head permutation against the reference's models/utils.py:12-38, plus what
    test_rtf.py:98-132 does around it (drop *_kernel, fix the amplicudes typo, PULF layout)."""
    import torch
    from golf_amd.ckpt import convert_ismir_state_dict, ismir2interspeech_ckpt, permute_head_rows

    g = golden("g24_noiseband_ckpt_biquads")
    sd = {k[len("ck_in/"):]: torch.from_numpy(np.asarray(g[k])) for k in g.files if k.startswith("ck_in/")}
    out = ismir2interspeech_ckpt(sd, lpc_order=22, h_size=8)
    want = {k[len("ck_out/"):]: g[k] for k in g.files if k.startswith("ck_out/")}
    assert set(out) == set(want)
    for k in want:
        np.testing.assert_array_equal(out[k].numpy(), want[k])
    cfg = {"decoder": {"init_args": {
        "harm_oscillator": {"class_path": "models.synth.DownsampledIndexedGlottalFlowTable", "init_args": {"in_channels": 8}},
        "harm_filter": {"class_path": "models.filters.LTVMinimumPhaseFilter", "init_args": {"lpc_order": 22}}}}}
    sd2 = dict(sd)
    sd2["decoder.harm_filter._kernel"] = torch.zeros(3)
    sd2["decoder.harm_oscillator.amplicudes"] = torch.ones(2)
    conv = convert_ismir_state_dict(sd2, cfg)
    assert "decoder.harm_filter._kernel" not in conv and "decoder.harm_oscillator.amplitudes" in conv
    np.testing.assert_array_equal(conv["encoder.backbone.out_linear.bias"].numpy(), want["encoder.backbone.out_linear.bias"])
    # PULF layout: [voice_lpc, voice_gain, noise_lpc, noise_gain] -> [voice_gain, voice_lpc, noise_gain, noise_lpc]
    v = torch.arange(3 + 4 + 1 + 5 + 1, dtype=torch.float32)
    got = permute_head_rows({"x.out_linear.bias": v}, [4, 1, 5, 1], [1, 0, 3, 2])["x.out_linear.bias"]
    assert got.tolist() == [0, 1, 2, 7, 3, 4, 5, 6, 13, 8, 9, 10, 11, 12]


def test_biquads_dump_g24(golden, tmp_path):
    """biquads.py: slicing of the encoder logits by the control protocol + 'coef' sections + the .pt key scheme."""
    import types
    import torch
    from golf_amd.biquads import dump_biquads, get_biquads

    g = golden("g24_noiseband_ckpt_biquads")
    logits = torch.from_numpy(g["bq_logits"])
    enc = types.SimpleNamespace(
        split_sizes=((1,), (1,), (8,), (), (1, 22), (1, 22), ()),
        args_keys=("f0", "voicing_logits", "harm_oscillator_params", "noise_generator_params", "harm_filter_params",
                   "noise_filter_params", "end_filter_params"),
        trsfms=(None, None, lambda h: (torch.sigmoid(h.mean(-1)),), None, None, None, None))
    res = get_biquads(logits, enc)
    names = ("bq_voicing", "bq_harm_log_gain", "bq_harm_biquads", "bq_noise_log_gain", "bq_noise_biquads",
             "bq_table_select_weight")
    assert len(res) == 6
    for got, name in zip(res, names):
        np.testing.assert_allclose(got.numpy(), g[name], rtol=1e-6, atol=1e-7, err_msg=name)
    path = str(tmp_path / "dump.pt")
    dump_biquads([("utt", 0, res), ("utt", 1, res[:5])], path)
    back = torch.load(path)
    assert set(back) == {f"utt_0.{n}" for n in ("voicing", "harm_log_gain", "harm_biquads", "noise_log_gain",
                                                  "noise_biquads", "table_select_weight")} | \
        {f"utt_1.{n}" for n in ("voicing", "harm_log_gain", "harm_biquads", "noise_log_gain", "noise_biquads")}
    assert back["utt_0.harm_biquads"].shape == (2, 7, 11, 3)


def test_lsp2lpc_construction():
    """LSP -> LPC (stand-in for diffsptk.functional.lsp2lpc; models/filters.py:82-86): the closed form for M = 2, the
    defining property for M = 22 and M = 5 -- the roots of P = A + z^-(M+1) A(1/z) and Q = A - z^-(M+1) A(1/z) lie on the
    unit circle at exactly the given frequencies, interlaced -- minimum phase, and the module's control transform."""
    import torch
    from golf_amd.utils import lsp2lpc

    w = torch.tensor([[0.7, 0.4, 1.9]], dtype=torch.float64)
    a = lsp2lpc(w)
    c1, c2 = np.cos(0.4), np.cos(1.9)
    np.testing.assert_allclose(a.numpy(), [[0.7, -(c1 + c2), 1 - c1 + c2]], rtol=1e-12)
    rng = np.random.default_rng(5)
    for M in (22, 5):
        freq = np.sort(rng.uniform(0.05, np.pi - 0.05, (3, M)), axis=-1)
        out = lsp2lpc(torch.from_numpy(np.concatenate([np.ones((3, 1)), freq], -1))).numpy()
        assert out.shape == (3, M + 1)
        for row, fr in zip(out, freq):
            A = np.concatenate([[1.0], row[1:]])
            P = np.concatenate([A, [0.0]]) + np.concatenate([[0.0], A[::-1]])
            Q = np.concatenate([A, [0.0]]) - np.concatenate([[0.0], A[::-1]])
            ang = lambda poly: np.sort(np.angle(np.roots(poly)))
            pa, qa = ang(P), ang(Q)
            assert np.allclose(np.abs(np.roots(P)), 1, atol=1e-6) and np.allclose(np.abs(np.roots(Q)), 1, atol=1e-6)
            pos = lambda v: np.sort(v[(v > 1e-6) & (v < np.pi - 1e-6)])
            np.testing.assert_allclose(pos(pa), fr[0::2], atol=1e-7)
            np.testing.assert_allclose(pos(qa), fr[1::2], atol=1e-7)
            assert np.all(np.abs(np.roots(A)) < 1)                    # interlaced LSPs => minimum phase
    from golf_amd.filters import LTVMinimumPhaseFilterPrecise
    from golf_amd.audiotensor import AudioTensor
    m = LTVMinimumPhaseFilterPrecise(lpc_order=10, lpc_parameterisation="lsp2lpc")
    split, trs, _ = m.ctrl(lambda s_, t_: (s_, t_, None))((), ())
    assert split == ((1, 11),)
    logits = torch.randn(2, 5, 11, requires_grad=True)
    gain, a = trs[0](AudioTensor(torch.zeros(2, 5), 240), AudioTensor(logits, 240))
    assert a.shape == (2, 5, 10) and a.hop_length == 240
    a.as_tensor().sum().backward()
    assert torch.isfinite(logits.grad).all()
    for row in a.as_tensor().detach().reshape(-1, 10).numpy():
        assert np.all(np.abs(np.roots(np.concatenate([[1.0], row]))) < 1)


def test_trainable_shape_grid():
    from golf_amd.functional import ss_is_trainable

    assert ss_is_trainable(22, 240) and ss_is_trainable(26, 240) and ss_is_trainable(38, 240) and ss_is_trainable(30, 256)
    assert not ss_is_trainable(39, 240) and not ss_is_trainable(22, 100) and not ss_is_trainable(31, 256)
    assert not ss_is_trainable(22, 240, F=1)
