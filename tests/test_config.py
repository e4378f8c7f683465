"""The reference's plugin surface is YAML (``class_path`` + ``init_args``): the same trees must build on
this package's classes.  The YAML below is written for this test in the shape of the reference's decoder / model files
(cfg/ae/decoder/golf*.yaml, ckpts/interspeech24/*/config.yaml); when the reference checkout is present (this container
only) its own shipped files are instantiated as well."""
import glob
import os

import numpy as np
import pytest
import torch

from conftest import rel_err

OSC = """
      class_path: models.synth.DownsampledIndexedGlottalFlowTable
      init_args: {hop_rate: 10, in_channels: 64, oversampling: 4, equal_energy: true, table_size: 100,
                  table_type: derivative, normalize_method: constant_power, align_peak: true, trainable: false,
                  min_R_d: 0.3, max_R_d: 2.7, lf_v2: true, points: 2048}
"""
GOLF_FF = """
decoder:
  class_path: models.sf.SourceFilterSynth
  init_args:
    harm_oscillator: %s
    noise_generator: {class_path: models.noise.StandardNormalNoise}
    noise_filter:
      class_path: models.filters.LTVZeroPhaseFIRFilter
      init_args: {window: hanning, conv_method: direct, n_mag: 256}
    end_filter:
      class_path: models.filters.LTVMinimumPhaseFilter
      init_args:
        window: ${decoder.init_args.noise_filter.init_args.window}
        window_length: 960
        lpc_order: 22
        lpc_parameterisation: rc2lpc
    room_filter:
      class_path: models.filters.LTIAcousticFilter
      init_args: {length: 128, conv_method: fft}
    subtract_harmonics: false
""" % OSC
GOLF_V1 = """
decoder:
  class_path: models.hpn.HarmonicPlusNoiseSynth
  init_args:
    harm_oscillator: %s
    noise_generator: {class_path: models.noise.StandardNormalNoise}
    harm_filter:
      class_path: models.filters.LTVMinimumPhaseFilter
      init_args: {window: hanning, window_length: 960, lpc_order: 22, lpc_parameterisation: rc2lpc}
    noise_filter:
      class_path: models.filters.LTVZeroPhaseFIRFilter
      init_args: {window: hanning, conv_method: direct, n_mag: 256}
    end_filter:
      class_path: models.filters.LTIAcousticFilter
      init_args: {length: 128, conv_method: fft}
""" % OSC
MODEL = """
model:
  class_path: ltng.ae.VoiceAutoEncoder
  init_args:
    decoder: %s
    criterion:
      class_path: loss.spec.MSSLoss
      init_args: {n_ffts: [509, 1021, 2053], alpha: 1.0, ratio: 1.0, overlap: 0.75, window: hanning, win_length: null,
                  pad: 0, normalized: false, wkwargs: null, center: true, pad_mode: reflect, onesided: true,
                  return_complex: null}
    encoder_class_path: models.enc.VocoderParameterEncoderInterface
    encoder_init_args: {f0_min: 60.0, f0_max: 1000.0, backbone_type: models.unet.UNetEncoder, n_fft: 1024,
                        hop_length: 240, channels: [32, 64, 128, 256], strides: [4, 4, 4, 4], lstm_hidden_size: 256,
                        num_layers: 3, dropout: 0.1, learn_voicing: false, learn_f0: false}
    sample_rate: 24000
    detach_f0: true
    detach_voicing: true
    train_with_true_f0: true
    f0_loss_weight: 1.0
    voicing_loss_weight: 1.0
optimizer:
  class_path: torch.optim.Adam
  init_args: {lr: 0.0001}
""" % GOLF_FF.split("decoder:", 1)[1].replace("\n", "\n    ").replace("${decoder.", "${model.init_args.decoder.")


def test_decoder_yaml_builds_on_drop_in_classes():
    from golf_amd.config import build_model
    from golf_amd.filters import LTIAcousticFilter, LTVMinimumPhaseFilter, LTVZeroPhaseFIRFilter
    from golf_amd.sf import HarmonicPlusNoiseSynth, SourceFilterSynth

    ff = build_model(GOLF_FF)
    assert type(ff) is SourceFilterSynth and type(ff.end_filter) is LTVMinimumPhaseFilter
    assert ff.end_filter.window_length == 960 if hasattr(ff.end_filter, "window_length") else True
    assert ff.split_sizes_and_trsfms[0] == ((64,), (), (256,), (1, 22), ())
    v1 = build_model(GOLF_V1)
    assert type(v1) is HarmonicPlusNoiseSynth and type(v1.end_filter) is LTIAcousticFilter
    assert type(v1.noise_filter) is LTVZeroPhaseFIRFilter
    assert v1.split_sizes_and_trsfms[0] == ((64,), (), (1, 22), (256,), ())
    assert v1.split_sizes_and_trsfms[2] == ("harm_oscillator_params", "noise_generator_params", "harm_filter_params",
                                            "noise_filter_params", "end_filter_params")


def test_model_yaml_builds_the_autoencoder():
    from golf_amd.ae import VoiceAutoEncoder
    from golf_amd.config import build_model, instantiate, load_yaml

    model = build_model(MODEL)
    assert type(model) is VoiceAutoEncoder
    assert model.encoder.backbone.out_linear.out_features == 343
    assert [l.spec.n_fft for l in model.criterion.losses] == [509, 1021, 2053]
    cfg = load_yaml(MODEL)
    opt_cls = instantiate({"class_path": "torch.optim.Adam", "init_args": {"params": [torch.nn.Parameter(torch.zeros(1))],
                                                                        **cfg["optimizer"]["init_args"]}})
    assert isinstance(opt_cls, torch.optim.Adam) and opt_cls.defaults["lr"] == 1e-4


def test_unknown_class_is_reported():
    from golf_amd.config import build_model

    with pytest.raises(NotImplementedError, match="LTVMLSAFilter"):
        build_model({"decoder": {"class_path": "models.filters.LTVMLSAFilter", "init_args": {}}})


REF = "/root/reference"
GOLF_FILES = ["cfg/ae/decoder/golf.yaml", "cfg/ae/decoder/golf-precise.yaml", "cfg/ae/decoder/golf-v1.yaml",
              "cfg/ae/decoder/ddsp.yaml", "ckpts/interspeech24/golf-ss/config.yaml",
              "ckpts/interspeech24/golf-ff/config.yaml", "ckpts/interspeech24/golf-v1/config.yaml",
              "ckpts/interspeech24/ddsp/config.yaml", "cfg/ae/decoder/nhv.yaml",
              "ckpts/interspeech24/nhv/config.yaml", "cfg/ae/decoder/world.yaml",
              "ckpts/interspeech24/world/config.yaml"] + [f"ckpts/ismir23/{m}_{v}/config.yaml"
                                                         for m in ("glottal_d", "ddsp", "pulse", "sawsing")
                                                         for v in ("f1", "m1")]
ISMIR_SPLITS = {"glottal_d": ((64,), (), (1, 22), (1, 22), ()), "ddsp": ((1, 150), (), (), (80,), ()),
                "pulse": ((), (), (1, 26), (1, 22), ()), "sawsing": ((), (), (256,), (80,), ())}


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not present (GPU box)")
@pytest.mark.parametrize("rel", GOLF_FILES)
def test_shipped_golf_configs_instantiate(rel):
    """Every GOLF / DDSP / NHV / WORLD config the reference ships builds unchanged (the MLSA baseline needs diffsptk's
    MLSA filter, which is out of scope and raises NotImplementedError, see test_unknown_class_is_reported)."""
    from golf_amd.config import build_model

    model = build_model(os.path.join(REF, rel))
    dec = getattr(model, "decoder", model)
    if "ismir23" in rel:   # ISMIR'23 models: the decoder of each (the Lightning module around it is control plane)
        assert dec.split_sizes_and_trsfms[0] == ISMIR_SPLITS[rel.split("/")[2].rsplit("_", 1)[0]]
        return
    total = sum(s for grp in dec.split_sizes_and_trsfms[0] for s in grp)
    # nhv: 241 cepstra + 256 magnitudes; world: 256 magnitudes + 80 mel bands
    assert total == (412 if "ddsp" in rel else 497 if "nhv" in rel else 336 if "world" in rel else 343)
    if hasattr(model, "encoder"):
        assert model.encoder.backbone.out_linear.out_features == total


def _params(dec, inp, device):
    from golf_amd.audiotensor import AudioTensor

    A = lambda k, hop=1: AudioTensor(inp[k], hop)
    return dict(phase=A("phase"), harm_oscillator_params=(A("wsel", inp["w_hop"]),), noise_generator_params=(),
                noise_filter_params=(A("log_mag", 240),))


@pytest.mark.gpu
def test_yaml_built_golf_v1_and_golf_ff_decoders_vs_oracle():
    """golf-v1 (HarmonicPlusNoiseSynth: frame-wise LPC on the oscillator + filtered noise -> room filter) and golf
    (SourceFilterSynth with the frame-wise end filter), built from YAML, against the float64 oracle composition."""
    from golf_amd.audiotensor import AudioTensor
    from golf_amd.config import build_model
    from golf_amd.noise import NoiseInterface
    from golf_amd.synthetic import make_inputs
    from oracle import golf_oracle as O

    B = 2
    inp = make_inputs(B=B, T=24000, device="cuda", with_noise_filter=True)

    class Fixed(NoiseInterface):
        def forward(self, ref, *args, **kwargs):
            return AudioTensor(inp["noise"][:, : ref.shape[1]])

    c = lambda t: t.double().cpu().numpy()
    win960 = torch.hann_window(960, dtype=torch.float64).numpy()
    win510 = torch.hann_window(510, dtype=torch.float64).numpy()
    kern = O.zero_phase_fir_kernels(c(inp["log_mag"]), win510)
    for text in (GOLF_V1, GOLF_FF):
        dec = build_model(text).cuda()
        dec.noise_generator = Fixed()
        room = dec.end_filter if "hpn" in text else dec.room_filter
        with torch.no_grad():
            room.kernel.copy_(inp["room_kernel"])
        osc = dec.harm_oscillator
        src = O.indexed_glottal_forward(c(inp["phase"]), 1, c(inp["wsel"]), inp["w_hop"], c(osc.table), 4, True,
                                        decim_taps=c(osc.decimater.taps))["out"]
        nz = O.ltv_fir_frames_forward(c(inp["noise"])[:, : src.shape[1]], kern, 240)
        lpc = (AudioTensor(inp["gain"], 240), AudioTensor(inp["a"], 240))
        if "hpn" in text:
            y = dec(**_params(dec, inp, "cuda"), harm_filter_params=lpc)
            harm = O.lti_frames_ola_forward(src, c(inp["gain"]), c(inp["a"]), 240, win960)[0]
            n = min(harm.shape[1], nz.shape[1])
            ref = O.lti_acoustic_filter_forward(harm[:, :n] + nz[:, :n], c(inp["room_kernel"]))
        else:
            y = dec(**_params(dec, inp, "cuda"), end_filter_params=lpc)
            n = min(src.shape[1], nz.shape[1])
            ref = O.lti_acoustic_filter_forward(
                O.lti_frames_ola_forward(src[:, :n] + nz[:, :n], c(inp["gain"]), c(inp["a"]), 240, win960)[0],
                c(inp["room_kernel"]))
        y = y.as_tensor().detach().cpu().numpy()
        assert y.shape == ref.shape, (y.shape, ref.shape)
        emax, el2 = rel_err(y, ref)
        print(type(dec).__name__, "vs oracle", emax, el2)
        assert emax < 1e-4 and el2 < 1e-4


GLOTTAL_D = """
decoder:
  class_path: models.hpn.HarmonicPlusNoiseSynth
  init_args:
    harm_oscillator:
      class_path: models.synth.DownsampledIndexedGlottalFlowTable
      init_args: {hop_rate: 10, in_channels: 64, table_size: 100, table_type: derivative,
                  normalize_method: constant_power, align_peak: true, trainable: false, min_R_d: 0.3, max_R_d: 2.7,
                  T_0: 5.0, n_iter_eps: 5, n_iter_a: 100, points: 2048}
    noise_generator: {class_path: models.noise.StandardNormalNoise}
    harm_filter:
      class_path: models.filters.LTVMinimumPhaseFilter
      init_args: {window: hanning, window_length: 480, centred: false, lpc_order: 22, lpc_parameterisation: coef,
                  max_abs_value: 0.99}
    noise_filter:
      class_path: models.filters.LTVMinimumPhaseFilter
      init_args: {window: hanning, window_length: 480, centred: false, lpc_order: 22, lpc_parameterisation: coef,
                  max_abs_value: 0.99}
    end_filter: {class_path: models.ctrl.PassThrough}
"""


@pytest.mark.gpu
def test_yaml_built_ismir23_glottal_decoder_vs_oracle():
    """The ISMIR'23 GOLF decoder (ckpts/ismir23/glottal_d_*: LF-v1 wavetable without oversampling, two frame-wise LPC
    filters with a 480-sample window, ``centred: false``, biquad ("coef") parameterisation) through the encoder-side
    control transforms, against the float64 oracle composition."""
    from golf_amd.audiotensor import AudioTensor
    from golf_amd.config import build_model
    from golf_amd.noise import NoiseInterface
    from golf_amd.synthetic import make_inputs
    from oracle import golf_oracle as O

    B, T, hop = 2, 12000, 240
    inp = make_inputs(B=B, T=T, device="cuda")
    F = T // hop
    g = torch.Generator().manual_seed(5)
    dec = build_model(GLOTTAL_D).cuda()

    class Fixed(NoiseInterface):
        def forward(self, ref, *args, **kwargs):
            return AudioTensor(inp["noise"][:, : ref.shape[1]])

    dec.noise_generator = Fixed()
    split_sizes, trsfms, keys = dec.split_sizes_and_trsfms
    assert split_sizes == ((64,), (), (1, 22), (1, 22), ())
    # per-utterance spread of the biquad logits: 0.2 keeps the poles where an ORDER-22 DIRECT FORM is well conditioned in
    # fp32 (the reference runs the same fp32 direct form): measured rel-max 4e-7 / 1.7e-5 / 2e-3 at spread 0.1 / 0.25 / 0.5
    SPREAD = 0.2
    # filter parameters through the modules' own control transforms (log-gain, 11 x 2 biquad logits -> direct form)
    params = []
    for k in range(2):
        lg = AudioTensor((torch.randn(B, F, generator=g) * 0.1 - 2).cuda(), hop)
        lo = AudioTensor((torch.cumsum(torch.randn(B, F, 22, generator=g) * 0.03, 1) + torch.randn(B, 1, 22, generator=g) * SPREAD).cuda(), hop)
        params.append(trsfms[2 + k](lg, lo))
    y = dec(phase=AudioTensor(inp["phase"]), harm_oscillator_params=(AudioTensor(inp["wsel"][:, : T // 2400 + 1], 2400),),
            noise_generator_params=(), harm_filter_params=params[0], noise_filter_params=params[1])
    y = y.as_tensor().detach().cpu().numpy()
    c = lambda t: t.double().cpu().numpy()
    osc = dec.harm_oscillator
    src = O.indexed_glottal_forward(c(inp["phase"]), 1, c(inp["wsel"][:, : T // 2400 + 1]), 2400, c(osc.table), 1, False)["out"]
    win = torch.hann_window(480, dtype=torch.float64).numpy()
    ga = [(c(p[0].as_tensor()), c(p[1].as_tensor())) for p in params]
    harm = O.lti_frames_ola_forward(src, ga[0][0], ga[0][1], hop, win, centred=False)[0]
    nz = O.lti_frames_ola_forward(c(inp["noise"])[:, : src.shape[1]], ga[1][0], ga[1][1], hop, win, centred=False)[0]
    n = min(harm.shape[1], nz.shape[1])
    ref = harm[:, :n] + nz[:, :n]
    assert y.shape == ref.shape, (y.shape, ref.shape)
    emax, el2 = rel_err(y, ref)
    print("ISMIR'23 glottal_d decoder vs oracle", emax, el2)
    assert emax < 1e-4 and el2 < 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["golf-ff", "golf-ss"])
def test_rtf_harness_on_one_clip(which):
    """BASELINE configs[0] — the reference's test_rtf.py protocol (one 2 s clip, f0 at hop sr // 200 so that the
    oscillator gets its phase at hop 120, analysis then synthesis, drop fastest/slowest) on a model built from YAML."""
    from golf_amd.rtf import run

    from golf_amd.config import load_yaml

    cfg = load_yaml(MODEL)
    if which == "golf-ss":
        cfg["model"]["init_args"]["decoder"]["init_args"]["end_filter"] = {
            "class_path": "models.filters.LTVMinimumPhaseFilterPrecise",
            "init_args": {"lpc_order": 22, "lpc_parameterisation": "rc2lpc"}}
    r = run(cfg, num=5, duration=2.0)
    print(which, {k: (round(v, 6) if isinstance(v, float) else v) for k, v in r.items()})
    assert r["duration"] == 2.0 and 47000 < r["samples_out"] <= 48000
    assert 0 < r["synthesis_rtf"] < 0.05 and 0 < r["analysis_rtf"] < 0.5   # far below real time on an MI355X


@pytest.mark.parametrize("mode", ["zero", "min"])
def test_cep_filter_matches_reference_g21(golden, mode):
    """LTVCepFilter (the NHV baseline's harmonic filter, models/filters.py:559-623; stock PyTorch here as there) against
    the reference's own run in float64: output and both gradients."""
    from golf_amd.audiotensor import AudioTensor
    from golf_amd.filters import LTVCepFilter

    g = golden("g21_cep_filter")
    flt = LTVCepFilter(filter_order=24, n_fft=128, window="hanning", hop_length=32, phase=mode).double()
    ex = torch.from_numpy(g[f"{mode}_ex"]).requires_grad_(True)
    ceps = torch.from_numpy(g[f"{mode}_ceps"]).requires_grad_(True)
    y = flt(AudioTensor(ex), AudioTensor(ceps, 32)).as_tensor()
    assert y.shape == g[f"{mode}_y"].shape
    (y * torch.from_numpy(g[f"{mode}_gy"])).sum().backward()
    for what, got, want in (("y", y.detach(), g[f"{mode}_y"]), ("g_ex", ex.grad, g[f"{mode}_g_ex"]),
                            ("g_ceps", ceps.grad, g[f"{mode}_g_ceps"])):
        emax, el2 = rel_err(got.numpy(), want)
        print("g21", mode, what, emax, el2)
        assert emax < 1e-9 and el2 < 1e-9


@pytest.mark.gpu
def test_nhv_decoder_runs_on_gpu():
    """cfg/ae/decoder/nhv.yaml in this test's own YAML: additive pulse train (HIP) -> cepstral filter (rocFFT) + filtered
    noise (HIP) -> room filter (HIP); the cepstral branch is checked against its float64 CPU evaluation."""
    from golf_amd.audiotensor import AudioTensor
    from golf_amd.config import build_model

    text = """
decoder:
  class_path: models.hpn.HarmonicPlusNoiseSynth
  init_args:
    harm_oscillator: {class_path: models.synth.AdditivePulseTrain, init_args: {num_harmonics: 155}}
    noise_generator: {class_path: models.noise.StandardNormalNoise}
    noise_filter: {class_path: models.filters.LTVZeroPhaseFIRFilter, init_args: {window: hanning, n_mag: 256}}
    harm_filter:
      class_path: models.filters.LTVCepFilter
      init_args: {n_fft: 1024, window: "${decoder.init_args.noise_filter.init_args.window}", filter_order: 240,
                  hop_length: 240, phase: min}
    end_filter: {class_path: models.filters.LTIAcousticFilter, init_args: {length: 128, conv_method: fft}}
"""
    dec = build_model(text).cuda()
    assert dec.split_sizes_and_trsfms[0] == ((), (), (241,), (256,), ())
    B, T, F = 2, 12000, 51
    gen = torch.Generator().manual_seed(8)
    phase = AudioTensor((torch.rand(B, 1, generator=gen) * 0.01 + 0.004).expand(B, T).contiguous().cuda())
    ceps = AudioTensor((torch.randn(B, F, 241, generator=gen) * 0.05 / (1 + torch.arange(241))).cuda(), 240)
    lm = AudioTensor((torch.randn(B, F, 256, generator=gen) * 0.3 - 4).cuda(), 240)
    y = dec(phase=phase, harm_oscillator_params=(), noise_generator_params=(), harm_filter_params=(ceps,),
            noise_filter_params=(lm,)).as_tensor()
    assert y.shape[0] == B and abs(y.shape[1] - T) <= 240 and torch.isfinite(y).all()
    # the cepstral branch alone, GPU float32 vs CPU float64
    src = dec.harm_oscillator(phase)
    got = dec.harm_filter(src, ceps).as_tensor().cpu().numpy()
    ref_filter = type(dec.harm_filter)(filter_order=240, n_fft=1024, window="hanning", hop_length=240, phase="min").double()
    ref = ref_filter(AudioTensor(src.as_tensor().double().cpu()), AudioTensor(ceps.as_tensor().double().cpu(), 240))
    emax, el2 = rel_err(got, ref.as_tensor().numpy())
    print("nhv cepstral branch gpu vs cpu64", emax, el2)
    assert emax < 1e-4 and el2 < 1e-4


def test_world_sp_filter_matches_reference_g22(golden):
    """DiffWorldSPFilter (WORLD baseline, models/filters.py:717-760) against the reference's own run: the rectified
    pseudo-inverse filterbank, the output and both gradients (float32 like the reference)."""
    from golf_amd.audiotensor import AudioTensor
    from golf_amd.filters import DiffWorldSPFilter

    g = golden("g22_world_sp_filter")
    flt = DiffWorldSPFilter(n_mels=12, n_fft=128, hop_length=32, f_min=0.0, f_max=4000.0, center=True, window="hanning",
                            sample_rate=8000, norm=None, mel_scale="htk")
    np.testing.assert_allclose(flt.fb.numpy(), g["inv_fb"], rtol=1e-4, atol=1e-5)
    (split, trs) = flt.ctrl(lambda s_, t_: (s_, t_))((), ())
    assert tuple(split[0]) == tuple(g["split"])
    ex = torch.from_numpy(g["ex"]).requires_grad_(True)
    logmel = torch.from_numpy(g["logmel"]).requires_grad_(True)
    (mel_sp,) = trs[0](AudioTensor(logmel, 32))
    y = flt(AudioTensor(ex), mel_sp).as_tensor()
    (y * torch.from_numpy(g["gy"])).sum().backward()
    for what, got, want in (("y", y.detach(), g["y"]), ("g_ex", ex.grad, g["g_ex"])):
        emax, el2 = rel_err(got.numpy(), want)
        print("g22", what, emax, el2)
        assert emax < 2e-5 and el2 < 2e-5
    # the reference's own gradient w.r.t. the envelope is NaN here: relu(pinv(fb)) leaves frequency bins with zero
    # gain and d sqrt(0) is infinite.  Same formula, same NaNs — reproduced, not "fixed".
    assert np.isnan(g["g_logmel"]).all() and torch.isnan(logmel.grad).all()
