"""Encoder, encoder interface, spectral loss and the training step against the reference's own
run (tests/golden/g20; models/unet.py:86-224, models/enc.py:33-100, loss/spec.py:11-67).  Stock-PyTorch modules, so the
parity part runs on the CPU; the GPU test runs the whole training step on the HIP decoder."""
import numpy as np
import pytest
import torch

from conftest import rel_err

ENC_ARGS = dict(n_fft=64, hop_length=16, channels=[4, 8], strides=[2, 2], lstm_hidden_size=8, num_layers=2)


def toy_decoder():
    """The toy GOLF-ss decoder of g14 (same split sizes as the reference object g20's interface was built from)."""
    from golf_amd.filters import LTIAcousticFilter, LTVMinimumPhaseFilterPrecise, LTVZeroPhaseFIRFilter
    from golf_amd.noise import StandardNormalNoise
    from golf_amd.sf import SourceFilterSynth
    from golf_amd.synth import IndexedGlottalFlowTable

    return SourceFilterSynth(
        harm_oscillator=IndexedGlottalFlowTable(table_size=7, table_type="derivative", normalize_method="constant_power",
                                                align_peak=True, lf_v2=True, points=16, oversampling=1,
                                                equal_energy=True),
        noise_generator=StandardNormalNoise(), noise_filter=LTVZeroPhaseFIRFilter(window="hanning", n_mag=9),
        end_filter=LTVMinimumPhaseFilterPrecise(lpc_order=6), room_filter=LTIAcousticFilter(length=8),
        subtract_harmonics=False)


def make_iface():
    from golf_amd.enc import VocoderParameterEncoderInterface

    split_sizes, trsfms, args_keys = toy_decoder().split_sizes_and_trsfms
    # the reference's YAML path resolves to this package's class
    return VocoderParameterEncoderInterface(backbone_type="models.unet.UNetEncoder", learn_voicing=True, learn_f0=True,
                                            f0_min=60.0, f0_max=1000.0, split_sizes=split_sizes, trsfms=trsfms,
                                            args_keys=args_keys, **ENC_ARGS)


def close(y, ref, what, tol=2e-5):
    emax, el2 = rel_err(np.asarray(y), np.asarray(ref))
    print(f"{what}: rel-max {emax:.2e} rel-l2 {el2:.2e}")
    assert emax <= tol and el2 <= tol, (what, emax, el2)


DEVICES = ["cpu", pytest.param("cuda", marks=pytest.mark.gpu)]


@pytest.mark.parametrize("device", DEVICES)
def test_encoder_matches_reference_g20(golden, device):
    from golf_amd.audiotensor import AudioTensor

    g = golden("g20_encoder_and_loss")
    iface = make_iface()
    assert [s for grp in iface.split_sizes for s in grp] == list(g["split_sizes"])
    assert [len(grp) for grp in iface.split_sizes] == list(g["group_lengths"])
    assert list(iface.args_keys) == [str(k) for k in g["args_keys"]]
    state0 = {k[len("state0/"):]: torch.from_numpy(np.asarray(g[k])) for k in g.files if k.startswith("state0/")}
    assert set(state0) == set(iface.state_dict())          # the reference's checkpoint keys, exactly
    iface.load_state_dict(state0)
    iface.to(device)                                        # "cuda": MIOpen / rocBLAS / rocFFT run the same modules
    x = [torch.from_numpy(g[f"x{i}"]).to(device) for i in range(3)]
    f0 = [torch.from_numpy(g[f"f0_{i}"]).to(device) for i in range(3)]
    iface.train()
    for i in range(2):
        h = iface.backbone(AudioTensor(x[i]), f0=AudioTensor(f0[i]))
        assert h.hop_length == 16
        close(h.as_tensor().detach().cpu(), g[f"h_train{i}"], f"h_train{i}")
    iface.eval()
    close(iface.backbone(AudioTensor(x[2]), f0=AudioTensor(f0[2])).as_tensor().detach().cpu(), g["h_eval"], "h_eval")
    sd = iface.state_dict()
    for k in g.files:                                       # running extrema and BatchNorm statistics evolved alike
        if k.startswith("state1/"):
            np.testing.assert_allclose(sd[k[len("state1/"):]].cpu().numpy(), g[k], rtol=1e-5, atol=1e-6, err_msg=k)
    params = iface(AudioTensor(x[2]), f0=AudioTensor(f0[2]))
    seen = 0
    for key, val in params.items():
        for j, t in enumerate(val if isinstance(val, tuple) else (val,)):
            close(t.as_tensor().detach().cpu(), g[f"param/{key}/{j}"], f"param {key}[{j}]")
            assert t.hop_length == int(g[f"param_hop/{key}/{j}"])
            seen += 1
    assert seen == sum(1 for k in g.files if k.startswith("param/"))


def test_out_linear_starts_at_zero():
    """Training starts from the decoder's neutral parameters (models/enc.py:26-27)."""
    iface = make_iface()
    assert float(iface.backbone.out_linear.weight.detach().abs().max()) == 0.0
    assert float(iface.backbone.out_linear.bias.detach().abs().max()) == 0.0


@pytest.mark.parametrize("device", DEVICES)
def test_mss_loss_matches_reference_g20(golden, device):
    from golf_amd.audiotensor import AudioTensor
    from golf_amd.loss import MSSLoss

    g = golden("g20_encoder_and_loss")
    crit = MSSLoss([61, 127, 251], alpha=1.0, window="hanning", center=True).to(device)
    assert [l.spec.hop_length for l in crit.losses] == list(g["loss_hops"])
    pred = torch.from_numpy(g["loss_pred"]).to(device).requires_grad_(True)
    true = torch.from_numpy(g["loss_true"]).to(device)
    val = crit(pred, true)
    val.backward()
    np.testing.assert_allclose(float(val), float(g["loss_value"]), rtol=2e-6 if device == "cpu" else 2e-5)
    close(pred.grad.cpu(), g["loss_g_pred"], "d loss / d pred", 1e-4)
    # AudioTensor in -> AudioTensor out (the reference's step calls .as_tensor() on it, ltng/ae.py:116-118)
    val2 = crit(AudioTensor(pred.detach()), AudioTensor(true))
    assert float(val2.as_tensor()) == pytest.approx(float(val), rel=1e-6)


def test_spectrogram_is_the_dft():
    """The restated front end against a direct DFT of reflect-padded, windowed frames (float64)."""
    from golf_amd.loss import Spectrogram

    rng = np.random.default_rng(3)
    x = rng.normal(0, 1, (2, 300))
    n_fft, hop = 61, 15
    for power in (1, 2.0):
        s = Spectrogram(n_fft=n_fft, hop_length=hop, power=power).double()
        s.window = torch.hann_window(n_fft, dtype=torch.float64)
        got = s(torch.from_numpy(x)).numpy()
        xp = np.pad(x, ((0, 0), (n_fft // 2, n_fft // 2)), mode="reflect")
        nfr = 1 + (xp.shape[1] - n_fft) // hop
        w = torch.hann_window(n_fft, dtype=torch.float64).numpy()
        k = np.arange(n_fft // 2 + 1)[:, None] * np.arange(n_fft)[None, :]
        dft = np.exp(-2j * np.pi * k / n_fft)
        ref = np.stack([np.abs(dft @ (xp[:, f * hop: f * hop + n_fft] * w).T).T for f in range(nfr)], -1) ** power
        assert got.shape == ref.shape == (2, n_fft // 2 + 1, nfr)
        np.testing.assert_allclose(got, ref, rtol=1e-9, atol=1e-9)


def _autoencoder(decoder, hop, n_fft, n_ffts, **enc_over):
    from golf_amd.ae import VoiceAutoEncoder
    from golf_amd.loss import MSSLoss

    enc_args = dict(f0_min=60.0, f0_max=1000.0, backbone_type="models.unet.UNetEncoder", n_fft=n_fft, hop_length=hop,
                    channels=[4, 8], strides=[2, 2], lstm_hidden_size=8, num_layers=2, dropout=0.0,
                    learn_voicing=False, learn_f0=False)
    enc_args.update(enc_over)
    return VoiceAutoEncoder(decoder=decoder, criterion=MSSLoss(n_ffts, alpha=1.0, window="hanning", center=True),
                            encoder_class_path="models.enc.VocoderParameterEncoderInterface",
                            encoder_init_args=enc_args, sample_rate=24000, detach_f0=True, detach_voicing=True,
                            train_with_true_f0=True)


def test_autoencoder_wiring_cpu():
    """Encoder head sized by the decoder's control protocol; vctk.yaml's encoder has out_channels = 343."""
    from golf_amd.synthetic import make_decoder

    model = _autoencoder(make_decoder(), hop=240, n_fft=1024, n_ffts=[509, 1021, 2053], channels=[32, 64, 128, 256],
                         strides=[4, 4, 4, 4], lstm_hidden_size=256, num_layers=3, dropout=0.1)
    assert model.encoder.backbone.out_linear.out_features == 343
    assert model.encoder.args_keys == ("harm_oscillator_params", "noise_generator_params", "noise_filter_params",
                                       "end_filter_params", "room_filter_params")
    assert model.encoder.backbone.lstm.input_size == 2 * 256 + 1
    keys = set(model.state_dict())
    assert {"encoder.backbone.out_linear.weight", "encoder.backbone.cnns.12.weight", "decoder.room_filter.kernel",
            "criterion.losses.2.spec.window", "encoder.backbone.log_spec_max"} <= keys


@pytest.mark.gpu
def test_training_step_runs_and_learns():
    """Config-5 step at toy size on the HIP decoder: loss -> backward through the custom backward kernels -> gradient
    clipping 0.5 -> Adam(1e-4); finite gradients reach the encoder, and repeating the step on one batch lowers the loss."""
    from golf_amd.ae import train_step
    from golf_amd.synthetic import make_decoder

    torch.manual_seed(0)
    B, T, hop = 4, 4800, 240
    model = _autoencoder(make_decoder(), hop=hop, n_fft=256, n_ffts=[127, 251, 509]).cuda()
    with torch.no_grad():
        model.encoder.backbone.out_linear.weight.normal_(0, 0.02)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    t = torch.arange(T, device="cuda") / 24000
    f0 = (120 + 40 * torch.rand(B, 1, device="cuda")) * (1 + 0.03 * torch.sin(2 * np.pi * 5.5 * t))[None]
    f0[:, :600] = 0                                            # an unvoiced stretch
    x = 0.1 * torch.sin(2 * np.pi * torch.cumsum(f0 / 24000, 1)) + 0.01 * torch.randn(B, T, device="cuda")
    uv = torch.full((B, 1), 200.0, device="cuda")
    model.train()
    losses = [float(train_step(model, opt, (x, f0), clip=0.5, unvoiced_f0=uv)) for _ in range(25)]
    print("losses", [round(v, 3) for v in losses[::6]])
    assert all(np.isfinite(losses))
    assert losses[-1] < losses[0]
    g = [p.grad for p in model.encoder.parameters() if p.grad is not None]
    assert g and all(torch.isfinite(v).all() for v in g)
    total = torch.sqrt(sum((v.float() ** 2).sum() for v in g))
    assert float(total) <= 0.5 * 1.001                         # clipped


# ---- data-parallel training step (config 5 at N > 1): DistributedDataParallel over gloo, world size 2, on the CPU.
# The HIP decoder cannot run here, so a small PyTorch decoder speaking the same control protocol stands in for it: what
# is under test is the wrapper (golf_amd.ae.data_parallel + train_step), i.e. that gradients are averaged across ranks.
def _toy_autoencoder():
    import torch.nn as nn

    from golf_amd.ae import VoiceAutoEncoder
    from golf_amd.audiotensor import AudioTensor
    from golf_amd.ctrl import Controllable, Synth, wrap_ctrl_fn
    from golf_amd.loss import MSSLoss

    class ToyOsc(Controllable):
        def __init__(self):
            super().__init__()
            self.scale = nn.Parameter(torch.tensor(0.1))
            self.ctrl = wrap_ctrl_fn(split_size=(1, 3), trsfm_fn=lambda g, h: (g.new_tensor(torch.exp(g.as_tensor())), h))

        def forward(self, phase, gain, harm):
            ph = torch.cumsum(phase.as_tensor(), 1) * (2 * np.pi)
            g = gain.reduce_hop_length().as_tensor()
            a = harm.reduce_hop_length().as_tensor()
            n = min(ph.shape[1], g.shape[1])
            sig = sum(a[:, :n, k] * torch.sin((k + 1) * ph[:, :n]) for k in range(3))
            return AudioTensor(self.scale * g[:, :n] * sig)

    class ToyDecoder(Synth):
        def __init__(self):
            super().__init__()
            self.osc = ToyOsc()

        def forward(self, phase, osc_params, **_):
            return self.osc(phase, *osc_params)

    torch.manual_seed(3)
    model = VoiceAutoEncoder(
        decoder=ToyDecoder(), criterion=MSSLoss([61, 127], alpha=1.0, window="hanning", center=True),
        encoder_class_path="models.enc.VocoderParameterEncoderInterface",
        encoder_init_args=dict(backbone_type="models.unet.UNetEncoder", n_fft=64, hop_length=16, channels=[4, 8],
                               strides=[2, 2], lstm_hidden_size=8, num_layers=1, learn_voicing=False, learn_f0=False),
        sample_rate=8000, train_with_true_f0=True)
    with torch.no_grad():
        model.encoder.backbone.out_linear.weight.normal_(0, 0.1)
    return model


def _ddp_worker(rank, world, port, q):
    import os
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch.distributed as dist

    from golf_amd.ae import data_parallel, train_step

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    model = _toy_autoencoder()                       # same seed on every rank; DDP broadcasts rank 0's weights anyway
    ddp = data_parallel(model)
    opt = torch.optim.Adam(ddp.parameters(), lr=1e-2)
    g = torch.Generator().manual_seed(100 + rank)    # every rank its own shard of the batch
    x = 0.1 * torch.randn(3, 800, generator=g)
    f0 = 100 + 100 * torch.rand(3, 1, generator=g) * torch.ones(1, 800)
    uv = torch.full((3, 1), 150.0)
    before = [p.detach().clone() for p in model.parameters()]
    losses = [float(train_step(ddp, opt, (x, f0), clip=0.5, unvoiced_f0=uv)) for _ in range(3)]
    flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    gathered = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    same = all(torch.equal(gathered[0], t) for t in gathered)            # averaged gradients -> identical weights
    moved = any(not torch.equal(b, p.detach()) for b, p in zip(before, model.parameters()))
    if rank == 0:
        q.put((same, moved, losses))
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_training_step_gloo_world2():
    import socket

    import torch.multiprocessing as mp

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_ddp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    same, moved, losses = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert same, "ranks diverged: gradients were not averaged"
    assert moved and all(np.isfinite(losses))


@pytest.mark.gpu
def test_config5_training_step_full_size_vs_oracle():
    """BASELINE configs[4]: one optimisation step of the golf-precise autoencoder at B = 64 x 2 s (cfg/ae/vctk.yaml:
    U-Net/LSTM encoder -> control transforms -> golf-precise decoder -> MSSLoss(509, 1021, 2053) -> backward -> clip 0.5
    -> Adam 1e-4; reference ltng/ae.py:86-143).  The decoder is what this package replaces, so it is what is pinned:
    on the control tensors the encoder produced, the decoder output equals the float64 oracle composition
    (oracle.golf_ss_decoder) and the gradients arriving at the encoder head (d loss / d table_select_weight, log_mag,
    gain, a) and at the room filter's taps equal the float64 closed-form backward of the same composition driven by the
    same d loss / d x_hat.  Then the real train_step runs (finite, clipped, parameters move)."""
    from golf_amd.ae import train_step
    from golf_amd.audiotensor import AudioTensor
    from golf_amd.synthetic import make_decoder, make_inputs
    from oracle import golf_oracle as O

    torch.manual_seed(2434)
    B, T, hop = 64, 48000, 240
    inp = make_inputs(B=B, device="cuda")
    noise = inp["noise"]
    dec = make_decoder(injected_noise=noise)
    model = _autoencoder(dec, hop=hop, n_fft=1024, n_ffts=[509, 1021, 2053], channels=[32, 64, 128, 256],
                         strides=[4, 4, 4, 4], lstm_hidden_size=256, num_layers=3, dropout=0.1).cuda()
    with torch.no_grad():   # the reference starts the head at exactly 0 (constant parameters); make them vary
        model.encoder.backbone.out_linear.weight.normal_(0, 0.02)
        model.decoder.room_filter.kernel.copy_(make_inputs(B=1, with_noise_filter=True)["room_kernel"])
    model.train()
    phase = inp["phase"]
    f0 = phase * 24000
    f0[:, :4800] = 0                                           # an unvoiced stretch, driven at uv Hz
    uv = torch.empty(B, 1, device="cuda").uniform_(50, 500)
    ph64 = torch.cumsum(phase.double(), 1)
    x = (sum(torch.sin(2 * np.pi * h * ph64) / h for h in range(1, 9)).float() * 0.05 + 0.005 * noise)

    # ---- the step, opened up at the encoder/decoder interface
    params = model.encoder(AudioTensor(x), f0=AudioTensor(f0))
    assert set(params) >= {"harm_oscillator_params", "noise_filter_params", "end_filter_params"}
    leaves = {}
    for key, val in params.items():
        new = []
        for j, t in enumerate(val):
            leaf = t.as_tensor().detach().clone().requires_grad_(True)
            leaves[(key, j)] = leaf
            new.append(AudioTensor(leaf, t.hop_length))
        params[key] = tuple(new)
    drive = torch.where(f0 == 0, uv, f0) / 24000
    x_hat = model.decoder(phase=AudioTensor(drive), **params).as_tensor()
    x_hat.retain_grad()
    n = min(x.shape[1], x_hat.shape[1])
    loss = model.criterion(x_hat[:, :n], x[:, :n])
    loss.backward()
    torch.cuda.synchronize()
    assert x_hat.shape == (B, 47760) and torch.isfinite(loss)
    g_xhat = x_hat.grad.double().cpu().numpy()

    wsel, log_mag = leaves[("harm_oscillator_params", 0)], leaves[("noise_filter_params", 0)]
    gain, a = leaves[("end_filter_params", 0)], leaves[("end_filter_params", 1)]
    assert wsel.shape == (B, 21) and log_mag.shape == (B, 200, 256) and gain.shape == (B, 200) and a.shape == (B, 200, 22)
    osc = model.decoder.harm_oscillator
    table, taps = osc.table.cpu().numpy(), osc.decimater.taps.cpu().numpy()
    room = model.decoder.room_filter.kernel.detach().cpu().numpy()
    win = torch.hann_window(510, dtype=torch.float64).numpy()
    npf = lambda t: t.detach().cpu().numpy().astype(np.float64)
    # ---- float64 forward of the same composition
    src, y_lpc = O.golf_ss_decoder(npf(drive), 1, npf(wsel), 2400, table, npf(noise), npf(log_mag), win, npf(gain),
                                   npf(a), hop, room_kernel=None, oversampling=4, equal_energy=True, decim_taps=taps)
    y_ref = O.lti_acoustic_filter_forward(y_lpc, room)
    close(x_hat.detach().cpu().numpy(), y_ref, "config-5 decoder output B=64", 1e-4)
    # ---- float64 backward, stage by stage, driven by the loss gradient of the GPU run
    g_y, g_room = O.lti_acoustic_filter_backward(g_xhat, y_lpc, room)
    g_src, g_gain, g_a = O.ltv_allpole_ss_backward(g_y, src, npf(gain), npf(a), hop)
    nz = npf(noise)[:, :48000]
    g_nz_in, g_log_mag = O.ltv_fir_frames_backward(g_src, nz, npf(log_mag), win, hop)
    g_osc = np.zeros((B, 48000))
    g_osc[:, : g_src.shape[1]] = g_src
    g_w = O.indexed_glottal_backward(g_osc, npf(drive), 1, npf(wsel), 2400, table, 4, True, None, taps)["g_weight"]
    for name, got, ref, tol in (("gain", gain.grad, g_gain, 2e-4), ("a", a.grad, g_a, 2e-4),
                                ("log_mag", log_mag.grad, g_log_mag, 2e-4), ("table_select_weight", wsel.grad, g_w, 2e-4),
                                ("room kernel", model.decoder.room_filter.kernel.grad, g_room, 2e-4)):
        close(got.cpu().numpy(), ref, f"config-5 d loss / d {name} (B=64)", tol)

    # ---- and the step itself, as bench.py --workload golf-ss-train-step times it
    opt = torch.optim.Adam(model.parameters(), lr=1e-4, fused=True)
    w0 = model.encoder.backbone.out_linear.weight.detach().clone()
    l1, _ = train_step(model, opt, (x, f0), clip=0.5, unvoiced_f0=uv, return_output=True)
    assert torch.isfinite(l1)            # (not compared with `loss`: dropout draws differ between the two calls)
    g = [p.grad for p in model.parameters() if p.grad is not None]
    total = torch.sqrt(sum((v.float() ** 2).sum() for v in g))
    assert float(total) <= 0.5 * 1.001
    assert not torch.equal(model.encoder.backbone.out_linear.weight, w0)
