"""GPU: the drop-in nn.Modules assembled like the reference decoder (models/sf.py) against golden g11."""
import numpy as np
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu


def test_source_filter_synth_g11(golden):
    from golf_amd.audiotensor import AudioTensor
    from golf_amd.ctrl import PassThrough
    from golf_amd.filters import LTVMinimumPhaseFilterPrecise
    from golf_amd.noise import NoiseInterface
    from golf_amd.sf import SourceFilterSynth
    from golf_amd.synth import IndexedGlottalFlowTable

    g = golden("g11_source_filter")
    dev = lambda x: torch.as_tensor(np.asarray(x), dtype=torch.float32).cuda()
    noise = dev(g["noise"])

    class FixedNoise(NoiseInterface):
        def forward(self, ref, *args, **kwargs):
            return AudioTensor(noise[:, : ref.shape[1]])

    osc = IndexedGlottalFlowTable(table_size=7, lf_v2=True, points=16, oversampling=1, equal_energy=True)
    np.testing.assert_allclose(osc.table.numpy(), g["table"], rtol=0, atol=1e-6)
    dec = SourceFilterSynth(harm_oscillator=osc, noise_generator=FixedNoise(), noise_filter=PassThrough(),
                            end_filter=LTVMinimumPhaseFilterPrecise(lpc_order=6), room_filter=None,
                            subtract_harmonics=False).cuda()
    hop, w_hop = int(g["hop"]), int(g["w_hop"])
    kw = dict(phase=AudioTensor(dev(g["phase"])), harm_oscillator_params=(AudioTensor(dev(g["w"]), w_hop),),
              noise_generator_params=(), noise_filter_params=(),
              end_filter_params=(AudioTensor(dev(g["gain"]), hop), AudioTensor(dev(g["a"]), hop)))
    y = dec(**kw).as_tensor().cpu().numpy()
    emax, el2 = rel_err(y, g["y"])
    print("g11 composition", emax, el2)
    assert emax < 1e-4 and el2 < 1e-4
    yv = dec(voicing=AudioTensor(dev(g["voicing"]), hop), **kw).as_tensor().cpu().numpy()
    emax, el2 = rel_err(yv, g["y_voiced"])
    print("g11 voiced", emax, el2)
    assert emax < 1e-4 and el2 < 1e-4


def test_golf_ss_decoder_trains():
    """One optimisation step through the whole GOLF-ss source+filter path: loss decreases, grads finite."""
    from golf_amd.audiotensor import AudioTensor
    from golf_amd.ctrl import PassThrough
    from golf_amd.filters import LTVMinimumPhaseFilterPrecise
    from golf_amd.noise import StandardNormalNoise
    from golf_amd.sf import SourceFilterSynth
    from golf_amd.synth import DownsampledIndexedGlottalFlowTable
    from golf_amd.synthetic import make_inputs

    torch.manual_seed(0)
    dec = SourceFilterSynth(
        harm_oscillator=DownsampledIndexedGlottalFlowTable(hop_rate=10, in_channels=8, oversampling=4,
                                                           equal_energy=True, lf_v2=True, points=2048),
        noise_generator=StandardNormalNoise(), noise_filter=PassThrough(),
        end_filter=LTVMinimumPhaseFilterPrecise(lpc_order=22), room_filter=None, subtract_harmonics=False).cuda()
    split_sizes, trsfms, keys = dec.split_sizes_and_trsfms
    assert split_sizes == ((8,), (), (), (1, 22), ())
    inp = make_inputs(B=4, T=9600, device="cuda")
    F = 40
    h = torch.zeros(4, F, 8 + 1 + 22, device="cuda", requires_grad=True)
    target = torch.randn(4, 9361, device="cuda") * 0.01
    opt = torch.optim.Adam([h] + list(dec.parameters()), lr=1e-2)
    losses = []
    for _ in range(3):
        chunks = torch.split(h, [8, 1, 22], dim=2)
        hh = AudioTensor(chunks[0], 240)
        (w,) = trsfms[0](hh)
        gain, a = trsfms[3](AudioTensor(chunks[1].squeeze(2), 240), AudioTensor(chunks[2], 240))
        y = dec(phase=AudioTensor(inp["phase"]), harm_oscillator_params=(w,), noise_generator_params=(),
                noise_filter_params=(), end_filter_params=(gain, a)).as_tensor()
        loss = (y[:, :9361] - target).square().mean()
        opt.zero_grad()
        loss.backward()
        assert torch.isfinite(h.grad).all()
        opt.step()
        losses.append(loss.item())
    print(losses)
    assert losses[-1] < losses[0]


def test_prefetch_overlap_is_bit_identical():
    """The side-stream transition prefetch must not change a single bit of the output, and a stale handle
    (different coefficients / shape) must be ignored."""
    from golf_amd import functional as GF
    from golf_amd.synthetic import make_inputs

    inp = make_inputs(B=4, T=9600, device="cuda")
    ex, gain, a = inp["noise"], inp["gain"], inp["a"]
    y0 = GF.ltv_allpole_ss(ex, gain, a, 240, fast_inference=False)  # same fp64 transitions as the prepared path
    for overlap in (False, True):
        prep = GF.ltv_allpole_prepare(a, 240, y0.shape[1], overlap=overlap)
        y1 = GF.ltv_allpole_ss(ex, gain, a, 240, prep)
        torch.cuda.synchronize()
        assert torch.equal(y0, y1), overlap
    prep = GF.ltv_allpole_prepare(a, 240, y0.shape[1], overlap=True)
    a2 = make_inputs(B=4, T=9600, device="cuda", seed=7)["a"]  # another STABLE coefficient set
    y2 = GF.ltv_allpole_ss(ex, gain, a2, 240, prep)  # handle is for `a`: must be ignored
    y2_ref = GF.ltv_allpole_ss(ex, gain, a2, 240)
    y_fast = GF.ltv_allpole_ss(ex, gain, a, 240)      # inference path: not bit-identical, but within fp32 accuracy
    assert (y_fast - y0).abs().max() <= 1e-4 * y0.abs().max()
    torch.cuda.synchronize()
    assert torch.equal(y2, y2_ref) and not torch.equal(y2, y0)
    # backward through a prepared forward
    ag = a.clone().requires_grad_(True)
    # (the training step runs on the inference path's fp32 matrices: the handle must keep what the backward reads)
    prep = GF.ltv_allpole_prepare(ag, 240, y0.shape[1], overlap=True, fast=True, training=True)
    GF.ltv_allpole_ss(ex, gain, ag, 240, prep).square().sum().backward()
    ag2 = a.clone().requires_grad_(True)
    GF.ltv_allpole_ss(ex, gain, ag2, 240).square().sum().backward()
    torch.cuda.synchronize()
    assert torch.equal(ag.grad, ag2.grad)


@pytest.mark.parametrize("fast", [True, False])
@pytest.mark.parametrize("mode", [None, "flat-scan"])
def test_prefetch_with_two_level_scan_is_bit_identical(fast, mode):
    """Long utterances (two-level boundary scan): transitions AND group composites prepared on the side stream, the
    forward then runs only the zero-state halves of the pre-pass -- same bits as the all-in-one forward."""
    from golf_amd import functional as GF
    from golf_amd.synthetic import make_inputs

    inp = make_inputs(B=3, T=24000, device="cuda")
    ex, gain, a = inp["noise"], inp["gain"], inp["a"]
    y0 = GF.ltv_allpole_ss(ex, gain, a, 240, fast_inference=fast, mode=mode)
    for overlap in (False, True):
        prep = GF.ltv_allpole_prepare(a, 240, y0.shape[1], overlap=overlap, fast=fast, mode=mode)
        y1 = GF.ltv_allpole_ss(ex, gain, a, 240, prep, fast_inference=fast, mode=mode)
        torch.cuda.synchronize()
        assert torch.equal(y0, y1), overlap
    # a handle prepared for the other scan is not picked up (its workspace may lack the composites)
    other = "flat-scan" if mode is None else None
    prep = GF.ltv_allpole_prepare(a, 240, y0.shape[1], overlap=True, fast=fast, mode=other)
    y2 = GF.ltv_allpole_ss(ex, gain, a, 240, prep, fast_inference=fast, mode=mode)
    torch.cuda.synchronize()
    assert torch.equal(y0, y2)


def test_room_filter_g14(golden):
    """LTIAcousticFilter (golf_lti_fir_f32 + adjoint + taps gradient) against the reference's own run."""
    from golf_amd.audiotensor import AudioTensor
    from golf_amd.filters import LTIAcousticFilter

    g = golden("g14_room_and_full_decoder")
    dev = lambda x: torch.as_tensor(np.asarray(x), dtype=torch.float32).cuda()
    room = LTIAcousticFilter(length=16).cuda()
    assert list(room.state_dict().keys()) == ["kernel"] and room.kernel.shape == (15,)
    ex = dev(g["room_ex"]).requires_grad_(True)
    np.testing.assert_array_equal(room(AudioTensor(ex)).as_tensor().detach().cpu().numpy(),
                                  ex.detach().cpu().numpy())  # zero-initialised kernel = identity
    with torch.no_grad():
        room.kernel.copy_(dev(g["room_kernel"]))
    y = room(AudioTensor(ex)).as_tensor()
    (y * dev(g["room_gy"])).sum().backward()
    for got, ref, what in ((y, "room_y", "y"), (ex.grad, "room_g_ex", "g_ex"), (room.kernel.grad, "room_g_kernel", "g_kernel")):
        emax, el2 = rel_err(got.detach().cpu().numpy(), g[ref])
        print("g14 room", what, emax, el2)
        assert emax < 1e-5 and el2 < 1e-5
    np.testing.assert_allclose(room.impulse_response.detach().cpu().numpy()[0], 1.0)


@pytest.mark.parametrize("B,T,K", [(3, 5000, 127), (2, 253, 5), (1, 960 * 3 + 2, 130), (2, 100, 300)])
def test_room_filter_vs_oracle(B, T, K):
    from golf_amd.audiotensor import AudioTensor
    from golf_amd.filters import LTIAcousticFilter
    from oracle import golf_oracle as O

    rng = np.random.default_rng(K)
    ex, kern, gy = rng.normal(0, 1, (B, T)), rng.normal(0, 0.1, K), rng.normal(0, 1, (B, T))
    dev = lambda x: torch.as_tensor(np.asarray(x), dtype=torch.float32).cuda()
    room = LTIAcousticFilter(length=K + 1).cuda()
    with torch.no_grad():
        room.kernel.copy_(dev(kern))
    x = dev(ex).requires_grad_(True)
    y = room(AudioTensor(x)).as_tensor()
    (y * dev(gy)).sum().backward()
    ref = O.lti_acoustic_filter_forward(ex, kern)
    rgx, rgk = O.lti_acoustic_filter_backward(gy, ex, kern)
    for got, r, what in ((y, ref, "y"), (x.grad, rgx, "g_ex"), (room.kernel.grad, rgk, "g_kernel")):
        emax, el2 = rel_err(got.detach().cpu().numpy(), r)
        print(f"room B{B} T{T} K{K}", what, emax, el2)
        assert emax < 2e-5 and el2 < 2e-5


def test_full_golf_ss_decoder_g14(golden):
    """The complete decoder of cfg/ae/decoder/golf-precise.yaml (oscillator + noise -> zero-phase FIR noise filter,
    LPC end filter, room filter) against the reference's own run at toy size."""
    from golf_amd.audiotensor import AudioTensor
    from golf_amd.filters import LTIAcousticFilter, LTVMinimumPhaseFilterPrecise, LTVZeroPhaseFIRFilter
    from golf_amd.noise import NoiseInterface
    from golf_amd.sf import SourceFilterSynth
    from golf_amd.synth import IndexedGlottalFlowTable

    g = golden("g14_room_and_full_decoder")
    dev = lambda x: torch.as_tensor(np.asarray(x), dtype=torch.float32).cuda()
    noise = dev(g["noise"])

    class FixedNoise(NoiseInterface):
        def forward(self, ref, *args, **kwargs):
            return AudioTensor(noise[:, : ref.shape[1]])

    osc = IndexedGlottalFlowTable(table_size=7, lf_v2=True, points=16, oversampling=1, equal_energy=True)
    room = LTIAcousticFilter(length=8)
    with torch.no_grad():
        room.kernel.copy_(torch.as_tensor(g["room2_kernel"], dtype=torch.float32))
    dec = SourceFilterSynth(harm_oscillator=osc, noise_generator=FixedNoise(),
                            noise_filter=LTVZeroPhaseFIRFilter(window="hanning", n_mag=9),
                            end_filter=LTVMinimumPhaseFilterPrecise(lpc_order=6), room_filter=room,
                            subtract_harmonics=False).cuda()
    hop, w_hop = int(g["hop"]), int(g["w_hop"])
    y = dec(phase=AudioTensor(dev(g["phase"])), harm_oscillator_params=(AudioTensor(dev(g["w"]), w_hop),),
            noise_generator_params=(), noise_filter_params=(AudioTensor(dev(g["log_mag"]), hop),),
            end_filter_params=(AudioTensor(dev(g["gain"]), hop), AudioTensor(dev(g["a"]), hop)))
    y = y.as_tensor().detach().cpu().numpy()
    emax, el2 = rel_err(y, g["y"])
    print("g14 full decoder", emax, el2)
    assert emax < 1e-4 and el2 < 1e-4


def test_full_size_decoder_vs_oracle():
    """golf-precise.yaml decoder at the BASELINE shape (B=32, 2 s) with injected noise vs the float64 oracle on a
    slice of the batch."""
    from golf_amd.audiotensor import AudioTensor
    from golf_amd.synthetic import make_decoder, make_inputs
    from oracle import golf_oracle as O

    inp = make_inputs(B=32, device="cuda", with_noise_filter=True)
    dec = make_decoder(noise_filter=True, room_filter=True, injected_noise=inp["noise"]).cuda()
    with torch.no_grad():
        dec.room_filter.kernel.copy_(inp["room_kernel"])
    y = dec(phase=AudioTensor(inp["phase"]), harm_oscillator_params=(AudioTensor(inp["wsel"], inp["w_hop"]),),
            noise_generator_params=(), noise_filter_params=(AudioTensor(inp["log_mag"], 240),),
            end_filter_params=(AudioTensor(inp["gain"], 240), AudioTensor(inp["a"], 240))).as_tensor()
    assert y.shape == (32, 47760)
    nb = 2
    c = lambda k: inp[k][:nb].double().cpu().numpy()
    osc = dec.harm_oscillator
    win = torch.hann_window(510, dtype=torch.float64).numpy()
    _, ref = O.golf_ss_decoder(c("phase"), 1, c("wsel"), inp["w_hop"], osc.table.double().cpu().numpy(), c("noise"),
                               c("log_mag"), win, c("gain"), c("a"), 240,
                               room_kernel=inp["room_kernel"].double().cpu().numpy(), oversampling=4,
                               equal_energy=True, decim_taps=osc.decimater.kernel.double().cpu().numpy().ravel())
    emax, el2 = rel_err(y[:nb].detach().cpu().numpy(), ref)
    print("full-size decoder", emax, el2)
    assert emax < 1e-4 and el2 < 1e-4


def test_ddsp_decoder_vs_oracle():
    """cfg/ae/decoder/ddsp.yaml (HarmonicPlusNoiseSynth: AdditiveSynthesizer(155) + LTVZeroPhaseFIRFilter noise +
    LTIAcousticFilter) assembled from the drop-in classes, against the float64 oracle composition."""
    from golf_amd.audiotensor import AudioTensor
    from golf_amd.synthetic import make_ddsp_decoder, make_harmonic_amplitudes, make_inputs
    from oracle import golf_oracle as O

    B, H = 3, 155
    inp = make_inputs(B=B, T=24000, device="cuda", with_noise_filter=True)
    amps = make_harmonic_amplitudes(B, 101, H, device="cuda")
    dec = make_ddsp_decoder(H, injected_noise=inp["noise"]).cuda()
    with torch.no_grad():
        dec.end_filter.kernel.copy_(inp["room_kernel"])
    split_sizes, _, keys = dec.split_sizes_and_trsfms
    assert split_sizes == ((1, 155), (), (), (256,), ()) and sum(sum(s) for s in split_sizes) == 412
    y = dec(phase=AudioTensor(inp["phase"]), harm_oscillator_params=(AudioTensor(amps, 240),),
            noise_generator_params=(), harm_filter_params=(), noise_filter_params=(AudioTensor(inp["log_mag"], 240),))
    y = y.as_tensor().detach().cpu().numpy()
    c = lambda t: t.double().cpu().numpy()
    phase = c(inp["phase"])
    harm = O.harmonic_oscillator_forward(phase, 1, c(amps) , 240) * np.sqrt(2 * phase[:, :24000])
    win = torch.hann_window(510, dtype=torch.float64).numpy()
    nz = O.ltv_fir_frames_forward(c(inp["noise"])[:, : harm.shape[1]], O.zero_phase_fir_kernels(c(inp["log_mag"]), win), 240)
    n = min(harm.shape[1], nz.shape[1])
    ref = O.lti_acoustic_filter_forward(harm[:, :n] + nz[:, :n], c(inp["room_kernel"]))
    assert y.shape == ref.shape
    emax, el2 = rel_err(y, ref)
    print("ddsp decoder", emax, el2)
    assert emax < 1e-4 and el2 < 1e-4


def test_replay_pipeline_matches_eager():
    """golf_amd.pipeline.ReplayPipeline (hipGraph replay of the whole decoder, 4 slots on 4 streams) returns bit for bit
    what eager calls return, also after the static inputs of a slot were refilled."""
    from golf_amd.audiotensor import AudioTensor
    from golf_amd.pipeline import ReplayPipeline
    from golf_amd.synthetic import make_decoder, make_inputs

    B = 4
    base = make_inputs(B=B, T=12000, device="cuda", with_noise_filter=True)
    dec = make_decoder(noise_filter=True, room_filter=True, injected_noise=base["noise"]).cuda()

    def fn(inp):
        return dec(phase=AudioTensor(inp["phase"]), harm_oscillator_params=(AudioTensor(inp["wsel"], base["w_hop"]),),
                   noise_generator_params=(), noise_filter_params=(AudioTensor(inp["log_mag"], 240),),
                   end_filter_params=(AudioTensor(inp["gain"], 240), AudioTensor(inp["a"], 240))).as_tensor()

    keys = ("phase", "wsel", "log_mag", "gain", "a")
    pipe = ReplayPipeline(fn, lambda: {k: base[k].clone() for k in keys}, n_slots=4)
    variants = []
    for k in range(6):   # six different batches through four slots: every slot is reused at least once
        v = {"phase": base["phase"] * (1 + 0.01 * k), "wsel": (base["wsel"] + 0.03 * k).clamp(0, 1),
             "log_mag": base["log_mag"] - 0.1 * k, "gain": base["gain"] * (1 + 0.05 * k), "a": base["a"]}
        variants.append(v)
    outs = []
    for v in variants:
        slot = pipe.slots[pipe._next]
        slot.load(**v)
        s = pipe.submit()
        s.stream.synchronize()
        outs.append(s.output.clone())
    for v, y in zip(variants, outs):
        assert torch.equal(y, fn(v))


def test_replay_pipeline_submit_refreshes_inputs_vs_oracle():
    """The serving API: ``ReplayPipeline.submit(batch)`` copies the batch into the slot's static inputs on the
    slot's stream and replays the captured GOLF-ss synthesis step behind it.  Five DIFFERENT batches through two slots -- as
    dicts (one copy per tensor) and packed (one copy) -- and every output is checked against the float64 oracle of ITS batch
    (oscillator + injected noise -> sample-wise LPC filter), so a stale or half-refreshed slot cannot pass."""
    from golf_amd import functional as GF
    from golf_amd.pipeline import ReplayPipeline
    from golf_amd.synth import DownsampledIndexedGlottalFlowTable
    from golf_amd.synthetic import make_inputs
    from oracle import golf_oracle as O

    B, T = 3, 7200
    osc = DownsampledIndexedGlottalFlowTable(hop_rate=10, in_channels=64, oversampling=4, equal_energy=True, lf_v2=True,
                                             points=2048).cuda()
    table, taps = osc.table, osc.decimater.taps
    keys = ("phase", "wsel", "noise", "gain", "a")
    batches = [make_inputs(B=B, T=T, device="cuda", seed=100 + k) for k in range(5)]
    w_hop = batches[0]["w_hop"]

    def fn(inp):
        src = GF.glottal_osc(inp["phase"], inp["wsel"], table, taps, 1, w_hop, 4, True, add=inp["noise"])
        return GF.ltv_allpole_ss(src, inp["gain"], inp["a"], 240, fast_inference=True)

    def oracle(bt):
        c = {k: bt[k].cpu().numpy() for k in keys}
        src = O.indexed_glottal_forward(c["phase"], 1, c["wsel"], w_hop, table.cpu().numpy(), 4, True,
                                        decim_taps=taps.cpu().numpy())["out"]
        n = min(src.shape[1], c["noise"].shape[1])
        return O.ltv_allpole_ss_forward(src[:, :n] + c["noise"][:, :n], c["gain"], c["a"], 240)

    refs = [oracle(bt) for bt in batches]
    for packed in (False, True):
        # (the packed round also takes its slots idle-first: whichever slot has finished gets the next batch)
        pipe = ReplayPipeline(fn, lambda: {k: batches[0][k].clone() for k in keys}, n_slots=2, packed=packed,
                              issue="idle-first" if packed else "round-robin")
        got = []
        for bt in batches:
            s = pipe.submit(pipe.pack(bt) if packed else {k: bt[k] for k in keys})
            s.stream.synchronize()
            got.append(s.output.cpu().numpy())
        for k, (y, r) in enumerate(zip(got, refs)):
            n = min(y.shape[1], r.shape[1])
            err = np.abs(y[:, :n] - r[:, :n]).max() / np.abs(r).max()
            print(f"submit(batch {k}, packed={packed}): rel-max err vs oracle {err:.2e}")
            assert err < 1e-4, (k, packed, err)
        # different batches really gave different outputs (a slot replaying stale inputs would repeat itself)
        assert np.abs(got[0] - got[2]).max() > 1e-3 and np.abs(got[1] - got[3]).max() > 1e-3


@pytest.mark.gpu
@pytest.mark.parametrize("packed", [False, True])
def test_replay_pipeline_submit_waits_for_the_producer_of_the_batch(packed):
    """``submit(batch)`` copies on the slot's stream; the batch is produced on the caller's stream.  A long
    kernel in front of the producer's writes makes the race deterministic: without the stream edge the copy reads the
    poison the buffers held before.  The batch is dropped right after ``submit`` (``record_stream`` keeps the allocator from
    recycling it under the copy), and a second submit takes an explicit event as the producer."""
    from golf_amd import functional as GF
    from golf_amd.pipeline import ReplayPipeline
    from golf_amd.synthetic import make_inputs

    B, T = 4, 24000
    keys = ("noise", "gain", "a")
    base = make_inputs(B=B, T=T, device="cuda", seed=500)
    fresh = [make_inputs(B=B, T=T, device="cuda", seed=501 + k) for k in range(2)]

    def fn(inp):
        return GF.ltv_allpole_ss(inp["noise"], inp["gain"], inp["a"], 240, fast_inference=True)

    want = [fn({k: bt[k] for k in keys}).clone() for bt in fresh]
    pipe = ReplayPipeline(fn, lambda: {k: base[k].clone() for k in keys}, n_slots=2, packed=packed)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    for rep, bt in enumerate(fresh):
        prod = torch.cuda.current_stream() if rep == 0 else side
        with torch.cuda.stream(prod):
            staged = {k: torch.full_like(bt[k], float("nan")) for k in keys}     # poison first ...
            flat = pipe.new_flat().fill_(float("nan")) if packed else None
            torch.cuda._sleep(200_000_000)                                        # ... ~0.1 s of nothing ...
            for k in keys:
                staged[k].copy_(bt[k])                                            # ... then the real batch
            if packed:
                pipe.pack(staged, out=flat)
            ev = torch.cuda.Event()
            ev.record(prod)
        batch = flat if packed else staged
        slot = pipe.submit(batch, producer=None if rep == 0 else ev)
        del batch, staged, flat
        junk = [torch.full((1 << 20,), float("nan"), device="cuda") for _ in range(8)]   # would land in the freed blocks
        slot.stream.synchronize()
        assert torch.equal(slot.output, want[rep]), f"submit {rep}: the copy ran ahead of the producer"
        del junk


@pytest.mark.gpu
def test_replay_pipeline_flat_submit_refuses_uncovered_inputs():
    """a flat batch refreshes the fp32 inputs only; a slot with other tensor inputs says so instead of
    replaying them stale."""
    from golf_amd import functional as GF
    from golf_amd.pipeline import ReplayPipeline
    from golf_amd.synthetic import make_inputs

    base = make_inputs(B=2, T=4800, device="cuda", seed=7)

    def fn(inp):
        return GF.ltv_allpole_ss(inp["noise"] * inp["scale"].float(), inp["gain"], inp["a"], 240, fast_inference=True)

    def mk():
        d = {k: base[k].clone() for k in ("noise", "gain", "a")}
        d["scale"] = torch.ones(1, dtype=torch.float16, device="cuda")
        return d

    pipe = ReplayPipeline(fn, mk, n_slots=1, packed=True)
    assert pipe.slots[0].unpacked == ["scale"]
    with pytest.raises(ValueError, match="non-fp32"):
        pipe.submit(pipe.new_flat())
    pipe.submit(pipe.pack(base), partial_ok=True).stream.synchronize()
    with pytest.raises(ValueError, match="fp32"):
        ReplayPipeline(lambda i: i["x"].float(), lambda: {"x": torch.ones(4, dtype=torch.int32, device="cuda")}, n_slots=1, packed=True)


@pytest.mark.gpu
def test_replay_pipeline_lone_batch_chain_in_flight_is_bit_identical():
    """``ReplayPipeline(throughput=False)`` keeps the filter's lone-batch launch chain with several slots: the two chunk passes are
    then ONE launch whose waves wait for each other's flag words (lpc_fwdq2m_kernel), and three of those launches are in flight at
    once -- the placement its look-back rule has to survive.  Six different batches through three slots of either pipeline: the
    same bits from both chains, every time, and the oracle's values."""
    from golf_amd import functional as GF
    from golf_amd.pipeline import ReplayPipeline
    from golf_amd.synthetic import make_inputs
    from oracle import golf_oracle as O

    B, T = 4, 24000   # 99 chunk maps = 7 groups: the two-level scan and with it the merged kernel
    keys = ("noise", "gain", "a")
    batches = [make_inputs(B=B, T=T, device="cuda", seed=300 + k) for k in range(6)]

    def fn(inp):
        return GF.ltv_allpole_ss(inp["noise"], inp["gain"], inp["a"], 240, fast_inference=True)

    outs = {}
    for throughput in (None, False):
        pipe = ReplayPipeline(fn, lambda: {k: batches[0][k].clone() for k in keys}, n_slots=3, throughput=throughput)
        assert pipe.throughput == (throughput is None)
        got = []
        for rep in range(3):
            slots = [pipe.submit({k: bt[k] for k in keys}) for bt in batches[:3]] if rep % 2 == 0 else \
                    [pipe.submit({k: bt[k] for k in keys}) for bt in batches[3:]]
            pipe.synchronize()
            got.append([s.output.clone() for s in slots])
        outs[throughput] = got
    for ga, gb in zip(outs[None], outs[False]):
        for u, v in zip(ga, gb):
            assert torch.equal(u, v)
    for u, v in zip(outs[False][0], outs[False][2]):      # the same three batches, two rounds apart
        assert torch.equal(u, v)
    c = {k: batches[1][k].cpu().numpy() for k in keys}
    ref = O.ltv_allpole_ss_forward(c["noise"], c["gain"], c["a"], 240)
    y = outs[False][0][1].cpu().numpy()
    n = min(y.shape[1], ref.shape[1])
    err = np.abs(y[:, :n] - ref[:, :n]).max() / np.abs(ref).max()
    assert err < 1e-4, err


def test_long_utterance_decoder_vs_oracle():
    """One 12.5 s utterance through the whole golf-precise decoder (293 phase-scan tiles, 1250 LPC chunks) against the
    float64 oracle: nothing in the path is sized for 2 s clips."""
    from golf_amd.audiotensor import AudioTensor
    from golf_amd.synthetic import make_decoder, make_inputs
    from oracle import golf_oracle as O

    inp = make_inputs(B=1, T=300_000, device="cuda", with_noise_filter=True)
    dec = make_decoder(noise_filter=True, room_filter=True, injected_noise=inp["noise"]).cuda()
    with torch.no_grad():
        dec.room_filter.kernel.copy_(inp["room_kernel"])
    y = dec(phase=AudioTensor(inp["phase"]), harm_oscillator_params=(AudioTensor(inp["wsel"], inp["w_hop"]),),
            noise_generator_params=(), noise_filter_params=(AudioTensor(inp["log_mag"], 240),),
            end_filter_params=(AudioTensor(inp["gain"], 240), AudioTensor(inp["a"], 240))).as_tensor()
    c = lambda k: inp[k].double().cpu().numpy()
    osc = dec.harm_oscillator
    win = torch.hann_window(510, dtype=torch.float64).numpy()
    _, ref = O.golf_ss_decoder(c("phase"), 1, c("wsel"), inp["w_hop"], osc.table.double().cpu().numpy(), c("noise"),
                               c("log_mag"), win, c("gain"), c("a"), 240,
                               room_kernel=inp["room_kernel"].double().cpu().numpy(), oversampling=4,
                               equal_energy=True, decim_taps=osc.decimater.kernel.double().cpu().numpy().ravel())
    assert y.shape == ref.shape
    emax, el2 = rel_err(y.detach().cpu().numpy(), ref)
    print("12.5 s decoder", y.shape, emax, el2)
    assert emax < 1e-4 and el2 < 1e-4


def test_empty_batch_and_zero_samples():
    """B = 0 / zero excitation samples: the reference's tensor ops return empty results that stay in the autograd graph;
    the drop-in wrappers do the same (the C ABI itself rejects non-positive sizes with GOLF_EINVAL)."""
    from golf_amd import functional as GF
    from golf_amd import _lib
    from golf_amd.synthetic import make_inputs

    inp = make_inputs(B=2, T=2400, device="cuda")
    a0 = inp["a"][:0].clone().requires_grad_(True)
    y = GF.ltv_allpole_ss(inp["noise"][:0], inp["gain"][:0], a0, 240)
    T = GF.ss_output_length(inp["noise"].shape[1], inp["a"].shape[1], 240)
    assert y.shape == (0, T) and y.requires_grad
    y.sum().backward()
    assert a0.grad.shape == a0.shape
    y = GF.ltv_allpole_ss(inp["noise"][:, :0], inp["gain"], inp["a"], 240)
    assert y.shape == (2, 0)
    osc_args = (inp["wsel"][:0], torch.rand(8, 2048, device="cuda"), torch.ones(129, device="cuda") / 129, 1, inp["w_hop"], 4, True)
    out = GF.glottal_osc(inp["phase"][:0], *osc_args)
    assert out.shape == (0, GF.osc_lengths(inp["phase"].shape[1], 1, 4)[1])
    lib = _lib.load()
    rc = lib.golf_ltv_allpole_fwd_f32(0, 0, 0, 0, 0, 0, 0, 2400, 11, 22, 240, 0, 0, 0, 0, 0)
    assert rc != 0   # GOLF_EINVAL, no launch


def test_decoder_under_bf16_autocast_matches_fp32():
    """Mixed-precision entry (reference intent: models/synth.py:250-251 pins the phase cumsum to fp32 so that
    the decoder survives `precision: 16-mixed`).  The kernels are fp32: under torch.autocast every Function casts its
    floating-point inputs to fp32 and runs with autocast off (torch.amp.custom_fwd / custom_bwd).  A training step of the
    GOLF-ss decoder inside torch.autocast("cuda", torch.bfloat16), fed bf16 control tracks as a bf16 encoder head would
    produce them: runs, returns fp32 audio equal to the fp32 call on the same (bf16-rounded) values, and finite gradients
    in the inputs' dtype."""
    from golf_amd.audiotensor import AudioTensor
    from golf_amd.ctrl import PassThrough
    from golf_amd.filters import LTVMinimumPhaseFilterPrecise
    from golf_amd.noise import NoiseInterface
    from golf_amd.sf import SourceFilterSynth
    from golf_amd.synth import DownsampledIndexedGlottalFlowTable
    from golf_amd.synthetic import make_inputs

    torch.manual_seed(0)
    inp = make_inputs(B=4, T=9600, device="cuda")
    noise = inp["noise"]

    class FixedNoise(NoiseInterface):
        def forward(self, ref, *args, **kwargs):
            return AudioTensor(noise[:, : ref.shape[1]])

    dec = SourceFilterSynth(
        harm_oscillator=DownsampledIndexedGlottalFlowTable(hop_rate=10, in_channels=8, oversampling=4,
                                                           equal_energy=True, lf_v2=True, points=2048),
        noise_generator=FixedNoise(), noise_filter=PassThrough(),
        end_filter=LTVMinimumPhaseFilterPrecise(lpc_order=22), room_filter=None, subtract_harmonics=False).cuda()
    split_sizes, trsfms, keys = dec.split_sizes_and_trsfms
    F = 40
    h16 = (torch.randn(4, F, 8 + 1 + 22, device="cuda") * 0.1).to(torch.bfloat16)

    def run(h):
        chunks = torch.split(h, [8, 1, 22], dim=2)
        (w,) = trsfms[0](AudioTensor(chunks[0], 240))
        gain, a = trsfms[3](AudioTensor(chunks[1].squeeze(2), 240), AudioTensor(chunks[2], 240))
        return dec(phase=AudioTensor(inp["phase"]), harm_oscillator_params=(w,), noise_generator_params=(),
                   noise_filter_params=(), end_filter_params=(gain, a)).as_tensor()

    h_amp = h16.clone().requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y_amp = run(h_amp)
        loss = y_amp.float().square().mean()
    loss.backward()
    assert y_amp.dtype == torch.float32 and torch.isfinite(y_amp).all()
    assert h_amp.grad is not None and h_amp.grad.dtype == torch.bfloat16 and torch.isfinite(h_amp.grad.float()).all()
    assert h_amp.grad.float().abs().max() > 0
    # the same values through the plain fp32 path.  The control transforms' small linear layer runs in bf16 under autocast
    # (that is what autocast means for the encoder side), so the audio is compared loosely; the point is that the fp32-only
    # kernels are reached with fp32 tensors and nothing raises
    y_ref = run(h16.float())
    emax, el2 = rel_err(y_amp.detach().cpu().numpy(), y_ref.detach().cpu().numpy())
    print("bf16 autocast vs fp32", emax, el2)
    assert el2 < 5e-2


@pytest.mark.gpu
@pytest.mark.parametrize("noise_filter,room_filter", [(True, True), (False, False)])
def test_decoder_fuses_source_and_transition_maps_bit_identically(noise_filter, room_filter, monkeypatch):
    """SourceFilterSynth in inference: oscillator + the end filter's transition maps as one launch (golf_amd.sf.FUSE_SOURCE_MAPS,
    functional.source_filter_ss) gives the composition's bits -- the golf-precise decoder with its noise filter (common length
    47 760: the ``length`` path) and the bare source -> filter pair; under autograd and with batches in flight the modules
    compose as ever (same values again).  Reference: models/sf.py:47-64."""
    import golf_amd.sf as SF
    from golf_amd import functional as GF
    from golf_amd.audiotensor import AudioTensor
    from golf_amd.synthetic import make_decoder, make_inputs

    B = 4
    inp = make_inputs(B=B, device="cuda", seed=11, with_noise_filter=True)
    dec = make_decoder(noise_filter=noise_filter, room_filter=room_filter, injected_noise=inp["noise"]).cuda().eval()
    calls = []
    real = GF.source_filter_ss
    monkeypatch.setattr(GF, "source_filter_ss", lambda *a, **k: (calls.append(1), real(*a, **k))[1])

    def run():
        params = dict(phase=AudioTensor(inp["phase"]),
                      harm_oscillator_params=(AudioTensor(inp["wsel"], inp["w_hop"]),),
                      noise_generator_params=(),
                      noise_filter_params=(AudioTensor(inp["log_mag"], 240),) if noise_filter else (),
                      end_filter_params=(AudioTensor(inp["gain"], 240), AudioTensor(inp["a"], 240)))
        with torch.no_grad():
            return dec(**params).as_tensor().clone()

    monkeypatch.setattr(SF, "FUSE_SOURCE_MAPS", True)
    y_fused = run()
    assert len(calls) == 1, "the fused launch was expected to serve this decoder"
    monkeypatch.setattr(SF, "FUSE_SOURCE_MAPS", False)
    y_comp = run()
    assert len(calls) == 1
    torch.cuda.synchronize()
    assert torch.isfinite(y_fused).all() and torch.equal(y_fused, y_comp)
    monkeypatch.setattr(SF, "FUSE_SOURCE_MAPS", True)
    monkeypatch.setattr(GF, "THROUGHPUT_MODE", True)          # batches in flight: the composition (and its launch chain)
    y_thr = run()
    assert len(calls) == 1 and torch.equal(y_thr, y_comp)
