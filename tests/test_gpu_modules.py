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
    prep = GF.ltv_allpole_prepare(ag, 240, y0.shape[1], overlap=True)
    GF.ltv_allpole_ss(ex, gain, ag, 240, prep).square().sum().backward()
    ag2 = a.clone().requires_grad_(True)
    GF.ltv_allpole_ss(ex, gain, ag2, 240).square().sum().backward()
    torch.cuda.synchronize()
    assert torch.equal(ag.grad, ag2.grad)
