"""GPU tests of the noise sources (reference models/noise.py): the RNG generators cannot match the
reference's stream sample for sample, so their *moments* and support are checked; NoiseBand is checked exactly with the
band offsets injected (golden g24 from the reference's own run + the float64 oracle at the configured size)."""
import math

import numpy as np
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu


def dev(x, grad=False):
    return torch.as_tensor(np.asarray(x), dtype=torch.float32).cuda().requires_grad_(grad)


def test_generator_moments():
    from golf_amd.audiotensor import AudioTensor
    from golf_amd.noise import SignFlipNoise, StandardNormalNoise, UniformNoise

    torch.manual_seed(1234)
    ref = AudioTensor(torch.empty(32, 48000, device="cuda"))
    n = 32 * 48000
    se = 1 / math.sqrt(n)                                    # standard error of the mean of a unit-variance sample

    x = StandardNormalNoise()(ref).as_tensor().double()
    assert x.shape == (32, 48000) and x.is_cuda
    assert abs(float(x.mean())) < 5 * se
    assert abs(float(x.var()) - 1) < 5 * math.sqrt(2) * se   # var of the sample variance of a normal: 2/n
    assert abs(float((x ** 3).mean())) < 5 * math.sqrt(15) * se
    assert abs(float((x ** 4).mean()) - 3) < 5 * math.sqrt(96) * se
    assert float(x.abs().max()) < 7
    rows = x[:, :-1] * x[:, 1:]                              # white: no lag-1 correlation
    assert abs(float(rows.mean())) < 5 * se

    u = UniformNoise()(ref).as_tensor().double()
    assert float(u.min()) >= -math.sqrt(3) and float(u.max()) < math.sqrt(3)           # support [-sqrt3, sqrt3)
    assert float(u.max()) > math.sqrt(3) * 0.999 and float(u.min()) < -math.sqrt(3) * 0.999
    assert abs(float(u.mean())) < 5 * se
    assert abs(float(u.var()) - 1) < 5 * math.sqrt(0.8) * se                           # kurtosis 1.8: var(s^2) = 0.8/n
    assert abs(float((u ** 4).mean()) - 1.8) < 0.01

    s = SignFlipNoise()(ref).as_tensor()
    assert torch.all(s.abs() == 1)
    assert torch.all(s[:, 1:] == -s[:, :-1])                 # strict alternation along time
    first = s[:, 0]
    assert 0 < int((first > 0).sum()) < 32                   # one random sign per row, both occur over 32 rows
    many = SignFlipNoise()(AudioTensor(torch.empty(4096, 4, device="cuda"))).as_tensor()[:, 0]
    assert abs(float(many.mean())) < 5 / math.sqrt(4096)


def test_noise_band_golden_g24(golden):
    from golf_amd.audiotensor import AudioTensor
    from golf_amd.noise import NoiseBand

    g = golden("g24_noiseband_ckpt_biquads")
    torch.manual_seed(7)
    nb = NoiseBand(n_filters=12, fs=24000, attenuation=50).cuda()
    lg = dev(g["nb_log_gain"], True)
    y = nb(AudioTensor(torch.zeros(2, 330, device="cuda")), AudioTensor(lg, 64),
           rand_offset=torch.from_numpy(g["nb_offsets"])).as_tensor()
    assert y.shape == g["nb_out"].shape
    assert max(rel_err(y.detach().cpu().numpy(), g["nb_out"])) < 5e-5
    (y * dev(g["nb_gy"])).sum().backward()
    assert max(rel_err(lg.grad.cpu().numpy(), g["nb_g_log_gain"])) < 3e-4


@pytest.mark.parametrize("B,T,F,hop,K,L", [(3, 4801, 21, 240, 64, 2048), (2, 1000, 9, 128, 37, 512),
                                           (4, 48000, 201, 240, 256, 8192),
                                           # short gain hops, down to sample-rate gains: the block's gain rows exceed the LDS
                                           # budget and the bands are walked in several passes (round 2 returned EUNSUPPORTED)
                                           (2, 600, 600, 1, 128, 512), (2, 2000, 501, 4, 1024, 1024), (1, 900, 301, 3, 75, 256)])
def test_noise_band_vs_oracle(B, T, F, hop, K, L):
    from golf_amd import functional as GF
    from oracle import golf_oracle as O

    rng = np.random.default_rng(K)
    bands = rng.normal(0, 0.3, (K, L)).astype(np.float32)
    offs = rng.integers(0, L, (B, K))
    lg = rng.normal(-2, 0.5, (B, F, K)).astype(np.float32)
    lgt = dev(lg, True)
    y = GF.noise_band(dev(bands), torch.from_numpy(offs).cuda(), lgt, hop, T)
    ref = O.noise_band_forward(bands, offs, lg, hop, T)
    assert y.shape == ref.shape
    emax, el2 = rel_err(y.detach().cpu().numpy(), ref)
    assert emax < 1e-4 and el2 < 1e-4, (emax, el2)
    gy = rng.normal(0, 1, ref.shape).astype(np.float32)
    (y * dev(gy)).sum().backward()
    emax, el2 = rel_err(lgt.grad.cpu().numpy(), O.noise_band_backward(gy, bands, offs, lg, hop))
    assert emax < 2e-4 and el2 < 2e-4, (emax, el2)


def test_noise_band_random_offsets_and_statistics():
    """Without injected offsets every (utterance, band) starts somewhere else: two calls differ, rows differ, and the
    output power follows the gains (bands are near-orthogonal, so power adds)."""
    from golf_amd.audiotensor import AudioTensor
    from golf_amd.noise import NoiseBand

    torch.manual_seed(3)
    nb = NoiseBand(n_filters=16, fs=24000, attenuation=50).cuda()
    lg = torch.full((4, 41, 16), -1.0, device="cuda")
    ref = AudioTensor(torch.zeros(4, 9601, device="cuda"))
    a = nb(ref, AudioTensor(lg, 240)).as_tensor()
    b = nb(ref, AudioTensor(lg, 240)).as_tensor()
    assert a.shape == (4, 9601) and torch.isfinite(a).all()
    assert not torch.equal(a, b) and not torch.equal(a[0], a[1])
    louder = nb(ref, AudioTensor(lg + math.log(2.0), 240)).as_tensor()
    ratio = float(louder.square().mean() / a.square().mean())
    assert 3.0 < ratio < 5.5                                  # 4x the power for 2x the gains
