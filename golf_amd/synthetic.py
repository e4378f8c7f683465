"""Seeded synthetic GOLF decoder inputs — the reference ships no checkpoints
(.MISSING_LARGE_BLOBS), so every benchmark/parity run uses these.  Generated on CPU with a fixed
generator (seed 2434 = the reference's seed_everything, cfg/ae/vctk.yaml:2), then moved to the device.
"""
from __future__ import annotations

import math

import torch

from .utils import rc2lpc

SR = 24000


def make_inputs(B: int = 32, T: int = 48000, hop: int = 240, M: int = 22, w_hop_rate: int = 10, seed: int = 2434,
                device="cpu", dtype=torch.float32, with_noise_filter: bool = False, n_mag: int = 256,
                room_length: int = 128):
    """phase (B,T) hop 1 · wsel (B,Fw) hop hop*w_hop_rate · noise (B,T) · gain (B,F) · a (B,F,M) · logits.

    * f0: per-utterance base U(80,400) Hz x 3 % vibrato at 5.5 Hz  -> phase increment in [0.003, 0.02]
    * table_select_weight: sigmoid of a random walk (sigma 0.3 / control frame)
    * LPC logits: N(0,0.5^2) per utterance + random walk N(0,0.02^2) per frame -> tanh -> rc2lpc
      (smooth on purpose: interpolating unrelated stable frames is unstable)
    * log-gain: random walk around -3 (sigma 0.05)
    * with_noise_filter: also log_mag (B,F,n_mag) for the zero-phase FIR noise filter (smooth spectral envelope
      around -3 with a per-frame random walk) and room_kernel (room_length-1), an exponentially decaying tail.
      Drawn after everything else, so the other tensors do not depend on the flag.
    """
    g = torch.Generator().manual_seed(seed)
    F = T // hop
    t = torch.arange(T, dtype=torch.float64) / SR
    f0 = torch.empty(B, 1, dtype=torch.float64).uniform_(80, 400, generator=g)
    vib_phase = torch.empty(B, 1, dtype=torch.float64).uniform_(0, 2 * math.pi, generator=g)
    phase = (f0 * (1 + 0.03 * torch.sin(2 * math.pi * 5.5 * t + vib_phase)) / SR).to(dtype)
    w_hop = hop * w_hop_rate
    Fw = T // w_hop + 1
    wsel = torch.sigmoid(torch.cumsum(0.3 * torch.randn(B, Fw, generator=g), 1)).to(dtype)
    logits = 0.5 * torch.randn(B, 1, M, generator=g) + torch.cumsum(0.02 * torch.randn(B, F, M, generator=g), 1)
    a = rc2lpc(torch.tanh(logits.double())).to(dtype)
    log_gain = -3 + torch.cumsum(0.05 * torch.randn(B, F, generator=g), 1)
    gain = torch.exp(log_gain).to(dtype)
    noise = torch.randn(B, T, generator=g).to(dtype)
    out = dict(phase=phase, wsel=wsel, w_hop=w_hop, noise=noise, gain=gain, a=a, logits=logits.to(dtype),
               log_gain=log_gain.to(dtype), hop=hop)
    if with_noise_filter:
        env = torch.cumsum(0.15 * torch.randn(B, 1, n_mag, generator=g), 2)
        walk = torch.cumsum(0.03 * torch.randn(B, F, n_mag, generator=g), 1)
        out["log_mag"] = (-3.0 + env + walk).clamp(-8, 1).to(dtype)
        K = room_length - 1
        decay = torch.exp(-torch.arange(K, 0, -1, dtype=torch.float64) / 20.0)  # kernel[k] acts at delay K-k
        out["room_kernel"] = (0.3 * torch.randn(K, generator=g).double() * decay).to(dtype)
    return {k: (v.to(device) if isinstance(v, torch.Tensor) else v) for k, v in out.items()}


def make_decoder(noise_filter: bool = True, room_filter: bool = True, injected_noise: torch.Tensor = None,
                 in_channels: int = 64, framewise: bool = False):
    """The GOLF decoder as cfg/ae/decoder/golf-precise.yaml (or golf.yaml with ``framewise``) instantiates it,
    from this package's drop-in classes.  ``injected_noise`` replaces the RNG by a fixed tensor (parity runs)."""
    from .audiotensor import AudioTensor
    from .ctrl import PassThrough
    from .filters import (LTIAcousticFilter, LTVMinimumPhaseFilter, LTVMinimumPhaseFilterPrecise,
                          LTVZeroPhaseFIRFilter)
    from .noise import NoiseInterface, StandardNormalNoise
    from .sf import SourceFilterSynth
    from .synth import DownsampledIndexedGlottalFlowTable

    if injected_noise is not None:
        class _Fixed(NoiseInterface):
            uses_reference_values = False

            def forward(self, ref, *args, **kwargs):
                return AudioTensor(injected_noise[:, : ref.shape[1]])

        gen = _Fixed()
    else:
        gen = StandardNormalNoise()
    end = (LTVMinimumPhaseFilter(window="hanning", window_length=960, lpc_order=22, lpc_parameterisation="rc2lpc")
           if framewise else LTVMinimumPhaseFilterPrecise(lpc_order=22, lpc_parameterisation="rc2lpc"))
    return SourceFilterSynth(
        harm_oscillator=DownsampledIndexedGlottalFlowTable(
            hop_rate=10, in_channels=in_channels, oversampling=4, equal_energy=True, table_type="derivative",
            normalize_method="constant_power", align_peak=True, trainable=False, min_R_d=0.3, max_R_d=2.7,
            lf_v2=True, points=2048),
        noise_generator=gen,
        noise_filter=LTVZeroPhaseFIRFilter(window="hanning", n_mag=256) if noise_filter else PassThrough(),
        end_filter=end,
        room_filter=LTIAcousticFilter(length=128, conv_method="direct") if room_filter else None,
        subtract_harmonics=False)


def make_ddsp_decoder(num_harmonics: int = 155, injected_noise: torch.Tensor = None):
    """The DDSP baseline decoder as cfg/ae/decoder/ddsp.yaml instantiates it: additive synthesiser + filtered noise
    + room filter, from this package's drop-in classes."""
    from .audiotensor import AudioTensor
    from .ctrl import PassThrough
    from .filters import LTIAcousticFilter, LTVZeroPhaseFIRFilter
    from .noise import NoiseInterface, StandardNormalNoise
    from .sf import HarmonicPlusNoiseSynth
    from .synth import AdditiveSynthesizer

    if injected_noise is not None:
        class _Fixed(NoiseInterface):
            uses_reference_values = False

            def forward(self, ref, *args, **kwargs):
                return AudioTensor(injected_noise[:, : ref.shape[1]])

        gen = _Fixed()
    else:
        gen = StandardNormalNoise()
    return HarmonicPlusNoiseSynth(
        harm_oscillator=AdditiveSynthesizer(num_harmonics=num_harmonics), noise_generator=gen,
        harm_filter=PassThrough(), noise_filter=LTVZeroPhaseFIRFilter(window="hanning", n_mag=256),
        end_filter=LTIAcousticFilter(length=128, conv_method="fft"))


def make_harmonic_amplitudes(B: int, F: int, num_harmonics: int = 155, seed: int = 2434, device="cpu"):
    """Smooth synthetic harmonic amplitudes (B,F,H) as AdditiveSynthesizer.ctrl produces them:
    exp(log_gain) * sigmoid(logits), with a 1/h spectral tilt and a slow random walk per frame."""
    g = torch.Generator().manual_seed(seed + 1)
    tilt = -torch.log(torch.arange(1, num_harmonics + 1, dtype=torch.float32))
    logits = tilt + torch.cumsum(0.05 * torch.randn(B, F, num_harmonics, generator=g), 1) \
        + 0.5 * torch.randn(B, 1, num_harmonics, generator=g)
    log_gain = -3 + torch.cumsum(0.05 * torch.randn(B, F, 1, generator=g), 1)
    return (torch.exp(log_gain) * torch.sigmoid(logits)).to(device)
