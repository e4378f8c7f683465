"""Noise sources (reference models/noise.py:19-213).  RNG is torch's device generator — RNG streams
cannot match across devices, so parity tests inject the noise tensor / the band offsets and check
the generators' moments."""
from __future__ import annotations

import math

import torch

from .audiotensor import AudioTensor
from .ctrl import Controllable

__all__ = ["NoiseInterface", "StandardNormalNoise", "UniformNoise", "SignFlipNoise", "NoiseBand"]


class NoiseInterface(Controllable):
    # False: forward() takes only shape / dtype / device from ``ref``.  SourceFilterSynth then generates the noise before
    # the oscillator has run and fuses the sum into the oscillator's last kernel; generators that read the values of
    # ``ref`` keep the default and the reference's order of operations.
    uses_reference_values = True

    def forward(self, ref: AudioTensor, *args, **kwargs) -> AudioTensor:
        raise NotImplementedError


class StandardNormalNoise(NoiseInterface):
    uses_reference_values = False

    def forward(self, ref: AudioTensor, *args, **kwargs) -> AudioTensor:
        return torch.randn_like(ref)


class UniformNoise(NoiseInterface):
    """Zero-mean unit-variance uniform noise on [-sqrt(3), sqrt(3))."""

    uses_reference_values = False

    def forward(self, ref: AudioTensor, *args, **kwargs) -> AudioTensor:
        return (torch.rand_like(ref) - 0.5) * (2 * math.sqrt(3))


class SignFlipNoise(NoiseInterface):
    """+s, -s, +s, ... with one random sign s per row."""

    uses_reference_values = False

    def forward(self, ref: AudioTensor, *args, **kwargs) -> AudioTensor:
        data = ref.as_tensor()
        sign = torch.where(torch.rand(data.shape[:-1] + (1,), device=data.device) < 0.5, -1.0, 1.0).to(data.dtype)
        alt = 1.0 - 2.0 * (torch.arange(data.shape[-1], device=data.device) % 2).to(data.dtype)
        return ref.new_tensor(sign * alt)


class NoiseBand(NoiseInterface):
    """Sum of ``n_filters`` pre-filtered, loopable noise bands, each scaled by a frame-rate gain (reference
    models/noise.py:58-213, the filtered-noise source of NoiseBandNet-style decoders).

    Init (host, once): a bank of Kaiser-window FIR filters -- one low-pass up to the first band edge, a band-pass per
    interior band, one high-pass from the last edge -- designed with scipy exactly as the reference designs them
    (``kaiserord`` for length/beta from ``attenuation`` and 20 % of the bandwidth, ``firwin``); each filter's magnitude
    response is given a random phase (torch's global generator, like the reference: seed it to reproduce a bank) and
    transformed back, which yields one period of stationary noise with that band's spectrum.  Buffers ``band_centers``
    and ``noise_bands`` are persistent, so checkpoints carry the bank.
    Forward (HIP, golf_noise_band_fwd_f32): random start offset per (utterance, band), then
    ``out[b,t] = sum_k noise_bands[k, (t + off[b,k]) % L] * up(exp(log_gain))[b,t,k]``; differentiable w.r.t. log_gain.
    ``rand_offset`` (B, K) injects the offsets (parity runs)."""

    uses_reference_values = False

    def __init__(self, n_filters: int = 1024, fs: int = 44100, attenuation: float = 50,
                 normalize_noise_bands: bool = True):
        super().__init__()
        from scipy import signal

        edges = torch.linspace(0, fs / 2, n_filters + 1)
        bands = torch.stack((edges[1:-2], edges[2:-1]), dim=1)          # interior bands (n_filters - 2, 2)
        centers = torch.cat([bands[0, :1] / 2, bands.mean(dim=1), ((fs / 2) + bands[-1, 1:]) / 2])
        self.register_buffer("band_centers", centers.float())

        def design(cutoff, pass_zero):
            if cutoff.numel() > 1:
                bandwidth = abs(cutoff[1] - cutoff[0])
            else:
                bandwidth = cutoff if pass_zero else abs((fs / 2) - cutoff)
            width = (bandwidth / (fs / 2)) * 0.2                        # transition width: 20 % of the bandwidth
            numtaps, beta = signal.kaiserord(ripple=attenuation, width=width)
            numtaps = 2 * (numtaps // 2) + 1
            return torch.from_numpy(signal.firwin(numtaps=numtaps, cutoff=cutoff, window=("kaiser", beta), scale=True,
                                                  fs=fs, pass_zero=pass_zero))

        filters = [design(bands[0, 0], True)] + [design(band, False) for band in bands] + [design(bands[-1, 1], False)]
        longest = max(len(h) for h in filters)
        period = 2 ** math.ceil(math.log2(longest))
        padded = torch.stack([torch.cat([torch.zeros(period - len(h)), h]) for h in filters])
        magnitude = torch.fft.rfft(padded).abs()
        rotation = torch.exp(1j * (torch.rand_like(magnitude) * 2 * torch.pi))
        rotation[:, 0] = 0
        rotation[:, -1] = 0
        loops = torch.fft.irfft(magnitude * rotation)
        if normalize_noise_bands:
            loops = loops / loops.abs().max()
        self.register_buffer("noise_bands", loops.float())
        self.n_filters = n_filters

        def ctrl_fn(other_split_trsfm):
            def split_and_trsfm(split_sizes, trsfm_fns):
                return other_split_trsfm(split_sizes + ((n_filters,),), trsfm_fns + ((lambda x: (x,)),))

            return split_and_trsfm

        self.ctrl = ctrl_fn

    def forward(self, ref: AudioTensor, log_gain: AudioTensor, rand_offset: torch.Tensor = None) -> AudioTensor:
        from . import functional as GF

        B, T = ref.shape
        K, period = self.noise_bands.shape
        if rand_offset is None:
            rand_offset = torch.randint(0, period, (B, K), device=self.noise_bands.device)
        hop = int(log_gain.hop_length)
        return AudioTensor(GF.noise_band(self.noise_bands, rand_offset.to(self.noise_bands.device), log_gain.as_tensor(),
                                         hop, T))
