"""Noise sources (reference models/noise.py:19-55).  RNG is torch's device generator — RNG streams
cannot match across devices, so parity tests inject the noise tensor."""
from __future__ import annotations

import math

import torch

from .audiotensor import AudioTensor
from .ctrl import Controllable

__all__ = ["NoiseInterface", "StandardNormalNoise", "UniformNoise", "SignFlipNoise"]


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
