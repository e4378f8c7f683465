"""Source-filter decoder assemblies (reference models/sf.py:13-64, models/hpn.py:11-57):
composition only — the oscillator and the end filter are the HIP-backed modules of this package."""
from __future__ import annotations

from typing import Optional, Tuple, Union

import torch
import torch.nn.functional as F

from .audiotensor import AudioTensor
from .ctrl import PassThrough, Synth

__all__ = ["SourceFilterSynth", "HarmonicPlusNoiseSynth"]


class SourceFilterSynth(Synth):
    """GOLF-ss / GOLF-ff decoder: (oscillator + filtered noise) -> end filter -> room filter."""

    def __init__(self, harm_oscillator, noise_generator, noise_filter, end_filter, room_filter=None,
                 subtract_harmonics: bool = True, check_ranges: bool = False, overlap_prefetch: bool = False):
        super().__init__()
        self.overlap_prefetch = overlap_prefetch  # see LTVMinimumPhaseFilterPrecise.prefetch (off: does not pay at B=32)
        self.subtract_harmonics = subtract_harmonics
        self.check_ranges = check_ranges  # the reference's host-syncing asserts, off by default
        # attribute order defines the encoder channel layout (models/ctrl.py:59-69)
        self.harm_oscillator = harm_oscillator
        self.noise_generator = noise_generator
        self.noise_filter = noise_filter
        self.end_filter = end_filter
        self.room_filter = room_filter if room_filter is not None else PassThrough()

    def forward(self, phase: AudioTensor, harm_oscillator_params: Tuple[AudioTensor, ...],
                noise_generator_params: Tuple[AudioTensor, ...], noise_filter_params: Tuple[AudioTensor, ...],
                end_filter_params: Tuple[AudioTensor, ...], voicing: Optional[AudioTensor] = None,
                target: Optional[AudioTensor] = None, **other_params) -> AudioTensor:
        if (self.overlap_prefetch and target is None and hasattr(self.end_filter, "prefetch")
                and hasattr(self.harm_oscillator, "output_length")):
            # overlap the filter's excitation-independent phase with the oscillator (second HIP stream)
            self.end_filter.prefetch(*end_filter_params, n_samples=self.harm_oscillator.output_length(phase))
        if (voicing is None and not self.subtract_harmonics
                and getattr(self.harm_oscillator, "supports_fused_add", False)
                and not getattr(self.noise_generator, "uses_reference_values", True)):
            # src = harm_osc + noise_filter(noise): the noise branch needs the oscillator output only for its shape,
            # so it runs first and the sum is fused into the oscillator's last kernel
            n = self.harm_oscillator.output_length(phase)
            ref = AudioTensor(phase.as_tensor().new_empty((phase.shape[0], n)))   # shape/device carrier, never read
            nz = self.noise_filter(self.noise_generator(ref, *noise_generator_params), *noise_filter_params)
            src = self.harm_oscillator(phase, *harm_oscillator_params, add=nz)
        else:
            harm_osc = self.harm_oscillator(phase, *harm_oscillator_params)
            if voicing is not None:
                if self.check_ranges:
                    assert torch.all(voicing >= 0) and torch.all(voicing <= 1)
                harm_osc = harm_osc * F.threshold(voicing, 0.5, 0)
            src = harm_osc + self.noise_filter(self.noise_generator(harm_osc, *noise_generator_params),
                                               *noise_filter_params)
            if self.subtract_harmonics:
                src = src - self.noise_filter(harm_osc, *noise_filter_params)
        if target is not None:
            return self.end_filter.reverse(src, target, *end_filter_params)
        return self.room_filter(self.end_filter(src, *end_filter_params))


class HarmonicPlusNoiseSynth(Synth):
    """GOLF-v1 decoder: filtered oscillator + filtered noise -> static end filter."""

    def __init__(self, harm_oscillator, noise_generator, harm_filter, noise_filter, end_filter,
                 check_ranges: bool = False):
        super().__init__()
        self.check_ranges = check_ranges
        self.harm_oscillator = harm_oscillator
        self.noise_generator = noise_generator
        self.harm_filter = harm_filter
        self.noise_filter = noise_filter
        self.end_filter = end_filter

    def forward(self, phase: AudioTensor, harm_oscillator_params, noise_generator_params, harm_filter_params,
                noise_filter_params, voicing: Optional[AudioTensor] = None, **other_params) -> AudioTensor:
        if voicing is not None:
            if self.check_ranges:
                assert torch.all(voicing >= 0) and torch.all(voicing <= 1)
            phase = phase * voicing
        harm_osc = self.harm_oscillator(phase, *harm_oscillator_params)
        noise = self.noise_generator(harm_osc, *noise_generator_params)
        out = self.harm_filter(harm_osc, *harm_filter_params) + self.noise_filter(noise, *noise_filter_params)
        return self.end_filter(out)
