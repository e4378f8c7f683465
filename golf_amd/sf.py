"""Source-filter decoder assemblies (reference models/sf.py:13-64, models/hpn.py:11-57):
composition only — the oscillator and the end filter are the HIP-backed modules of this package."""
from __future__ import annotations

from typing import Optional, Tuple, Union

import torch
import torch.nn.functional as F

from .audiotensor import AudioTensor
from .ctrl import PassThrough, Synth

__all__ = ["SourceFilterSynth", "HarmonicPlusNoiseSynth"]

# Round 6: in inference, for a caller without batches in flight, the glottal oscillator and the sample-wise end filter's transition
# maps run as ONE launch (functional.source_filter_ss / golf_source_transitions_f32, ABI 6): bit-identical to the composition,
# a lone B = 32 batch ~4 us sooner.  False: always the composition.
FUSE_SOURCE_MAPS = True


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
            fused = self._source_and_end_filter(phase, harm_oscillator_params, end_filter_params, nz) if target is None else None
            if fused is not None:
                return self.room_filter(fused)
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


    def _source_and_end_filter(self, phase, osc_params, filt_params, nz) -> Optional[AudioTensor]:
        """``end_filter(harm_oscillator(phase, w, add=nz), gain, a)`` through functional.source_filter_ss where that is the same
        computation: the indexed glottal oscillator on its fused path (oversampled, no phase offset), the sample-wise end filter
        (not its frame-wise subclass) with nothing prefetched and no health monitor, inference, no batches in flight.  None:
        the caller composes the modules as ever."""
        from . import functional as GF
        from .filters import LTVMinimumPhaseFilterPrecise
        from .synth import IndexedGlottalFlowTable

        osc, filt = self.harm_oscillator, self.end_filter
        if not (FUSE_SOURCE_MAPS and not GF.THROUGHPUT_MODE and type(filt) is LTVMinimumPhaseFilterPrecise
                and isinstance(osc, IndexedGlottalFlowTable) and osc.oversampling > 1 and len(osc_params) == 1
                and len(filt_params) == 2 and getattr(filt, "_prepared", None) is None and not filt.health_check
                and not osc.check_ranges and phase.ndim == 2 and nz.hop_length == 1):
            return None
        w, (gain, a) = osc_params[0], filt_params
        ts = [t.as_tensor() for t in (phase, w, nz, gain, a)] + [osc.table]
        if not all(t.is_cuda for t in ts) or (torch.is_grad_enabled() and any(t.requires_grad for t in ts)):
            return None
        if (w.dim() != 2 or a.dim() != 3 or gain.dim() != 2 or int(gain.hop_length) != int(a.hop_length)
                or a.shape[1] != gain.shape[1] or a.shape[0] != phase.shape[0] or gain.shape[0] != phase.shape[0]):
            return None   # (the modules' own asserts speak for malformed inputs)
        n_osc = osc.output_length(phase)
        y = GF.source_filter_ss(ts[0], ts[1], osc.table, osc.decimater.taps, int(phase.hop_length), int(w.hop_length),
                                osc.oversampling, osc.equal_energy, ts[3], ts[4], int(a.hop_length), add=ts[2],
                                length=nz.shape[1] if nz.shape[1] < n_osc else None)
        return AudioTensor(y)


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
