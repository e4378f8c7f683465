"""Glottal-flow wavetable oscillators (reference models/synth.py:40-340), MI355X-native.

``IndexedGlottalFlowTable`` / ``DownsampledIndexedGlottalFlowTable`` keep the reference's constructor
arguments, buffers (``table``, ``R_d_values`` persistent; ``decimater.kernel`` non-persistent),
``model.{1,3}`` parameter names and ``.ctrl`` transforms, so reference checkpoints load unchanged.
``forward`` is one fused HIP path (golf_glottal_osc_fwd_f32): table blend, phase accumulation,
bilinear lookup, equal-energy scaling and decimation -- forward and the gradient w.r.t. table_select_weight, which is
all GOLF training needs (f0 is data).  When the phase, a phase offset or a trainable table must receive gradients
(reference models/synth.py:59-70,213-218) the same module routes through the general differentiable form
(functional.wavetable_osc: HIP table lookup + HIP decimator with their own backward kernels, phase accumulation as
tensor ops); the other table oscillators of the reference (Weighted / WrappedPhase / PulseTrain) build on the same pieces.
"""
from __future__ import annotations

import math

import torch
from torch import Tensor, nn

from . import functional as GF
from .audiotensor import AudioTensor
from .ctrl import Controllable, wrap_ctrl_fn
from .utils import get_transformed_lf, get_transformed_lf_v2

__all__ = ["OscillatorInterface", "GlottalFlowTable", "IndexedGlottalFlowTable", "WeightedGlottalFlowTable",
           "DownsampledIndexedGlottalFlowTable", "DownsampledWeightedGlottalFlowTable",
           "WrappedPhaseDownsampledIndexedGlottalFlowTable", "PulseTrain", "Decimate", "get_downsampler",
           "HarmonicOscillator",
           "AdditiveSynthesizer", "V1AdditiveSynthesizer", "SawToothOscillator", "AdditivePulseTrain"]


class Decimate(nn.Module):
    """Anti-aliased decimation filter holder — stand-in for ``kazane.Decimate(q)`` (third-party, absent,
    taps unknown: decimation parity is unpinned).  Own design: Hann-windowed sinc,
    K = 2*zeros*q+1 taps, cutoff rolloff/(2q), unity DC gain.  ``kernel`` has kazane's (1,1,K) shape so the
    real taps can be copied in; the filtering itself is fused into the oscillator kernel."""

    def __init__(self, q: int = 2, zeros: int = 16, rolloff: float = 0.945):
        super().__init__()
        self.q = q
        n = torch.arange(-zeros * q, zeros * q + 1, dtype=torch.float64)
        h = torch.sinc(n * rolloff / q) * (0.5 + 0.5 * torch.cos(math.pi * n / (zeros * q + 1)))
        self.register_buffer("kernel", (h / h.sum()).float().view(1, 1, -1))

    @property
    def taps(self) -> Tensor:
        return self.kernel.reshape(-1)


class OscillatorInterface(Controllable):
    """``check_ranges=True`` restores the reference's host-syncing input asserts
    (models/synth.py:26-36: 2-D phase in [0, 0.5]); off by default to keep the stream asynchronous."""

    def __init__(self, check_ranges: bool = False) -> None:
        super().__init__()
        self.check_ranges = check_ranges

    def forward(self, phase: AudioTensor, *args, **kwargs) -> AudioTensor:
        raise NotImplementedError


class GlottalFlowTable(OscillatorInterface):
    """(table_size, points) wavetable of LF glottal-flow (derivative) pulses over a log grid of R_d
    (reference models/synth.py:58-120)."""

    def __init__(self, table_size: int = 100, table_type: str = "derivative",
                 normalize_method: str = "constant_power", align_peak: bool = True, trainable: bool = False,
                 min_R_d: float = 0.3, max_R_d: float = 2.7, lf_v2: bool = False, check_ranges: bool = False,
                 **kwargs):
        super().__init__(check_ranges=check_ranges)
        self.register_buffer("R_d_values",
                             torch.exp(torch.linspace(math.log(min_R_d), math.log(max_R_d), table_size)))
        if lf_v2:
            table = get_transformed_lf_v2(self.R_d_values, **kwargs)
        else:
            table = torch.stack([get_transformed_lf(R_d=r, **kwargs) for r in self.R_d_values])
        if table_type == "flow":
            table = table.cumsum(dim=1)
        elif table_type != "derivative":
            raise ValueError(f"unknown table_type: {table_type}")
        if align_peak:
            peak = table.argmin(dim=1) if table_type == "derivative" else table.argmax(dim=1)
            target = int(peak.max())
            table = torch.stack([torch.roll(row, target - int(p)) for row, p in zip(table, peak)])
        if normalize_method == "constant_power":
            table = table / table.norm(dim=1, keepdim=True) * math.sqrt(table.shape[1])
        elif normalize_method == "peak":
            if table_type == "flow":
                table = table / table.max(dim=1, keepdim=True).values
        elif normalize_method is not None:
            raise ValueError(f"unknown normalize_method: {normalize_method}")
        if trainable:
            self.register_parameter("table", nn.Parameter(table))
        else:
            self.register_buffer("table", table)


class IndexedGlottalFlowTable(GlottalFlowTable):
    def __init__(self, *args, oversampling: int = 1, equal_energy: bool = False, **kwargs):
        super().__init__(*args, **kwargs)
        self.ctrl = wrap_ctrl_fn(split_size=(1,), trsfm_fn=lambda x: (torch.sigmoid(x),))
        self.equal_energy = equal_energy
        self.oversampling = oversampling
        if oversampling > 1:
            self.decimater = Decimate(oversampling)
            # non-persistent, like the reference's re-registration (models/synth.py:209-211)
            kernel = self.decimater.kernel
            del self.decimater._buffers["kernel"]
            self.decimater.register_buffer("kernel", kernel, persistent=False)

    def output_length(self, phase: AudioTensor) -> int:
        """Number of samples forward() will return for this phase input."""
        return GF.osc_lengths(phase.shape[1], phase.hop_length, self.oversampling)[1]

    supports_fused_add = True   # forward(..., add=x) returns oscillator + x (hop 1), truncated to the shorter

    def forward(self, phase: AudioTensor, table_select_weight: AudioTensor, phase_offset: AudioTensor = None,
                return_pre: bool = False, add: AudioTensor = None) -> AudioTensor:
        assert phase.ndim == 2, phase.shape
        assert table_select_weight.dim() == 2
        if self.check_ranges:
            assert torch.all(phase >= 0) and torch.all(phase <= 0.5)
            assert torch.all(table_select_weight >= 0) and torch.all(table_select_weight <= 1)
        taps = self.decimater.taps if self.oversampling > 1 else None
        ph, w = phase.as_tensor(), table_select_weight.as_tensor()
        grads = torch.is_grad_enabled() and (ph.requires_grad or self.table.requires_grad)
        if grads or phase_offset is not None:
            # general path (reference semantics under autograd, models/synth.py:213-263): a differentiable phase,
            # a phase offset or a trainable table.  Tables are blended with tensor ops (gradients to the weight and the
            # table), the lookup and the decimation are HIP kernels with their own backward.
            off = None if phase_offset is None else (phase_offset.as_tensor() if isinstance(phase_offset, AudioTensor)
                                                      else phase_offset)
            res = GF.wavetable_osc(ph, int(phase.hop_length), GF.blend_tables(self.table, w),
                                   int(table_select_weight.hop_length), self.oversampling, self.equal_energy, off, taps,
                                   decimate=not return_pre)
            if return_pre:
                pre = res
                res = GF.decimate_fir(pre, taps, self.oversampling) if self.oversampling > 1 else pre
                return AudioTensor(res), pre
            if add is not None:
                n = min(res.shape[1], add.shape[1])
                return AudioTensor(res[:, :n] + add.as_tensor()[:, :n])
            return AudioTensor(res)
        res = GF.glottal_osc(ph, w, self.table, taps, phase_hop=phase.hop_length, w_hop=table_select_weight.hop_length,
                             oversampling=self.oversampling, equal_energy=self.equal_energy, return_pre=return_pre,
                             add=None if add is None else add.as_tensor())
        if add is not None:   # what AudioTensor addition of two hop-1 signals does: truncate to the shorter
            assert add.hop_length == 1 and not return_pre
            # (no slice when there is nothing to cut: a full-range slice still records a SliceBackward, whose backward is a
            #  6 MB fill + a 6 MB copy in front of the oscillator's backward)
            return AudioTensor(res if add.shape[1] >= res.shape[1] else res[:, : add.shape[1]])
        if return_pre:
            return AudioTensor(res[0]), res[1]
        return AudioTensor(res)


class WeightedGlottalFlowTable(GlottalFlowTable):
    """Soft table selection (reference models/synth.py:266-294): per-frame tables = softmax weights @ table, looked up at
    the running phase (no oversampling, no equal-energy scaling).  Differentiable w.r.t. everything."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.ctrl = wrap_ctrl_fn(split_size=(self.table.shape[0],), trsfm_fn=lambda x: (torch.softmax(x, 2),))

    def forward(self, phase: AudioTensor, table_select_weight: AudioTensor, phase_offset: AudioTensor = None) -> AudioTensor:
        assert table_select_weight.dim() == 3 and table_select_weight.shape[2] == self.table.shape[0]
        if self.check_ranges:
            assert torch.all(phase >= 0) and torch.all(phase <= 0.5)
            assert torch.all(table_select_weight >= 0) and torch.all(table_select_weight <= 1)
        tables = table_select_weight.as_tensor() @ self.table
        off = None if phase_offset is None else (phase_offset.as_tensor() if isinstance(phase_offset, AudioTensor)
                                                  else phase_offset)
        return AudioTensor(GF.wavetable_osc(phase.as_tensor(), int(phase.hop_length), tables,
                                            int(table_select_weight.hop_length), 1, False, off))


def get_downsampler(hop_rate: int, in_channels: int, output_channels: int) -> nn.Sequential:
    """AvgPool(hop_rate) -> 1x1 conv -> GLU -> 1x1 conv (reference models/synth.py:297-315);
    indices 1 and 3 carry the parameters, matching checkpoint keys ``model.1.*`` / ``model.3.*``."""
    return nn.Sequential(
        nn.AvgPool1d(kernel_size=hop_rate, stride=hop_rate, padding=hop_rate // 2),
        nn.Conv1d(in_channels, in_channels * 2, kernel_size=1),
        nn.GLU(dim=1),
        nn.Conv1d(in_channels, output_channels, kernel_size=1),
    )


class DownsampledIndexedGlottalFlowTable(IndexedGlottalFlowTable):
    """Table selection weight predicted at hop_rate x the encoder hop from ``in_channels`` features."""

    def __init__(self, hop_rate: int, in_channels: int, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.hop_rate = hop_rate
        self.model = get_downsampler(hop_rate, in_channels, 1)
        self.ctrl = wrap_ctrl_fn(
            split_size=(in_channels,),
            trsfm_fn=lambda h: (
                AudioTensor(
                    self.model(torch.transpose(h.as_tensor(), 1, 2)).squeeze(1).sigmoid(),
                    hop_length=h.hop_length * self.hop_rate,
                ),
            ),
        )


class WrappedPhaseDownsampledIndexedGlottalFlowTable(DownsampledIndexedGlottalFlowTable):
    """Indexed table lookup at a GIVEN wrapped phase (reference models/synth.py:343-375): no accumulation, no
    oversampling, no range hook on the first argument."""

    def forward(self, wrapped_phase: AudioTensor, table_select_weight: AudioTensor) -> AudioTensor:
        assert wrapped_phase.hop_length == 1
        assert table_select_weight.dim() == 2
        if self.check_ranges:
            assert torch.all(table_select_weight >= 0) and torch.all(table_select_weight <= 1)
        tables = GF.blend_tables(self.table, table_select_weight.as_tensor())
        return AudioTensor(GF.wavetable_lookup(wrapped_phase.as_tensor().float(), tables,
                                               int(table_select_weight.hop_length)))


class DownsampledWeightedGlottalFlowTable(WeightedGlottalFlowTable):
    """Soft table weights predicted at hop_rate x the encoder hop (reference models/synth.py:378-400)."""

    def __init__(self, hop_rate: int, in_channels: int, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.hop_rate = hop_rate
        self.model = get_downsampler(hop_rate, in_channels, self.table.shape[0])
        self.ctrl = wrap_ctrl_fn(
            split_size=(in_channels,),
            trsfm_fn=lambda h: (
                AudioTensor(
                    self.model(torch.transpose(h.as_tensor(), 1, 2)).softmax(dim=1).transpose(1, 2),
                    hop_length=h.hop_length * self.hop_rate,
                ),
            ),
        )


class PulseTrain(OscillatorInterface):
    """One impulse of height rsqrt(phase increment) at every sample where the running phase wraps (reference
    models/synth.py:507-523).  The wrap instants are a comparison of neighbouring wrapped phases -- elementwise work on
    the accumulated phase (float64 here, float32 in the reference); gradients reach the impulse heights only, as there."""

    def forward(self, phase: AudioTensor, phase_offset: AudioTensor = None) -> AudioTensor:
        if self.check_ranges:
            assert torch.all(phase >= 0) and torch.all(phase <= 0.5)
        up = GF.linear_upsample(phase.as_tensor(), int(phase.hop_length))
        inst = torch.cumsum(up.double(), 1)
        if phase_offset is not None:
            off = phase_offset.as_tensor() if isinstance(phase_offset, AudioTensor) else phase_offset
            n = min(inst.shape[1], off.shape[1])
            inst, up = inst[:, :n] + off[:, :n].double(), up[:, :n]
        wrapped = torch.remainder(inst, 1.0)
        hit = (wrapped[:, 1:] - wrapped[:, :-1]) < 0
        out = torch.zeros_like(up)
        out[:, 1:] = torch.where(hit, torch.rsqrt(up[:, 1:]), torch.zeros_like(up[:, 1:]))
        return AudioTensor(out)


# ---------------------------------------------------------------------------------------------
# harmonic oscillator bank (reference models/synth.py:403-547): the sources of the DDSP / NHV / WORLD / MLSA /
# SawSing / PULF baselines.  One fused kernel (golf_harmonic_osc_{fwd,bwd_amp}_f32): the reference's (B,T,H)
# phase / mask / amplitude / sine tensors are never formed.
# ---------------------------------------------------------------------------------------------
class HarmonicOscillator(OscillatorInterface):
    """out = sum_h sin(2 pi cumsum(h * up(phase))) * amplitudes[..., h], harmonics at or above Nyquist muted."""

    def _run(self, phase: AudioTensor, num_harmonics: int, amplitudes: AudioTensor = None, tscale: AudioTensor = None,
             hscale: Tensor = None, initial_phase=None, phase_offset=None) -> AudioTensor:
        assert phase.ndim == 2, phase.shape
        if initial_phase is not None and isinstance(initial_phase, AudioTensor):
            initial_phase = initial_phase.as_tensor()
        if self.check_ranges:
            assert torch.all(phase >= 0) and torch.all(phase <= 0.5)
        y = GF.harmonic_osc(phase.as_tensor(), num_harmonics, phase_hop=int(phase.hop_length),
                            amp=None if amplitudes is None else amplitudes.as_tensor(),
                            amp_hop=1 if amplitudes is None else int(amplitudes.hop_length),
                            tscale=None if tscale is None else tscale.as_tensor(),
                            ts_hop=1 if tscale is None else int(tscale.hop_length), hscale=hscale,
                            phase_offset=None if phase_offset is None else phase_offset.as_tensor(),
                            po_hop=1 if phase_offset is None else int(phase_offset.hop_length),
                            initial_phase=initial_phase)
        return AudioTensor(y)

    def forward(self, phase: AudioTensor, amplitudes: AudioTensor, initial_phase=None, phase_offset=None) -> AudioTensor:
        return self._run(phase, amplitudes.shape[-1], amplitudes, initial_phase=initial_phase,
                         phase_offset=phase_offset)


def _sqrt_two_phase(phase: AudioTensor) -> AudioTensor:
    """rsqrt(0.5 / phase): the equal-energy factor of the reference (synth.py:465-466, 543-544), at the phase's hop."""
    return phase.new_tensor(torch.rsqrt(0.5 / phase.as_tensor()))


class AdditiveSynthesizer(HarmonicOscillator):
    def __init__(self, num_harmonics: int = 150, **kwargs) -> None:
        super().__init__(**kwargs)
        self.ctrl = wrap_ctrl_fn(
            split_size=(1, num_harmonics),
            trsfm_fn=lambda log_gain, amplitudes_logits: (torch.exp(log_gain) * torch.sigmoid(amplitudes_logits),))

    def forward(self, phase: AudioTensor, amplitudes: AudioTensor, **kwargs) -> AudioTensor:
        scale = _sqrt_two_phase(phase)
        if phase.hop_length == 1:
            # amplitudes are upsampled to the sample rate and multiplied point-wise: the factor can ride along as a
            # per-sample scale instead of materialising (B, T, H)
            return self._run(phase, amplitudes.shape[-1], amplitudes, tscale=scale, **kwargs)
        # coarser phase: the reference forms the product at the finer of the two hops and upsamples that
        return self._run(phase, amplitudes.shape[-1], amplitudes * torch.unsqueeze(scale, -1), **kwargs)


class V1AdditiveSynthesizer(HarmonicOscillator):
    def __init__(self, num_harmonics: int = 150, **kwargs) -> None:
        super().__init__(**kwargs)

        def normalised(log_gain, amplitudes_logits):
            s = torch.sigmoid(amplitudes_logits.as_tensor())
            return (torch.exp(log_gain) * amplitudes_logits.new_tensor(s / s.sum(-1, keepdim=True)),)

        self.ctrl = wrap_ctrl_fn(split_size=(1, num_harmonics), trsfm_fn=normalised)


class SawToothOscillator(HarmonicOscillator):
    """Band-limited sawtooth: harmonic h with amplitude 1/h (the reference's ``gain`` argument is unused there too)."""

    def __init__(self, num_harmonics: int, gain: float = 0.4, **kwargs) -> None:
        super().__init__(**kwargs)
        self.gain = gain
        self.register_buffer("amplitudes", 1 / torch.arange(1, num_harmonics + 1))

    def forward(self, phase: AudioTensor, initial_phase=None, phase_offset=None, **kwargs) -> AudioTensor:
        return self._run(phase, self.amplitudes.numel(), hscale=self.amplitudes, initial_phase=initial_phase,
                         phase_offset=phase_offset)


class AdditivePulseTrain(HarmonicOscillator):
    def __init__(self, num_harmonics: int = 155, **kwargs) -> None:
        super().__init__(**kwargs)
        self.num_harmonics = num_harmonics

    def forward(self, phase: AudioTensor, initial_phase=None, phase_offset=None, **kwargs) -> AudioTensor:
        # all harmonics share rsqrt(0.5/phase) (at the phase's hop, linearly upsampled like any amplitude)
        return self._run(phase, self.num_harmonics, tscale=_sqrt_two_phase(phase), initial_phase=initial_phase,
                         phase_offset=phase_offset)
