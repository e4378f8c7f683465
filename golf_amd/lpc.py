"""Frame-wise LPC synthesisers of the reference's models/lpc.py, MI355X-native.

``BatchSecondOrderLPCSynth`` (reference models/lpc.py:94-131) is the reference's statement of the *cascaded-biquad*
all-pole filter: every frame runs through K second-order sections.  Here the cascade is a systolic pipeline across
the lanes of a DPP row (golf_biquad_frames_ola_fwd_f32, csrc/lpc_ff.hip), and so is its backward
(golf_biquad_frames_ola_bwd_f32: the adjoint of a cascade is the reversed cascade on the reversed signal), so the module is
differentiable w.r.t. the excitation, the gains and the section coefficients like the reference's (autograd through its K
lfilter calls, models/lpc.py:115-118).
"""
from __future__ import annotations

import torch
from torch import Tensor, nn

from . import functional as GF
from .utils import get_window_fn

__all__ = ["BatchSecondOrderLPCSynth"]


class BatchSecondOrderLPCSynth(nn.Module):
    def __init__(self, hop_length: int, window_size: int = None, window: str = "hann"):
        super().__init__()
        self.hop_length = hop_length
        self.window_size = hop_length * 4 if window_size is None else window_size
        self.padding = (self.window_size - self.hop_length) // 2
        # the reference keeps diag(window) as a (W,1,W) conv kernel `_kernel`; only its diagonal is ever used
        self.register_buffer("_window", get_window_fn(window)(self.window_size).float(), persistent=False)

    def forward(self, ex: Tensor, gain: Tensor, biquads: Tensor) -> Tensor:
        assert ex.ndim == 2
        assert gain.ndim == 2
        assert biquads.ndim == 4 and biquads.shape[-1] == 3
        return GF.biquad_frames_ola(ex, gain, biquads, self._window, self.hop_length, pad=self.padding, frame_gain=True)
