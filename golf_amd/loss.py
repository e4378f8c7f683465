"""Spectral training criteria (reference loss/spec.py:11-67) and the magnitude/power spectrogram
front end both the criterion and the encoder use (the reference takes it from torchaudio.transforms.Spectrogram,
a third-party module: restated here on torch.stft = rocFFT, same arguments, same ``window`` buffer name so that
checkpoints keep their keys).

Stock PyTorch by design: nothing here is on the synthesis hot path; what the rewrite changes is only that no
step forces a host sync.  Written independently of the reference; parity pinned by tests/golden/g20.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Union

import torch
import torch.nn as nn
import torch.nn.functional as F

from .audiotensor import AudioTensor
from .utils import get_window_fn

__all__ = ["Spectrogram", "SSSLoss", "MSSLoss"]


def _plain(x) -> torch.Tensor:
    return x.as_tensor() if isinstance(x, AudioTensor) else x


class Spectrogram(nn.Module):
    """|STFT|^power, (..., T) -> (..., n_fft // 2 + 1, 1 + (T + 2 * (n_fft // 2) - n_fft) // hop) for
    ``center=True``.  Argument names and defaults of torchaudio.transforms.Spectrogram (win_length = n_fft, hop = win_length // 2, periodic Hann window,
    power 2, reflect padding, one-sided, not normalised)."""

    def __init__(self, n_fft: int = 400, win_length: Optional[int] = None, hop_length: Optional[int] = None,
                 pad: int = 0, window_fn: Callable[..., torch.Tensor] = torch.hann_window,
                 power: Optional[float] = 2.0, normalized: bool = False, wkwargs: Optional[dict] = None,
                 center: bool = True, pad_mode: str = "reflect", onesided: bool = True,
                 return_complex: Optional[bool] = None):  # (deprecated torchaudio argument, saved in old configs)
        super().__init__()
        if normalized:
            raise NotImplementedError("Spectrogram(normalized=True) is not used by any GOLF config")
        self.n_fft = n_fft
        self.win_length = win_length if win_length is not None else n_fft
        self.hop_length = hop_length if hop_length is not None else self.win_length // 2
        self.pad, self.power, self.center, self.pad_mode, self.onesided = pad, power, center, pad_mode, onesided
        window = window_fn(self.win_length) if wkwargs is None else window_fn(self.win_length, **wkwargs)
        self.register_buffer("window", window.float())

    def forward(self, waveform) -> torch.Tensor:
        x = _plain(waveform)
        if self.pad > 0:
            x = F.pad(x, (self.pad, self.pad))
        lead = x.shape[:-1]
        z = torch.stft(x.reshape(-1, x.shape[-1]), self.n_fft, self.hop_length, self.win_length, self.window,
                       center=self.center, pad_mode=self.pad_mode, normalized=False, onesided=self.onesided,
                       return_complex=True)
        z = z.reshape(lead + z.shape[-2:])
        if self.power is None:
            return z
        return z.abs() if self.power == 1 else z.abs().pow(self.power)


class SSSLoss(nn.Module):
    """Single-scale spectral loss: L1 of the magnitudes + alpha * L1 of their log2 (loss/spec.py:11-29)."""

    eps = 1e-8

    def __init__(self, alpha: float = 1.0, window: str = "hann", **kwargs):
        super().__init__()
        self.alpha = alpha
        self.spec = Spectrogram(power=1, window_fn=get_window_fn(window), **kwargs)

    def forward(self, pred, target) -> torch.Tensor:
        s_true, s_pred = self.spec(target), self.spec(pred)
        lin = F.l1_loss(s_pred, s_true)
        log = F.l1_loss((s_true + self.eps).log2(), (s_pred + self.eps).log2())
        return lin + self.alpha * log


class MSSLoss(nn.Module):
    """Multi-scale spectral loss: ratio * sum of SSSLoss over ``n_ffts`` with hop = int(n_fft - n_fft * overlap)
    (loss/spec.py:32-67; cfg/ae/vctk.yaml:58-67 uses the prime sizes 509 / 1021 / 2053, window "hanning")."""

    def __init__(self, n_ffts: List[int], alpha: float = 1.0, ratio: float = 1.0, overlap: float = 0.75, **kwargs):
        super().__init__()
        self.losses = nn.ModuleList(
            [SSSLoss(alpha=alpha, n_fft=n, hop_length=int(n - n * overlap), **kwargs) for n in n_ffts])
        self.ratio = ratio

    def forward(self, x_pred, x_true) -> Union[torch.Tensor, AudioTensor]:
        total = self.ratio * sum(loss(x_pred, x_true) for loss in self.losses)
        # the reference's training step calls .as_tensor() on the result (ltng/ae.py:116-118)
        return AudioTensor(total) if isinstance(x_pred, AudioTensor) else total
