"""Spectrogram U-Net-style encoder backbone (reference models/unet.py:86-224, the variant every
shipped config uses: ``include_env_features=False``, LSTM).

(power spectrogram) -> log -> running min/max normalisation -> [Conv2d (2s+1, 3) + BatchNorm + ReLU + MaxPool (s,1)]
per stage along the frequency axis -> flatten (channels x remaining bins) -> concat log1p(f0) -> bidirectional LSTM ->
LayerNorm -> out_linear.  Stock PyTorch (MIOpen convolutions / LSTM, rocFFT).  Module and buffer names are the
reference's, so its checkpoints load: ``spectrogram.window``, ``cnns.{0,1,4,5,...}``, ``lstm.*``, ``norm.*``,
``out_linear.*``, ``log_spec_min``, ``log_spec_max``.

Two deliberate differences: the convolution stack is kept in channels-last memory format (layout only), and the
running extrema of the log spectrogram are updated ON THE DEVICE
(``torch.minimum`` into the buffers) — the reference reads them back with ``.item()`` every training step
(models/unet.py:207-209), a host sync per step that would also break hipGraph capture.  Values are identical.
"""
from __future__ import annotations

from functools import reduce
from typing import List

import torch
import torch.nn as nn

from .audiotensor import AudioTensor
from .enc import BackboneModelInterface
from .loss import Spectrogram

__all__ = ["UNetEncoder"]


class UNetEncoder(BackboneModelInterface):
    def __init__(self, out_channels: int, n_fft: int = 1024, hop_length: int = 256,
                 channels: List[int] = (16, 32, 64, 128), strides: List[int] = (4, 4, 4, 4),
                 lstm_hidden_size: int = 128, include_env_features: bool = False, num_harmonics: int = 150,
                 sample_rate: int = 22050, f0_conditioning: bool = True, use_lru: bool = False, **lstm_kwargs):
        if include_env_features or use_lru:
            raise NotImplementedError("UNetEncoder: include_env_features / use_lru are not used by any shipped config")
        super().__init__(lstm_hidden_size * 2, out_channels)
        self.n_fft, self.hop_length, self.f0_conditioning = n_fft, hop_length, f0_conditioning
        self.spectrogram = Spectrogram(n_fft=n_fft, hop_length=hop_length, center=True)
        stages, c_in = [], 1
        for c_out, s in zip(channels, strides):
            stages += [nn.Conv2d(c_in, c_out, (2 * s + 1, 3), padding=(s, 1)), nn.BatchNorm2d(c_out), nn.ReLU(),
                       nn.MaxPool2d((s, 1), stride=(s, 1))]
            c_in = c_out
        # NHWC weights: MIOpen's fp32 implicit-GEMM convolutions are NHWC kernels; with NCHW tensors it transposes around
        # every call (measured on the config-5 step, B=64: 68.0 -> 60.4 ms).  Shapes, values and state_dict are unchanged.
        self.cnns = nn.Sequential(*stages).to(memory_format=torch.channels_last)
        flat = (n_fft // 2 + 1) // reduce(lambda p, q: p * q, strides) * c_in
        self.lstm = nn.LSTM(flat + (1 if f0_conditioning else 0), lstm_hidden_size, batch_first=True,
                            bidirectional=True, **lstm_kwargs)
        self.norm = nn.LayerNorm(lstm_hidden_size * 2)
        self.register_buffer("log_spec_min", torch.tensor(torch.inf))
        self.register_buffer("log_spec_max", torch.tensor(-torch.inf))

    def forward(self, x: AudioTensor, f0: AudioTensor = None) -> AudioTensor:
        assert x.hop_length == 1
        spec = self.spectrogram(x.as_tensor())                       # (B, bins, frames)
        f0_frames = None
        if self.f0_conditioning and f0 is not None:
            f0_frames = f0.set_hop_length(self.hop_length).truncate(spec.size(2)).as_tensor()
            spec = spec[..., : f0_frames.size(-1)]
        log_spec = spec.unsqueeze(1).add(1e-8).log()
        if self.training:
            with torch.no_grad():                                    # device-side running extrema, no host sync
                self.log_spec_min.copy_(torch.minimum(self.log_spec_min, log_spec.min()))
                self.log_spec_max.copy_(torch.maximum(self.log_spec_max, log_spec.max()))
        feature = (log_spec - self.log_spec_min) / (self.log_spec_max - self.log_spec_min)
        h = self.cnns(feature)                                       # (B, C, bins', frames)
        h = torch.flatten(h, 1, 2).mT                                # (B, frames, C * bins')
        if f0_frames is not None:
            h = torch.cat([h, torch.log1p(f0_frames).unsqueeze(-1)], dim=-1)
        h = self.norm(self.lstm(h)[0])
        return AudioTensor(super().forward(h), hop_length=self.hop_length)
