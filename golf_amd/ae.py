"""Analysis-synthesis autoencoder and its training step (reference ltng/ae.py:24-143,
cfg/ae/vctk.yaml): encoder(x, f0) -> decoder parameters -> decoder -> multi-scale spectral loss (+ optional f0 /
voicing heads), Adam(1e-4), gradient-norm clipping 0.5.

A plain ``nn.Module`` (the reference is a LightningModule; Lightning, logging, data loading, validation metrics and
checkpoint callbacks are control plane and out of scope).  Same constructor arguments, same ``forward`` and
``training_step`` contract, same sub-module names (``decoder``, ``criterion``, ``encoder``) so state_dicts carry
over.  Written independently.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple, Union

import torch
import torch.nn.functional as F
from torch import nn

from .audiotensor import AudioTensor
from .enc import resolve_class

__all__ = ["VoiceAutoEncoder", "train_step", "data_parallel"]


class VoiceAutoEncoder(nn.Module):
    def __init__(self, decoder: nn.Module, criterion: nn.Module, encoder_class_path: str,
                 encoder_init_args: Optional[Dict] = None, sample_rate: int = 24000, detach_f0: bool = False,
                 detach_voicing: bool = False, train_with_true_f0: bool = True, f0_loss_weight: float = 1.0,
                 voicing_loss_weight: float = 1.0):
        super().__init__()
        self.decoder, self.criterion = decoder, criterion
        split_sizes, trsfms, args_keys = decoder.split_sizes_and_trsfms
        self.encoder = resolve_class(encoder_class_path)(split_sizes=split_sizes, trsfms=trsfms, args_keys=args_keys,
                                                         **(encoder_init_args or {}))
        self.sample_rate = sample_rate
        self.f0_loss_weight, self.voicing_loss_weight = f0_loss_weight, voicing_loss_weight
        self.detach_f0, self.detach_voicing, self.train_with_true_f0 = detach_f0, detach_voicing, train_with_true_f0

    def forward(self, x: AudioTensor = None, f0: AudioTensor = None, params: Dict = None):
        """Analysis (if ``x`` is given) + synthesis; returns (audio, encoder parameters or None) — ltng/ae.py:60-81."""
        params = {} if params is None else params
        enc_params = None
        if x is not None:
            enc_params = self.encoder(x, f0=f0)
            params.update(enc_params)
            if "phase" not in params:
                params["phase"] = params["f0"] / self.sample_rate
            params.pop("f0", None)
            logits = params.pop("voicing_logits", None)
            if logits is not None:
                params["voicing"] = torch.sigmoid(logits)
        return self.decoder(**params), enc_params

    @staticmethod
    def f0_loss(f0_hat: torch.Tensor, f0: torch.Tensor) -> torch.Tensor:
        return F.l1_loss(torch.log(f0_hat + 1e-3), torch.log(f0 + 1e-3))

    def training_step(self, batch: Tuple[torch.Tensor, torch.Tensor], batch_idx: int = 0,
                      unvoiced_f0: Optional[torch.Tensor] = None, return_output: bool = False):
        """The loss of one batch ``(x (B,T), f0 in Hz (B,T))`` — ltng/ae.py:86-143.  Unvoiced samples (f0 == 0) are
        driven at one random frequency per utterance, U(50, 500) Hz; ``unvoiced_f0`` (B,1) injects that draw (parity
        runs, graph capture)."""
        x, f0_hz = AudioTensor(batch[0]), AudioTensor(batch[1])
        params = self.encoder(x, f0=f0_hz if self.train_with_true_f0 else None)
        f0_hat = params.pop("f0", None)
        if self.train_with_true_f0:
            f0t = f0_hz.as_tensor()
            if unvoiced_f0 is None:
                unvoiced_f0 = f0t.new_empty(f0t.shape[0], 1).uniform_(50, 500)
            phase = AudioTensor(torch.where(f0t == 0, unvoiced_f0, f0t) / self.sample_rate)
        elif self.detach_f0:
            phase = f0_hat.detach() / self.sample_rate
        else:
            phase = f0_hat / self.sample_rate
        params["phase"] = phase
        logits = params.pop("voicing_logits", None)
        if logits is not None:
            voicing = torch.sigmoid(logits)
            params["voicing"] = voicing.detach() if self.detach_voicing else voicing
        x_hat = self.decoder(**params)
        n = min(x.shape[1], x_hat.shape[1])
        loss = self.criterion(x_hat[:, :n], x[:, :n])
        loss = loss.as_tensor() if isinstance(loss, AudioTensor) else loss
        if f0_hat is not None:
            target = f0_hz.as_tensor()[:, :: f0_hat.hop_length][:, : f0_hat.shape[1]]
            pred = f0_hat.as_tensor()[:, : target.shape[1]]
            mask = target > 50
            loss = loss + self.f0_loss_weight * self.f0_loss(pred[mask], target[mask])
        if logits is not None:
            target = (f0_hz.as_tensor() > 50).float()[:, :: logits.hop_length][:, : logits.shape[1]]
            loss = loss + self.voicing_loss_weight * F.binary_cross_entropy_with_logits(
                logits.as_tensor()[:, : target.shape[1]], target)
        return (loss, x_hat.as_tensor().detach()) if return_output else loss


class _StepModule(nn.Module):
    """forward = training_step, so that DistributedDataParallel's gradient hooks see the call (DDP instruments
    ``forward``; Lightning wraps its module the same way)."""

    def __init__(self, model: VoiceAutoEncoder):
        super().__init__()
        self.model = model

    def forward(self, x, f0, unvoiced_f0=None):
        return self.model.training_step((x, f0), unvoiced_f0=unvoiced_f0, return_output=True)


def data_parallel(model: VoiceAutoEncoder, device_ids=None):
    """Data-parallel training as the reference's ``strategy: auto`` on several GPUs gives it: one process per GPU, the
    batch sharded across ranks, gradients (the ~6 M encoder parameters, 24 MB) averaged by DDP's bucketed all-reduce
    over RCCL, overlapped with the backward.  The decoder shards with the batch and has no exchange of its own.
    Returns the wrapped step module; pass it to ``train_step`` in place of the model."""
    from torch.nn.parallel import DistributedDataParallel

    return DistributedDataParallel(_StepModule(model), device_ids=device_ids)


def train_step(model, optimizer: torch.optim.Optimizer, batch, clip: float = 0.5,
               unvoiced_f0: Optional[torch.Tensor] = None, return_output: bool = False):
    """One optimisation step as cfg/ae/vctk.yaml configures it: loss -> backward -> clip the global gradient norm at
    ``clip`` (trainer.gradient_clip_val: 0.5) -> optimizer step (Adam, lr 1e-4).  No host sync.  ``model`` is a
    VoiceAutoEncoder or the module returned by ``data_parallel``."""
    optimizer.zero_grad(set_to_none=True)
    if isinstance(model, VoiceAutoEncoder):
        loss, x_hat = model.training_step(batch, unvoiced_f0=unvoiced_f0, return_output=True)
    else:
        loss, x_hat = model(batch[0], batch[1], unvoiced_f0)
    loss.backward()
    torch.nn.utils.clip_grad_norm_(model.parameters(), clip, foreach=True)
    optimizer.step()
    return (loss.detach(), x_hat) if return_output else loss.detach()
