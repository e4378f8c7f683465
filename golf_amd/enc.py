"""Encoder interface (reference models/enc.py:18-100): a backbone producing one logit vector per
frame, split by the decoder's control protocol (golf_amd/ctrl.py) into the keyword arguments of the decoder.

Stock PyTorch; state_dict keys (``backbone.*``, ``backbone.out_linear.*``) are the reference's.  Written
independently; the split / transform order is pinned by tests/golden/g20 and g12.
"""
from __future__ import annotations

import math
from importlib import import_module
from itertools import accumulate
from typing import Any, Callable, Dict, Tuple, Union

import torch
from torch import nn

from .audiotensor import AudioTensor

__all__ = ["BackboneModelInterface", "VocoderParameterEncoderInterface"]

# the reference's YAML names its own module paths; the same configs resolve to this package
_MODULE_ALIASES = {"models.unet": "golf_amd.unet", "models.enc": "golf_amd.enc"}


def resolve_class(path: str):
    module_path, class_name = path.rsplit(".", 1)
    return getattr(import_module(_MODULE_ALIASES.get(module_path, module_path)), class_name)


class BackboneModelInterface(nn.Module):
    """A backbone ends in ``out_linear`` (hidden -> sum of all split sizes), zero-initialised so that training starts
    from the decoder's neutral parameters (models/enc.py:23-28)."""

    def __init__(self, linear_in_channels: int, linear_out_channels: int):
        super().__init__()
        self.out_linear = nn.Linear(linear_in_channels, linear_out_channels)
        nn.init.zeros_(self.out_linear.weight)
        nn.init.zeros_(self.out_linear.bias)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.out_linear(x)


class VocoderParameterEncoderInterface(nn.Module):
    """backbone(x, f0) -> (B, F, C) logits -> dict of decoder arguments.

    ``split_sizes`` / ``trsfms`` / ``args_keys`` come from ``decoder.split_sizes_and_trsfms``; optional f0 and
    voicing heads are PREPENDED (f0 first), as in models/enc.py:48-64.  A group of sizes ``(1, 22)`` yields two
    tensors — size-1 splits are squeezed to (B, F) — which the group's transform maps to a tuple of AudioTensors.
    """

    def __init__(self, backbone_type: str, learn_voicing: bool = False, learn_f0: bool = True, f0_min: float = 80,
                 f0_max: float = 1000, split_sizes: Tuple[Tuple[int, ...], ...] = (),
                 trsfms: Tuple[Callable[..., Tuple[torch.Tensor, ...]], ...] = (), args_keys: Tuple[str, ...] = (),
                 **kwargs):
        super().__init__()
        lo, hi = math.log(f0_min), math.log(f0_max)
        split_sizes, trsfms, args_keys = tuple(split_sizes), tuple(trsfms), tuple(args_keys)
        if learn_voicing:
            split_sizes, trsfms, args_keys = ((1,),) + split_sizes, (lambda x: x,) + trsfms, ("voicing_logits",) + args_keys
        if learn_f0:
            f0_head = lambda logits: torch.exp(torch.sigmoid(logits) * (hi - lo) + lo)  # noqa: E731
            split_sizes, trsfms, args_keys = ((1,),) + split_sizes, (f0_head,) + trsfms, ("f0",) + args_keys
        self.split_sizes, self.trsfms, self.args_keys = split_sizes, trsfms, args_keys
        flat = [s for group in split_sizes for s in group]
        self.backbone = resolve_class(backbone_type)(out_channels=sum(flat), **kwargs)

    def forward(self, x: AudioTensor, *args: Any, **kwargs: Any) -> Dict[str, Union[AudioTensor, Tuple[AudioTensor]]]:
        h = self.backbone(x, *args, **kwargs)
        flat = [s for group in self.split_sizes for s in group]
        pieces = [h.new_tensor(t.squeeze(2) if t.shape[2] == 1 else t)
                  for t in torch.split(h.as_tensor(), flat, dim=2)]
        bounds = list(accumulate((len(g) for g in self.split_sizes), initial=0))
        return {key: fn(*pieces[i:j]) for key, fn, i, j in zip(self.args_keys, self.trsfms, bounds[:-1], bounds[1:])}
