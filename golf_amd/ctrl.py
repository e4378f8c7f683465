"""Control-split protocol of the GOLF decoder (reference models/ctrl.py:32-69).

Every DSP module carries a ``.ctrl`` closure that says how many encoder channels it consumes
(``split_size``) and how the raw logits become DSP parameters (``trsfm_fn``).  A ``Synth`` folds the
closures of its ``Controllable`` children, in attribute-assignment order, into
``(split_sizes, trsfm_fns, arg_keys)`` which sizes the encoder head (343 channels for GOLF-ss/ff,
SURVEY.md §8a-15).  Same continuation-passing contract as the reference so that reference modules
and these modules can be mixed inside one decoder.
"""
from __future__ import annotations

from typing import Callable, Tuple

import torch

from .audiotensor import AudioTensor

__all__ = ["Controllable", "PassThrough", "Synth", "wrap_ctrl_fn", "default_ctrl_fn", "DUMMY_SPLIT_TRSFM"]

TRSFM_TYPE = Callable[..., Tuple[AudioTensor, ...]]


def DUMMY_SPLIT_TRSFM(split_sizes, trsfm_fns):
    """Terminal continuation: returns what it is given."""
    return split_sizes, trsfm_fns


def wrap_ctrl_fn(split_size: Tuple[int, ...] = (), trsfm_fn: TRSFM_TYPE = lambda *x: ()):
    """Build a ``.ctrl`` closure: appends (split_size, trsfm_fn) and defers to the continuation."""

    def ctrl_fn(next_fn):
        def split_and_trsfm(split_sizes, trsfm_fns):
            return next_fn(tuple(split_sizes) + (split_size,), tuple(trsfm_fns) + (trsfm_fn,))

        return split_and_trsfm

    return ctrl_fn


def default_ctrl_fn(next_fn):
    """A module that consumes no channels (reference models/ctrl.py:20-29)."""
    return wrap_ctrl_fn()(next_fn)


class Controllable(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.ctrl = wrap_ctrl_fn()


class PassThrough(Controllable):
    def forward(self, x: AudioTensor, *args, **kwargs) -> AudioTensor:
        return x


class Synth(torch.nn.Module):
    @property
    def split_sizes_and_trsfms(self):
        children = [(name, m) for name, m in self.named_children() if isinstance(m, Controllable)]
        fn = DUMMY_SPLIT_TRSFM
        for _, m in reversed(children):  # innermost continuation = last child
            fn = m.ctrl(fn)
        split_sizes, trsfm_fns = fn((), ())
        keys = tuple(name + "_params" for name, _ in children)
        return split_sizes, trsfm_fns, keys
