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


class _CtrlStep:
    """One link of the fold.  ``step(next_fn)`` is a function of the (split sizes, transforms) gathered so far that
    appends this module's pair and hands the longer tuples to ``next_fn`` — the calling convention the reference's
    modules use, so links of both code bases chain."""

    __slots__ = ("split_size", "trsfm_fn")

    def __init__(self, split_size: Tuple[int, ...], trsfm_fn: TRSFM_TYPE):
        self.split_size = tuple(split_size)
        self.trsfm_fn = trsfm_fn

    def __call__(self, next_fn):
        mine = (self.split_size, self.trsfm_fn)
        return lambda sizes, fns: next_fn((*sizes, mine[0]), (*fns, mine[1]))


def _no_params(*_):
    return ()


def wrap_ctrl_fn(split_size: Tuple[int, ...] = (), trsfm_fn: TRSFM_TYPE = _no_params):
    """The ``.ctrl`` attribute of a module that consumes ``split_size`` encoder channels (one tensor per entry) and
    turns them into its parameters with ``trsfm_fn``."""
    return _CtrlStep(split_size, trsfm_fn)


def default_ctrl_fn(next_fn):
    """A module that consumes no channels (reference models/ctrl.py:20-29)."""
    return _CtrlStep((), _no_params)(next_fn)


class Controllable(torch.nn.Module):
    """Base of every DSP module: consumes nothing unless it overrides ``ctrl``."""

    def __init__(self):
        super().__init__()
        self.ctrl = _CtrlStep((), _no_params)


class PassThrough(Controllable):
    def forward(self, x: AudioTensor, *args, **kwargs) -> AudioTensor:
        return x


class Synth(torch.nn.Module):
    @property
    def split_sizes_and_trsfms(self):
        """(split_sizes, trsfm_fns, arg_keys) of the Controllable children, in attribute-assignment order."""
        links = [(name, module.ctrl) for name, module in self.named_children() if isinstance(module, Controllable)]
        chain = DUMMY_SPLIT_TRSFM
        for _, link in links[::-1]:   # built back to front: the first child's link runs first
            chain = link(chain)
        sizes, fns = chain((), ())
        return sizes, fns, tuple(f"{name}_params" for name, _ in links)
