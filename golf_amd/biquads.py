"""Counterpart of the reference's ``biquads.py``: recover, from an ISMIR'23 GOLF encoder, the
per-frame second-order sections of the harmonic and noise filters together with gains, voicing and the table selection
weight, and write them in the reference's ``.pt`` dictionary format.

The arithmetic is ``utils.get_logits2biquads("coef")`` (reference models/utils.py:487-525, default pole bound 0.99 as
biquads.py:10); what is specific here is the slicing of the encoder logits by the control protocol (biquads.py:13-58)
and the key scheme of the dump (biquads.py:80-103).  Host logic only; pinned by tests/golden/g24."""
from __future__ import annotations

from typing import Dict, Iterable, Optional, Tuple

import torch

from .utils import get_logits2biquads

__all__ = ["get_biquads", "biquad_records", "dump_biquads"]

_logits2biquads = get_logits2biquads("coef")


@torch.no_grad()
def get_biquads(logits: torch.Tensor, enc) -> Tuple[torch.Tensor, ...]:
    """``logits`` (B, F, C): the encoder backbone's output; ``enc``: anything with ``split_sizes`` / ``args_keys`` /
    ``trsfms`` (a VocoderParameterEncoderInterface).  Returns (voicing, harm_log_gain, harm_biquads (B,F,K,3),
    noise_log_gain, noise_biquads[, table_select_weight]) exactly as the reference's get_biquads does."""
    widths = [sum(group) for group in enc.split_sizes]
    keys = list(enc.args_keys)

    def piece(key: str) -> Optional[torch.Tensor]:
        i = keys.index(key)
        start = sum(widths[:i])
        return logits[..., start:start + widths[i]] if widths[i] else None

    def sections(key: str):
        block = piece(key)
        log_gain = block[..., 0]
        return log_gain, _logits2biquads(block[..., 1:].reshape(*logits.shape[:-1], -1, 2))

    harm_gain, harm_bq = sections("harm_filter_params")
    noise_gain, noise_bq = sections("noise_filter_params")
    voicing = torch.sigmoid(logits[..., 1])          # the voicing head sits right after the f0 head (models/enc.py:48-64)
    result = (voicing, harm_gain, harm_bq, noise_gain, noise_bq)
    osc = piece("harm_oscillator_params")
    if osc is not None:
        i = keys.index("harm_oscillator_params")
        params = enc.trsfms[i](*torch.split(osc, list(enc.split_sizes[i]), dim=-1))
        if len(params):
            first = params[0]
            result = result + (first.as_tensor() if hasattr(first, "as_tensor") else first,)
    return result


def biquad_records(stem: str, index: int, values: Tuple[torch.Tensor, ...]) -> Dict[str, torch.Tensor]:
    """One chunk's entries of the dump, keyed ``{stem}_{index}.{name}`` (biquads.py:88-101)."""
    names = ("voicing", "harm_log_gain", "harm_biquads", "noise_log_gain", "noise_biquads", "table_select_weight")
    return {f"{stem}_{index}.{name}": value for name, value in zip(names, values)}


def dump_biquads(chunks: Iterable[Tuple[str, int, Tuple[torch.Tensor, ...]]], outfile: str) -> Dict[str, torch.Tensor]:
    """Write (stem, chunk index, get_biquads(...) result) triples as one ``torch.save`` dictionary."""
    out: Dict[str, torch.Tensor] = {}
    for stem, index, values in chunks:
        out.update(biquad_records(stem, index, values))
    torch.save(out, outfile)
    return out
