"""On-disk format compatibility: the checkpoint key / head-row remaps between the reference's model
generations.  Pure state_dict logic (no kernels); pinned by tests/golden/g24 against the reference's own function.

The ISMIR'23 models lay the encoder head out as [..., voice LPC, voice gain, noise LPC, noise gain, table features];
the Interspeech'24 control protocol (golf_amd/ctrl.py, models/ctrl.py:59-69) orders the same channels as
[..., table features, voice gain, voice LPC, noise gain, noise LPC].  Loading an old checkpoint therefore means
permuting the trailing rows of every ``out_linear`` tensor (reference models/utils.py:12-38, test_rtf.py:35-132,
convert2v2.py)."""
from __future__ import annotations

from typing import Dict, Mapping, Optional, Sequence

import torch

__all__ = ["permute_head_rows", "ismir2interspeech_ckpt", "convert_ismir_state_dict"]


def permute_head_rows(state_dict: Mapping[str, torch.Tensor], old_sizes: Sequence[int], new_order: Sequence[int],
                      match: str = "out_linear") -> Dict[str, torch.Tensor]:
    """Split the LAST ``sum(old_sizes)`` rows of every tensor whose key contains ``match`` into groups of ``old_sizes``
    and write them back in ``new_order`` (indices into the old groups); everything else is passed through."""
    total = sum(old_sizes)
    out = {}
    for key, value in state_dict.items():
        if match in key:
            head, tail = value[:-total], value[-total:]
            groups = torch.split(tail, list(old_sizes), dim=0)
            value = torch.cat([head] + [groups[i] for i in new_order], dim=0)
        out[key] = value
    return out


def ismir2interspeech_ckpt(ckpt: Mapping[str, torch.Tensor], lpc_order: int, h_size: int) -> Dict[str, torch.Tensor]:
    """GOLF (ISMIR'23) -> Interspeech'24 head layout; same signature as the reference's models/utils.py:12."""
    # old: voice_lpc, voice_gain, noise_lpc, noise_gain, h   ->   new: h, voice_gain, voice_lpc, noise_gain, noise_lpc
    return permute_head_rows(ckpt, [lpc_order, 1, lpc_order, 1, h_size], [4, 1, 0, 3, 2])


def _mentions(tree, needle: str) -> bool:
    if isinstance(tree, dict):
        return any(needle in str(k) or _mentions(v, needle) for k, v in tree.items())
    if isinstance(tree, (list, tuple)):
        return any(_mentions(v, needle) for v in tree)
    return isinstance(tree, str) and needle in tree


def convert_ismir_state_dict(state_dict: Mapping[str, torch.Tensor], model_configs: Optional[dict]) -> Dict[str, torch.Tensor]:
    """What the reference's RTF script does to an ISMIR'23 checkpoint before loading it (test_rtf.py:98-132): drop the
    non-persistent OLA kernels that old checkpoints still carry (``*_kernel``), repair the ``amplicudes`` typo, and
    permute the head rows -- the GOLF layout when the config holds a DownsampledIndexedGlottalFlowTable, the PULF layout
    (voice LPC, voice gain, noise LPC, noise gain -> gains first) when it holds an AdditivePulseTrain in front of
    LTVMinimumPhaseFilters, nothing otherwise."""
    sd = {k.replace("amplicudes", "amplitudes"): v for k, v in state_dict.items() if not k.endswith("_kernel")}
    if model_configs is None:
        return sd
    init = lambda name: model_configs["decoder"]["init_args"][name]["init_args"]  # noqa: E731
    if _mentions(model_configs, "DownsampledIndexedGlottalFlowTable"):
        return ismir2interspeech_ckpt(sd, init("harm_filter")["lpc_order"], init("harm_oscillator")["in_channels"])
    if _mentions(model_configs, "AdditivePulseTrain") and _mentions(model_configs, "LTVMinimumPhaseFilter"):
        return permute_head_rows(sd, [init("harm_filter")["lpc_order"], 1, init("noise_filter")["lpc_order"], 1],
                                 [1, 0, 3, 2])
    return sd
