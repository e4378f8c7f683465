"""Instantiate GOLF configs on this package's classes (the reference's plugin surface is a dotted
``class_path`` + ``init_args`` tree in YAML, resolved by jsonargparse / LightningCLI — cfg/ae/decoder/*.yaml,
ckpts/*/*/config.yaml).  The same trees resolve here: ``models.*`` / ``loss.spec`` / ``ltng.ae`` paths are mapped to
``golf_amd``; everything else (``torch.optim.Adam`` …) is imported as written.  Trainer / data / logger sections are
Lightning control plane and are ignored.
"""
from __future__ import annotations

from importlib import import_module
from typing import Any, Mapping, Union

__all__ = ["resolve_class", "instantiate", "build_model", "load_yaml"]

_ALIASES = (("models.audiotensor", "golf_amd.audiotensor"), ("models.", "golf_amd."), ("loss.spec", "golf_amd.loss"),
            ("ltng.ae", "golf_amd.ae"))


def resolve_class(path: str):
    """``models.filters.LTVMinimumPhaseFilterPrecise`` -> ``golf_amd.filters.LTVMinimumPhaseFilterPrecise``."""
    module_path, name = path.rsplit(".", 1)
    for old, new in _ALIASES:
        if module_path == old.rstrip(".") or module_path.startswith(old if old.endswith(".") else old + "."):
            module_path = new.rstrip(".") + module_path[len(old.rstrip(".")):]
            break
    try:
        return getattr(import_module(module_path), name)
    except (ImportError, AttributeError) as e:
        raise NotImplementedError(f"golf_amd: no counterpart of {path} ({e})") from e


def instantiate(node: Any) -> Any:
    """Recursively turn ``{class_path, init_args}`` nodes into objects; lists and plain dicts are walked, scalars kept."""
    if isinstance(node, Mapping):
        if "class_path" in node:
            kwargs = {k: instantiate(v) for k, v in (node.get("init_args") or {}).items()}
            return resolve_class(node["class_path"])(**kwargs)
        return {k: instantiate(v) for k, v in node.items()}
    if isinstance(node, (list, tuple)):
        return [instantiate(v) for v in node]
    return node


def _interpolate(root: Any) -> Any:
    """``${a.b.c}`` values refer to other nodes of the same file (the reference parses its YAML in omegaconf mode:
    cfg/ae/decoder/golf.yaml:29 takes the end filter's window from the noise filter's)."""
    import re

    ref = re.compile(r"^\$\{([\w.]+)\}$")

    def lookup(dotted: str):
        node = root
        for key in dotted.split("."):
            node = node[int(key)] if isinstance(node, list) else node[key]
        return resolve(node)

    def resolve(node):
        if isinstance(node, str):
            m = ref.match(node)
            return lookup(m.group(1)) if m else node
        if isinstance(node, dict):
            return {k: resolve(v) for k, v in node.items()}
        if isinstance(node, list):
            return [resolve(v) for v in node]
        return node

    return resolve(root)


def load_yaml(path_or_text: str) -> dict:
    import os

    import yaml

    if "\n" not in path_or_text:
        if not os.path.exists(path_or_text):
            raise FileNotFoundError(f"golf_amd.config: no such config file: {path_or_text}")
        with open(path_or_text) as f:
            return _interpolate(yaml.safe_load(f))
    return _interpolate(yaml.safe_load(path_or_text))


def build_model(config: Union[str, Mapping]):
    """The ``model:`` section of a LightningCLI config (file path, YAML text or parsed dict) -> module.  A bare decoder
    file (cfg/ae/decoder/*.yaml: top-level ``decoder:``) yields the decoder."""
    cfg = load_yaml(config) if isinstance(config, str) else config
    if "model" in cfg:
        model = cfg["model"]
        if "class_path" not in model and "decoder" in model:   # ckpts/ismir23/*: a Lightning module of ltng/vocoder.py
            return instantiate(model["decoder"])                # (control plane) around the decoder: build the decoder
        return instantiate(model)
    if "decoder" in cfg:
        return instantiate(cfg["decoder"])
    return instantiate(cfg)
