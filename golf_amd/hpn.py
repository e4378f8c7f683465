"""Dotted-path twin of the reference's models/hpn.py (``class_path: golf_amd.hpn.HarmonicPlusNoiseSynth``)."""
from .sf import HarmonicPlusNoiseSynth

__all__ = ["HarmonicPlusNoiseSynth"]
