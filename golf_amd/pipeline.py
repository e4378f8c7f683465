"""Replay pipeline: the execution mode of a synthesis serving loop on MI355X.

One batch of 32 utterances cannot fill the chip — the serial phases of the LPC filter keep a few dozen waves busy for
tens of microseconds — so a serving loop keeps several independent batches in flight: ``n_slots`` copies of the step
are captured as hipGraphs (one launch instead of ~8 kernel launches plus allocator traffic each) and replayed
round-robin on ``n_slots`` HIP streams.  Each slot owns static input tensors (refill them with ``slot.load(...)``) and a
static output (valid until the slot comes round again).

Two facts of the ROCm runtime are built in: streams are created AFTER capture (streams map round-robin onto a few
hardware queues; streams that alias a queue serialise — measured 131 vs 100 us/step), and 4 slots (or 8) is what the
default 4 hardware queues reward (2/3/5/6 slots: 111/90/99/91 us/step against 80).
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional

import torch

__all__ = ["ReplayPipeline", "Slot"]


class Slot:
    """One captured copy of the step: static inputs, static output, its graph and (after ``ReplayPipeline.start``) stream."""

    def __init__(self, inputs: Dict[str, torch.Tensor]):
        self.inputs = inputs
        self.output: Optional[torch.Tensor] = None
        self.graph: Optional[torch.cuda.CUDAGraph] = None
        self.stream: Optional[torch.cuda.Stream] = None

    def load(self, **tensors: torch.Tensor) -> None:
        """Copy new data into the static inputs, ordered on this slot's stream (after its previous replay)."""
        with torch.cuda.stream(self.stream):
            for name, value in tensors.items():
                self.inputs[name].copy_(value, non_blocking=True)


class ReplayPipeline:
    """``fn(inputs) -> tensor`` captured ``n_slots`` times; ``submit()`` replays the next slot and returns it.

    ``make_inputs()`` must return a fresh dict of device tensors per call (each slot needs its own static inputs);
    ``fn`` must be capturable (no host syncs: every op of this package qualifies) and, with ``check=True``, replay is
    asserted bit-identical to an eager call on the same inputs.
    """

    def __init__(self, fn: Callable[[Dict[str, torch.Tensor]], torch.Tensor],
                 make_inputs: Callable[[], Dict[str, torch.Tensor]], n_slots: int = 4, check: bool = True,
                 use_graphs: bool = True):
        self.fn, self.use_graphs = fn, use_graphs
        self.slots: List[Slot] = [Slot(make_inputs()) for _ in range(max(1, n_slots))]
        # several batches in flight: ask the sample-wise filter for the launch chain that costs the least chip time
        # (GOLF_SS_THROUGHPUT, include/golf_amd.h; bit-identical results).  The flag is read when a step is issued or
        # captured, so it is set around the capture / the eager submits of THIS pipeline only.
        self.throughput = len(self.slots) > 1
        from . import functional as _GF

        prev_mode, _GF.THROUGHPUT_MODE = _GF.THROUGHPUT_MODE, self.throughput
        try:
            self._capture(fn, check)
        finally:
            _GF.THROUGHPUT_MODE = prev_mode

    def _capture(self, fn, check: bool) -> None:
        use_graphs = self.use_graphs
        self._next = 0
        device = next(t for t in self.slots[0].inputs.values() if isinstance(t, torch.Tensor)).device
        if use_graphs:
            for slot in self.slots:
                warm = torch.cuda.Stream(device=device)       # warm-up off the capture stream (allocator, lazy init)
                warm.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(warm):
                    for _ in range(2):
                        fn(slot.inputs)
                torch.cuda.current_stream().wait_stream(warm)
                slot.graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(slot.graph):
                    slot.output = fn(slot.inputs)
            torch.cuda.synchronize()
            if check:
                ref = fn(self.slots[0].inputs)
                self.slots[0].graph.replay()
                torch.cuda.synchronize()
                assert torch.equal(self.slots[0].output, ref), "hipGraph replay differs from eager execution"
        for slot in self.slots:                                 # streams AFTER capture (see module docstring)
            slot.stream = torch.cuda.Stream(device=device)

    def submit(self) -> Slot:
        """Launch the next slot's step on its stream (asynchronous) and return the slot; its ``output`` is complete once
        ``slot.stream`` has been waited on / synchronised, and stays valid until ``n_slots`` further submits."""
        slot = self.slots[self._next]
        self._next = (self._next + 1) % len(self.slots)
        with torch.cuda.stream(slot.stream):
            if self.use_graphs:
                slot.graph.replay()
            else:
                from . import functional as _GF

                prev_mode, _GF.THROUGHPUT_MODE = _GF.THROUGHPUT_MODE, self.throughput
                try:
                    slot.output = self.fn(slot.inputs)
                finally:
                    _GF.THROUGHPUT_MODE = prev_mode
        return slot

    def synchronize(self) -> None:
        for slot in self.slots:
            slot.stream.synchronize()
