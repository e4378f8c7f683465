"""Replay pipeline: the execution mode of a synthesis serving loop on MI355X.

One batch of 32 utterances cannot fill the chip — the serial phases of the LPC filter keep a few dozen waves busy for
tens of microseconds — so a serving loop keeps several independent batches in flight: ``n_slots`` copies of the step
are captured as hipGraphs (one launch instead of ~8 kernel launches plus allocator traffic each) and replayed
round-robin on ``n_slots`` HIP streams.  Each slot owns static input tensors and a static output (valid until the slot
comes round again).  A serving loop hands every batch to ``submit(batch)``: the batch is copied into the slot's static
inputs ON THE SLOT'S STREAM (ordered after the slot's previous replay, in flight beside the other slots' compute) and the
graph is replayed behind it.  With ``packed=True`` a slot's fp32 inputs are views of ONE flat buffer, so a batch that
arrives packed the same way (``pipe.pack(batch)``, or an encoder writing into ``pipe.views(flat)`` directly) is
refreshed by a single device-to-device copy instead of one per tensor.

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

    # ---- packed inputs: one flat fp32 buffer, the tensors of ``inputs`` are views of it ------------------------------
    flat: Optional[torch.Tensor] = None
    layout: Optional[Dict[str, tuple]] = None     # name -> (offset in floats, shape)

    def pack(self) -> None:
        """Re-home every fp32 tensor of ``inputs`` in one flat buffer (64-float = 256-byte aligned pieces; call before the
        capture).  Other entries (ints, tensors of other types) stay as they are."""
        names = [k for k, v in self.inputs.items() if isinstance(v, torch.Tensor) and v.dtype == torch.float32]
        off, layout = 0, {}
        for k in names:
            layout[k] = (off, tuple(self.inputs[k].shape))
            off += (self.inputs[k].numel() + 63) // 64 * 64
        dev = self.inputs[names[0]].device
        flat = torch.zeros(off, dtype=torch.float32, device=dev)
        for k in names:
            o, shp = layout[k]
            view = flat[o:o + self.inputs[k].numel()].view(shp)
            view.copy_(self.inputs[k])
            self.inputs[k] = view
        self.flat, self.layout = flat, layout

    def load_flat(self, flat_src: torch.Tensor) -> None:
        """One copy refreshes every packed input (ordered on this slot's stream); ``flat_src`` has this slot's layout."""
        with torch.cuda.stream(self.stream):
            # (a copy kernel of our own -- 16 bytes per lane, a grid over the whole chip -- was measured against this
            #  hipMemcpyDtoDAsync with four batches in flight: 82.7 vs 78.9 us/step; the runtime's copy stays)
            self.flat.copy_(flat_src, non_blocking=True)


class ReplayPipeline:
    """``fn(inputs) -> tensor`` captured ``n_slots`` times; ``submit()`` replays the next slot and returns it.

    ``make_inputs()`` must return a fresh dict of device tensors per call (each slot needs its own static inputs);
    ``fn`` must be capturable (no host syncs: every op of this package qualifies) and, with ``check=True``, replay is
    asserted bit-identical to an eager call on the same inputs.
    """

    def __init__(self, fn: Callable[[Dict[str, torch.Tensor]], torch.Tensor],
                 make_inputs: Callable[[], Dict[str, torch.Tensor]], n_slots: int = 4, check: bool = True,
                 use_graphs: bool = True, packed: bool = False, throughput: Optional[bool] = None):
        self.fn, self.use_graphs, self.packed = fn, use_graphs, packed
        self.slots: List[Slot] = [Slot(make_inputs()) for _ in range(max(1, n_slots))]
        if packed:
            for slot in self.slots:
                slot.pack()
        # several batches in flight: ask the sample-wise filter for the launch chain that costs the least chip time
        # (GOLF_SS_THROUGHPUT, include/golf_amd.h; bit-identical results).  The flag is read when a step is issued or
        # captured, so it is set around the capture / the eager submits of THIS pipeline only.
        # (``throughput=False`` keeps the lone-batch chain whatever the slot count: a diagnostic -- tools/soak.py runs four of
        #  the one-launch chunk-pass kernels concurrently that way)
        self.throughput = len(self.slots) > 1 if throughput is None else bool(throughput)
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

    def pack(self, batch: Dict[str, torch.Tensor]) -> torch.Tensor:
        """A batch laid out like a packed slot (one flat fp32 tensor on the slots' device): what ``submit`` takes for the
        single-copy refresh.  A producer on the same GPU can instead write into ``views(flat)`` of a buffer it owns."""
        lay = self.slots[0].layout
        assert lay is not None, "ReplayPipeline(packed=True) packs its slots"
        flat = torch.zeros_like(self.slots[0].flat)
        for k, view in self.views(flat).items():
            view.copy_(batch[k])
        return flat

    def views(self, flat: torch.Tensor) -> Dict[str, torch.Tensor]:
        """Named views of a flat buffer with the slots' layout."""
        out = {}
        for k, (o, shp) in self.slots[0].layout.items():
            n = 1
            for d in shp:
                n *= d
            out[k] = flat[o:o + n].view(shp)
        return out

    def submit(self, batch=None) -> Slot:
        """Launch the next slot's step on its stream (asynchronous) and return the slot; its ``output`` is complete once
        ``slot.stream`` has been waited on / synchronised, and stays valid until ``n_slots`` further submits.

        ``batch``: None replays the slot on whatever its static inputs hold; a dict of tensors (device or pinned host) is
        copied into them first, tensor by tensor; a flat tensor (``pack``) refreshes a packed slot with one copy.  The copies
        are issued on the slot's stream: ordered after the slot's previous replay (which read the old values) and before the
        new one, overlapping the other slots' work.  The caller keeps ``batch`` alive until the slot's stream has passed the
        copy (e.g. until the slot's output is consumed)."""
        slot = self.slots[self._next]
        self._next = (self._next + 1) % len(self.slots)
        if batch is not None:
            if isinstance(batch, torch.Tensor):
                slot.load_flat(batch)
            else:
                slot.load(**{k: v for k, v in batch.items() if isinstance(v, torch.Tensor) and k in slot.inputs})
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
