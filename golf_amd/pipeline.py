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

    def _order_after_producer(self, sources, producer) -> None:
        """The copies run on this slot's stream; the data they read was produced elsewhere.  ``producer``: the stream (or an
        event recorded on it) behind which the batch is complete -- default: the caller's current stream, which is where
        ``pipe.pack``, a ``torch.full`` or an encoder's forward enqueue their writes.  Device-side sources are also handed to
        the caching allocator as in use on this stream, so that a batch freed by the caller right after ``submit`` is not
        recycled under the copy."""
        if producer is None:
            producer = torch.cuda.current_stream(self.stream.device)
        if isinstance(producer, torch.cuda.Event):
            self.stream.wait_event(producer)
        elif producer != self.stream:
            self.stream.wait_stream(producer)
        for t in sources:
            if isinstance(t, torch.Tensor) and t.is_cuda:
                t.record_stream(self.stream)

    def load(self, _producer=None, **tensors: torch.Tensor) -> None:
        """Copy new data into the static inputs, ordered on this slot's stream after its previous replay AND after the
        producer of the data (see ``_order_after_producer``)."""
        self._order_after_producer(tensors.values(), _producer)
        with torch.cuda.stream(self.stream):
            for name, value in tensors.items():
                dst = self.inputs[name]
                if tuple(value.shape) != tuple(dst.shape):
                    raise ValueError(f"input {name!r}: batch has shape {tuple(value.shape)}, the slot's static input {tuple(dst.shape)}")
                dst.copy_(value, non_blocking=True)

    # ---- packed inputs: one flat fp32 buffer, the tensors of ``inputs`` are views of it ------------------------------
    flat: Optional[torch.Tensor] = None
    layout: Optional[Dict[str, tuple]] = None     # name -> (offset in floats, shape)

    def pack(self) -> None:
        """Re-home every fp32 tensor of ``inputs`` in one flat buffer (64-float = 256-byte aligned pieces; call before the
        capture).  Other entries (ints, tensors of other types) stay as they are."""
        names = [k for k, v in self.inputs.items() if isinstance(v, torch.Tensor) and v.dtype == torch.float32]
        if not names:
            raise ValueError("packed=True needs at least one fp32 tensor among the slot's inputs")
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
        # tensor inputs the flat buffer does NOT cover (other dtypes): submit(flat) leaves them as they are -- they have to be
        # refreshed through the dict form; ReplayPipeline.submit refuses a flat batch for a slot that has any unless told so
        self.unpacked = [k for k, v in self.inputs.items() if isinstance(v, torch.Tensor) and k not in layout]

    unpacked: List[str] = []

    def load_flat(self, flat_src: torch.Tensor, _producer=None) -> None:
        """One copy refreshes every packed input (ordered on this slot's stream, after the producer of ``flat_src``);
        ``flat_src`` has this slot's layout."""
        if flat_src.shape != self.flat.shape or flat_src.dtype != self.flat.dtype:
            raise ValueError(f"flat batch of {tuple(flat_src.shape)} {flat_src.dtype}, the slot's buffer is {tuple(self.flat.shape)} {self.flat.dtype}")
        self._order_after_producer((flat_src,), _producer)
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
                 use_graphs: bool = True, packed: bool = False, throughput: Optional[bool] = None,
                 issue: str = "round-robin"):
        """``issue``: "round-robin" (default) or "idle-first" -- ``submit`` then takes the first slot whose previous step has
        finished (work-conserving: what slots of unequal speed need; the slot's output stays valid until THAT slot is taken again)."""
        assert issue in ("round-robin", "idle-first")
        self.fn, self.use_graphs, self.packed, self.issue = fn, use_graphs, packed, issue
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

    def pack(self, batch: Dict[str, torch.Tensor], out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """A batch laid out like a packed slot (one flat fp32 tensor on the slots' device): what ``submit`` takes for the
        single-copy refresh.  ``out``: a flat buffer to reuse (``pipe.new_flat()`` once, then no allocation and no fill per
        batch; the alignment gaps between the pieces are never read).  A producer on the same GPU can instead write into
        ``views(flat)`` of a buffer it owns."""
        lay = self.slots[0].layout
        assert lay is not None, "ReplayPipeline(packed=True) packs its slots"
        flat = self.new_flat() if out is None else out
        if flat.shape != self.slots[0].flat.shape or flat.dtype != torch.float32:
            raise ValueError("out= must come from new_flat()")
        for k, view in self.views(flat).items():
            view.copy_(batch[k])
        return flat

    def new_flat(self) -> torch.Tensor:
        """An uninitialised flat buffer with the slots' layout (for ``pack(out=)`` / ``views``)."""
        assert self.slots[0].layout is not None, "ReplayPipeline(packed=True) packs its slots"
        return torch.empty_like(self.slots[0].flat)

    def views(self, flat: torch.Tensor) -> Dict[str, torch.Tensor]:
        """Named views of a flat buffer with the slots' layout."""
        out = {}
        for k, (o, shp) in self.slots[0].layout.items():
            n = 1
            for d in shp:
                n *= d
            out[k] = flat[o:o + n].view(shp)
        return out

    def submit(self, batch=None, producer=None, partial_ok: bool = False) -> Slot:
        """Launch the next slot's step on its stream (asynchronous) and return the slot; its ``output`` is complete once
        ``slot.stream`` has been waited on / synchronised, and stays valid until ``n_slots`` further submits.

        ``batch``: None replays the slot on whatever its static inputs hold; a dict of tensors (device or pinned host) is
        copied into them first, tensor by tensor; a flat tensor (``pack``) refreshes a packed slot with one copy.  The copies
        are issued on the slot's stream: ordered after the slot's previous replay (which read the old values) and before the
        new one, overlapping the other slots' work -- and after ``producer``, the stream (or an event on it) that wrote the
        batch; default: the caller's current stream (without that edge the copy could read a batch still being
        written).  Device tensors of the batch are marked as in use on the slot's stream (``record_stream``), so the caller may
        drop them right after the call; pinned host tensors must stay alive until the slot's stream has passed the copy.
        A flat batch for a slot that also has tensor inputs outside the flat buffer (non-fp32) raises unless ``partial_ok``:
        those would silently keep their old values."""
        if self.issue == "idle-first":
            n, i = len(self.slots), self._next
            while getattr(self.slots[i], "done", None) is not None and not self.slots[i].done.query():
                i = (i + 1) % n
            self._next = i
        slot = self.slots[self._next]
        self._next = (self._next + 1) % len(self.slots)
        if batch is not None:
            if isinstance(batch, torch.Tensor):
                if slot.unpacked and not partial_ok:
                    raise ValueError(f"a flat batch does not refresh the slot's non-fp32 inputs {slot.unpacked}: "
                                     "submit them as a dict, or pass partial_ok=True")
                slot.load_flat(batch, producer)
            else:
                slot.load(producer, **{k: v for k, v in batch.items() if isinstance(v, torch.Tensor) and k in slot.inputs})
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
        if self.issue == "idle-first":
            if getattr(slot, "done", None) is None:
                slot.done = torch.cuda.Event()
            slot.done.record(slot.stream)
        return slot

    def synchronize(self) -> None:
        for slot in self.slots:
            slot.stream.synchronize()
