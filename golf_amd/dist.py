"""Multi-GPU: the hot path shards over utterances (every utterance is independent in every kernel —
the recursion couples only along t), one process per GPU, no data-path collective.  The only exchange
is the all-gather of the synthesised audio (north_star / BASELINE configs[3]): 6.1 MB per rank at
B=32, one RCCL all_gather_into_tensor over xGMI.  Backend-agnostic so the same code runs under gloo
(CPU, tests) and nccl (= RCCL on ROCm).
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch

__all__ = ["shard_bounds", "shard_inputs", "gather_audio", "gather_audio_async", "synth_sharded", "StagedGather",
           "PeerStoreGather"]


def shard_bounds(n: int, rank: int, world: int) -> Tuple[int, int, int]:
    """Contiguous equal shards of ceil(n/world) rows (the tail shard is padded by the caller).
    Returns (start, stop, per_rank)."""
    per = (n + world - 1) // world
    start = min(rank * per, n)
    return start, min(start + per, n), per


def shard_inputs(inputs: Dict, rank: int, world: int) -> Dict:
    """Slice every batch-major tensor of an input dict to this rank's utterances.  A ragged tail is padded
    by repeating the last utterance so that all ranks run the same shapes (trimmed again by gather_audio)."""
    tensors = [v for v in inputs.values() if isinstance(v, torch.Tensor)]
    n = tensors[0].shape[0]
    start, stop, per = shard_bounds(n, rank, world)
    out = {}
    for k, v in inputs.items():
        if isinstance(v, torch.Tensor) and v.shape[0] == n:
            s = v[start:stop]
            if s.shape[0] < per:
                filler = (s[-1:] if s.shape[0] else v[-1:]).expand(per - s.shape[0], *v.shape[1:])
                s = torch.cat([s, filler], 0)
            out[k] = s
        else:
            out[k] = v
    return out


def gather_audio(y: torch.Tensor, out: Optional[torch.Tensor] = None, total: Optional[int] = None) -> torch.Tensor:
    """All-gather per-rank audio (B_local, T) into (world*B_local, T), trimmed to ``total`` rows."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return y if total is None else y[:total]
    world = dist.get_world_size()
    y = y.contiguous()
    if out is None:
        out = torch.empty((world * y.shape[0],) + tuple(y.shape[1:]), dtype=y.dtype, device=y.device)
    try:
        dist.all_gather_into_tensor(out, y)
    except (RuntimeError, NotImplementedError):  # backends without the flat variant
        dist.all_gather(list(out.chunk(world, 0)), y)
    return out if total is None else out[:total]


def gather_audio_async(y: torch.Tensor, out: torch.Tensor):
    """Start the all-gather of ``y`` into ``out`` without blocking the compute stream and return a handle with
    ``.wait()`` (None when there is nothing to do).  A serving loop issues it after step k and waits before reusing
    ``out`` (double buffering), so the 6 MB/rank exchange over xGMI hides behind step k+1's kernels — these are
    latency-bound and leave most CUs idle for RCCL's channels."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return None
    y = y.contiguous()
    try:
        return dist.all_gather_into_tensor(out, y, async_op=True)
    except (RuntimeError, NotImplementedError):
        return dist.all_gather(list(out.chunk(dist.get_world_size(), 0)), y, async_op=True)


def synth_sharded(synth_fn, inputs: Dict, gather: bool = True) -> torch.Tensor:
    """Run ``synth_fn(local_inputs) -> (B_local, T)`` on this rank's shard and (optionally) gather."""
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized():
        rank, world = dist.get_rank(), dist.get_world_size()
    else:
        rank, world = 0, 1
    tensors = [v for v in inputs.values() if isinstance(v, torch.Tensor)]
    total = tensors[0].shape[0]
    y = synth_fn(shard_inputs(inputs, rank, world))
    return gather_audio(y, total=total) if gather else y


class StagedGather:
    """Exchange the audio of ``every`` consecutive steps in ONE all-gather.

    Why: the synthesis step got fast enough (~75 us at B=32) that a 6.1 MB all-gather per step asks every rank to take
    in (N-1) x 6.1 MB per step -- at 8 GPUs about 0.57 TB/s inbound, the aggregate of the seven xGMI links.  xGMI is
    point to point and a collective's cost has a fixed per-call part (launch, channel setup, the ring/tree latency), so
    fewer, larger messages are what the fabric rewards: ``every`` = 8 turns eight 6 MB collectives into one 49 MB one.
    The price is latency (a step's audio reaches the peers up to ``every`` steps later) and ``every`` x the staging
    memory -- irrelevant next to 288 GB of HBM.

    ``push(y)`` copies a step's (rows, T) output into the staging buffer (device-to-device, on the current stream) and,
    when the buffer is full, starts the collective asynchronously and returns its handle (None otherwise).  Two
    staging/receive buffer pairs alternate so that staging step k+1 never overwrites data still being sent.
    ``result(i)`` views receive buffer i as (world, every, rows, T): [r, k] = what rank r produced at step k of that
    group.  Backend-agnostic (gloo on CPU in the tests, nccl = RCCL on the GPUs)."""

    def __init__(self, rows: int, T: int, every: int, device, dtype=torch.float32, world: Optional[int] = None):
        import torch.distributed as dist

        self.world = world if world is not None else (dist.get_world_size() if dist.is_initialized() else 1)
        self.rows, self.T, self.every = rows, T, max(1, int(every))
        self.stage = [torch.empty(self.every * rows, T, device=device, dtype=dtype) for _ in range(2)]
        self.recv = [torch.empty(self.world * self.every * rows, T, device=device, dtype=dtype) for _ in range(2)]
        self.pending = [None, None]
        self.cur, self.fill, self.groups_sent, self.last_fill = 0, 0, 0, 0

    @property
    def bytes_per_collective(self) -> int:
        return self.every * self.rows * self.T * self.stage[0].element_size()

    def push(self, y: torch.Tensor):
        i = self.cur
        if self.fill == 0 and self.pending[i] is not None:      # this pair's previous collective must be done
            self.pending[i].wait()
            self.pending[i] = None
        self.stage[i][self.fill * self.rows:(self.fill + 1) * self.rows].copy_(y, non_blocking=True)
        self.fill += 1
        if self.fill < self.every:
            return None
        return self._send()

    def _send(self):
        i = self.cur
        n = self.fill * self.rows                      # a partial group (flush) sends only what was staged
        src = self.stage[i][:n]
        if self.fill == self.every:
            dst = self.recv[i]
        else:                                          # rank r's rows land at [r*n, (r+1)*n): a compact prefix of recv
            dst = self.recv[i][: self.world * n]
        self.last_fill = self.fill
        handle = gather_audio_async(src, dst) if self.world > 1 else None
        if self.world == 1:
            dst.copy_(src)
        self.pending[i] = handle
        self.cur, self.fill = 1 - i, 0
        self.groups_sent += 1
        return handle

    def flush(self):
        """Send a partially filled group (rows beyond ``fill`` carry stale data) and wait for everything in flight."""
        if self.fill:
            self._send()
        for i in (0, 1):
            if self.pending[i] is not None:
                self.pending[i].wait()
                self.pending[i] = None

    def result(self, i: int, steps: Optional[int] = None) -> torch.Tensor:
        """(world, steps, rows, T) view of receive buffer ``i``; ``steps`` = how many steps the group held (``every`` for
        a full group, fewer for the partial group a flush sent)."""
        k = self.every if steps is None else steps
        return self.recv[i][: self.world * k * self.rows].view(self.world, k, self.rows, self.T)


class _DeviceView:
    """Raw device memory as a tensor (torch.as_tensor reads __cuda_array_interface__): the buffers of PeerStoreGather
    are allocated / IPC-mapped by libgolf_hip.so, not by torch's allocator."""

    def __init__(self, ptr: int, shape, typestr: str):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2, "strides": None}


_LEAKED_ON_ERROR: list = []   # receive buffers of PeerStoreGather objects left by an exception (see __exit__)


class PeerStoreGather:
    """The audio exchange as peer-to-peer stores over xGMI instead of an all-gather (flag-gated alternative to
    StagedGather / gather_audio; include/golf_amd.h golf_peer_*).

    Every rank owns a receive buffer (depth, world, rows, T) and a flag array, both fine-grained device memory exported
    through HIP IPC and mapped by every other rank of the node.  ``push(y)`` -- on the current stream, no host
    synchronisation --
      1. waits until every peer has released the slot this step is about to overwrite (their acknowledgement flags),
      2. stores the (rows, T) block into slot [step % depth, rank] of EVERY rank's buffer with one kernel (source read
         once; xGMI is point to point, so the world-1 remote stores run on world-1 different links at once),
      3. publishes the step's sequence number into each rank's flag array (system-scope release after the stores).
    ``wait()`` blocks the stream until the oldest outstanding step has arrived from every rank and returns it as a
    (world, rows, T) view of the local buffer; ``release()`` hands the slot back (acknowledgement to every rank).
    ``push`` consumes-and-releases the oldest step by itself when ``depth`` steps are outstanding, so a producer loop
    that never looks at the data (bench.py) cannot overrun; ``flush()`` drains.  Waits are bounded (``timeout_s``): a
    missing peer raises at the next ``check()`` / ``flush()`` instead of hanging the GPU.

    Needs one process per GPU on ONE node, an initialised process group (any backend: it only carries the 64-byte IPC
    handles, once) and HSA_ENABLE_IPC_MODE_LEGACY=0.  Exercised with two processes sharing one GPU
    (tests/test_gpu_peer.py); not yet measured across GPUs."""

    def __init__(self, rows: int, T: int, depth: int = 4, device=None, timeout_s: float = 20.0):
        import ctypes

        import torch.distributed as dist

        from . import _lib

        assert dist.is_initialized(), "PeerStoreGather needs an initialised process group"
        self.lib = _lib.load()
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        assert self.world <= 16, "at most GOLF_MAX_PEERS = 16 ranks"
        self.rows, self.T, self.depth = int(rows), int(T), max(1, int(depth))
        self.device = torch.device(device if device is not None else "cuda")
        self.timeout_us = int(timeout_s * 1e6)
        W, D = self.world, self.depth
        self.slot_elems = self.rows * self.T
        data_bytes = 4 * D * W * self.slot_elems
        # flags: [0, D*W) data-arrived sequence numbers (slot, source); [D*W, 2*D*W) slot-released ones (slot, consumer)
        flag_bytes = 4 * 2 * D * W
        self._local, self._opened, failure, handles = [], [], None, None
        self._data, self._flags = [None] * W, [None] * W
        try:
            with torch.cuda.device(self.device):
                for nbytes in (data_bytes, flag_bytes):
                    p = ctypes.c_void_p()
                    _lib.check(self.lib.golf_peer_alloc(nbytes, ctypes.byref(p)), "golf_peer_alloc")
                    self._local.append(p.value)
                hs = []
                for ptr in self._local:
                    h = ctypes.create_string_buffer(64)
                    _lib.check(self.lib.golf_peer_export(ptr, h), "golf_peer_export")
                    hs.append(h.raw)
                handles = hs
        except Exception as e:   # noqa: BLE001 -- agreed on below
            failure = e
        everyone = [None] * W
        dist.all_gather_object(everyone, handles)   # every rank reaches this, with or without handles
        try:
            if failure is None and any(h is None for h in everyone):
                raise RuntimeError("a peer could not allocate or export its buffers")
            with torch.cuda.device(self.device):
                for r in range(W if failure is None else 0):
                    if r == self.rank:
                        self._data[r], self._flags[r] = self._local
                        continue
                    mapped = []
                    for raw in everyone[r]:
                        p = ctypes.c_void_p()
                        _lib.check(self.lib.golf_peer_open(raw, ctypes.byref(p)), "golf_peer_open")
                        mapped.append(p.value)
                        self._opened.append(p.value)
                    self._data[r], self._flags[r] = mapped
        except Exception as e:   # noqa: BLE001 -- a rank that cannot map a peer must not leave the others in a barrier
            failure = e
        # every rank has mapped every buffer before the first store -- or every rank learns that one could not (the exchange of
        # the 64-byte handles above is a collective all ranks reach; an IPC error after it used to strand the peers in the barrier)
        ok = torch.tensor([0 if failure is not None else 1], dtype=torch.int32,
                          device=self.device if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0:
            for p in self._opened:
                self.lib.golf_peer_close(p)
            for p in self._local:
                self.lib.golf_peer_free(p)
            self._opened, self._local = [], []
            raise RuntimeError(f"PeerStoreGather: rank {self.rank}: " + (repr(failure) if failure is not None
                                                                         else "another rank could not map its peers' buffers"))
        self.recv = torch.as_tensor(_DeviceView(self._local[0], (D, W, self.rows, self.T), "<f4"), device=self.device)
        self.status = torch.zeros(1, dtype=torch.int32, device=self.device)
        self.pushed = 0       # steps pushed so far; step k uses slot k % depth and sequence number k + 1
        self.consumed = 0     # steps waited for
        self.released = 0     # steps released
        self._PtrArray = ctypes.c_void_p * W

    # ---- addresses ----------------------------------------------------------------------------
    def _flag_addr(self, owner: int, kind: int, slot: int, who: int) -> int:
        return self._flags[owner] + 4 * ((kind * self.depth + slot) * self.world + who)

    @property
    def bytes_out_per_step(self) -> int:
        """Bytes this rank stores into OTHER ranks' memory per step."""
        return 4 * self.slot_elems * (self.world - 1)

    # ---- protocol -----------------------------------------------------------------------------
    def push(self, y: torch.Tensor) -> None:
        from . import _lib

        assert y.shape == (self.rows, self.T) and y.dtype == torch.float32 and y.stride(1) == 1
        if self.pushed - self.released >= self.depth:      # the ring is full: consume the oldest step ourselves
            if self.consumed == self.released:
                self.wait()
            self.release()
        k, W = self.pushed, self.world
        slot, stream = k % self.depth, _lib.stream_ptr()
        if k >= self.depth:   # every rank has released what it received in this slot `depth` steps ago
            _lib.check(self.lib.golf_peer_wait_u32(self._flag_addr(self.rank, 1, slot, 0), W, 1, k - self.depth + 1,
                                                   self.timeout_us, self.status.data_ptr(), stream), "golf_peer_wait_u32")
        dst = self._PtrArray(*[self._data[r] + 4 * (slot * W + self.rank) * self.slot_elems for r in range(W)])
        _lib.check(self.lib.golf_peer_store_f32(y.data_ptr(), y.stride(0), self.rows, self.T, dst, self.T, W, stream),
                   "golf_peer_store_f32")
        flags = self._PtrArray(*[self._flag_addr(r, 0, slot, self.rank) for r in range(W)])
        _lib.check(self.lib.golf_peer_signal_u32(flags, W, k + 1, stream), "golf_peer_signal_u32")
        self.pushed += 1

    def wait(self) -> torch.Tensor:
        """Block the current stream until the oldest un-waited step is here from every rank; (world, rows, T) view."""
        from . import _lib

        assert self.consumed < self.pushed, "nothing outstanding"
        k = self.consumed
        slot = k % self.depth
        _lib.check(self.lib.golf_peer_wait_u32(self._flag_addr(self.rank, 0, slot, 0), self.world, 1, k + 1,
                                               self.timeout_us, self.status.data_ptr(), _lib.stream_ptr()),
                   "golf_peer_wait_u32")
        self.consumed += 1
        return self.recv[slot]

    def release(self) -> None:
        """The oldest waited-for step has been read: its slot may be overwritten (acknowledgement to every rank)."""
        from . import _lib

        assert self.released < self.consumed, "release() follows wait()"
        k = self.released
        slot = k % self.depth
        flags = self._PtrArray(*[self._flag_addr(r, 1, slot, self.rank) for r in range(self.world)])
        _lib.check(self.lib.golf_peer_signal_u32(flags, self.world, k + 1, _lib.stream_ptr()), "golf_peer_signal_u32")
        self.released += 1

    def flush(self) -> None:
        """Wait for and release everything outstanding, then check that no wait timed out (synchronises the stream)."""
        while self.released < self.pushed:
            if self.consumed == self.released:
                self.wait()
            self.release()
        self.check()

    def check(self) -> None:
        st = int(self.status.item())
        if st:
            raise RuntimeError(f"PeerStoreGather: rank {self.rank} timed out waiting for rank {st - 1} "
                               f"(pushed {self.pushed}, consumed {self.consumed})")

    def __enter__(self):
        return self

    def __exit__(self, exc_type, exc, tb):
        # leaving by an exception: the peers may be gone, do not add a barrier they will never join.  Our mappings of THEIR buffers
        # are closed; our OWN receive buffers are deliberately NOT freed: a peer that is still alive -- the exception
        # may be this rank's alone -- can still be storing into them, and a store into freed or re-used memory is worse than
        # depth x B x T floats held until the process exits.  They stay referenced from a module-level list.
        if exc_type is None:
            self.close()
        else:
            torch.cuda.synchronize(self.device)
            for p in self._opened:
                self.lib.golf_peer_close(p)
            _LEAKED_ON_ERROR.extend(self._local)
            self._opened, self._local, self.recv = [], [], None
        return False

    def close(self) -> None:
        """Unmap the peers' buffers and free the local ones (all ranks, after a barrier: nobody stores any more)."""
        import torch.distributed as dist

        torch.cuda.synchronize(self.device)
        dist.barrier()
        self.recv = None
        for p in self._opened:
            self.lib.golf_peer_close(p)
        self._opened = []
        dist.barrier()
        for p in self._local:
            self.lib.golf_peer_free(p)
        self._local = []
