"""Multi-GPU: the hot path shards over utterances (every utterance is independent in every kernel —
the recursion couples only along t), one process per GPU, no data-path collective.  The only exchange
is the all-gather of the synthesised audio (north_star / BASELINE configs[3]): 6.1 MB per rank at
B=32, one RCCL all_gather_into_tensor over xGMI.  Backend-agnostic so the same code runs under gloo
(CPU, tests) and nccl (= RCCL on ROCm).
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch

__all__ = ["shard_bounds", "shard_inputs", "gather_audio", "gather_audio_async", "synth_sharded", "StagedGather"]


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
