"""AudioTensor: a tensor tagged with its hop length (samples per time step on dim 1).

The reference imports this type from the git submodule ``models/audiotensor`` which is EMPTY in the
snapshot (/root/reference/.gitmodules:4-6); its contract is recovered from the in-tree
``LegacyAudioTensor`` (models/utils.py:41-305) and from how the DSP modules use it:

* ``AudioTensor(data, hop_length=1)``; 1-D data is "timeless" (hop = INT64_MAX);
* element-wise binary ops / ``torch.where`` / ``matmul`` between different hops first bring every
  operand to the finest hop by *linear* upsampling (align_corners, length (F-1)*k+1), right-pad
  missing trailing dims, and truncate to the shortest number of steps (utils.py:213-241,270-296);
* any other torch function runs on the raw data and is re-wrapped with the common hop.

Own implementation (explicit dispatch table), not a copy.  The hot-path kernels never call
``reduce_hop_length`` on LPC coefficients — interpolation is fused into the HIP kernels.
"""
from __future__ import annotations

from typing import Tuple

import torch
import torch.nn.functional as F

__all__ = ["AudioTensor"]

_TIMELESS = 9223372036854775807

_ALIGNING = {
    torch.add, torch.sub, torch.mul, torch.div, torch.floor_divide, torch.remainder,
    torch.lt, torch.le, torch.gt, torch.ge, torch.eq, torch.ne, torch.where, torch.matmul,
}


def _lerp_up(x: torch.Tensor, k: int) -> torch.Tensor:
    """Linear upsampling of dim 1 by integer factor k: (B,F,...) -> (B,(F-1)k+1,...)."""
    if k == 1 or x.ndim < 2:
        return x
    t = x.transpose(1, -1) if x.ndim > 2 else x
    lead = t.shape[:-1]
    n = t.shape[-1]
    out = F.interpolate(t.reshape(-1, 1, n), size=(n - 1) * k + 1, mode="linear", align_corners=True)
    out = out.reshape(*lead, -1)
    return out.transpose(1, -1) if x.ndim > 2 else out


class AudioTensor:
    __slots__ = ("_data", "hop_length")

    def __init__(self, data, hop_length: int = 1, **kwargs):
        self._data = torch.as_tensor(data, **kwargs)
        self.hop_length = int(hop_length) if self._data.ndim > 1 else _TIMELESS

    # ---- plain accessors ------------------------------------------------------------------
    def as_tensor(self) -> torch.Tensor:
        return self._data

    def new_tensor(self, data: torch.Tensor) -> "AudioTensor":
        return AudioTensor(data, hop_length=self.hop_length)

    shape = property(lambda self: self._data.shape)
    ndim = property(lambda self: self._data.ndim)
    device = property(lambda self: self._data.device)
    dtype = property(lambda self: self._data.dtype)
    size = property(lambda self: self._data.size)
    names = property(lambda self: self._data.names)
    requires_grad = property(lambda self: self._data.requires_grad)

    def dim(self) -> int:
        return self._data.dim()

    @property
    def steps(self) -> int:
        return self._data.size(1) if self._data.ndim >= 2 else 1

    def __repr__(self):
        return f"AudioTensor(hop_length={self.hop_length}, data={self._data!r})"

    def __getitem__(self, index):
        return AudioTensor(self._data[index], hop_length=self.hop_length)

    def __len__(self):
        return len(self._data)

    def float(self):
        return self.new_tensor(self._data.float())

    def double(self):
        return self.new_tensor(self._data.double())

    def half(self):
        return self.new_tensor(self._data.half())

    def detach(self):
        return self.new_tensor(self._data.detach())

    def to(self, *a, **k):
        return self.new_tensor(self._data.to(*a, **k))

    # ---- hop manipulation -----------------------------------------------------------------
    def unfold(self, size: int, step: int = 1) -> "AudioTensor":
        assert self.ndim == 2
        return AudioTensor(self._data.unfold(1, size, step), hop_length=self.hop_length * step)

    def truncate(self, steps: int) -> "AudioTensor":
        if steps >= self.steps:
            return self
        return AudioTensor(self._data.narrow(1, 0, steps), hop_length=self.hop_length)

    def reduce_hop_length(self, factor: int = None) -> "AudioTensor":
        if factor is None:
            factor = self.hop_length
        else:
            assert factor <= self.hop_length and self.hop_length % factor == 0
        if factor == 1 or self.ndim < 2:
            return self
        return AudioTensor(_lerp_up(self._data, factor), hop_length=self.hop_length // factor)

    def increase_hop_length(self, factor: int) -> "AudioTensor":
        assert factor > 0
        if factor == 1 or self.ndim < 2:
            return self
        return AudioTensor(self._data[:, ::factor], hop_length=self.hop_length * factor)

    def set_hop_length(self, hop_length: int) -> "AudioTensor":
        assert hop_length > 0
        if hop_length > self.hop_length:
            assert hop_length % self.hop_length == 0
            return self.increase_hop_length(hop_length // self.hop_length)
        if hop_length < self.hop_length:
            assert self.hop_length % hop_length == 0
            return self.reduce_hop_length(self.hop_length // hop_length)
        return self

    # ---- alignment ------------------------------------------------------------------------
    @staticmethod
    def broadcasting(*tensors: "AudioTensor") -> Tuple["AudioTensor", ...]:
        assert tensors
        finest = min(t.hop_length for t in tensors)
        assert all(t.hop_length % finest == 0 for t in tensors), "hop lengths must divide each other"
        ups = [t.reduce_hop_length(t.hop_length // finest) if t.hop_length > finest else t for t in tensors]
        nd = max(t.ndim for t in ups)
        out = []
        for t in ups:
            if t.ndim < nd:
                t = AudioTensor(t._data.reshape(t._data.shape + (1,) * (nd - t.ndim)), hop_length=t.hop_length)
            out.append(t)
        return tuple(out)

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if func in (torch.cat, torch.stack):
            raise NotImplementedError("AudioTensor does not support torch.cat / torch.stack")
        args = list(args)
        if func in _ALIGNING:
            pos = [i for i, a in enumerate(args) if isinstance(a, AudioTensor)]
            aligned = cls.broadcasting(*(args[i] for i in pos))
            n = min(a.steps for a in aligned)
            for i, a in zip(pos, aligned):
                args[i] = a.truncate(n)
        hops = []
        for a in args:
            if isinstance(a, AudioTensor):
                hops.append(a.hop_length)
            elif isinstance(a, (tuple, list)):
                hops.extend(x.hop_length for x in a if isinstance(x, AudioTensor))
        assert hops and all(h == hops[0] for h in hops), f"mismatching hop lengths {hops}"
        raw = tuple(a._data if isinstance(a, AudioTensor) else a for a in args)
        ret = func(*raw, **kwargs)
        if isinstance(ret, torch.Tensor) and ret.ndim != 0:
            return AudioTensor(ret, hop_length=hops[0])
        return ret

    # ---- operators (all routed through __torch_function__ for alignment) -------------------
    def __neg__(self): return torch.neg(self)
    def __add__(self, o): return torch.add(self, o)
    def __radd__(self, o): return torch.add(o, self)
    def __sub__(self, o): return torch.sub(self, o)
    def __rsub__(self, o): return torch.sub(o, self)
    def __mul__(self, o): return torch.mul(self, o)
    def __rmul__(self, o): return torch.mul(o, self)
    def __truediv__(self, o): return torch.div(self, o)
    def __rtruediv__(self, o): return torch.div(o, self)
    def __floordiv__(self, o): return torch.floor_divide(self, o)
    def __rfloordiv__(self, o): return torch.floor_divide(o, self)
    def __mod__(self, o): return torch.remainder(self, o)
    def __rmod__(self, o): return torch.remainder(o, self)
    def __matmul__(self, o): return torch.matmul(self, o)
    def __rmatmul__(self, o): return torch.matmul(o, self)
    def __lt__(self, o): return torch.lt(self, o)
    def __le__(self, o): return torch.le(self, o)
    def __gt__(self, o): return torch.gt(self, o)
    def __ge__(self, o): return torch.ge(self, o)
    def __eq__(self, o): return torch.eq(self, o)
    def __ne__(self, o): return torch.ne(self, o)
    __hash__ = None
