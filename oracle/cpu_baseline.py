"""Timed CPU baseline of the GOLF-ss source+filter step (TEST/BENCH INFRASTRUCTURE, kind = "port").

The reference's Python cannot travel to the GPU box and its third-party kernels are absent anyway, so the
baseline restates the reference's CPU *structure*:
  * oscillator: the exact PyTorch-CPU op sequence of IndexedGlottalFlowTable.forward /
    GlottalFlowTable.generate (reference models/synth.py:213-263, 124-177): table gather+blend,
    F.interpolate, fp32 cumsum, remainder, F.grid_sample, rsqrt, strided conv1d decimation
    (kazane.Decimate is a strided F.conv1d — stand-in taps, see golf_oracle.default_decimation_taps);
  * filter: AudioTensor gain broadcast + reduce_hop_length (F.interpolate) then
    oracle/golf_oracle.c::golf_oracle_sample_wise_lpc_f32 = torchlpc's CPU shape (sequential taps,
    OpenMP over the batch only).
Only bench.py's cpu_baseline leg and tests import this.
"""
from __future__ import annotations

import ctypes
import os

import numpy as np
import torch
import torch.nn.functional as F

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libgolf_oracle.so")
        if not os.path.exists(path):
            import subprocess

            subprocess.run(["make", "-s", "-C", _HERE], check=True)
        _LIB = ctypes.CDLL(path)
        _LIB.golf_oracle_num_threads.restype = ctypes.c_int
    return _LIB


def num_threads() -> int:
    return int(lib().golf_oracle_num_threads())


def _p(t: torch.Tensor):
    return ctypes.c_void_p(t.data_ptr())


def sample_wise_lpc_c(x: torch.Tensor, A: torch.Tensor) -> torch.Tensor:
    """fp32 or fp64, x (B,T), A (B,T,M) contiguous CPU tensors."""
    x, A = x.contiguous(), A.contiguous()
    y = torch.empty_like(x)
    B, T = x.shape
    fn = lib().golf_oracle_sample_wise_lpc_f32 if x.dtype == torch.float32 else lib().golf_oracle_sample_wise_lpc_f64
    fn(_p(x), _p(A), _p(y), B, T, A.shape[2])
    return y


def ltv_ss_c(ex: torch.Tensor, gain: torch.Tensor, a: torch.Tensor, hop: int) -> torch.Tensor:
    """Whole LTVMinimumPhaseFilterPrecise.forward in C (fp32/fp64)."""
    ex, gain, a = ex.contiguous(), gain.contiguous(), a.contiguous()
    B, Tx = ex.shape
    F_, M = a.shape[1], a.shape[2]
    T = min(Tx, (F_ - 1) * hop + 1)
    y = torch.empty(B, T, dtype=ex.dtype)
    scratch = torch.empty(B * T * (M + 2), dtype=ex.dtype)
    fn = lib().golf_oracle_ltv_ss_f32 if ex.dtype == torch.float32 else lib().golf_oracle_ltv_ss_f64
    fn(_p(ex), ctypes.c_int64(ex.stride(0)), _p(gain), _p(a), _p(y), B, T, F_, M, hop, _p(scratch))
    return y


def ltv_ss_bwd_c(gy, y, ex, gain, a, hop):
    """float64 closed-form backward (checker)."""
    gy, y, ex, gain, a = (t.double().contiguous() for t in (gy, y, ex, gain, a))
    B, T = y.shape
    F_, M = a.shape[1], a.shape[2]
    g_ex = torch.zeros(B, T, dtype=torch.float64)
    g_gain = torch.zeros(B, F_, dtype=torch.float64)
    g_a = torch.zeros(B, F_, M, dtype=torch.float64)
    scratch = torch.empty(B * T * (M + 2), dtype=torch.float64)
    lib().golf_oracle_ltv_ss_bwd_f64(_p(gy), _p(y), _p(ex), ctypes.c_int64(ex.stride(0)), _p(gain), _p(a), _p(g_ex),
                                     _p(g_gain), _p(g_a), B, T, F_, M, hop, _p(scratch))
    return g_ex, g_gain, g_a


def _upsample(x: torch.Tensor, k: int) -> torch.Tensor:
    """AudioTensor.reduce_hop_length (models/utils.py:171-191, 538-544)."""
    n = x.shape[1]
    if x.ndim == 2:
        return F.interpolate(x[:, None], (n - 1) * k + 1, mode="linear", align_corners=True)[:, 0]
    t = x.transpose(1, 2)
    return F.interpolate(t, (n - 1) * k + 1, mode="linear", align_corners=True).transpose(1, 2)


def oscillator_reference_ops(phase, wsel, w_hop, table, decim_kernel, oversampling=4, equal_energy=True):
    """models/synth.py:213-263 + 124-177 with plain tensors (phase at hop 1)."""
    n_tab, L = table.shape
    idx = wsel * (n_tab - 1)
    i0 = idx.long().clip_(0, n_tab - 2)
    p = (idx - i0).unsqueeze(-1)
    tables = table[i0.flatten()].view(*i0.shape, L) * (1 - p) + table[i0.flatten() + 1].view(*i0.shape, L) * p
    hop_t = w_hop * oversampling
    up = _upsample(phase / oversampling, oversampling)
    inst = torch.cumsum(up.float(), 1)
    wrapped = inst % 1
    B, N = wrapped.shape
    blocks = (N + hop_t - 1) // hop_t
    if tables.shape[1] < blocks + 1:
        tables = F.pad(tables, (0, 0, 0, blocks - tables.shape[1] + 1), "replicate")
    else:
        tables = tables[:, : blocks + 1]
    padded = torch.cat([tables, tables[:, :, :1]], dim=2)
    gx = wrapped * 2 - 1
    gy = torch.arange(N, dtype=wrapped.dtype).view(1, -1).broadcast_to(B, -1) / (hop_t * blocks) * 2 - 1
    grid = torch.stack([gx, gy], dim=2).unsqueeze(2)
    y = F.grid_sample(padded.unsqueeze(1), grid, mode="bilinear", align_corners=True).squeeze(-1).squeeze(1)
    if equal_energy:
        y = y * torch.rsqrt(up)
    if oversampling > 1:
        K = decim_kernel.shape[-1]
        y = F.conv1d(y[:, None], decim_kernel.view(1, 1, K), stride=oversampling, padding=(K - 1) // 2)[:, 0]
    return y


def golf_ss_synth_cpu(inp, table, decim_kernel):
    """One GOLF-ss source+filter step on the host: reference op structure, fp32."""
    osc = oscillator_reference_ops(inp["phase"], inp["wsel"], inp["w_hop"], table, decim_kernel, 4, True)
    src = osc + inp["noise"][:, : osc.shape[1]]
    hop = inp["hop"]
    G = _upsample(inp["gain"], hop)
    T = min(src.shape[1], G.shape[1])
    x = src[:, :T] * G[:, :T]
    A = _upsample(inp["a"], hop)[:, :T].contiguous()
    return sample_wise_lpc_c(x.contiguous(), A)
