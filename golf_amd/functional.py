"""torch.autograd.Functions over the C ABI of libgolf_hip.so (include/golf_amd.h).

Plain tensors in, plain tensors out; AudioTensor bookkeeping lives in the nn.Modules.
Every function raises if the tensors are not fp32 ROCm-device tensors — no CPU path exists.
"""
from __future__ import annotations

import math

import torch

from . import _lib

__all__ = ["ltv_allpole_ss", "ltv_allpole_prepare", "ltv_inverse", "lti_frames_ola", "glottal_osc",
           "ss_output_length", "ff_output_length", "osc_lengths", "PreparedTransitions", "ss_status",
           "zero_phase_fir_basis", "zero_phase_fir_kernels", "ltv_fir_frames", "zero_phase_fir_filter",
           "zero_phase_fir_filter_precise",
           "fir_frames_length", "lti_fir", "harmonic_osc", "biquad_frames_ola"]

HAVE_TRANSITIONS = 1
FAST_TRANSITIONS = 2
TRAINING = 64            # GOLF_SS_TRAINING
MAPS_ONLY = 128          # GOLF_SS_MAPS_ONLY (ABI 4)
THROUGHPUT = 256         # GOLF_SS_THROUGHPUT (ABI 4)
ZERO_TAIL = 512          # GOLF_SS_ZERO_TAIL (ABI 5): the backward zeroes g_ex beyond the output length itself
OSC_THROUGHPUT = 4       # GOLF_OSC_THROUGHPUT (ABI 6): batches in flight -> the oscillator's phase scan as a launch of its own
OSC_WS_KEPT = 2          # GOLF_OSC_WS_KEPT (ABI 5): the oscillator's backward reuses the totals in the forward's saved workspace
FORK_TRANSITIONS = False
SPLIT_P1 = False   # diagnostic: bench.py --split-p1  # set True to run the transition kernel beside the zero-state pass (streams)
_side_streams = {}

# Mixed-precision entry (reference intent: models/synth.py:250-251 pins the phase cumsum to fp32 so that the decoder survives
# `precision: 16-mixed`).  The kernels are fp32: under torch.autocast every Function casts its floating-point inputs to fp32
# and runs with autocast off; the backward runs in the same mode.  Outside autocast a non-fp32 tensor is still an error.
_amp_fwd = torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
_amp_bwd = torch.amp.custom_bwd(device_type="cuda")


def _side_stream(device) -> "torch.cuda.Stream":
    """One extra HIP stream per device for the excitation-independent phase (transition matrices)."""
    key = torch.device(device).index
    if key is None:
        key = torch.cuda.current_device()
    s = _side_streams.get(key)
    if s is None:
        s = _side_streams[key] = torch.cuda.Stream(device=key)
    return s


class PreparedTransitions:
    """Handle returned by ltv_allpole_prepare: the workspace with the transition matrices in flight on the side
    stream, plus the shape key they are valid for."""

    def __init__(self, ws, key, stream, a, fast=False, training=False, maps_only=False):
        self.ws, self.key, self.stream, self.a, self.fast, self.training = ws, key, stream, a, fast, training
        self.maps_only = maps_only   # the matrices alone: the forward still runs their fix-up and the group composites


def ltv_allpole_prepare(a: torch.Tensor, hop: int, T: int, overlap: bool = False,
                        fast: bool = False, mode=None, training: bool = False,
                        maps_only: bool = False) -> PreparedTransitions:
    """Compute the transition matrices for coefficients ``a`` (B,F,M) and output length ``T`` ahead of the
    excitation; pass the handle to ltv_allpole_ss(..., prepared=handle) (e.g. to filter several signals with the
    same coefficients, or to start the most expensive, excitation-independent phase early).
    ``overlap=True`` launches on a second HIP stream that the forward joins right before its boundary scan.
    With the two-level boundary scan the group composites (which need only the matrices) are computed here as well.
    Off by default: the HIP fork/join (two event record/wait pairs) measured ~25 us of latency on ROCm 7.2 -- more than the
    ~20 us of the transition kernel that the oscillator can hide at B = 32 (round 3: one batch alone 161 us forked vs 137 us
    on one stream; training 370 vs 283) -- and with several batches in flight the chip is full either way.  It pays when
    much more work precedes the filter on the main stream than the join costs (a long encoder).
    ``fast=True`` computes the fp32 matrices of the inference path; ``training=True`` (implied by ``fast=False``) also keeps
    what the backward reads, so the handle serves a forward whose gradients are needed.
    ``maps_only=True`` (with ``fast``): only the transition matrices; the forward then runs what the boundary scan still
    needs from them (fix-up of ill-conditioned matrices, group composites) in the launch of its zero-state pass."""
    _lib.require_device(a)
    lib = _lib.load()
    a = a.detach().contiguous()
    B, F, M = a.shape
    maps_only = bool(maps_only and fast)
    flags = (SS_MODES[mode] | (FAST_TRANSITIONS if fast else 0) | (TRAINING if (fast and training) else 0)
             | (MAPS_ONLY if maps_only else 0))
    ws = _workspace(lib.golf_ltv_allpole_workspace_bytes_ex(B, T, F, M, hop, flags), a.device)
    cur = torch.cuda.current_stream(a.device)
    side = None
    if overlap:
        side = _side_stream(a.device)
        side.wait_stream(cur)  # `a` (and the fresh workspace) are ordered after the current stream's work
        ws.record_stream(side)
        a.record_stream(side)
    rc = lib.golf_ltv_allpole_transitions_f32(a.data_ptr(), B, T, F, M, hop, ws.data_ptr(), ws.numel(),
                                              flags, (side or cur).cuda_stream)
    _lib.check(rc, "golf_ltv_allpole_transitions_f32")
    return PreparedTransitions(ws, (B, T, F, M, hop, a.data_ptr(), a._version, SS_MODES[mode]), side, a, fast,
                               training or not fast, maps_only)


def _rows(t: torch.Tensor) -> torch.Tensor:
    """2-D tensor with unit inner stride (row stride may exceed the width)."""
    assert t.ndim == 2
    if t.stride(1) == 1 and t.stride(0) >= t.shape[1]:
        return t
    if t.shape[0] == 1 and t.stride(1) == 1:
        # a single row may carry any row stride (0 after numpy's [None]) and still count as contiguous
        return t.as_strided(t.shape, (t.shape[1], 1))
    return t.contiguous()


def _workspace(nbytes: int, device) -> torch.Tensor:
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


def ss_output_length(Tx: int, F: int, hop: int) -> int:
    return min(Tx, (F - 1) * hop + 1)


# (ring width, largest order it serves) of the sample-wise filter's kernels (csrc/lpc_ss.hip kTable): the fast path --
# and with it the custom backward -- needs one ring width that divides the hop and exceeds the order.
SS_RINGS = ((8, 6), (16, 14), (24, 22), (32, 30), (40, 38))


def ss_is_trainable(M: int, hop: int, F: int = 2) -> bool:
    """True when (lpc order, hop) has a fast-path kernel, i.e. gradients are available.  Other shapes still run the
    forward (serial generic kernel) but cannot be trained through."""
    return F >= 2 and any(M <= order and hop % ring == 0 for ring, order in SS_RINGS)


# ------------------------------------------------------------------------------------------------
class _LTVAllPoleSS(torch.autograd.Function):
    @staticmethod
    @_amp_fwd
    def forward(ctx, ex, gain, a, hop, prepared, fast_inference, mode=0, status=None, length=None):
        _lib.require_device(ex, gain, a)
        lib = _lib.load()
        ex = _rows(ex)
        gain = gain.contiguous()
        a = a.contiguous()
        B, Tx = ex.shape
        F, M = a.shape[1], a.shape[2]
        assert gain.shape == (B, F) and a.shape[0] == B
        T = ss_output_length(Tx, F, hop)
        if length is not None:   # only the first `length` samples of the excitation rows (no slice op on the caller's side)
            T = min(T, int(length))
            if T < 1:
                raise _lib.GolfError(f"ltv_allpole_ss: length={length} leaves nothing to filter")
        y = torch.empty(B, T, dtype=torch.float32, device=ex.device)
        side, flags = None, 0
        needs_grad = any(ctx.needs_input_grad[:3])
        if needs_grad and not ss_is_trainable(M, hop, F):
            # fail before the forward, not at the first backward: name the supported grid
            raise _lib.GolfError(
                f"golf_amd: the sample-wise LPC filter has no backward for lpc_order={M}, hop={hop}, frames={F}: training "
                f"needs >= 2 frames and a ring width W in {[r for r, _ in SS_RINGS]} with hop % W == 0 and lpc_order <= W - 2 "
                f"(e.g. hop 240 -> orders up to 38, hop 256 -> up to 30, hop 100 -> none)")
        if (prepared is not None and prepared.key == (B, T, F, M, hop, a.data_ptr(), a._version, mode)
                and not (prepared.fast and needs_grad and not prepared.training)):
            # (a caller with batches in flight keeps the thin pair of chunk passes with prepared maps too: without the flag the
            #  forward would take the one-launch form meant for a lone batch -- 50 vs 43 us/step measured, round 5)
            ws, flags, side = prepared.ws, HAVE_TRANSITIONS | mode | (THROUGHPUT if THROUGHPUT_MODE else 0), prepared.stream
            if prepared.fast:
                flags |= FAST_TRANSITIONS | (TRAINING if prepared.training else 0) | (MAPS_ONLY if prepared.maps_only else 0)
        else:
            if (THROUGHPUT_MODE and mode == 0 and not needs_grad and B >= SS_THROUGHPUT_SERIAL_MIN
                    and ex.stride(0) < SS_SERIAL_MAX_STRIDE):
                # several batches in flight: the plan that costs the least CHIP time (see the constant).  An explicit GOLF_SS_SERIAL
                # turns the library's quiet fallback for huge row strides into GOLF_EUNSUPPORTED, hence the stride test here (the
                # output is allocated dense above) -- a caller who asked for "auto" never sees that error.
                mode = SS_MODES["serial"]
            ws = _workspace(lib.golf_ltv_allpole_workspace_bytes_ex(B, T, F, M, hop, mode), ex.device)
            flags = mode | (THROUGHPUT if THROUGHPUT_MODE else 0)
            if fast_inference:
                # fp32 transition matrices (hot chunks recomputed from fp64 trajectories) + one refinement sweep; with a
                # gradient pending the forward also keeps the adjoint-orientation maps for the backward's own sweep
                flags |= FAST_TRANSITIONS | (TRAINING if needs_grad else 0)
            if SPLIT_P1:
                flags |= 4                # GOLF_SS_SPLIT_P1 (diagnostic): two launches instead of the fused one
            if FORK_TRANSITIONS:
                # the transition kernel (needs only `a`) and the zero-state pass (needs `ex`) are independent: the
                # library forks the former onto a side stream and joins before the boundary scan
                side = _side_stream(ex.device)
                ws.record_stream(side)
        rc = lib.golf_ltv_allpole_fwd_f32(ex.data_ptr(), ex.stride(0), gain.data_ptr(), a.data_ptr(), y.data_ptr(),
                                          y.stride(0), B, T, F, M, hop, ws.data_ptr(), ws.numel(), flags,
                                          side.cuda_stream if side is not None else 0, _lib.stream_ptr())
        _lib.check(rc, "golf_ltv_allpole_fwd_f32")
        if status is not None:
            # conditioning / health words of this forward (include/golf_amd.h golf_ltv_allpole_status_u32): written
            # asynchronously on the stream into the caller's 4-word device tensor, read by the host when it likes
            if not (status.is_cuda and status.numel() >= 4 and status.element_size() == 4 and status.is_contiguous()):
                raise _lib.GolfError("ltv_allpole_ss: `status` must be a contiguous device tensor of >= 4 32-bit words")
            rc = lib.golf_ltv_allpole_status_u32(ws.data_ptr(), ws.numel(), B, T, F, M, hop, flags, status.data_ptr(),
                                                 _lib.stream_ptr())
            _lib.check(rc, "golf_ltv_allpole_status_u32")
        ctx.hop, ctx.mode = hop, mode
        ctx.save_for_backward(ex, gain, a, y, ws)
        return y

    @staticmethod
    @_amp_bwd
    def backward(ctx, gy):
        ex, gain, a, y, ws = ctx.saved_tensors
        lib = _lib.load()
        hop = ctx.hop
        B, Tx = ex.shape
        F, M = a.shape[1], a.shape[2]
        T = y.shape[1]
        gy = _rows(gy.float())
        # the kernels write columns [0, T); the excitation's tail beyond the output length needs zeros: GOLF_SS_ZERO_TAIL has
        # the backward write them (a full-size fill was a 6 MB memset in front of every backward, a strided fill of the
        # tail alone still a launch: ~5 us of the B = 32 training step either way)
        g_ex = torch.empty(B, Tx, dtype=torch.float32, device=ex.device)
        g_gain = torch.empty_like(gain)
        g_a = torch.empty_like(a)
        rc = lib.golf_ltv_allpole_bwd_f32(gy.data_ptr(), gy.stride(0), y.data_ptr(), y.stride(0), ex.data_ptr(),
                                          ex.stride(0), gain.data_ptr(), a.data_ptr(), g_ex.data_ptr(),
                                          g_ex.stride(0), g_gain.data_ptr(), g_a.data_ptr(), B, T, F, M, hop,
                                          ws.data_ptr(), ws.numel(), ctx.mode | (ZERO_TAIL if Tx > T else 0),
                                          _lib.stream_ptr())
        _lib.check(rc, "golf_ltv_allpole_bwd_f32")
        return g_ex, g_gain, g_a, None, None, None, None, None, None


SS_MODES = {None: 0, "auto": 0, "serial": 8, "chunked": 16, "flat-scan": 16 | 32}   # GOLF_SS_SERIAL / _CHUNKED / _FLAT_SCAN
# Set by a caller that keeps several batches in flight (bench.py's pipelined loop, a serving loop): GOLF_SS_THROUGHPUT
# is added to every sample-wise filter call.  Bit-identical results WITHIN one algorithm (the flag only changes the launch
# chain of the time-chunked scan); from SS_THROUGHPUT_SERIAL_MIN utterances on this module additionally switches an inference
# forward to the serial kernels, which are another algorithm: same accuracy class, different bits (and a forward through a
# PreparedTransitions handle keeps the chunked scan its maps were made for).  A lone batch takes ~10 us longer, four in
# flight finish ~2 % more per second (include/golf_amd.h).
THROUGHPUT_MODE = False
# With batches in flight (THROUGHPUT_MODE) an inference forward of this many utterances or more takes the batch-parallel
# serial kernels instead of the time-chunked scan (the library's own switch, for a lone batch, is at 2048).  The chunked
# scan does ~12 x the sequential arithmetic to finish ONE batch soon and saturates the chip at ~29 G samples/s whatever the
# batch; the serial kernels do 1 x, take ~2.4 ms per batch whatever its size, and leave the chip to the other batches.
# Whole synthesis step, MI355X, G samples/s (round 5): B = 256 chunked 28.8 | serial x 8 streams 24.5;  B = 1024 chunked 29.0 |
# serial x 4 streams 42.6, x 8 streams 57.4;  B = 2048 (serial either way) x 1 stream 29.0, x 4 streams 57.4.
SS_THROUGHPUT_SERIAL_MIN = 512
SS_SERIAL_MAX_STRIDE = 1 << 24   # the serial kernels address 16 rows through one 32-bit buffer descriptor (lpc_ss.hip serial_strides_ok)


def ltv_allpole_ss(ex: torch.Tensor, gain: torch.Tensor, a: torch.Tensor, hop: int,
                   prepared: "PreparedTransitions" = None, fast_inference: bool = True,
                   mode: str = None, status: torch.Tensor = None, length: int = None) -> torch.Tensor:
    """y[t] = ex[t]*up(gain)[t] - sum_i up(a)[t,i] y[t-1-i]; ex (B,Tx), gain (B,F), a (B,F,M) at hop.
    Output (B, min(Tx,(F-1)*hop+1)).  Differentiable w.r.t. ex, gain, a (custom HIP backward).
    ``prepared``: handle from ltv_allpole_prepare(a, hop, T) (ignored if it does not match).
    ``fast_inference``: when no input requires grad, use fp32 transition matrices + one refinement sweep instead of
    fp64 matrices (same accuracy class as a sequential fp32 recursion, ~4x less work in the dominant kernel).
    ``mode``: None/"auto" picks the algorithm by batch size (time-chunked scan below 2048 utterances, batch-parallel
    serial recursion from there on: include/golf_amd.h GOLF_SS_SERIAL); "serial" / "chunked" force one.  "flat-scan"
    forces the chunked algorithm's flat boundary scan.  That is not only a diagnostic: "auto" itself takes the flat scan for
    every lone batch of more than ~40 utterances below the serial threshold (lpc_ss.hip use_two_level_scan; the two-level scan
    is the faster form only while B x groups <= 2 x the CU count) and whenever an utterance has fewer than 48 chunk maps.  Both
    scans are held to the same bound -- 3 e_sequential_fp32 + 1e-4 of the float64 oracle forward, + 2e-4 for the gradients -- by
    the suite's soak (tests/test_gpu_lpc_ss.py::test_conditioning_soak_bounded, ::test_round5_soak_exceedances) and by
    tools/fuzz_tiers.py over eleven seeds x 120 cases (round 6: no row beyond it on either scan).
    ``length``: filter only the first ``length`` samples of ``ex`` (output (B, min(length, natural length))): what
    ``ltv_allpole_ss(ex[:, :length], ...)`` computes, without the slice -- whose backward would be a full-size fill and a
    full-size copy in front of the producer's backward (the gradient of the unused tail is written as zeros by the filter's
    own backward, GOLF_SS_ZERO_TAIL).
    ``status``: optional int32 device tensor of >= 4 elements that receives, asynchronously, the conditioning / health
    words of this call (decode with ss_status): utterances with recomputed chunk maps, utterances on the fp64 boundary
    scan, non-finite output flag, largest transition-matrix entry."""
    if ex.dim() == 2 and (ex.shape[0] == 0 or ex.shape[1] == 0):
        # an empty batch / zero samples: what the reference's tensor ops return (an empty result that stays in the graph);
        # the C ABI itself rejects non-positive sizes
        _lib.require_device(ex, gain, a)
        T = ss_output_length(ex.shape[1], a.shape[1], int(hop)) if ex.shape[1] else 0
        if length is not None:
            T = min(T, max(int(length), 0))
        return ex[:, :T] * 1.0 + 0.0 * (gain.sum() + a.sum())
    return _LTVAllPoleSS.apply(ex, gain, a, int(hop), prepared, bool(fast_inference), SS_MODES[mode], status, length)


def ss_status(status: torch.Tensor, warn: bool = True) -> dict:
    """Decode the 4 status words of ltv_allpole_ss(..., status=t) (synchronises: reads the device tensor).  ``warn=False``:
    the caller reports a fix-up timeout itself (the module's health monitor: one warning per event, not two)."""
    w = status.detach().to("cpu").view(torch.int32)[:4]
    if warn and int(w[2]) & 2:   # never observed; a wait that ran out means the maps of a hot utterance may be half-updated
        import warnings

        warnings.warn("golf_amd: a device-side wait for the chunk-map fix-up ran out (golf_ltv_allpole_status_u32 bit 1): "
                      "the output of hot utterances of this forward is not trustworthy", RuntimeWarning)
    return {"hot_utterances": int(w[0]), "tier3_utterances": int(w[1]), "nonfinite": bool(int(w[2]) & 1),
            "fixup_timeout": bool(int(w[2]) & 2), "scan_mismatch": bool(int(w[2]) & 4),
            "max_phi": float(w[3:4].view(torch.float32)[0])}


class _LTVInverse(torch.autograd.Function):
    @staticmethod
    @_amp_fwd
    def forward(ctx, y, a, hop):
        _lib.require_device(y, a)
        lib = _lib.load()
        y = _rows(y)
        a = a.contiguous()
        B, Ty = y.shape
        F, M = a.shape[1], a.shape[2]
        T = ss_output_length(Ty, F, hop)
        e = torch.empty(B, T, dtype=torch.float32, device=y.device)
        rc = lib.golf_ltv_inverse_f32(y.data_ptr(), y.stride(0), a.data_ptr(), e.data_ptr(), e.stride(0), B, T, F, M,
                                      hop, _lib.stream_ptr())
        _lib.check(rc, "golf_ltv_inverse_f32")
        ctx.save_for_backward(y, a)
        ctx.geom = (hop, T)
        return e

    @staticmethod
    @_amp_bwd
    def backward(ctx, g_e):
        y, a = ctx.saved_tensors
        hop, T = ctx.geom
        lib = _lib.load()
        g_e = _rows(g_e)
        B, Ty = y.shape
        F, M = a.shape[1], a.shape[2]
        need_y, need_a = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        g_y = (torch.empty_like(y) if Ty == T else torch.zeros_like(y)) if need_y else None
        g_a = torch.empty_like(a) if need_a else None
        rc = lib.golf_ltv_inverse_bwd_f32(g_e.data_ptr(), g_e.stride(0), y.data_ptr(), y.stride(0), a.data_ptr(),
                                          g_y.data_ptr() if need_y else None, g_y.stride(0) if need_y else 0,
                                          g_a.data_ptr() if need_a else None, B, T, F, M, hop, _lib.stream_ptr())
        _lib.check(rc, "golf_ltv_inverse_bwd_f32")
        return g_y, g_a, None


def ltv_inverse(y: torch.Tensor, a: torch.Tensor, hop: int) -> torch.Tensor:
    """e[t] = y[t] + sum_i up(a)[t,i] y[t-1-i] (analysis filter); differentiable w.r.t. y and a."""
    return _LTVInverse.apply(y, a, hop)


# ------------------------------------------------------------------------------------------------
def ff_output_length(Tx: int, F: int, hop: int, W: int):
    """(Tx_used, nfr, Ty) of the frame-wise filter (reference models/filters.py:147-180)."""
    Tx = min(Tx, (F - 1) * hop + 1)
    pad = W // 2
    nfr = (Tx + 2 * pad - W) // hop + 1
    Ty = (nfr - 1) * hop + W - 2 * pad
    return Tx, nfr, Ty


class _LTIFramesOLA(torch.autograd.Function):
    @staticmethod
    @_amp_fwd
    def forward(ctx, ex, gain, a, window, hop):
        _lib.require_device(ex, gain, a, window)
        lib = _lib.load()
        ex = _rows(ex)
        gain = gain.contiguous()
        a = a.contiguous()
        window = window.contiguous()
        B, Tx0 = ex.shape
        F, M = a.shape[1], a.shape[2]
        W = window.numel()
        Tx, nfr, Ty = ff_output_length(Tx0, F, hop, W)
        if nfr > F:
            raise _lib.GolfError(f"frame-wise filter: {nfr} frames needed but only {F} coefficient frames")
        y = torch.empty(B, Ty, dtype=torch.float32, device=ex.device)
        ws = _workspace(lib.golf_lti_frames_workspace_bytes(B, Tx, F, M, hop, W), ex.device)
        rc = lib.golf_lti_frames_ola_fwd_f32(ex.data_ptr(), ex.stride(0), gain.data_ptr(), a.data_ptr(),
                                             window.data_ptr(), y.data_ptr(), y.stride(0), B, Tx, F, M, hop, W, Ty,
                                             ws.data_ptr(), ws.numel(), _lib.stream_ptr())
        _lib.check(rc, "golf_lti_frames_ola_fwd_f32")
        ctx.save_for_backward(ex, gain, a, window, ws)  # ws holds the filtered frames the backward needs
        ctx.geom = (hop, Tx, Ty)
        return y

    @staticmethod
    @_amp_bwd
    def backward(ctx, gy):
        ex, gain, a, window, ws_fwd = ctx.saved_tensors
        hop, Tx, Ty = ctx.geom
        lib = _lib.load()
        gy = _rows(gy)
        B, Tx0 = ex.shape
        F, M = a.shape[1], a.shape[2]
        W = window.numel()
        g_ex = torch.empty(B, Tx0, dtype=torch.float32, device=ex.device)   # written in full by the backward (zeros in the tail)
        g_gain = torch.empty_like(gain)
        g_a = torch.empty_like(a)
        ws = _workspace(lib.golf_lti_frames_bwd_workspace_bytes(B, Tx, F, M, hop, W), ex.device)
        rc = lib.golf_lti_frames_ola_bwd_f32(gy.data_ptr(), gy.stride(0), ex.data_ptr(), ex.stride(0),
                                             gain.data_ptr(), a.data_ptr(), window.data_ptr(), g_ex.data_ptr(),
                                             g_ex.stride(0), Tx0, g_gain.data_ptr(), g_a.data_ptr(), B, Tx, F, M, hop,
                                             W, Ty, ws_fwd.data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream_ptr())
        _lib.check(rc, "golf_lti_frames_ola_bwd_f32")
        return g_ex, g_gain, g_a, None, None


def lti_frames_ola(ex, gain, a, window, hop: int) -> torch.Tensor:
    return _LTIFramesOLA.apply(ex, gain, a, window, int(hop))


# ------------------------------------------------------------------------------------------------
# control transform of the LPC filters: logits -> direct-form coefficients (reference models/utils.py:581-593)
# ------------------------------------------------------------------------------------------------
class _RC2LPC(torch.autograd.Function):
    @staticmethod
    @_amp_fwd
    def forward(ctx, logits, max_abs, apply_tanh):
        _lib.require_device(logits)
        lib = _lib.load()
        x = logits.float().contiguous()
        M = x.shape[-1]
        N = x.numel() // M
        a = torch.empty_like(x)
        _lib.check(lib.golf_rc2lpc_fwd_f32(x.data_ptr(), a.data_ptr(), N, M, float(max_abs), int(apply_tanh),
                                           _lib.stream_ptr()), "golf_rc2lpc_fwd_f32")
        ctx.cfg = (float(max_abs), int(apply_tanh))
        ctx.save_for_backward(x)
        return a

    @staticmethod
    @_amp_bwd
    def backward(ctx, g_a):
        (x,) = ctx.saved_tensors
        max_abs, apply_tanh = ctx.cfg
        lib = _lib.load()
        M = x.shape[-1]
        g = g_a.float().contiguous()
        out = torch.empty_like(x)
        _lib.check(lib.golf_rc2lpc_bwd_f32(x.data_ptr(), g.data_ptr(), out.data_ptr(), x.numel() // M, M, max_abs,
                                           apply_tanh, _lib.stream_ptr()), "golf_rc2lpc_bwd_f32")
        return out, None, None


def rc2lpc_logits(logits: torch.Tensor, max_abs: float = 1.0) -> torch.Tensor:
    """``rc2lpc(tanh(logits) * max_abs)`` (..., M) -> (..., M) as ONE kernel (forward and backward)."""
    return _RC2LPC.apply(logits, float(max_abs), True)


def rc2lpc(rc: torch.Tensor) -> torch.Tensor:
    """Levinson step-up of given reflection coefficients (..., M) -> a_1..a_M, one kernel, differentiable."""
    return _RC2LPC.apply(rc, 1.0, False)


_SOS_REP = {"coef": 0, "conj": 1, "real": 2}


class _SOS2LPC(torch.autograd.Function):
    @staticmethod
    @_amp_fwd
    def forward(ctx, logits, max_abs_pole, rep):
        _lib.require_device(logits)
        lib = _lib.load()
        x = logits.float().contiguous()
        M = x.shape[-1]
        assert M % 2 == 0, "two logits per second-order section"
        N = x.numel() // M
        a = torch.empty_like(x)
        _lib.check(lib.golf_sos2lpc_fwd_f32(x.data_ptr(), a.data_ptr(), N, M // 2, float(max_abs_pole), int(rep),
                                            _lib.stream_ptr()), "golf_sos2lpc_fwd_f32")
        ctx.cfg = (float(max_abs_pole), int(rep))
        ctx.save_for_backward(x)
        return a

    @staticmethod
    @_amp_bwd
    def backward(ctx, g_a):
        (x,) = ctx.saved_tensors
        rho, rep = ctx.cfg
        lib = _lib.load()
        M = x.shape[-1]
        g = g_a.float().contiguous()
        out = torch.empty_like(x)
        _lib.check(lib.golf_sos2lpc_bwd_f32(x.data_ptr(), g.data_ptr(), out.data_ptr(), x.numel() // M, M // 2, rho, rep,
                                            _lib.stream_ptr()), "golf_sos2lpc_bwd_f32")
        return out, None, None


def biquad_logits2lpc(logits: torch.Tensor, rep_type: str, max_abs_pole: float = 0.99) -> torch.Tensor:
    """``biquads2lpc(get_logits2biquads(rep_type, max_abs_pole)(logits.view(..., K, 2)))`` (..., 2K) -> (..., 2K) as
    ONE kernel, forward and backward."""
    return _SOS2LPC.apply(logits, float(max_abs_pole), _SOS_REP[rep_type])


# ------------------------------------------------------------------------------------------------
def osc_lengths(Tp: int, phase_hop: int, os: int):
    """(N oversampled length, Tout)."""
    P = phase_hop * os
    N = (Tp - 1) * P + 1 if P > 1 else Tp
    Tout = (N - 1) // os + 1 if os > 1 else N
    return N, Tout


_TAP_FRAGS: dict = {}   # (taps pointer, version, K, os, device) -> the taps' Toeplitz fragments (golf_glottal_osc_tap_fragments_f32)


def osc_tap_fragments(taps: torch.Tensor, os: int):
    """The decimation taps laid out for the matrix pipe, prepared once per tap set (ABI 6; the module's taps are a constant
    buffer, so this runs once per module and device).  None where the fused oscillator does not apply, and -- so that no
    allocation of a graph's private pool ends up in a process-wide cache -- when asked for the first time during a hipGraph
    capture: the library then lays them out inside the call, as it did every step until round 5."""
    if taps is None or taps.numel() == 0:
        return None
    key = (taps.data_ptr(), taps._version, taps.numel(), int(os), taps.device)
    hit = _TAP_FRAGS.get(key)
    if hit is not None:
        return hit[0]
    lib = _lib.load()
    n = lib.golf_glottal_osc_tap_fragments_bytes(taps.numel(), int(os))
    if n == 0 or torch.cuda.is_current_stream_capturing():
        return None
    frags = torch.empty(n, dtype=torch.uint8, device=taps.device)
    _lib.check(lib.golf_glottal_osc_tap_fragments_f32(taps.data_ptr(), taps.numel(), int(os), frags.data_ptr(), n,
                                                      _lib.stream_ptr()), "golf_glottal_osc_tap_fragments_f32")
    if len(_TAP_FRAGS) >= 16:
        _TAP_FRAGS.pop(next(iter(_TAP_FRAGS)))
    _TAP_FRAGS[key] = (frags, taps)   # (the taps tensor is kept alive with its fragments: the key is its address)
    return frags


class _GlottalOsc(torch.autograd.Function):
    @staticmethod
    @_amp_fwd
    def forward(ctx, phase, wsel, table, taps, phase_hop, w_hop, os, equal_energy, want_pre, add=None):
        _lib.require_device(phase, wsel, table, taps)
        if add is not None:
            assert os > 1 and add.ndim == 2 and add.shape[0] == phase.shape[0]
            _lib.require_device(add)
            add = _rows(add.float())
        if phase.requires_grad or table.requires_grad:
            raise NotImplementedError("golf_amd: the FUSED glottal oscillator differentiates w.r.t. table_select_weight "
                                      "only; use functional.wavetable_osc (IndexedGlottalFlowTable does so on its own) "
                                      "for gradients w.r.t. phase / the table")
        lib = _lib.load()
        phase = _rows(phase)
        wsel = wsel.contiguous()
        table = table.contiguous()
        B, Tp = phase.shape
        Fw = wsel.shape[1]
        n_tab, L = table.shape
        K = 0 if taps is None else taps.numel()
        if os > 1:
            assert taps is not None and K % 2 == 1, "decimation taps (odd length) required when oversampling > 1"
            taps = taps.contiguous()
        N, Tout = osc_lengths(Tp, phase_hop, os)
        out = torch.empty(B, Tout, dtype=torch.float32, device=phase.device)
        pre = torch.empty(B, N, dtype=torch.float32, device=phase.device) if (want_pre and os > 1) else None
        ws = _workspace(lib.golf_glottal_osc_workspace_bytes(B, Tp, phase_hop, Fw, w_hop, L, os), phase.device)
        frags = osc_tap_fragments(taps, os) if (os > 1 and not want_pre) else None
        rc = lib.golf_glottal_osc_fwd_f32(phase.data_ptr(), phase.stride(0), Tp, phase_hop, wsel.data_ptr(), Fw, w_hop,
                                          table.data_ptr(), n_tab, L, os,
                                          int(bool(equal_energy)) | (OSC_THROUGHPUT if THROUGHPUT_MODE else 0), _lib.ptr(taps), K,
                                          _lib.ptr(pre), out.data_ptr(), out.stride(0), B, Tout, ws.data_ptr(),
                                          ws.numel(), _lib.stream_ptr(), _lib.ptr(add),
                                          0 if add is None else add.stride(0), 0 if add is None else add.shape[1],
                                          _lib.ptr(frags))
        _lib.check(rc, "golf_glottal_osc_fwd_f32")
        ctx.frags = frags   # a backward that reuses the workspace takes the fragments its forward took
        ctx.cfg = (phase_hop, w_hop, os, bool(equal_energy))
        ctx.ws_kept = pre is None   # the saved workspace is private to this node: the backward may reuse the forward's totals
        ctx.add_len = None if add is None else add.shape[1]
        ctx.save_for_backward(phase, wsel, table, taps if taps is not None else phase.new_empty(0), ws)
        ctx.mark_non_differentiable(*([pre] if pre is not None else []))
        return (out, pre) if pre is not None else (out, None)

    @staticmethod
    @_amp_bwd
    def backward(ctx, g_out, _g_pre):
        phase, wsel, table, taps, ws = ctx.saved_tensors
        phase_hop, w_hop, os, eq = ctx.cfg
        lib = _lib.load()
        B, Tp = phase.shape
        Fw = wsel.shape[1]
        n_tab, L = table.shape
        K = taps.numel()
        g_out = _rows(g_out.float())
        g_w = torch.empty_like(wsel)
        rc = lib.golf_glottal_osc_bwd_wsel_f32(g_out.data_ptr(), g_out.stride(0), phase.data_ptr(), phase.stride(0), Tp,
                                               phase_hop, wsel.data_ptr(), Fw, w_hop, table.data_ptr(), n_tab, L, os,
                                               int(eq) | (OSC_WS_KEPT if ctx.ws_kept else 0), _lib.ptr(taps) if K else 0, K,
                                               g_w.data_ptr(), B,
                                               g_out.shape[1], ws.data_ptr(), ws.numel(), _lib.stream_ptr(),
                                               _lib.ptr(ctx.frags))
        _lib.check(rc, "golf_glottal_osc_bwd_wsel_f32")
        g_add = None
        if ctx.add_len is not None and ctx.needs_input_grad[9]:   # out[:, :Tadd] += add
            n = min(ctx.add_len, g_out.shape[1])
            g_add = g_out[:, :n] if n == ctx.add_len else torch.nn.functional.pad(g_out, (0, ctx.add_len - n))
        return None, g_w, None, None, None, None, None, None, None, g_add


def glottal_osc(phase, wsel, table, taps, phase_hop: int, w_hop: int, oversampling: int = 1,
                equal_energy: bool = False, return_pre: bool = False, add=None):
    """Indexed glottal-flow wavetable oscillator (see include/golf_amd.h golf_glottal_osc_fwd_f32).
    ``add`` (B, Tadd), oversampling > 1: fused ``out[:, :Tadd] += add`` (differentiable w.r.t. ``add``); the caller
    truncates to the common length as AudioTensor addition would."""
    if phase.dim() == 2 and phase.shape[0] == 0:   # an empty batch: empty outputs of the right widths, like the reference's ops
        _lib.require_device(phase, wsel, table)
        N, Tout = osc_lengths(phase.shape[1], int(phase_hop), int(oversampling))
        out = phase.new_zeros(0, Tout) + 0.0 * wsel.sum()
        return (out, phase.new_zeros(0, N)) if return_pre else out
    if add is not None and oversampling <= 1:   # no decimator to fuse into
        out, pre = _GlottalOsc.apply(phase, wsel, table, taps, int(phase_hop), int(w_hop), int(oversampling),
                                     bool(equal_energy), bool(return_pre))
        n = min(out.shape[1], add.shape[1])
        out = torch.cat([out[:, :n] + add[:, :n], out[:, n:]], 1)
        return (out, pre) if return_pre else out
    out, pre = _GlottalOsc.apply(phase, wsel, table, taps, int(phase_hop), int(w_hop), int(oversampling),
                                 bool(equal_energy), bool(return_pre), add)
    return (out, pre) if return_pre else out


def source_filter_ss(phase, wsel, table, taps, phase_hop: int, w_hop: int, oversampling: int, equal_energy: bool,
                     gain, a, hop: int, add=None, mode: str = None, status: torch.Tensor = None,
                     length: int = None) -> torch.Tensor:
    """``ltv_allpole_ss(glottal_osc(phase, wsel, table, taps, ..., add=add), gain, a, hop)`` -- the decoder's source into its end
    filter (reference: SourceFilterSynth.forward, models/sf.py:47-64) -- with the oscillator and the filter's chunk transition
    maps in ONE launch (include/golf_amd.h golf_source_transitions_f32, ABI 6): the maps need only ``a``, and for a lone batch
    they are the long pole of the chain (38 us of issue-bound waves on 160 of 256 CUs, which the oscillator's workgroups fill).
    Inference only and bit-identical to the composition; whenever a gradient is required, or the shapes fall outside the fused
    launch, this IS the composition.  ``length``: as ``ltv_allpole_ss(..., length=)`` -- filter only the first ``length`` samples of
    the source (the decoder's common length of oscillator and filtered noise)."""
    ts = (phase, wsel, table, gain, a) + ((add,) if add is not None else ())
    if (torch.is_grad_enabled() and any(t.requires_grad for t in ts)) or oversampling <= 1 or phase.dim() != 2 or phase.shape[0] == 0:
        src = glottal_osc(phase, wsel, table, taps, phase_hop, w_hop, oversampling, equal_energy, add=add)
        return ltv_allpole_ss(src, gain, a, hop, mode=mode, status=status, length=length)
    _lib.require_device(phase, wsel, table, taps, gain, a)
    lib = _lib.load()
    phase, wsel, table, taps = _rows(phase.float()), wsel.float().contiguous(), table.float().contiguous(), taps.float().contiguous()
    gain, a = gain.float().contiguous(), a.float().contiguous()
    if add is not None:
        _lib.require_device(add)
        add = _rows(add.float())
    B, Tp = phase.shape
    Fw = wsel.shape[1]
    n_tab, L = table.shape
    K = taps.numel()
    os_ = int(oversampling)
    _, Tout = osc_lengths(Tp, int(phase_hop), os_)
    F, M = a.shape[1], a.shape[2]
    T = ss_output_length(Tout, F, int(hop))
    if length is not None:
        T = min(T, int(length))
        if T < 1:
            raise _lib.GolfError(f"source_filter_ss: length={length} leaves nothing to filter")
    flags = SS_MODES[mode] | FAST_TRANSITIONS | MAPS_ONLY
    src = torch.empty(B, Tout, dtype=torch.float32, device=phase.device)
    osc_ws = _workspace(lib.golf_glottal_osc_workspace_bytes(B, Tp, int(phase_hop), Fw, int(w_hop), L, os_), phase.device)
    ss_ws = _workspace(lib.golf_ltv_allpole_workspace_bytes_ex(B, T, F, M, int(hop), flags), phase.device)
    frags = osc_tap_fragments(taps, os_)
    rc = lib.golf_source_transitions_f32(phase.data_ptr(), phase.stride(0), Tp, int(phase_hop), wsel.data_ptr(), Fw, int(w_hop),
                                         table.data_ptr(), n_tab, L, os_, int(bool(equal_energy)), taps.data_ptr(), K,
                                         src.data_ptr(), src.stride(0), B, Tout, osc_ws.data_ptr(), osc_ws.numel(),
                                         _lib.ptr(add), 0 if add is None else add.stride(0), 0 if add is None else add.shape[1],
                                         _lib.ptr(frags), a.data_ptr(), T, F, M, int(hop), ss_ws.data_ptr(), ss_ws.numel(), flags,
                                         _lib.stream_ptr())
    _lib.check(rc, "golf_source_transitions_f32")
    handle = PreparedTransitions(ss_ws, (B, T, F, M, int(hop), a.data_ptr(), a._version, SS_MODES[mode]), None, a, True, False, True)
    with torch.no_grad():
        return ltv_allpole_ss(src, gain, a, hop, prepared=handle, fast_inference=True, mode=mode, status=status, length=length)


# ------------------------------------------------------------------------------------------------
# generic (fully differentiable) table oscillator: wrapped phase -> bilinear table lookup (-> decimation)
# ------------------------------------------------------------------------------------------------
class _WavetableLookup(torch.autograd.Function):
    """GlottalFlowTable.generate (reference models/synth.py:124-177) on golf_wavetable_lookup_{fwd,bwd}_f32:
    out[b,n] = bilerp(tables[b], row n/hop_t, column wrapped[b,n]*L); differentiable w.r.t. both inputs."""

    @staticmethod
    @_amp_fwd
    def forward(ctx, wrapped, tables, hop_t):
        _lib.require_device(wrapped, tables)
        lib = _lib.load()
        wrapped = _rows(wrapped)
        tables = tables.contiguous()
        B, N = wrapped.shape
        assert tables.ndim == 3 and tables.shape[0] == B
        K, L = tables.shape[1], tables.shape[2]
        out = torch.empty(B, N, dtype=torch.float32, device=wrapped.device)
        rc = lib.golf_wavetable_lookup_fwd_f32(wrapped.data_ptr(), wrapped.stride(0), tables.data_ptr(), K, L, hop_t,
                                               out.data_ptr(), out.stride(0), B, N, _lib.stream_ptr())
        _lib.check(rc, "golf_wavetable_lookup_fwd_f32")
        ctx.hop_t = hop_t
        ctx.save_for_backward(wrapped, tables)
        return out

    @staticmethod
    @_amp_bwd
    def backward(ctx, g_out):
        wrapped, tables = ctx.saved_tensors
        lib = _lib.load()
        B, N = wrapped.shape
        K, L = tables.shape[1], tables.shape[2]
        g_out = _rows(g_out.float())
        g_w = torch.empty_like(wrapped) if ctx.needs_input_grad[0] else None
        g_t = torch.empty_like(tables) if ctx.needs_input_grad[1] else None
        rc = lib.golf_wavetable_lookup_bwd_f32(g_out.data_ptr(), g_out.stride(0), wrapped.data_ptr(), wrapped.stride(0),
                                               tables.data_ptr(), K, L, ctx.hop_t, _lib.ptr(g_w),
                                               0 if g_w is None else g_w.stride(0), _lib.ptr(g_t), B, N,
                                               _lib.stream_ptr())
        _lib.check(rc, "golf_wavetable_lookup_bwd_f32")
        return g_w, g_t, None


def wavetable_lookup(wrapped: torch.Tensor, tables: torch.Tensor, hop_t: int) -> torch.Tensor:
    """wrapped (B,N) in [0,1], tables (B,K,L) at hop ``hop_t`` (frames beyond K-1 replicate the last) -> (B,N)."""
    return _WavetableLookup.apply(wrapped, tables, int(hop_t))


class _DecimateFIR(torch.autograd.Function):
    """The oscillator's decimator on its own (golf_decimate_fir_f32 / its adjoint)."""

    @staticmethod
    @_amp_fwd
    def forward(ctx, x, taps, os):
        _lib.require_device(x, taps)
        lib = _lib.load()
        x = _rows(x)
        taps = taps.contiguous()
        B, N = x.shape
        Tout = (N - 1) // os + 1
        out = torch.empty(B, Tout, dtype=torch.float32, device=x.device)
        rc = lib.golf_decimate_fir_f32(x.data_ptr(), x.stride(0), N, taps.data_ptr(), taps.numel(), os, out.data_ptr(),
                                       out.stride(0), B, Tout, _lib.stream_ptr())
        _lib.check(rc, "golf_decimate_fir_f32")
        ctx.cfg = (N, os)
        ctx.save_for_backward(taps)
        return out

    @staticmethod
    @_amp_bwd
    def backward(ctx, g_out):
        (taps,) = ctx.saved_tensors
        N, os = ctx.cfg
        lib = _lib.load()
        g_out = _rows(g_out.float())
        B, Tout = g_out.shape
        g_x = torch.empty(B, N, dtype=torch.float32, device=g_out.device)
        rc = lib.golf_decimate_fir_adj_f32(g_out.data_ptr(), g_out.stride(0), Tout, taps.data_ptr(), taps.numel(), os,
                                           g_x.data_ptr(), N, B, _lib.stream_ptr())
        _lib.check(rc, "golf_decimate_fir_adj_f32")
        return g_x, None, None


def decimate_fir(x: torch.Tensor, taps: torch.Tensor, os: int) -> torch.Tensor:
    return _DecimateFIR.apply(x, taps, int(os))


def linear_upsample(z: torch.Tensor, hop: int) -> torch.Tensor:
    """AudioTensor.reduce_hop_length on a plain (B,F) tensor (models/utils.py:171-191, 538-544)."""
    if hop == 1 or z.shape[1] == 1:
        return z
    n = (z.shape[1] - 1) * hop + 1
    return torch.nn.functional.interpolate(z[:, None], n, mode="linear", align_corners=True)[:, 0]


class _PhaseAccumulate(torch.autograd.Function):
    """wrapped = frac(cumsum(up(phase / os)) + offset) on the device (golf_phase_accumulate_f32: the exact 64-bit fixed-point
    prefix of the fused oscillator).  Backward: d wrapped / d inst = 1, so g_phase = up^T(reverse cumsum(g)) / os and
    g_offset = g -- what autograd does through cumsum and F.interpolate in the reference (models/synth.py:239-255)."""

    @staticmethod
    @_amp_fwd
    def forward(ctx, phase, phase_offset, phase_hop, os):
        # float64 / half phases are accepted like the tensor-op version did (the reference forces .float() here,
        # models/synth.py:251): cast first, then the fp32 / device contract applies
        off_meta = None if phase_offset is None else (phase_offset.shape, phase_offset.dtype)
        phase = _rows(phase.float())
        if phase_offset is not None:
            phase_offset = _rows(phase_offset.float().reshape(phase.shape[0], -1))
        _lib.require_device(phase, phase_offset)
        lib = _lib.load()
        B, Tp = phase.shape
        P = phase_hop * os
        N = (Tp - 1) * P + 1 if P > 1 else Tp
        if phase_offset is not None:   # AudioTensor addition truncates to the shorter operand (utils.py:230-232)
            N = min(N, phase_offset.shape[1])
        out = torch.empty(B, N, dtype=torch.float32, device=phase.device)
        ws = _workspace(lib.golf_phase_accumulate_workspace_bytes(B, Tp), phase.device)
        rc = lib.golf_phase_accumulate_f32(phase.data_ptr(), phase.stride(0), Tp, phase_hop, os, _lib.ptr(phase_offset),
                                           0 if phase_offset is None else phase_offset.stride(0), out.data_ptr(),
                                           out.stride(0), B, N, ws.data_ptr(), ws.numel(), _lib.stream_ptr())
        _lib.check(rc, "golf_phase_accumulate_f32")
        ctx.geom = (Tp, P, os, N, phase_offset is not None and phase_offset.shape[1])
        ctx.off_meta = off_meta
        return out

    @staticmethod
    @_amp_bwd
    def backward(ctx, g):
        Tp, P, os, N, off_len = ctx.geom
        g_phase = g_off = None
        if ctx.needs_input_grad[0]:
            g_up = torch.flip(torch.cumsum(torch.flip(g.double(), [1]), 1), [1]).float()
            g_phase = upsample_adjoint(g_up, P, Tp) / os
        if ctx.needs_input_grad[1]:
            g_off = g.new_zeros(g.shape[0], off_len)
            g_off[:, :N] = g
            g_off = g_off.reshape(ctx.off_meta[0]).to(ctx.off_meta[1])   # the caller's shape and dtype
        return g_phase, g_off, None, None


def instantaneous_phase(phase: torch.Tensor, phase_hop: int, oversampling: int = 1, phase_offset: torch.Tensor = None):
    """(wrapped phase (B,N) float32 in [0,1], per-sample increment (B,N)): phase/os upsampled to the (oversampled) sample
    rate and accumulated exactly on the device (the reference: float32 cumsum, models/synth.py:250-251) plus the optional
    offset, wrapped; differentiable w.r.t. phase and phase_offset (custom backward: reverse cumulative sum + transposed
    upsampling, as in the reference).  The increment is a plain linear upsampling (it only scales the equal-energy factor)."""
    wrapped = _PhaseAccumulate.apply(phase, phase_offset, int(phase_hop), int(oversampling))
    ph = phase / oversampling if oversampling > 1 else phase
    up = linear_upsample(ph, phase_hop * oversampling)[:, : wrapped.shape[1]]
    return wrapped, up


def wavetable_osc(phase: torch.Tensor, phase_hop: int, tables: torch.Tensor, table_hop: int, oversampling: int = 1,
                  equal_energy: bool = False, phase_offset: torch.Tensor = None, taps: torch.Tensor = None,
                  decimate: bool = True) -> torch.Tensor:
    """The table oscillators of the reference in their general, fully differentiable form (models/synth.py:213-294):
    per-frame tables (B,K,L) at ``table_hop`` looked up at the running phase.  Differentiable w.r.t. ``phase``,
    ``tables`` and ``phase_offset``.  IndexedGlottalFlowTable routes here when anything but table_select_weight needs a
    gradient; the forward-only / weight-gradient case stays on the fused kernel (glottal_osc).
    ``decimate=False`` returns the oversampled signal."""
    wrapped, up = instantaneous_phase(phase, phase_hop, oversampling, phase_offset)
    v = wavetable_lookup(wrapped, tables, table_hop * oversampling)
    if equal_energy:
        v = v * torch.rsqrt(up)
    if oversampling > 1 and decimate:
        assert taps is not None, "decimation taps required when oversampling > 1"
        v = decimate_fir(v, taps, oversampling)
    return v


def blend_tables(table: torch.Tensor, wsel: torch.Tensor) -> torch.Tensor:
    """Two-row table blend of IndexedGlottalFlowTable (models/synth.py:223-237): (n_tab,L), (B,Fw) -> (B,Fw,L)."""
    n_tab = table.shape[0]
    idx = wsel * (n_tab - 1)
    i0 = idx.detach().long().clip_(0, n_tab - 2)
    p = (idx - i0).unsqueeze(-1)
    return table[i0] * (1 - p) + table[i0 + 1] * p


# ------------------------------------------------------------------------------------------------
# filtered-noise-band generator (reference models/noise.py:114-124)
# ------------------------------------------------------------------------------------------------
class _NoiseBand(torch.autograd.Function):
    @staticmethod
    @_amp_fwd
    def forward(ctx, log_gain, bands, offsets, hop, T):
        _lib.require_device(log_gain, bands)
        lib = _lib.load()
        log_gain, bands = log_gain.contiguous(), bands.contiguous()
        offsets = offsets.to(torch.int32).contiguous()
        B, F, K = log_gain.shape
        assert bands.shape[0] == K and offsets.shape == (B, K) and offsets.is_cuda
        Tout = min(T, (F - 1) * hop + 1) if F > 1 else T
        out = torch.empty(B, Tout, dtype=torch.float32, device=log_gain.device)
        rc = lib.golf_noise_band_fwd_f32(bands.data_ptr(), bands.shape[1], offsets.data_ptr(), log_gain.data_ptr(), F,
                                         hop, out.data_ptr(), out.stride(0), B, Tout, K, _lib.stream_ptr())
        _lib.check(rc, "golf_noise_band_fwd_f32")
        ctx.hop = hop
        ctx.save_for_backward(log_gain, bands, offsets)
        return out

    @staticmethod
    @_amp_bwd
    def backward(ctx, g_out):
        log_gain, bands, offsets = ctx.saved_tensors
        lib = _lib.load()
        B, F, K = log_gain.shape
        g_out = _rows(g_out.float())
        g_lg = torch.empty_like(log_gain)
        ws = _workspace(lib.golf_noise_band_workspace_bytes(B, F, K), log_gain.device)
        rc = lib.golf_noise_band_bwd_f32(g_out.data_ptr(), g_out.stride(0), bands.data_ptr(), bands.shape[1],
                                         offsets.data_ptr(), log_gain.data_ptr(), F, ctx.hop, g_lg.data_ptr(), B,
                                         g_out.shape[1], K, ws.data_ptr(), ws.numel(), _lib.stream_ptr())
        _lib.check(rc, "golf_noise_band_bwd_f32")
        return g_lg, None, None, None, None


def noise_band(bands: torch.Tensor, offsets: torch.Tensor, log_gain: torch.Tensor, hop: int, T: int) -> torch.Tensor:
    """bands (K, L) loopable noise periods, offsets (B,K) start indices, log_gain (B,F,K) at ``hop`` ->
    (B, min(T, (F-1)*hop+1)); differentiable w.r.t. log_gain (include/golf_amd.h golf_noise_band_*)."""
    return _NoiseBand.apply(log_gain, bands, offsets, int(hop), int(T))


# ------------------------------------------------------------------------------------------------
# zero-phase FIR noise filter (reference models/filters.py:286-384)
# ------------------------------------------------------------------------------------------------
_basis_cache = {}


def zero_phase_fir_basis(n_mag: int, device) -> torch.Tensor:
    """The constant cosine-transform matrix for ``n_mag`` bins (both orientations), built once per device."""
    dev = torch.device(device)
    key = (n_mag, dev.index if dev.index is not None else torch.cuda.current_device())
    b = _basis_cache.get(key)
    if b is None:
        lib = _lib.load()
        nbytes = lib.golf_zero_phase_fir_basis_bytes(n_mag)
        b = torch.empty(nbytes // 4, dtype=torch.float32, device=dev)
        _lib.check(lib.golf_zero_phase_fir_basis_f32(n_mag, b.data_ptr(), nbytes, _lib.stream_ptr()),
                   "golf_zero_phase_fir_basis_f32")
        _basis_cache[key] = b
    return b


def fir_frames_length(T: int, F: int, N: int, hop: int) -> int:
    lib = _lib.load()
    n = lib.golf_ltv_fir_frames_length(T, F, N, hop)
    if n < 0:
        raise _lib.GolfError(lib.golf_last_error().decode(errors="replace"))
    return n


def _zp_kernels_raw(lib, log_mag, window, basis):
    B, F, n_mag = log_mag.shape
    KS = lib.golf_zero_phase_fir_row_stride(n_mag)
    kern = torch.empty(B * F, KS, dtype=torch.float32, device=log_mag.device)
    _lib.check(lib.golf_zero_phase_fir_kernels_f32(log_mag.data_ptr(), window.data_ptr(), basis.data_ptr(),
                                                   kern.data_ptr(), B * F, n_mag, _lib.stream_ptr()),
               "golf_zero_phase_fir_kernels_f32")
    return kern


def zero_phase_fir_kernels(log_mag: torch.Tensor, window: torch.Tensor) -> torch.Tensor:
    """(B,F,n_mag) log magnitudes -> (B,F,N) windowed zero-phase FIR kernels, N = 2*(n_mag-1) (no autograd)."""
    _lib.require_device(log_mag, window)
    lib = _lib.load()
    log_mag = log_mag.detach().contiguous()
    B, F, n_mag = log_mag.shape
    N = 2 * (n_mag - 1)
    kern = _zp_kernels_raw(lib, log_mag, window.contiguous(), zero_phase_fir_basis(n_mag, log_mag.device))
    return kern.view(B, F, -1)[..., :N]


class _ZPKernels(torch.autograd.Function):
    """(B,F,n_mag) log magnitudes -> (B*F, row_stride) windowed zero-phase FIR rows (cosine transform on the MFMAs)."""

    @staticmethod
    @_amp_fwd
    def forward(ctx, log_mag, window):
        _lib.require_device(log_mag, window)
        lib = _lib.load()
        log_mag = log_mag.contiguous()
        window = window.contiguous()
        n_mag = log_mag.shape[2]
        if window.numel() != 2 * (n_mag - 1):
            raise _lib.GolfError(f"zero_phase_fir: window has {window.numel()} taps, expected {2 * (n_mag - 1)}")
        basis = zero_phase_fir_basis(n_mag, log_mag.device)
        kern = _zp_kernels_raw(lib, log_mag, window, basis)
        ctx.save_for_backward(log_mag, window, basis)
        return kern

    @staticmethod
    @_amp_bwd
    def backward(ctx, g_kern):
        log_mag, window, basis = ctx.saved_tensors
        lib = _lib.load()
        g_kern = g_kern.contiguous()
        B, F, n_mag = log_mag.shape
        g_lm = torch.empty_like(log_mag)
        _lib.check(lib.golf_zero_phase_fir_kernels_bwd_f32(g_kern.data_ptr(), log_mag.data_ptr(), window.data_ptr(),
                                                           basis.data_ptr(), g_lm.data_ptr(), B * F, n_mag,
                                                           _lib.stream_ptr()),
                   "golf_zero_phase_fir_kernels_bwd_f32")
        return g_lm, None


class _FIRFrames(torch.autograd.Function):
    """Per-frame FIR with kernel rows kern (B*F, row_stride): output frame f uses row f + frame0."""

    @staticmethod
    @_amp_fwd
    def forward(ctx, ex, kern, F, N, hop, frame0):
        _lib.require_device(ex, kern)
        lib = _lib.load()
        ex = _rows(ex)
        kern = kern.contiguous()
        B, T = ex.shape
        if kern.shape[0] != B * F:
            raise _lib.GolfError(f"fir_frames: {kern.shape[0]} kernel rows for B={B}, F={F}")
        Ty = fir_frames_length(T, F - frame0, N, hop)
        y = torch.empty(B, Ty, dtype=torch.float32, device=ex.device)
        _lib.check(lib.golf_ltv_fir_frames_fwd_f32(ex.data_ptr(), ex.stride(0), kern.data_ptr(), kern.shape[1],
                                                   y.data_ptr(), y.stride(0), B, T, F, N, hop, frame0,
                                                   _lib.stream_ptr()),
                   "golf_ltv_fir_frames_fwd_f32")
        ctx.save_for_backward(ex, kern)
        ctx.geom = (F, N, hop, frame0)
        return y

    @staticmethod
    @_amp_bwd
    def backward(ctx, gy):
        ex, kern = ctx.saved_tensors
        F, N, hop, frame0 = ctx.geom
        lib = _lib.load()
        gy = _rows(gy)
        B, T = ex.shape
        need_ex, need_k = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        g_ex = torch.empty_like(ex) if need_ex else None
        g_kern = torch.empty_like(kern) if need_k else None
        _lib.check(lib.golf_ltv_fir_frames_bwd_f32(gy.data_ptr(), gy.stride(0), ex.data_ptr(), ex.stride(0),
                                                   kern.data_ptr(), kern.shape[1],
                                                   g_ex.data_ptr() if need_ex else None,
                                                   g_ex.stride(0) if need_ex else 0,
                                                   g_kern.data_ptr() if need_k else None,
                                                   B, T, F, N, hop, frame0, _lib.stream_ptr()),
                   "golf_ltv_fir_frames_bwd_f32")
        # (the padding taps [N, row stride) carry no gradient: the kernel writes their zeros itself)
        return g_ex, g_kern, None, None, None, None


def zero_phase_fir_filter(ex: torch.Tensor, log_mag: torch.Tensor, window: torch.Tensor, hop: int) -> torch.Tensor:
    """LTVZeroPhaseFIRFilter.forward on plain tensors: ex (B,T), log_mag (B,F,n_mag) at ``hop`` -> (B, nfr*hop);
    differentiable w.r.t. ex and log_mag."""
    if log_mag.dim() != 3 or log_mag.shape[0] != ex.shape[0]:
        raise _lib.GolfError(f"zero_phase_fir_filter: ex {tuple(ex.shape)} vs log_mag {tuple(log_mag.shape)}")
    F, n_mag = log_mag.shape[1], log_mag.shape[2]
    kern = _ZPKernels.apply(log_mag, window)
    return _FIRFrames.apply(ex, kern, F, 2 * (n_mag - 1), hop, 0)


def zero_phase_fir_filter_precise(ex: torch.Tensor, log_mag: torch.Tensor, window: torch.Tensor, hop: int) -> torch.Tensor:
    """LTVZeroPhaseFIRFilterPrecise.forward (reference models/filters.py:308-337): the kernels are linearly
    interpolated to sample rate, y[t] = sum_k pad(ex)[t+k] * ((1-w_t) K_f[k] + w_t K_{f+1}[k]), f = t // hop,
    w_t = (t % hop)/hop, output length min(T, (F-1)*hop+1).  Evaluated as two frame FIRs (kernel rows f and f+1 over
    the same signal) blended per sample; the single sample t = (F-1)*hop is a dot product with the last kernel."""
    if log_mag.dim() != 3 or log_mag.shape[0] != ex.shape[0]:
        raise _lib.GolfError(f"zero_phase_fir_filter_precise: ex {tuple(ex.shape)} vs log_mag {tuple(log_mag.shape)}")
    B, T = ex.shape
    F, n_mag = log_mag.shape[1], log_mag.shape[2]
    if F < 2:
        raise _lib.GolfError("zero_phase_fir_filter_precise: need at least 2 frames")
    N = 2 * (n_mag - 1)
    P = (N - 1) // 2
    Tfull = (F - 1) * hop + 1
    Tout = min(T, Tfull)
    kern = _ZPKernels.apply(log_mag, window)
    # the samples the outputs t < Tfull can reach: ex[t + k - P], k < N  ->  indices < Tfull + N-1-P (zeros past T)
    Tin = Tfull + N - 1 - P
    x = ex[:, :Tin]
    if x.shape[1] < Tin:
        x = torch.nn.functional.pad(x, (0, Tin - x.shape[1]))
    ya = _FIRFrames.apply(x, kern, F, N, hop, 0)[:, : (F - 1) * hop]
    yb = _FIRFrames.apply(x, kern, F, N, hop, 1)[:, : (F - 1) * hop]
    w = (torch.arange((F - 1) * hop, device=ex.device) % hop).to(torch.float32) / hop
    main = ya + w * (yb - ya)
    tail = torch.nn.functional.pad(x, (P, 0))[:, Tfull - 1: Tfull - 1 + N]  # pad(ex)[Tfull-1 + k], k < N
    last = (tail * kern.view(B, F, -1)[:, F - 1, :N]).sum(-1, keepdim=True)
    return torch.cat([main, last], dim=1)[:, :Tout]


def ltv_fir_frames(ex: torch.Tensor, kernels: torch.Tensor, hop: int) -> torch.Tensor:
    """Frame-wise FIR with arbitrary per-frame kernels (B,F,N): y[b,f*hop+n] = sum_k pad(ex)[b,f*hop+n+k] *
    kernels[b,f,k]; differentiable w.r.t. both."""
    B, F, N = kernels.shape
    KS = (N + 15) // 16 * 16
    kern = torch.nn.functional.pad(kernels.reshape(B * F, N), (0, KS - N))
    return _FIRFrames.apply(ex, kern, F, N, hop, 0)


# ------------------------------------------------------------------------------------------------
# LTI FIR shared by the batch (room filter, reference models/filters.py:426-449)
# ------------------------------------------------------------------------------------------------
class _LTIFIR(torch.autograd.Function):
    @staticmethod
    @_amp_fwd
    def forward(ctx, ex, taps, lead):
        _lib.require_device(ex, taps)
        lib = _lib.load()
        ex = _rows(ex)
        taps = taps.contiguous()
        B, T = ex.shape
        y = torch.empty(B, T, dtype=torch.float32, device=ex.device)
        _lib.check(lib.golf_lti_fir_f32(ex.data_ptr(), ex.stride(0), taps.data_ptr(), taps.numel(), lead,
                                        y.data_ptr(), y.stride(0), B, T, _lib.stream_ptr()), "golf_lti_fir_f32")
        ctx.save_for_backward(ex, taps)
        ctx.lead = lead
        return y

    @staticmethod
    @_amp_bwd
    def backward(ctx, gy):
        ex, taps = ctx.saved_tensors
        lib = _lib.load()
        gy = _rows(gy)
        B, T = ex.shape
        n = taps.numel()
        g_ex = g_taps = None
        if ctx.needs_input_grad[0]:
            g_ex = torch.empty_like(ex)
            # the adjoint = the same FIR with the taps in reverse order: a negative tap count (ABI 5) instead of taps.flip(0)
            _lib.check(lib.golf_lti_fir_f32(gy.data_ptr(), gy.stride(0), taps.data_ptr(), -n, n - 1 - ctx.lead,
                                            g_ex.data_ptr(), g_ex.stride(0), B, T, _lib.stream_ptr()),
                       "golf_lti_fir_f32 (adjoint)")
        if ctx.needs_input_grad[1]:
            g_taps = torch.empty_like(taps)
            ws = _workspace(lib.golf_lti_fir_taps_grad_workspace_bytes(B, T, n), ex.device)
            _lib.check(lib.golf_lti_fir_taps_grad_f32(gy.data_ptr(), gy.stride(0), ex.data_ptr(), ex.stride(0),
                                                      g_taps.data_ptr(), n, ctx.lead, B, T, ws.data_ptr(),
                                                      ws.numel(), _lib.stream_ptr()),
                       "golf_lti_fir_taps_grad_f32")
        return g_ex, g_taps, None


def lti_fir(ex: torch.Tensor, taps: torch.Tensor, lead: int) -> torch.Tensor:
    """y[b,t] = sum_n taps[n] * ex[b, t-lead+n] (zero outside the signal); taps.numel() % 4 == 0; differentiable
    w.r.t. both arguments."""
    return _LTIFIR.apply(ex, taps, lead)


# ------------------------------------------------------------------------------------------------
# harmonic oscillator bank (reference models/synth.py:403-547)
# ------------------------------------------------------------------------------------------------
def _up_len(n: int, hop: int) -> int:
    return (n - 1) * hop + 1 if hop > 1 else n


class _HarmonicOsc(torch.autograd.Function):
    @staticmethod
    @_amp_fwd
    def forward(ctx, phase, amp, tscale, hscale, H, phase_hop, amp_hop, ts_hop, phase_offset=None, po_hop=1,
                initial_phase=None):
        _lib.require_device(phase, amp, tscale, hscale, phase_offset, initial_phase)
        lib = _lib.load()
        phase = _rows(phase)
        B, Tp = phase.shape
        Tout = _up_len(Tp, phase_hop)
        Fa = Fs = 1
        if amp is not None:
            amp = amp.contiguous()
            if amp.dim() != 3 or amp.shape[0] != B or amp.shape[2] != H:
                raise _lib.GolfError(f"harmonic_osc: amplitudes {tuple(amp.shape)} for B={B}, H={H}")
            Fa = amp.shape[1]
            Tout = min(Tout, _up_len(Fa, amp_hop))
        if tscale is not None:
            tscale = tscale.contiguous()
            Fs = tscale.shape[1]
            Tout = min(Tout, _up_len(Fs, ts_hop))
        if hscale is not None:
            hscale = hscale.contiguous()
        Fo = 1
        if phase_offset is not None:
            phase_offset = phase_offset.contiguous()
            Fo = phase_offset.shape[1]
            Tout = min(Tout, _up_len(Fo, po_hop))
        if initial_phase is not None:
            initial_phase = initial_phase.contiguous()
            if tuple(initial_phase.shape) != (B, H):
                raise _lib.GolfError(f"harmonic_osc: initial_phase {tuple(initial_phase.shape)} for B={B}, H={H}")
        out = torch.empty(B, Tout, dtype=torch.float32, device=phase.device)
        ws = _workspace(lib.golf_harmonic_osc_workspace_bytes(B, Tp, phase_hop, Fa if amp is not None else 0, H),
                        phase.device)
        rc = lib.golf_harmonic_osc_fwd_f32(phase.data_ptr(), phase.stride(0), Tp, phase_hop, _lib.ptr(amp), Fa, amp_hop,
                                           _lib.ptr(tscale), Fs, ts_hop, _lib.ptr(hscale), H, out.data_ptr(),
                                           out.stride(0), B, Tout, ws.data_ptr(), ws.numel(), _lib.stream_ptr(),
                                           _lib.ptr(phase_offset), Fo, po_hop, _lib.ptr(initial_phase))
        _lib.check(rc, "golf_harmonic_osc_fwd_f32")
        ctx.save_for_backward(phase, tscale, hscale, amp, phase_offset, initial_phase)
        ctx.geom = (H, phase_hop, amp_hop, ts_hop, Fa, Fs, Tout, amp is not None)
        ctx.po = (Fo, po_hop)
        return out

    @staticmethod
    def _run(lib, name, phase, amp, tscale, hscale, geom, phase_offset=None, po=(1, 1), initial_phase=None):
        H, phase_hop, amp_hop, ts_hop, Fa, Fs, Tout, has_amp = geom
        B, Tp = phase.shape
        out = torch.empty(B, Tout, dtype=torch.float32, device=phase.device)
        ws = _workspace(lib.golf_harmonic_osc_workspace_bytes(B, Tp, phase_hop, Fa if has_amp else 0, H), phase.device)
        rc = getattr(lib, name)(phase.data_ptr(), phase.stride(0), Tp, phase_hop, _lib.ptr(amp), Fa, amp_hop,
                                _lib.ptr(tscale), Fs, ts_hop, _lib.ptr(hscale), H, out.data_ptr(), out.stride(0), B, Tout,
                                ws.data_ptr(), ws.numel(), _lib.stream_ptr(), _lib.ptr(phase_offset), po[0], po[1],
                                _lib.ptr(initial_phase))
        _lib.check(rc, name)
        return out

    @staticmethod
    @_amp_bwd
    def backward(ctx, g_out):
        phase, tscale, hscale, amp, phase_offset, initial_phase = ctx.saved_tensors
        H, phase_hop, amp_hop, ts_hop, Fa, Fs, Tout, has_amp = ctx.geom
        if ctx.needs_input_grad[3]:
            raise NotImplementedError("golf_amd: the per-harmonic scale of the harmonic oscillator is a constant "
                                      "(SawToothOscillator's 1/h buffer)")
        lib = _lib.load()
        g_out = _rows(g_out.float())
        B, Tp = phase.shape
        g_phase = g_amp = g_ts = g_po = g_ip = None
        if ctx.needs_input_grad[10]:
            # d out / d initial_phase[b,h] = 2 pi sum_t g[t] A_h[t] cos(2 pi theta_h[t]), A = the (masked, scaled) amplitude
            # track.  cos(x) = sin(x + 1/4 cycle) and A_h[t] = sum_f hat_f(t) amp[b,f,h]: the amplitude-gradient kernel run
            # with initial_phase + 1/4 returns G[b,f,h] = sum_t hat_f(t) g[t] cos(.) (scales and Nyquist mask included), and
            # g_ip = 2 pi sum_f amp[b,f,h] G[b,f,h].  Without an amplitude track (A = 1): pseudo-frames 256 samples apart, whose
            # hat functions sum to one over the signal -- G summed over them is the plain sum over t.  (Two frames one whole
            # signal apart gave the kernel ONE segment: a single wave per utterance walking Tout x H sincos chains, milliseconds
            # per call -- ADVICE r3.)  (reference: autograd through models/synth.py:434-440)
            if has_amp:
                Fq, hq = Fa, amp_hop
            else:
                hq = 256 if Tout > 257 else max(Tout - 1, 1)
                Fq = -(-(Tout - 1) // hq) + 1 if Tout > 1 else 2
            G = torch.empty(B, Fq, H, dtype=torch.float32, device=phase.device)
            ws = _workspace(lib.golf_harmonic_osc_workspace_bytes(B, Tp, phase_hop, Fq, H), phase.device)
            shifted = (initial_phase + 0.25).contiguous()
            rc = lib.golf_harmonic_osc_bwd_amp_f32(g_out.data_ptr(), g_out.stride(0), phase.data_ptr(),
                                                   phase.stride(0), Tp, phase_hop, Fq, hq, _lib.ptr(tscale), Fs,
                                                   ts_hop, _lib.ptr(hscale), H, G.data_ptr(), B, Tout,
                                                   ws.data_ptr(), ws.numel(), _lib.stream_ptr(), _lib.ptr(phase_offset),
                                                   ctx.po[0], ctx.po[1], shifted.data_ptr())
            _lib.check(rc, "golf_harmonic_osc_bwd_amp_f32")
            g_ip = (2.0 * math.pi) * ((G * amp).sum(1) if has_amp else G.sum(1))
        if has_amp and ctx.needs_input_grad[1]:
            g_amp = torch.empty(B, Fa, H, dtype=torch.float32, device=phase.device)
            ws = _workspace(lib.golf_harmonic_osc_workspace_bytes(B, Tp, phase_hop, Fa, H), phase.device)
            rc = lib.golf_harmonic_osc_bwd_amp_f32(g_out.data_ptr(), g_out.stride(0), phase.data_ptr(),
                                                   phase.stride(0), Tp, phase_hop, Fa, amp_hop, _lib.ptr(tscale), Fs,
                                                   ts_hop, _lib.ptr(hscale), H, g_amp.data_ptr(), B, Tout,
                                                   ws.data_ptr(), ws.numel(), _lib.stream_ptr(), _lib.ptr(phase_offset),
                                                   ctx.po[0], ctx.po[1], _lib.ptr(initial_phase))
            _lib.check(rc, "golf_harmonic_osc_bwd_amp_f32")
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[8]:
            # out depends on the phase input through Phi = cumsum(up(phase)) (the Nyquist mask is piecewise constant):
            # g_phase = up^T( reverse-cumsum( g_out * d out / d Phi ) ), the derivative bank from the same kernel
            d = _HarmonicOsc._run(lib, "golf_harmonic_osc_dphase_f32", phase, amp if has_amp else None, tscale, hscale,
                                  ctx.geom, phase_offset, ctx.po, initial_phase)
            g_inst = (g_out[:, :Tout] * d).double()
            if ctx.needs_input_grad[0]:
                g_up = torch.flip(torch.cumsum(torch.flip(g_inst, [1]), 1), [1]).float()
                g_phase = upsample_adjoint(g_up, phase_hop, Tp)
            if ctx.needs_input_grad[8]:   # the offset enters every harmonic's phase like Phi does, without the cumsum
                g_po = upsample_adjoint(g_inst.float(), ctx.po[1], ctx.po[0])
        if tscale is not None and ctx.needs_input_grad[2]:
            # out = up(tscale) * S: S is the same kernel without the per-sample scale
            geom = (H, phase_hop, amp_hop, ts_hop, Fa, 1, Tout, has_amp)
            S = _HarmonicOsc._run(lib, "golf_harmonic_osc_fwd_f32", phase, amp if has_amp else None, None, hscale, geom,
                                  phase_offset, ctx.po, initial_phase)
            g_ts = upsample_adjoint(g_out[:, :Tout] * S, ts_hop, Fs)
        return g_phase, g_amp, g_ts, None, None, None, None, None, g_po, None, g_ip


def upsample_adjoint(v: torch.Tensor, hop: int, F: int) -> torch.Tensor:
    """Adjoint of linear_upsample along dim 1: v (B,T) with T <= (F-1)*hop+1 -> (B,F)."""
    B, T = v.shape
    out = v.new_zeros(B, F)
    if hop == 1 or F == 1:
        out[:, :T] = v
        return out
    n = torch.arange(T, device=v.device)
    f = torch.clamp(torch.div(n, hop, rounding_mode="floor"), max=F - 2)
    w = (n - f * hop).to(v.dtype) / hop
    out.index_add_(1, f, v * (1 - w))
    out.index_add_(1, f + 1, v * w)
    return out


def harmonic_osc(phase, H: int, phase_hop: int = 1, amp=None, amp_hop: int = 1, tscale=None, ts_hop: int = 1,
                 hscale=None, phase_offset=None, po_hop: int = 1, initial_phase=None) -> torch.Tensor:
    """out[t] = sum_h [h p(t) < 0.5] * up(amp)[t,h] * up(tscale)[t] * hscale[h]
                      * sin(2 pi (h (cumsum(p)[t] + up(phase_offset)[t]) + initial_phase[b,h])),   p = up(phase);
    differentiable w.r.t. ``amp``, ``phase``, ``tscale``, ``phase_offset`` and ``initial_phase`` (reference models/synth.py:403-446)."""
    return _HarmonicOsc.apply(phase, amp, tscale, hscale, H, phase_hop, amp_hop, ts_hop, phase_offset, int(po_hop),
                              initial_phase)


# ------------------------------------------------------------------------------------------------
# frame-wise all-pole synthesis as a cascade of biquads (reference models/lpc.py:94-131)
# ------------------------------------------------------------------------------------------------
class _BiquadFramesOLA(torch.autograd.Function):
    @staticmethod
    @_amp_fwd
    def forward(ctx, ex, gain, biquads, window, hop, pad, frame_gain):
        _lib.require_device(ex, gain, biquads, window)
        lib = _lib.load()
        ex = _rows(ex)
        gain, biquads, window = gain.contiguous(), biquads.contiguous(), window.contiguous()
        B, Tx0 = ex.shape
        F, K = biquads.shape[1], biquads.shape[2]
        W = window.numel()
        Tx = Tx0 if frame_gain else min(Tx0, (F - 1) * hop + 1)
        nfr = (Tx + 2 * pad - W) // hop + 1
        if nfr < 1 or nfr > F:
            raise _lib.GolfError(f"biquad_frames_ola: {nfr} frames for {F} coefficient frames")
        Ty = (nfr - 1) * hop + W - 2 * pad
        y = torch.empty(B, Ty, dtype=torch.float32, device=ex.device)
        ws = _workspace(B * nfr * W * 4 + 256, ex.device)
        rc = lib.golf_biquad_frames_ola_fwd_f32(ex.data_ptr(), ex.stride(0), gain.data_ptr(), biquads.data_ptr(),
                                                window.data_ptr(), y.data_ptr(), y.stride(0), B, Tx, F, K, hop, W, pad,
                                                1 if frame_gain else 0, Ty, ws.data_ptr(), ws.numel(), _lib.stream_ptr())
        _lib.check(rc, "golf_biquad_frames_ola_fwd_f32")
        ctx.geom = (hop, pad, bool(frame_gain), Tx, nfr, Ty)
        ctx.save_for_backward(ex, gain, biquads, window)
        return y

    @staticmethod
    @_amp_bwd
    def backward(ctx, gy):
        ex, gain, biquads, window = ctx.saved_tensors
        hop, pad, frame_gain, Tx, nfr, Ty = ctx.geom
        lib = _lib.load()
        B, Tx0 = ex.shape
        F, K = biquads.shape[1], biquads.shape[2]
        W = window.numel()
        # the overlap-add of the window (the normaliser of the forward), on the output samples
        norm = torch.nn.functional.conv_transpose1d(window.new_ones(1, 1, nfr), window.view(1, 1, W), stride=hop)
        norm = norm.view(-1)[pad: pad + Ty]
        gq = _rows((gy.float() / norm).contiguous())
        g_ex = torch.zeros(B, Tx0, dtype=torch.float32, device=ex.device)
        g_bq = torch.empty_like(biquads)
        g_gain_f = torch.empty_like(gain)
        gxe = torch.empty(B, Tx, dtype=torch.float32, device=ex.device)
        ws = _workspace(lib.golf_biquad_frames_bwd_workspace_bytes(B, Tx, F, K, hop, W, pad), ex.device)
        rc = lib.golf_biquad_frames_ola_bwd_f32(gq.data_ptr(), gq.stride(0), ex.data_ptr(), ex.stride(0), gain.data_ptr(),
                                                biquads.data_ptr(), window.data_ptr(), g_ex.data_ptr(), g_ex.stride(0),
                                                g_gain_f.data_ptr(), g_bq.data_ptr(), gxe.data_ptr(), B, Tx, F, K, hop, W,
                                                pad, 1 if frame_gain else 0, Ty, ws.data_ptr(), ws.numel(),
                                                _lib.stream_ptr())
        _lib.check(rc, "golf_biquad_frames_ola_bwd_f32")
        g_gain = g_gain_f if frame_gain else upsample_adjoint(gxe, hop, F)
        return g_ex, g_gain, g_bq, None, None, None, None


def biquad_frames_ola(ex: torch.Tensor, gain: torch.Tensor, biquads: torch.Tensor, window: torch.Tensor, hop: int,
                      pad: int = None, frame_gain: bool = True) -> torch.Tensor:
    """ex (B,Tx), gain (B,F), biquads (B,F,K,3) -> (B,Ty): every frame through its K sections 1/(a0+a1 z^-1+a2 z^-2),
    windowed overlap-add.  ``pad`` defaults to (W-hop)//2 and ``frame_gain`` to True (BatchSecondOrderLPCSynth);
    pad = W//2, frame_gain=False is LTVMinimumPhaseFilter's convention.  Differentiable w.r.t. ex, gain and biquads
    (custom HIP backward: the cascade's adjoint is the reversed cascade on the reversed signal)."""
    W = window.numel()
    pad = (W - hop) // 2 if pad is None else pad
    return _BiquadFramesOLA.apply(ex, gain, biquads, window, int(hop), int(pad), bool(frame_gain))
