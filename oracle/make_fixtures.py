#!/usr/bin/env python
"""Generate tests/golden/*.npz by running the REFERENCE'S OWN glue code (TEST INFRASTRUCTURE).

Runs ONLY in the build container (needs /root/reference; the GPU box never has it).  Usage:

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_fixtures.py

What is the reference's code and what is restated
-------------------------------------------------
/root/reference/models/{utils,ctrl,lpc,synth,noise,filters,sf}.py are imported verbatim.  Their
third-party imports are absent from this image, so ``sys.modules`` stubs
provide them:

* ``models.audiotensor.AudioTensor``      := the reference's own ``models.utils.LegacyAudioTensor``
  (the submodule directory is empty; .gitmodules:4-6).
* ``torchlpc.sample_wise_lpc``            := explicit float64 torch loop of its difference equation
  (autograd-able, so reference gradients are captured too).
* ``torchaudio.functional.lfilter``       := scipy.signal.lfilter per row (float64, independent witness).
* ``kazane.Decimate``                     := recorder: stores its input (the pre-decimation signal),
  returns x[..., ::q].  kazane's taps are unknown => decimation parity is unpinned.
* ``torchaudio.transforms.Spectrogram`` / ``InverseSpectrogram`` := torch.stft / torch.istft with torchaudio's
  documented arguments and defaults (only g20 — encoder / loss — and g21 — cepstral filter — ever call them).
* ``torch_fftconv``, ``diffsptk``, ``pyworld`` := inert placeholders
  (only needed so that ``import models.filters`` succeeds; never on the measured path).

Every array written is an input or an output of reference code — no reference source text.
"""
import hashlib
import os
import sys
import types

sys.dont_write_bytecode = True
os.environ["PYTHONDONTWRITEBYTECODE"] = "1"

import numpy as np
import scipy.signal
import torch
import torch.nn as nn

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


# ----------------------------------------------------------------------------- stubs
def _mod(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def sample_wise_lpc(x, a):
    B, T = x.shape
    M = a.shape[2]
    hist = [x.new_zeros(B) for _ in range(M)]  # hist[i] = y[t-1-i]
    ys = []
    for t in range(T):
        acc = x[:, t]
        for i in range(M):
            acc = acc - a[:, t, i] * hist[i]
        ys.append(acc)
        hist = [acc] + hist[:-1]
    return torch.stack(ys, 1)


def lfilter(x, a_coeffs, b_coeffs, clamp=True, batching=True):
    if x.requires_grad or a_coeffs.requires_grad:
        # autograd-able float64 loop of the same direct-form difference equation (used only to capture the
        # reference's gradients through its own glue; b = [1,0,...] at every call site of this path)
        a0 = a_coeffs[:, :1]
        an, bn = a_coeffs / a0, b_coeffs / a0
        R, W = x.shape
        K = an.shape[1]
        ys = []
        for n in range(W):
            acc = sum(bn[:, k] * x[:, n - k] for k in range(min(K, n + 1)))
            for k in range(1, min(K, n + 1)):
                acc = acc - an[:, k] * ys[n - k]
            ys.append(acc)
        return torch.stack(ys, 1)
    xn = x.detach().double().numpy()
    an = a_coeffs.detach().double().numpy()
    bn = b_coeffs.detach().double().numpy()
    out = np.stack([scipy.signal.lfilter(bn[r], an[r], xn[r]) for r in range(xn.shape[0])])
    return torch.from_numpy(out).to(x.dtype)


class _Dummy(nn.Module):
    def __init__(self, *a, **k):
        super().__init__()


class Spectrogram(nn.Module):
    """torchaudio.transforms.Spectrogram as documented: win_length = n_fft, hop = win_length // 2, periodic Hann,
    power 2 (None: complex STFT), centre-padded by reflection, unnormalised; ``window`` is a registered buffer."""

    def __init__(self, n_fft=400, win_length=None, hop_length=None, pad=0, window_fn=torch.hann_window, power=2.0,
                 normalized=False, wkwargs=None, center=True, pad_mode="reflect", onesided=True):
        super().__init__()
        assert pad == 0 and not normalized
        self.n_fft = n_fft
        self.win_length = n_fft if win_length is None else win_length
        self.hop_length = self.win_length // 2 if hop_length is None else hop_length
        self.power, self.center, self.pad_mode, self.onesided = power, center, pad_mode, onesided
        self.register_buffer("window", window_fn(self.win_length, **(wkwargs or {})))

    def forward(self, x):
        x = x.as_tensor() if hasattr(x, "as_tensor") else x
        z = torch.stft(x, self.n_fft, self.hop_length, self.win_length, self.window.to(x.dtype), center=self.center,
                       pad_mode=self.pad_mode, normalized=False, onesided=self.onesided, return_complex=True)
        if self.power is None:
            return z
        return z.abs() if self.power == 1 else z.abs().pow(self.power)


class InverseSpectrogram(nn.Module):
    """torchaudio.transforms.InverseSpectrogram as documented (torch.istft with the same window conventions)."""

    def __init__(self, n_fft=400, win_length=None, hop_length=None, pad=0, window_fn=torch.hann_window,
                 normalized=False, wkwargs=None, center=True, pad_mode="reflect", onesided=True):
        super().__init__()
        assert pad == 0 and not normalized
        self.n_fft = n_fft
        self.win_length = n_fft if win_length is None else win_length
        self.hop_length = self.win_length // 2 if hop_length is None else hop_length
        self.center, self.onesided = center, onesided
        self.register_buffer("window", window_fn(self.win_length, **(wkwargs or {})))

    def forward(self, z, length=None):
        return torch.istft(z, self.n_fft, self.hop_length, self.win_length, self.window.to(z.real.dtype),
                           center=self.center, normalized=False, onesided=self.onesided, length=length,
                           return_complex=False)


class Decimate(nn.Module):
    last_input = None

    def __init__(self, q=2, *a, **k):
        super().__init__()
        self.q = q
        self.register_buffer("kernel", torch.ones(1, 1, 1))  # must be a buffer (synth.py:209-211)

    def forward(self, x):
        Decimate.last_input = x.detach().clone()
        Decimate.last_input_live = x          # still attached to the autograd graph (g23: gradients of the oversampled signal)
        return x[..., :: self.q]


_mod("pyworld", dio=lambda *a, **k: None)
_mod("torchlpc", sample_wise_lpc=sample_wise_lpc)
_mod("torchaudio")
def melscale_fbanks(n_freqs, f_min, f_max, n_mels, sample_rate, norm=None, mel_scale="htk"):
    """torchaudio.functional.melscale_fbanks as published (HTK scale, no normalisation): triangular filters between
    n_mels + 2 points equally spaced in mel, on linspace(0, sample_rate // 2, n_freqs)."""
    assert norm is None and mel_scale == "htk"
    all_freqs = np.linspace(0, sample_rate // 2, n_freqs)
    mel = lambda f: 2595.0 * np.log10(1.0 + f / 700.0)
    f_pts = 700.0 * (10.0 ** (np.linspace(mel(f_min), mel(f_max), n_mels + 2) / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts[None, :] - all_freqs[:, None]
    fb = np.maximum(0.0, np.minimum(-slopes[:, :-2] / f_diff[:-1], slopes[:, 2:] / f_diff[1:]))
    return torch.from_numpy(fb).float()


_mod("torchaudio.functional", lfilter=lfilter, melscale_fbanks=melscale_fbanks)
_mod("torchaudio.transforms", Spectrogram=Spectrogram, InverseSpectrogram=InverseSpectrogram)
_mod("torch_fftconv")
_mod("torch_fftconv.functional", fft_conv1d=torch.nn.functional.conv1d)
_mod("diffsptk", MLSA=_Dummy, MelCepstralAnalysis=_Dummy, MelGeneralizedCepstrumToSpectrum=_Dummy,
     PQMF=_Dummy, IPQMF=_Dummy)
_mod("diffsptk.functional", lsp2lpc=lambda *a, **k: None)
_mod("kazane", Decimate=Decimate)

sys.path.insert(0, REF)
import models.utils as ru  # noqa: E402

_mod("models.audiotensor", AudioTensor=ru.LegacyAudioTensor)
import models.ctrl as rctrl  # noqa: E402,F401
import models.filters as rf  # noqa: E402
import models.noise as rn  # noqa: E402
import models.sf as rsf  # noqa: E402
import models.synth as rs  # noqa: E402

AT = ru.LegacyAudioTensor
torch.manual_seed(2434)
rng = np.random.default_rng(2434)


def save(name, **arrs):
    os.makedirs(OUT, exist_ok=True)
    arrs = {k: (v.detach().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in arrs.items()}
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **arrs)
    print(f"wrote {name}.npz:", {k: v.shape for k, v in arrs.items()})


def smooth_lpc(B, F, M, scale=0.5, step=0.02):
    """SURVEY §8d smooth random-walk logits -> stable, slowly varying direct-form a."""
    base = rng.normal(0, scale, (B, 1, M))
    walk = np.cumsum(rng.normal(0, step, (B, F, M)), axis=1)
    logits = torch.from_numpy((base + walk).astype(np.float32)).double()
    return logits, ru.rc2lpc(torch.tanh(logits))


# ----------------------------------------------------------------------------- g1 rc2lpc
rc = torch.tanh(torch.from_numpy(rng.normal(0, 1, (2, 3, 22)).astype(np.float32)).double())
save("g1_rc2lpc", rc=rc, lpc=ru.rc2lpc(rc), rc1=rc[..., :1], lpc1=ru.rc2lpc(rc[..., :1]))

# ----------------------------------------------------------------------------- g2 biquads
logits = torch.from_numpy(rng.normal(0, 1.5, (2, 3, 11, 2)).astype(np.float32)).double()
d = {"logits": logits}
for t in ("coef", "conj", "real"):
    bq = ru.get_logits2biquads(t)(logits)  # default max_abs_pole 0.99 (biquads.py:10)
    d["bq_" + t] = bq
    d["lpc_" + t] = ru.biquads2lpc(bq)
bq = ru.get_logits2biquads("coef", 0.9)(logits)
d["bq_coef_09"] = bq
save("g2_biquads", **d)

# ----------------------------------------------------------------------------- g3 LF tables
Rd = torch.exp(torch.linspace(np.log(0.3), np.log(2.7), 100))
v2 = ru.get_transformed_lf_v2(Rd, points=2048)
v1_rows = {f"v1_row{r}": ru.get_transformed_lf(R_d=Rd[r], points=2048) for r in (0, 49, 99)}
osc_v2 = rs.IndexedGlottalFlowTable(table_size=100, table_type="derivative", normalize_method="constant_power",
                                    align_peak=True, lf_v2=True, points=2048)
osc_v1 = rs.IndexedGlottalFlowTable(table_size=100, table_type="derivative", normalize_method="constant_power",
                                    align_peak=True, lf_v2=False, T_0=5.0, n_iter_eps=5, n_iter_a=100, points=2048)
osc_flow = rs.IndexedGlottalFlowTable(table_size=12, table_type="flow", normalize_method="peak",
                                      align_peak=True, lf_v2=True, points=256)
osc_none = rs.IndexedGlottalFlowTable(table_size=12, table_type="derivative", normalize_method=None,
                                      align_peak=False, lf_v2=False, points=256)
tb2 = osc_v2.table.numpy()
tb1 = osc_v1.table.numpy()
save("g3_lf_tables", Rd=Rd, v2_rows=v2[[0, 49, 99]], **v1_rows,
     table_v2_rows=tb2[[0, 1, 49, 98, 99]], table_v1_rows=tb1[[0, 1, 49, 98, 99]],
     table_v2_sha256=np.frombuffer(hashlib.sha256(tb2.tobytes()).digest(), dtype=np.uint8),
     table_v2_colsum=tb2.astype(np.float64).sum(0), table_v1_colsum=tb1.astype(np.float64).sum(0),
     table_v2_argmin=tb2.argmin(1), table_v1_argmin=tb1.argmin(1),
     table_flow=osc_flow.table, table_none=osc_none.table, Rd_small=osc_flow.R_d_values)

# ----------------------------------------------------------------------------- g4 upsample / mixed hop
z = torch.from_numpy(rng.normal(0, 1, (2, 100)).astype(np.float32)).double()
x10 = AT(z, hop_length=10)
x1 = x10.reduce_hop_length()
x2 = x10.reduce_hop_length(5)
x3 = x1 + x2 * x10
z3 = torch.from_numpy(rng.normal(0, 1, (2, 7, 3)).astype(np.float32)).double()
a3 = AT(z3, hop_length=4).reduce_hop_length()
ex = AT(torch.from_numpy(rng.normal(0, 1, (2, 40)).astype(np.float32)).double(), 1)
g = AT(z3[..., 0], hop_length=4)
save("g4_upsample", z=z, up10=x1.as_tensor(), up5=x2.as_tensor(), up5_hop=x2.hop_length, mixed=x3.as_tensor(),
     z3=z3, up3=a3.as_tensor(), ex=ex.as_tensor(), ex_times_g=(ex * g).as_tensor())

# ----------------------------------------------------------------------------- g5 wavetable generate
ph = torch.from_numpy(rng.uniform(0, 1, (2, 333)).astype(np.float32)).double()
tables = torch.from_numpy(rng.normal(0, 1, (2, 8, 64)).astype(np.float32)).double()
out = rs.GlottalFlowTable.generate(AT(ph, 1), AT(tables, 48))
tables_short = tables[:, :5]  # forces replicate-padding (synth.py:141-146)
out_short = rs.GlottalFlowTable.generate(AT(ph, 1), AT(tables_short, 48))
save("g5_generate", phase=ph, tables=tables, hop_t=48, out=out.as_tensor(), out_short=out_short.as_tensor())


# ----------------------------------------------------------------------------- g6 oscillator forward
def dyadic(shape, bits, lo, hi):
    """values k * 2^-bits so that the reference's float32 cumsum (synth.py:250-251) is exact."""
    k = rng.integers(int(lo * 2**bits), int(hi * 2**bits), shape)
    return torch.from_numpy(k / 2.0**bits).float()  # the reference oscillator only runs in float32 (synth.py:251)


d = {}
for name, os_, eq in (("os1", 1, False), ("os1_eq", 1, True), ("os4_eq", 4, True)):
    osc = rs.IndexedGlottalFlowTable(table_size=7, table_type="derivative", normalize_method="constant_power",
                                     align_peak=True, lf_v2=True, points=16, oversampling=os_, equal_energy=eq)
    d[name + "_table"] = osc.table
    # training style: phase hop 1, weight hop 16
    phase = dyadic((2, 64), 10, 0.01, 0.12)
    w = torch.from_numpy(rng.uniform(0, 1, (2, 5)).astype(np.float32))
    w[0, 0] = 0.0
    w[1, -1] = 1.0
    Decimate.last_input = None
    y = osc(AT(phase, 1), AT(w, 16))
    d[name + "_phase"], d[name + "_w"] = phase, w
    d[name + "_out"] = y.as_tensor()
    if os_ > 1:
        d[name + "_pre"] = Decimate.last_input
    # test_rtf style: phase hop 8, weight hop 32, too few table frames -> replicate pad
    phase2 = dyadic((2, 9), 10, 0.01, 0.12)
    w2 = torch.from_numpy(rng.uniform(0, 1, (2, 3)).astype(np.float32))
    Decimate.last_input = None
    y2 = osc(AT(phase2, 8), AT(w2, 32))
    d[name + "_phase2"], d[name + "_w2"] = phase2, w2
    d[name + "_out2"] = y2.as_tensor()
    if os_ > 1:
        d[name + "_pre2"] = Decimate.last_input
save("g6_oscillator", **d)

# ----------------------------------------------------------------------------- g7 frame-wise (ff) forward
B, F, hop, W, M = 2, 9, 8, 32, 4
_, a = smooth_lpc(B, F, M)
gain = torch.exp(torch.from_numpy(rng.normal(-1, 0.3, (B, F)).astype(np.float32)).double())
ex = torch.from_numpy(rng.normal(0, 1, (B, (F - 1) * hop + 5)).astype(np.float32)).double()
d = dict(ex=ex, gain=gain, a=a, hop=hop, W=W)
for centred in (True, False):
    m = rf.LTVMinimumPhaseFilter(window="hanning", window_length=W, centred=centred, lpc_order=M)
    m._kernel = m._kernel.double()
    y = m(AT(ex, 1), AT(gain, hop), AT(a, hop))
    d["y_centred" if centred else "y_uncentred"] = y.as_tensor()
# W == 2*hop edge and a wider M
m = rf.LTVMinimumPhaseFilter(window="hanning", window_length=16, centred=True, lpc_order=M)
m._kernel = m._kernel.double()
d["y_w16"] = m(AT(ex, 1), AT(gain, hop), AT(a, hop)).as_tensor()
# normaliser for the config shape (W=960, hop=240, F=200): OLA of ones, filters.py:171-177
win = torch.hann_window(960).double()
nfr = 200
full = (nfr - 1) * 240 + 960
norm = torch.zeros(full, dtype=torch.float64)
for f in range(nfr):
    norm[f * 240 : f * 240 + 960] += win
norm = norm[480 : full - 480]
mcfg = rf.LTVMinimumPhaseFilter(window="hanning", window_length=960, centred=True, lpc_order=2)
mcfg._kernel = mcfg._kernel.double()
ycfg = mcfg(AT(torch.ones(1, 48000, dtype=torch.float64), 1), AT(torch.ones(1, 200, dtype=torch.float64), 240),
            AT(torch.zeros(1, 200, 2, dtype=torch.float64), 240)).as_tensor()
d.update(norm_cfg_head=norm[:300], norm_cfg_tail=norm[-300:], norm_cfg_mid=norm[20000:20010], norm_cfg_len=norm.shape[0],
         y_cfg_ones_len=ycfg.shape[1], y_cfg_ones_maxdev=(ycfg - 1).abs().max(), window_f32=torch.hann_window(W))
save("g7_framewise", **d)

# ----------------------------------------------------------------------------- g8 sample-wise (ss) forward
B, F, hop, M = 2, 6, 8, 4
_, a = smooth_lpc(B, F, M)
gain = torch.exp(torch.from_numpy(rng.normal(-1, 0.3, (B, F)).astype(np.float32)).double())
ex = torch.from_numpy(rng.normal(0, 1, (B, 50)).astype(np.float32)).double()  # longer than (F-1)*hop+1 = 41
ex_short = ex[:, :30]
mss = rf.LTVMinimumPhaseFilterPrecise(lpc_order=M)
y = mss(AT(ex, 1), AT(gain, hop), AT(a, hop)).as_tensor()
y_short = mss(AT(ex_short, 1), AT(gain, hop), AT(a, hop)).as_tensor()
d = dict(ex=ex, gain=gain, a=a, hop=hop, y=y, y_short=y_short)
# mid-size, config order, used for fp32 kernels: B=3, F=12, hop=240, M=22
B2, F2, hop2, M2 = 3, 12, 240, 22
logits2, a2 = smooth_lpc(B2, F2, M2)
gain2 = torch.exp(torch.from_numpy((-3 + np.cumsum(rng.normal(0, 0.05, (B2, F2)), 1)).astype(np.float32)).double())
ex2 = torch.from_numpy(rng.normal(0, 1, (B2, (F2 - 1) * hop2 + 1)).astype(np.float32)).double()
y2 = mss(AT(ex2, 1), AT(gain2, hop2), AT(a2, hop2)).as_tensor()
d.update(logits2=logits2, a2=a2, gain2=gain2, ex2=ex2, hop2=hop2, y2=y2)
# the ctrl transform of the module (filters.py:90-97): logits -> (gain, a)
lg = torch.from_numpy(rng.normal(-3, 0.1, (B2, F2)).astype(np.float32)).double()
(split, trsfm) = mss.ctrl(lambda s, t: (s, t))((), ())
gg, aa = trsfm[0](AT(lg, hop2), AT(logits2, hop2))
d.update(ctrl_log_gain=lg, ctrl_gain=gg.as_tensor(), ctrl_a=aa.as_tensor(), ctrl_split=np.array(split[0]))
save("g8_samplewise", **d)

# ----------------------------------------------------------------------------- g9 reverse (inverse filter)
mff = rf.LTVMinimumPhaseFilter(window="hanning", window_length=32, lpc_order=4)
B, F, hop, M = 2, 6, 8, 4
_, a = smooth_lpc(B, F, M)
gain = torch.exp(torch.from_numpy(rng.normal(-1, 0.3, (B, F)).astype(np.float32)).double())
ex = torch.from_numpy(rng.normal(0, 1, (B, 48)).astype(np.float32)).double()
tgt = torch.from_numpy(rng.normal(0, 1, (B, 48)).astype(np.float32)).double()
exg, e = mff.reverse(AT(ex, 1), AT(tgt, 1), AT(gain, hop), AT(a, hop))
save("g9_reverse", ex=ex, target=tgt, gain=gain, a=a, hop=hop, ex_gain=exg.as_tensor(), e=e.as_tensor())

# ----------------------------------------------------------------------------- g10 gradients of a-1
B, F, hop, M = 2, 5, 8, 4
_, a = smooth_lpc(B, F, M)
a = a.detach().clone().requires_grad_(True)
gain = torch.exp(torch.from_numpy(rng.normal(-1, 0.3, (B, F)).astype(np.float32)).double()).requires_grad_(True)
ex = torch.from_numpy(rng.normal(0, 1, (B, 40)).astype(np.float32)).double().requires_grad_(True)
y = mss(AT(ex, 1), AT(gain, hop), AT(a, hop)).as_tensor()
gy = torch.from_numpy(rng.normal(0, 1, tuple(y.shape)).astype(np.float32)).double()
(y * gy).sum().backward()
d = dict(ex=ex, gain=gain, a=a, hop=hop, y=y, gy=gy, g_ex=ex.grad, g_gain=gain.grad, g_a=a.grad)
# config-order case: B=2, F=4, hop=24, M=22
B, F, hop, M = 2, 4, 24, 22
_, a = smooth_lpc(B, F, M)
a = a.detach().clone().requires_grad_(True)
gain = torch.exp(torch.from_numpy(rng.normal(-1, 0.3, (B, F)).astype(np.float32)).double()).requires_grad_(True)
ex = torch.from_numpy(rng.normal(0, 1, (B, 80)).astype(np.float32)).double().requires_grad_(True)
y = mss(AT(ex, 1), AT(gain, hop), AT(a, hop)).as_tensor()
gy = torch.from_numpy(rng.normal(0, 1, tuple(y.shape)).astype(np.float32)).double()
(y * gy).sum().backward()
d.update(ex22=ex, gain22=gain, a22=a, hop22=hop, y22=y, gy22=gy, g_ex22=ex.grad, g_gain22=gain.grad, g_a22=a.grad)
save("g10_ss_grads", **d)


# ----------------------------------------------------------------------------- g11 SourceFilterSynth composition
class FixedNoise(rn.NoiseInterface):
    def __init__(self, noise):
        super().__init__(torch.distributions.Normal(0, 1))
        self.noise = noise

    def forward(self, ref, *args, **kwargs):
        return AT(self.noise[:, : ref.shape[1]], 1)


B, T, hop, M = 2, 129, 16, 6
F = (T - 1) // hop + 1  # 9
phase = dyadic((B, T), 10, 0.01, 0.08)
w = torch.from_numpy(rng.uniform(0, 1, (B, 3)).astype(np.float32))
noise = torch.from_numpy(rng.normal(0, 1, (B, T)).astype(np.float32)).double()
_, a = smooth_lpc(B, F, M)
gain = torch.exp(torch.from_numpy(rng.normal(-1, 0.3, (B, F)).astype(np.float32)).double())
osc = rs.IndexedGlottalFlowTable(table_size=7, table_type="derivative", normalize_method="constant_power",
                                 align_peak=True, lf_v2=True, points=16, oversampling=1, equal_energy=True)
dec = rsf.SourceFilterSynth(harm_oscillator=osc, noise_generator=FixedNoise(noise), noise_filter=rctrl.PassThrough(),
                            end_filter=rf.LTVMinimumPhaseFilterPrecise(lpc_order=M), room_filter=None,
                            subtract_harmonics=False)
y = dec(phase=AT(phase, 1), harm_oscillator_params=(AT(w, 64),), noise_generator_params=(), noise_filter_params=(),
        end_filter_params=(AT(gain, hop), AT(a, hop)))
voicing = torch.from_numpy((rng.uniform(0, 1, (B, F)) > 0.4).astype(np.float64) * rng.uniform(0.5, 1, (B, F)))
yv = dec(phase=AT(phase, 1), harm_oscillator_params=(AT(w, 64),), noise_generator_params=(), noise_filter_params=(),
         end_filter_params=(AT(gain, hop), AT(a, hop)), voicing=AT(voicing, hop))
save("g11_source_filter", phase=phase, w=w, w_hop=64, noise=noise, gain=gain, a=a, hop=hop, table=osc.table,
     y=y.as_tensor(), voicing=voicing, y_voiced=yv.as_tensor())


# ----------------------------------------------------------------------------- g12 control protocol
def golf_decoder(end_filter):
    return rsf.SourceFilterSynth(
        harm_oscillator=rs.DownsampledIndexedGlottalFlowTable(
            hop_rate=10, in_channels=64, oversampling=4, equal_energy=True, table_type="derivative",
            normalize_method="constant_power", align_peak=True, trainable=False, min_R_d=0.3, max_R_d=2.7,
            lf_v2=True, points=64),
        noise_generator=rn.StandardNormalNoise(),
        noise_filter=rf.LTVZeroPhaseFIRFilter(window="hanning", n_mag=256),
        end_filter=end_filter,
        room_filter=rf.LTIAcousticFilter(length=128, conv_method="direct"),
        subtract_harmonics=False)


d = {}
for name, ef in (("ss", rf.LTVMinimumPhaseFilterPrecise(lpc_order=22, lpc_parameterisation="rc2lpc")),
                 ("ff", rf.LTVMinimumPhaseFilter(window="hanning", window_length=960, lpc_order=22,
                                                 lpc_parameterisation="rc2lpc")),
                 ("ss_coef", rf.LTVMinimumPhaseFilterPrecise(lpc_order=22, lpc_parameterisation="coef",
                                                             max_abs_value=0.99))):
    dec = golf_decoder(ef)
    split_sizes, trsfms, keys = dec.split_sizes_and_trsfms
    d[name + "_split_sizes"] = np.array([",".join(map(str, s)) for s in split_sizes])
    d[name + "_keys"] = np.array(keys)
    d[name + "_state_dict_keys"] = np.array(sorted(dec.state_dict().keys()))
# downsampler ctrl (synth.py:297-340): h (B,200,64) hop 240 -> w (B,21) hop 2400
dec = golf_decoder(rf.LTVMinimumPhaseFilterPrecise(lpc_order=22))
osc = dec.harm_oscillator
h = torch.from_numpy(rng.normal(0, 1, (2, 200, 64)).astype(np.float32))
(wout,) = osc.ctrl(lambda s, t: (s, t))((), ())[1][0](AT(h, 240))
sd = {k: v for k, v in osc.model.state_dict().items()}
d.update(ds_h=h, ds_w=wout.as_tensor(), ds_w_hop=wout.hop_length,
         **{"ds_model." + k: v for k, v in sd.items()})
save("g12_ctrl_protocol", **d)
# ----------------------------------------------------------------------------- g13 zero-phase FIR noise filter
# LTVZeroPhaseFIRFilter.forward (filters.py:340-384) in float64, with autograd gradients.
def run_zp(B, T, F, hop, n_mag, window="hanning"):
    flt = rf.LTVZeroPhaseFIRFilter(window=window, conv_method="direct", n_mag=n_mag)
    ex = torch.from_numpy(rng.normal(0, 1, (B, T)).astype(np.float32)).double().requires_grad_(True)
    lm = torch.from_numpy(rng.normal(-1, 0.7, (B, F, n_mag)).astype(np.float32)).double().requires_grad_(True)
    y = flt(AT(ex, 1), AT(lm, hop)).as_tensor()
    gy = torch.from_numpy(rng.normal(0, 1, tuple(y.shape)).astype(np.float32)).double()
    (y * gy).sum().backward()
    kern = flt.windowing(flt.get_zero_phase_fir(lm.detach()))
    return dict(ex=ex, log_mag=lm, hop=hop, y=y, gy=gy, g_ex=ex.grad, g_log_mag=lm.grad, kernel=kern)


d = {}
# even n_fft (N = 16), more frames than the unfold yields, and fewer (kernel[:, :nfr] / unfolded[:, :F] truncation)
for tag, (B, T, F, hop, n_mag) in (("a", (2, 61, 9, 8, 9)), ("b", (2, 100, 5, 8, 9)), ("c", (1, 200, 6, 24, 33))):
    for k, v in run_zp(B, T, F, hop, n_mag).items():
        d[f"{tag}_{k}"] = v
for k, v in run_zp(2, 75, 8, 8, 9, window="hamming").items():
    d[f"h_{k}"] = v
save("g13_zero_phase_fir", **d)

# ----------------------------------------------------------------------------- g14 room filter + full GOLF-ss decoder
# LTIAcousticFilter.forward (filters.py:426-449) and the complete SourceFilterSynth of golf-precise.yaml at toy size
room = rf.LTIAcousticFilter(length=16, conv_method="direct").double()
with torch.no_grad():
    room.kernel.copy_(torch.from_numpy(rng.normal(0, 0.2, (15,))).double())
ex = torch.from_numpy(rng.normal(0, 1, (2, 50)).astype(np.float32)).double().requires_grad_(True)
y = room(AT(ex, 1)).as_tensor()
gy = torch.from_numpy(rng.normal(0, 1, tuple(y.shape)).astype(np.float32)).double()
(y * gy).sum().backward()
d = dict(room_ex=ex, room_kernel=room.kernel.detach(), room_y=y, room_gy=gy, room_g_ex=ex.grad,
         room_g_kernel=room.kernel.grad)

B, T, hop, M, n_mag = 2, 161, 16, 6, 9
F = (T - 1) // hop + 1  # 11
phase = dyadic((B, T), 10, 0.01, 0.08)
w = torch.from_numpy(rng.uniform(0, 1, (B, 3)).astype(np.float32))
noise = torch.from_numpy(rng.normal(0, 1, (B, T)).astype(np.float32)).double()
_, a = smooth_lpc(B, F, M)
gain = torch.exp(torch.from_numpy(rng.normal(-1, 0.3, (B, F)).astype(np.float32)).double())
lm = torch.from_numpy(rng.normal(-1, 0.7, (B, F, n_mag)).astype(np.float32)).double()
osc = rs.IndexedGlottalFlowTable(table_size=7, table_type="derivative", normalize_method="constant_power",
                                 align_peak=True, lf_v2=True, points=16, oversampling=1, equal_energy=True)
room2 = rf.LTIAcousticFilter(length=8, conv_method="direct").double()
with torch.no_grad():
    room2.kernel.copy_(torch.from_numpy(rng.normal(0, 0.2, (7,))).double())
dec = rsf.SourceFilterSynth(harm_oscillator=osc, noise_generator=FixedNoise(noise),
                            noise_filter=rf.LTVZeroPhaseFIRFilter(window="hanning", n_mag=n_mag),
                            end_filter=rf.LTVMinimumPhaseFilterPrecise(lpc_order=M), room_filter=room2,
                            subtract_harmonics=False)
y = dec(phase=AT(phase, 1), harm_oscillator_params=(AT(w, 64),), noise_generator_params=(),
        noise_filter_params=(AT(lm, hop),), end_filter_params=(AT(gain, hop), AT(a, hop)))
d.update(phase=phase, w=w, w_hop=64, noise=noise, gain=gain, a=a, log_mag=lm, hop=hop, table=osc.table,
         room2_kernel=room2.kernel.detach(), y=y.as_tensor())
save("g14_room_and_full_decoder", **d)
# ----------------------------------------------------------------------------- g15 gradients of the ff filter (a-4)
d = {}
for tag, (B, F, hop, W, M, centred, Tx) in (("c", (2, 7, 8, 32, 4, True, None)), ("u", (2, 7, 8, 32, 4, False, None)),
                                            ("s", (1, 6, 8, 16, 3, True, 33))):
    _, a = smooth_lpc(B, F, M)
    a = a.detach().clone().requires_grad_(True)
    gain = torch.exp(torch.from_numpy(rng.normal(-1, 0.3, (B, F)).astype(np.float32)).double()).requires_grad_(True)
    Tx = Tx or (F - 1) * hop + 3
    ex = torch.from_numpy(rng.normal(0, 1, (B, Tx)).astype(np.float32)).double().requires_grad_(True)
    m = rf.LTVMinimumPhaseFilter(window="hanning", window_length=W, centred=centred, lpc_order=M)
    m._kernel = m._kernel.double()
    y = m(AT(ex, 1), AT(gain, hop), AT(a, hop)).as_tensor()
    gy = torch.from_numpy(rng.normal(0, 1, tuple(y.shape)).astype(np.float32)).double()
    (y * gy).sum().backward()
    d.update({f"{tag}_ex": ex, f"{tag}_gain": gain, f"{tag}_a": a, f"{tag}_hop": hop, f"{tag}_W": W,
              f"{tag}_centred": int(centred), f"{tag}_y": y, f"{tag}_gy": gy, f"{tag}_g_ex": ex.grad,
              f"{tag}_g_gain": gain.grad, f"{tag}_g_a": a.grad})
save("g15_ff_grads", **d)
# ----------------------------------------------------------------------------- g16 sample-wise zero-phase FIR filter
# LTVZeroPhaseFIRFilterPrecise.forward (filters.py:308-337) in float64, with autograd gradients
d = {}
for tag, (B, T, F, hop, n_mag) in (("a", (2, 57, 8, 8, 9)), ("b", (2, 40, 9, 8, 9)), ("c", (1, 150, 6, 24, 17))):
    flt = rf.LTVZeroPhaseFIRFilterPrecise(window="hanning", n_mag=n_mag)
    ex = torch.from_numpy(rng.normal(0, 1, (B, T)).astype(np.float32)).double().requires_grad_(True)
    lm = torch.from_numpy(rng.normal(-1, 0.7, (B, F, n_mag)).astype(np.float32)).double().requires_grad_(True)
    y = flt(AT(ex, 1), AT(lm, hop)).as_tensor()
    gy = torch.from_numpy(rng.normal(0, 1, tuple(y.shape)).astype(np.float32)).double()
    (y * gy).sum().backward()
    d.update({f"{tag}_ex": ex, f"{tag}_log_mag": lm, f"{tag}_hop": hop, f"{tag}_y": y, f"{tag}_gy": gy,
              f"{tag}_g_ex": ex.grad, f"{tag}_g_log_mag": lm.grad})
save("g16_zero_phase_fir_precise", **d)
# ----------------------------------------------------------------------------- g17 gradients of reverse() (a-5)
mff = rf.LTVMinimumPhaseFilter(window="hanning", window_length=32, lpc_order=4)
d = {}
for tag, (B, F, hop, M, Ty) in (("a", (2, 6, 8, 4, 48)), ("b", (2, 5, 24, 22, 97))):
    _, a = smooth_lpc(B, F, M)
    a = a.detach().clone().requires_grad_(True)
    gain = torch.exp(torch.from_numpy(rng.normal(-1, 0.3, (B, F)).astype(np.float32)).double())
    ex = torch.from_numpy(rng.normal(0, 1, (B, Ty)).astype(np.float32)).double()
    tgt = torch.from_numpy(rng.normal(0, 1, (B, Ty)).astype(np.float32)).double().requires_grad_(True)
    _, e = mff.reverse(AT(ex, 1), AT(tgt, 1), AT(gain, hop), AT(a, hop))
    e = e.as_tensor()
    ge = torch.from_numpy(rng.normal(0, 1, tuple(e.shape)).astype(np.float32)).double()
    (e * ge).sum().backward()
    d.update({f"{tag}_target": tgt, f"{tag}_a": a, f"{tag}_hop": hop, f"{tag}_e": e, f"{tag}_g_e": ge,
              f"{tag}_g_target": tgt.grad, f"{tag}_g_a": a.grad})
save("g17_reverse_grads", **d)
# ----------------------------------------------------------------------------- g18 harmonic oscillator bank (a-11)
# HarmonicOscillator / AdditiveSynthesizer / SawToothOscillator / AdditivePulseTrain (synth.py:403-547).  The reference
# cumsums in float32 (synth.py:426-427): dyadic phase increments make that exact for these short signals, and
# power-of-two hops keep the interpolated h*phase exact too, so that harmonics sitting exactly AT Nyquist (h*p == 0.5)
# are masked identically by every implementation instead of by the rounding of one particular interpolation routine.
d = {}
H = 6
for tag, (B, Tp, ph, Fa, ah) in (("t", (2, 97, 1, 7, 16)), ("r", (2, 7, 16, 4, 32)), ("q", (1, 25, 4, 3, 32))):
    phase = dyadic((B, Tp), 10, 0.01, 0.12)
    # float32 throughout: the reference casts the harmonic phases to float32 (synth.py:427), a float64 amplitude
    # tensor does not even type-check in its matmul
    amp = torch.from_numpy(rng.uniform(0, 1, (B, Fa, H)).astype(np.float32)).requires_grad_(True)
    y = rs.HarmonicOscillator()(AT(phase, ph), AT(amp, ah)).as_tensor()
    gy = torch.from_numpy(rng.normal(0, 1, tuple(y.shape)).astype(np.float32))
    (y * gy).sum().backward()
    d.update({f"{tag}_phase": phase, f"{tag}_phase_hop": ph, f"{tag}_amp": amp, f"{tag}_amp_hop": ah, f"{tag}_y": y,
              f"{tag}_gy": gy, f"{tag}_g_amp": amp.grad})
    d[f"{tag}_additive"] = rs.AdditiveSynthesizer(num_harmonics=H)(AT(phase, ph), AT(amp.detach(), ah)).as_tensor()
    d[f"{tag}_saw"] = rs.SawToothOscillator(num_harmonics=H)(AT(phase, ph)).as_tensor()
    d[f"{tag}_pulse"] = rs.AdditivePulseTrain(num_harmonics=H)(AT(phase, ph)).as_tensor()
add_syn = rs.AdditiveSynthesizer(num_harmonics=H)
(split, trsfm) = add_syn.ctrl(lambda s_, t_: (s_, t_))((), ())
lg = torch.from_numpy(rng.normal(-1, 0.3, (2, 7)).astype(np.float32))
lo = torch.from_numpy(rng.normal(0, 1, (2, 7, H)).astype(np.float32))
(amp_c,) = trsfm[0](AT(lg, 16), AT(lo, 16))
# (V1AdditiveSynthesizer's transform calls .sum() on the AudioTensor, which models.utils.LegacyAudioTensor lacks: it
# cannot be captured with the submodule absent)
d.update(ctrl_log_gain=lg, ctrl_logits=lo, ctrl_amp=amp_c.as_tensor(), ctrl_split=np.array(split[0]))
save("g18_harmonic_oscillators", **d)
# ----------------------------------------------------------------------------- g19 cascaded-biquad frame synthesis (a-6)
# BatchSecondOrderLPCSynth.forward (models/lpc.py:94-131): the reference's statement of the cascaded-biquad all-pole
# filter; biquads from its own get_logits2biquads("coef").
import models.lpc as rlpc
d = {}
for tag, (B, F, K, hop, W, T) in (("a", (2, 9, 3, 8, 32, 64)), ("b", (1, 6, 5, 16, 48, 80))):
    logits = torch.from_numpy(rng.normal(0, 1, (B, F, K, 2)).astype(np.float32)).double()
    bq = ru.get_logits2biquads("coef", 0.95)(logits)                      # (B,F,K,3), a0 = 1
    gain = torch.exp(torch.from_numpy(rng.normal(-1, 0.3, (B, F)).astype(np.float32)).double())
    ex = torch.from_numpy(rng.normal(0, 1, (B, T)).astype(np.float32)).double()
    syn = rlpc.BatchSecondOrderLPCSynth(hop_length=hop, window_size=W, window="hanning").double()
    y = syn(ex, gain, bq)
    d.update({f"{tag}_ex": ex, f"{tag}_gain": gain, f"{tag}_biquads": bq, f"{tag}_hop": hop, f"{tag}_W": W, f"{tag}_y": y,
              f"{tag}_lpc": ru.biquads2lpc(bq)})
save("g19_biquad_cascade", **d)
# ----------------------------------------------------------------------------- g20 encoder, encoder interface, loss (f-3)
# UNetEncoder.forward (models/unet.py:86-224), VocoderParameterEncoderInterface (models/enc.py:33-100) on the toy
# GOLF-ss decoder of g14, MSSLoss (loss/spec.py:32-67).  float32 like the reference runs them.
_mod("models.lru", LRU=_Dummy)
import models.enc as renc  # noqa: E402
import models.unet as runet  # noqa: E402,F401
import loss.spec as rloss  # noqa: E402

torch.manual_seed(20)
d = {}
enc_args = dict(n_fft=64, hop_length=16, channels=[4, 8], strides=[2, 2], lstm_hidden_size=8, num_layers=2)
iface = renc.VocoderParameterEncoderInterface(
    backbone_type="models.unet.UNetEncoder", learn_voicing=True, learn_f0=True, f0_min=60.0, f0_max=1000.0,
    split_sizes=dec.split_sizes_and_trsfms[0], trsfms=dec.split_sizes_and_trsfms[1],
    args_keys=dec.split_sizes_and_trsfms[2], **enc_args).float()
with torch.no_grad():
    iface.backbone.out_linear.weight.normal_(0, 0.3)
    iface.backbone.out_linear.bias.normal_(0, 0.1)
d["split_sizes"] = np.array([s for g_ in iface.split_sizes for s in g_])
d["group_lengths"] = np.array([len(g_) for g_ in iface.split_sizes])
d["args_keys"] = np.array(list(iface.args_keys))
for k, v in iface.state_dict().items():
    d["state0/" + k] = v.clone()
xs = [torch.from_numpy(rng.normal(0, 0.3, (3, 400)).astype(np.float32)) for _ in range(3)]
f0s = [torch.from_numpy(rng.uniform(80, 300, (3, 400)).astype(np.float32)) for _ in range(3)]
iface.train()
for i in range(2):          # two training-mode calls: running extrema and BatchNorm statistics evolve
    d[f"h_train{i}"] = iface.backbone(AT(xs[i], 1), f0=AT(f0s[i], 1)).as_tensor()
iface.eval()
d["h_eval"] = iface.backbone(AT(xs[2], 1), f0=AT(f0s[2], 1)).as_tensor()
params = iface(AT(xs[2], 1), f0=AT(f0s[2], 1))
for k, v in params.items():
    for j, t in enumerate(v if isinstance(v, tuple) else (v,)):
        d[f"param/{k}/{j}"] = t.as_tensor()
        d[f"param_hop/{k}/{j}"] = t.hop_length
for k, v in iface.state_dict().items():
    if "running" in k or "log_spec" in k or "num_batches" in k:
        d["state1/" + k] = v.clone()
d.update({f"x{i}": xs[i] for i in range(3)})
d.update({f"f0_{i}": f0s[i] for i in range(3)})
crit = rloss.MSSLoss([61, 127, 251], alpha=1.0, window="hanning", center=True)
pred = torch.from_numpy(rng.normal(0, 0.3, (2, 1500)).astype(np.float32)).requires_grad_(True)
true = torch.from_numpy(rng.normal(0, 0.3, (2, 1500)).astype(np.float32))
val = crit(pred, true)
val.backward()
d.update(loss_pred=pred, loss_true=true, loss_value=val, loss_g_pred=pred.grad,
         loss_hops=np.array([l_.spec.hop_length for l_ in crit.losses]))
save("g20_encoder_and_loss", **d)
# ----------------------------------------------------------------------------- g21 cepstral filter of the NHV baseline
# LTVCepFilter.forward (models/filters.py:559-623), both phase modes, with gradients (float64)
d = {}
for tag, phase_mode in (("zero", "zero"), ("min", "min")):
    flt = rf.LTVCepFilter(filter_order=24, n_fft=128, window="hanning", hop_length=32, phase=phase_mode).double()
    ex = torch.from_numpy(rng.normal(0, 1, (2, 640)).astype(np.float32)).double().requires_grad_(True)
    ceps = torch.from_numpy((rng.normal(0, 0.2, (2, 21, 25)) / (1 + np.arange(25))).astype(np.float32)).double()
    ceps.requires_grad_(True)
    y = flt(AT(ex, 1), AT(ceps, 32)).as_tensor()
    gy = torch.from_numpy(rng.normal(0, 1, tuple(y.shape)).astype(np.float32)).double()
    (y * gy).sum().backward()
    d.update({f"{tag}_ex": ex, f"{tag}_ceps": ceps, f"{tag}_y": y, f"{tag}_gy": gy, f"{tag}_g_ex": ex.grad,
              f"{tag}_g_ceps": ceps.grad})
save("g21_cep_filter", **d)
# ----------------------------------------------------------------------------- g22 spectral-envelope filter (WORLD baseline)
# DiffWorldSPFilter.forward (models/filters.py:717-760): the reference's glue (pinv().relu() of the filterbank, sqrt, STFT
# gain); the filterbank itself comes from the restated torchaudio formula above.
flt = rf.DiffWorldSPFilter(n_mels=12, n_fft=128, hop_length=32, f_min=0.0, f_max=4000.0, center=True, window="hanning",
                           sample_rate=8000, norm=None, mel_scale="htk")
ex = torch.from_numpy(rng.normal(0, 1, (2, 640)).astype(np.float32)).requires_grad_(True)
logmel = torch.from_numpy(rng.normal(-1, 0.5, (2, 21, 12)).astype(np.float32)).requires_grad_(True)
(split, trs) = flt.ctrl(lambda s_, t_: (s_, t_))((), ())
(mel_sp,) = trs[0](AT(logmel, 32))
y = flt(AT(ex, 1), mel_sp).as_tensor()
gy = torch.from_numpy(rng.normal(0, 1, tuple(y.shape)).astype(np.float32))
(y * gy).sum().backward()
save("g22_world_sp_filter", ex=ex, logmel=logmel, y=y, gy=gy, g_ex=ex.grad, g_logmel=logmel.grad, inv_fb=flt.fb,
     split=np.array(split[0]))
# ----------------------------------------------------------------------------- g23 oscillator surface (a-8 gradients, a-11)
# What the reference differentiates through its oscillators and the table oscillators no GOLF config uses:
# IndexedGlottalFlowTable with a trainable table / differentiable phase / phase_offset (models/synth.py:59-70,213-263),
# WeightedGlottalFlowTable (:266-294), WrappedPhaseDownsampledIndexedGlottalFlowTable (:343-375), PulseTrain (:507-523).
# The reference runs these in float32 (synth.py:251 forces .float()); phases are dyadic so that its fp32 cumsum is exact.
d = {}
for name, os_, eq, with_off in (("ix1", 1, True, True), ("ix4", 4, True, False)):
    osc = rs.IndexedGlottalFlowTable(table_size=7, table_type="derivative", normalize_method="constant_power",
                                     align_peak=True, lf_v2=True, points=16, oversampling=os_, equal_energy=eq,
                                     trainable=True)
    phase = dyadic((2, 64), 10, 0.01, 0.12).requires_grad_(True)
    w = torch.from_numpy(rng.uniform(0.02, 0.98, (2, 5)).astype(np.float32)).requires_grad_(True)
    n_fine = (64 - 1) * os_ + 1
    off = dyadic((2, n_fine), 10, 0.0, 1.0).requires_grad_(True) if with_off else None
    Decimate.last_input = None
    y = osc(AT(phase, 1), AT(w, 16), AT(off, 1) if with_off else None).as_tensor()
    sig = Decimate.last_input_live if os_ > 1 else y     # gradients are taken on the pre-decimation signal (kazane absent)
    gy = torch.from_numpy(rng.normal(0, 1, tuple(sig.shape)).astype(np.float32))
    (sig * gy).sum().backward()
    d.update({f"{name}_table": osc.table.detach().clone(), f"{name}_phase": phase.detach(), f"{name}_w": w.detach(),
              f"{name}_sig": sig.detach(), f"{name}_gy": gy, f"{name}_g_phase": phase.grad, f"{name}_g_w": w.grad,
              f"{name}_g_table": osc.table.grad})
    if with_off:
        d.update({f"{name}_off": off.detach(), f"{name}_g_off": off.grad})
# weighted table mix
osc = rs.WeightedGlottalFlowTable(table_size=7, table_type="derivative", normalize_method="constant_power",
                                  align_peak=True, lf_v2=True, points=16, trainable=True)
phase = dyadic((2, 64), 10, 0.01, 0.12).requires_grad_(True)
wl = torch.from_numpy(rng.normal(0, 1, (2, 5, 7)).astype(np.float32)).requires_grad_(True)
(split, trs) = osc.ctrl(lambda s_, t_: (s_, t_))((), ())
(wsm,) = trs[0](AT(wl, 16))
y = osc(AT(phase, 1), wsm).as_tensor()
gy = torch.from_numpy(rng.normal(0, 1, tuple(y.shape)).astype(np.float32))
(y * gy).sum().backward()
d.update(wt_table=osc.table.detach().clone(), wt_phase=phase.detach(), wt_logits=wl.detach(), wt_w=wsm.as_tensor().detach(),
         wt_out=y.detach(), wt_gy=gy, wt_g_phase=phase.grad, wt_g_logits=wl.grad, wt_g_table=osc.table.grad,
         wt_split=np.array(split[0]))
# phase at a coarser hop (test_rtf style) through the weighted oscillator, forward only
phase8 = dyadic((2, 9), 10, 0.01, 0.12)
w8 = torch.softmax(torch.from_numpy(rng.normal(0, 1, (2, 3, 7)).astype(np.float32)), 2)
d.update(wt_phase8=phase8, wt_w8=w8, wt_out8=osc(AT(phase8, 8), AT(w8, 32)).as_tensor().detach())
# wrapped-phase input
osc = rs.WrappedPhaseDownsampledIndexedGlottalFlowTable(hop_rate=2, in_channels=3, table_size=7, table_type="derivative",
                                                        normalize_method="constant_power", align_peak=True, lf_v2=True,
                                                        points=16)
wp = torch.from_numpy(rng.uniform(0, 1, (2, 64)).astype(np.float32)).requires_grad_(True)
w = torch.from_numpy(rng.uniform(0.02, 0.98, (2, 5)).astype(np.float32)).requires_grad_(True)
y = osc(AT(wp, 1), AT(w, 16)).as_tensor()
gy = torch.from_numpy(rng.normal(0, 1, tuple(y.shape)).astype(np.float32))
(y * gy).sum().backward()
d.update(wp_table=osc.table.clone(), wp_phase=wp.detach(), wp_w=w.detach(), wp_out=y.detach(), wp_gy=gy,
         wp_g_phase=wp.grad, wp_g_w=w.grad, wp_state_keys=np.array(sorted(osc.state_dict().keys())))
# pulse train
pt = rs.PulseTrain()
phase = dyadic((2, 200), 10, 0.01, 0.12)
d.update(pt_phase=phase, pt_out=pt(AT(phase, 1)).as_tensor())
phase8 = dyadic((2, 30), 10, 0.01, 0.12)
off8 = dyadic((2, 233), 10, 0.0, 1.0)
# (offset as a plain tensor: the stand-in AudioTensor type has no Tensor methods on indexed results, models/utils.py:41-56)
d.update(pt_phase8=phase8, pt_off8=off8, pt_out8=pt(AT(phase8, 8), off8).as_tensor())
save("g23_oscillator_surface", **d)
# ----------------------------------------------------------------------------- g24 NoiseBand (a-12), checkpoint key remaps and the
# biquads.py dump (f-4 on-disk formats)
import io
import contextlib

torch.manual_seed(7)                      # NoiseBand draws its band phases from torch's global generator (noise.py:199)
with contextlib.redirect_stdout(io.StringIO()):
    nb = rn.NoiseBand(n_filters=12, fs=24000, attenuation=50, normalize_noise_bands=True)
(split, trs) = nb.ctrl(lambda s_, t_: (s_, t_))((), ())
log_gain = torch.from_numpy(rng.normal(-2, 0.5, (2, 6, 12)).astype(np.float32)).requires_grad_(True)
ref_sig = AT(torch.zeros(2, 330), 1)
torch.manual_seed(11)
offs = torch.randint(0, nb.noise_bands.shape[1], (2, nb.noise_bands.shape[0]))     # the draw forward() makes first
torch.manual_seed(11)
y = nb(ref_sig, AT(log_gain, 64))
y = y.as_tensor() if hasattr(y, "as_tensor") else y
gy = torch.from_numpy(rng.normal(0, 1, tuple(y.shape)).astype(np.float32))
(y * gy).sum().backward()
d = dict(nb_band_centers=nb.band_centers, nb_noise_bands=nb.noise_bands, nb_split=np.array(split[0]),
         nb_log_gain=log_gain.detach(), nb_offsets=offs, nb_out=y.detach(), nb_gy=gy, nb_g_log_gain=log_gain.grad,
         nb_state_keys=np.array(sorted(nb.state_dict().keys())))
# ISMIR'23 -> Interspeech'24 head permutation (models/utils.py:12-38) on a random state dict
sd = {"encoder.backbone.out_linear.weight": torch.from_numpy(rng.normal(0, 1, (2 + 2 * 23 + 8, 5)).astype(np.float32)),
      "encoder.backbone.out_linear.bias": torch.from_numpy(rng.normal(0, 1, (2 + 2 * 23 + 8,)).astype(np.float32)),
      "encoder.backbone.norm.weight": torch.from_numpy(rng.normal(0, 1, (5,)).astype(np.float32))}
sd2 = ru.ismir2interspeech_ckpt(sd, lpc_order=22, h_size=8)
d.update({"ck_in/" + k: v for k, v in sd.items()})
d.update({"ck_out/" + k: v for k, v in sd2.items()})
# biquads.py get_biquads on a stand-in model (its imports need torchaudio / harm_and_noise, absent: inert placeholders)
_mod("harm_and_noise", loader=lambda *a, **k: None)
if "torchaudio" in sys.modules and not hasattr(sys.modules["torchaudio"], "load"):
    sys.modules["torchaudio"].load = lambda *a, **k: None
import biquads as rbq  # noqa: E402

bq_logits = torch.from_numpy(rng.normal(0, 1, (2, 7, 2 + 8 + 23 + 23)).astype(np.float32))
enc = types.SimpleNamespace(
    backbone=lambda h: bq_logits, split_sizes=((1,), (1,), (8,), (), (1, 22), (1, 22), ()),
    args_keys=("f0", "voicing_logits", "harm_oscillator_params", "noise_generator_params", "harm_filter_params",
               "noise_filter_params", "end_filter_params"),
    trsfms=(None, None, lambda h: (torch.sigmoid(h.mean(-1)),), None, None, None, None))
fake = types.SimpleNamespace(feature_trsfm=lambda x: x, encoder=enc)
res = rbq.get_biquads(fake, torch.zeros(1, 10))
d.update(bq_logits=bq_logits, bq_voicing=res[0], bq_harm_log_gain=res[1], bq_harm_biquads=res[2], bq_noise_log_gain=res[3],
         bq_noise_biquads=res[4], bq_table_select_weight=res[5])
save("g24_noiseband_ckpt_biquads", **d)
# ----------------------------------------------------------------------------- g26 gradients of the biquad cascade (a-6)
# BatchSecondOrderLPCSynth is differentiable through its K lfilter calls (models/lpc.py:115-118): the reference's own
# autograd gradients w.r.t. ex, gain and the section coefficients, in float64 through its own glue (the lfilter stand-in
# above is an autograd-able loop of the same difference equation).  Own generator: g1..g24 stay bit-identical.
rng26 = np.random.default_rng(26)
d = {}
for tag, (B, F, K, hop, W, T) in (("a", (2, 9, 3, 8, 32, 64)), ("b", (1, 6, 5, 16, 48, 80))):
    logits = torch.from_numpy(rng26.normal(0, 1, (B, F, K, 2)).astype(np.float32)).double()
    bq = ru.get_logits2biquads("coef", 0.95)(logits).clone()
    bq[..., 0] = torch.from_numpy(rng26.uniform(0.8, 1.25, (B, F, K)))       # a0 != 1 as well
    bq.requires_grad_(True)
    gain = torch.exp(torch.from_numpy(rng26.normal(-1, 0.3, (B, F)).astype(np.float32)).double()).requires_grad_(True)
    ex = torch.from_numpy(rng26.normal(0, 1, (B, T)).astype(np.float32)).double().requires_grad_(True)
    syn = rlpc.BatchSecondOrderLPCSynth(hop_length=hop, window_size=W, window="hanning").double()
    y = syn(ex, gain, bq)
    gy = torch.from_numpy(rng26.normal(0, 1, tuple(y.shape)).astype(np.float32)).double()
    (y * gy).sum().backward()
    d.update({f"{tag}_ex": ex.detach(), f"{tag}_gain": gain.detach(), f"{tag}_biquads": bq.detach(), f"{tag}_hop": hop,
              f"{tag}_W": W, f"{tag}_y": y.detach(), f"{tag}_gy": gy, f"{tag}_g_ex": ex.grad, f"{tag}_g_gain": gain.grad,
              f"{tag}_g_biquads": bq.grad})
save("g26_biquad_cascade_grads", **d)
# ----------------------------------------------------------------------------- g27 harmonic oscillator with phase terms (a-11)
# HarmonicOscillator.forward with initial_phase (B,H) and phase_offset (hop-rate AudioTensor), models/synth.py:429-435:
# values and the reference's autograd gradients w.r.t. the amplitudes, the phase offset and the initial phase.  Own generator.
rng27 = np.random.default_rng(27)
d = {}
H = 6


class ATu(AT):
    """models.utils.LegacyAudioTensor lacks the .unsqueeze method synth.py:432 calls on the offset (the absent submodule's
    AudioTensor has it): route it through torch.unsqueeze, which the class's __torch_function__ handles."""

    def unsqueeze(self, dim):
        return torch.unsqueeze(self, dim)


for tag, (B, Tp, ph, Fa, ah, Fo, oh) in (("t", (2, 97, 1, 7, 16, 13, 8)), ("r", (2, 7, 16, 4, 32, 97, 1))):
    phase = dyadic((B, Tp), 10, 0.01, 0.12)
    amp = torch.from_numpy(rng27.uniform(0, 1, (B, Fa, H)).astype(np.float32)).requires_grad_(True)
    off = torch.from_numpy(rng27.uniform(-1.5, 1.5, (B, Fo)).astype(np.float32)).requires_grad_(True)
    ip = torch.from_numpy(rng27.uniform(-1, 1, (B, H)).astype(np.float32)).requires_grad_(True)
    y = rs.HarmonicOscillator()(AT(phase, ph), AT(amp, ah), initial_phase=ip, phase_offset=ATu(off, oh)).as_tensor()
    gy = torch.from_numpy(rng27.normal(0, 1, tuple(y.shape)).astype(np.float32))
    (y * gy).sum().backward()
    d.update({f"{tag}_phase": phase, f"{tag}_phase_hop": ph, f"{tag}_amp": amp, f"{tag}_amp_hop": ah, f"{tag}_offset": off,
              f"{tag}_offset_hop": oh, f"{tag}_initial_phase": ip, f"{tag}_y": y, f"{tag}_gy": gy, f"{tag}_g_amp": amp.grad,
              f"{tag}_g_offset": off.grad, f"{tag}_g_initial_phase": ip.grad})
save("g27_harmonic_phase_terms", **d)
print("done")
