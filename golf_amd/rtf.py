"""Real-time-factor harness — the counterpart of the reference's ``test_rtf.py`` (BASELINE configs[0]): build a model
from a LightningCLI YAML, run analysis (encoder) and synthesis (decoder) on one clip ``-n`` times, drop the fastest and
the slowest run, report mean time and RTF = time / clip duration for each and for the total.

    python -m golf_amd.rtf CONFIG.yaml [--ckpt STATE.pt] [--wav CLIP.wav] [-n 10] [--duration 6.0]

Same protocol as the reference (test_rtf.py:163-253): f0 is a constant 150 Hz track at hop sr // 200, so the oscillator
receives its phase at hop 120 (480 oversampled samples per phase sample), and the encoder's output drives the decoder
through the control transforms.  Differences: the device is synchronised around every run (the reference reads the
clock without a sync, i.e. measures launch time on a GPU); without ``--wav`` a synthetic harmonic clip is used and
without ``--ckpt`` the weights are those of ``__init__`` plus a small random encoder head (there is no network here for
either); the YAML is built on this package's classes (golf_amd.config).
"""
from __future__ import annotations

import argparse
import math
import time
from typing import Callable, Dict, Optional

import numpy as np
import torch

from .audiotensor import AudioTensor
from .config import build_model, load_yaml

__all__ = ["measure", "run", "main"]


def measure(runner: Callable[[], object], num: int):
    """``num`` timed runs, device-synchronised, fastest and slowest dropped (test_rtf.py:163-172)."""
    times, out = [], None
    for _ in range(num):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = runner()
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    kept = sorted(times)[1:-1] if num > 2 else times
    return float(np.mean(kept)), out


def synthetic_clip(sr: int, duration: float, device) -> torch.Tensor:
    t = torch.arange(int(sr * duration), device=device) / sr
    f0 = 150.0
    x = sum(torch.sin(2 * math.pi * f0 * h * t) / h for h in range(1, 12))
    return (0.05 * x + 0.005 * torch.randn_like(x)).unsqueeze(0)


@torch.no_grad()
def run(config, ckpt: Optional[str] = None, wav: Optional[str] = None, num: int = 10, duration: float = 6.0,
        device="cuda", warmup: int = 2) -> Dict[str, float]:
    cfg = load_yaml(config) if isinstance(config, str) else config
    model = build_model(cfg).to(device)
    if ckpt:
        state = torch.load(ckpt, map_location=device)
        model.load_state_dict(state.get("state_dict", state))
    else:
        torch.manual_seed(2434)
        model.encoder.backbone.out_linear.weight.normal_(0, 0.02)   # the zero head of a fresh model is degenerate
    model.eval()
    sr = int(model.sample_rate)
    if wav:
        import soundfile  # optional dependency, only for --wav

        data, wav_sr = soundfile.read(wav, dtype="float32", always_2d=True)
        assert wav_sr == sr, f"clip is {wav_sr} Hz, model expects {sr}"
        x = torch.from_numpy(data[: int(sr * duration), 0]).to(device).unsqueeze(0)
    else:
        x = synthetic_clip(sr, duration, device)
    clip = x.shape[1] / sr
    x = AudioTensor(x)
    f0_hop = sr // 200
    f0 = AudioTensor(torch.full((1, x.shape[1] // f0_hop + 1), 150.0, device=device), f0_hop)

    def analysis():
        params = model.encoder(x, f0=f0 if model.train_with_true_f0 else None)
        f0_hat = params.pop("f0", None)
        params["phase"] = (f0_hat if f0_hat is not None else f0) / sr
        logits = params.pop("voicing_logits", None)
        if logits is not None:
            params["voicing"] = torch.sigmoid(logits)
        return params

    for _ in range(warmup):
        params = analysis()
        model.decoder(**params)
    t_ana, params = measure(analysis, num)
    t_syn, y = measure(lambda: model.decoder(**params), num)
    return {"duration": clip, "analysis_s": t_ana, "synthesis_s": t_syn, "analysis_rtf": t_ana / clip,
            "synthesis_rtf": t_syn / clip, "total_rtf": (t_ana + t_syn) / clip, "samples_out": int(y.shape[1])}


def main(argv=None):
    ap = argparse.ArgumentParser("Real-time factor of a GOLF model on one clip (counterpart of the reference's test_rtf.py)")
    ap.add_argument("config", help="LightningCLI YAML (ckpts/*/*/config.yaml, cfg/ae/*.yaml with a decoder) or YAML text")
    ap.add_argument("--ckpt", default=None, help="state_dict / Lightning checkpoint (optional)")
    ap.add_argument("--wav", default=None, help="clip at the model's sample rate (optional; synthetic otherwise)")
    ap.add_argument("-n", "--num", type=int, default=10)
    ap.add_argument("--duration", type=float, default=6.0)
    args = ap.parse_args(argv)
    r = run(args.config, args.ckpt, args.wav, args.num, args.duration)
    print(f"Test duration: {r['duration']:.3f}")
    print(f"Average analysis time: {r['analysis_s']:.6f}")
    print(f"Real time factor: {r['analysis_rtf']:.6f}")
    print(f"Average synthesis time: {r['synthesis_s']:.6f}")
    print(f"Real time factor: {r['synthesis_rtf']:.6f}")
    print(f"Total time: {r['analysis_s'] + r['synthesis_s']:.6f}")
    print(f"Total real time factor: {r['total_rtf']:.6f}")
    return r


if __name__ == "__main__":
    main()
