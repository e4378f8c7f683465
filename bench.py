#!/usr/bin/env python
"""bench.py — GOLF-ss synthesis throughput on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

A "step" = one pass of the hot path over one synthetic batch already resident in HBM:
  glottal wavetable oscillator (4x oversampled, equal energy, decimated) + injected N(0,1) noise
  -> sample-wise LTV all-pole filter (GOLF-ss end filter), B=32 utterances x 2 s @ 24 kHz per GPU.
For N>1 the batch is sharded 32/GPU (weak scaling, no data-path collective inside the path); the
synthesised audio is all-gathered once per step over RCCL (BASELINE configs[3]).
value = whole-job audio samples/s = N * B * T_out / max-over-ranks(step time).

Extra objects on the JSON line (contract §4):
  roofline      dominant kernel: algorithmic bytes per launch / its HIP-event duration vs 8 TB/s
  cpu_baseline  the C restatement of the reference algorithm (oracle/, "port") on the host cores,
                bounded sample, rank 0 / N=1 only
  stages_us     HIP-event time of each stage (informational)
"""
import argparse
import ctypes
import json
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)
SR = 24000


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=32, help="utterances per GPU")
    ap.add_argument("--workload", default="golf-ss-synth",
                    choices=["golf-ss-synth", "golf-ss-train", "golf-ff-synth", "golf-ff-train", "lpc-ss-fwd",
                             "golf-ss-decoder", "golf-ss-decoder-train", "ddsp-decoder", "golf-ss-train-step", "osc-only", "lpc-ss-fast", "golf-ss-decoder-logits", "golf-ss-synth-have-maps"],
                    help="golf-ss-synth (default, BASELINE metric): oscillator + noise + LPC-ss filter; "
                         "golf-ss-decoder: the whole golf-precise.yaml decoder (adds the zero-phase FIR noise filter "
                         "and the room filter); golf-ss-train-step (BASELINE config 5, use --batch 64): one optimisation "
                         "step of the autoencoder of cfg/ae/vctk.yaml -- U-Net/LSTM encoder and multi-scale spectral "
                         "loss in stock PyTorch (MIOpen, rocFFT) around the HIP decoder, Adam(1e-4), grad-clip 0.5")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--streams", type=int, default=4,
                    help="independent batches in flight: the K steps are issued round-robin on this many HIP streams")
    ap.add_argument("--no-graphs", action="store_true",
                    help="issue every step eagerly instead of replaying one captured hipGraph per stream")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, default) | gloo (dry runs)")
    ap.add_argument("--single-device", action="store_true",
                    help="dev: map every rank to cuda:0 (control-flow dry run of the N>1 path on a 1-GPU box)")
    ap.add_argument("--no-gather", action="store_true", help="skip the RCCL all-gather of the audio (N>1)")
    ap.add_argument("--overlap-transitions", action="store_true",
                    help="golf-ss-decoder: start the (excitation-independent) transition kernel on a side stream at the "
                         "top of the step so that it overlaps the oscillator and the noise filter")
    ap.add_argument("--fork-transitions", action="store_true",
                    help="run the transition kernel on a side stream beside the zero-state pass (fork/join inside the step)")
    ap.add_argument("--split-p1", action="store_true",
                    help="diagnostic: transition kernel and zero-state pass as two launches instead of the fused one")
    ap.add_argument("--fp64-transitions", action="store_true",
                    help="inference with the training path's fp64 transition matrices instead of fp32 + refinement sweep")
    ap.add_argument("--gather-mode", default="pipelined", choices=["pipelined", "sync", "peer-store"],
                    help="pipelined: the gather of step k overlaps step k+1 (double-buffered); sync: inside each step; "
                         "peer-store: no collective -- every step's audio is stored straight into the peers' receive "
                         "buffers over xGMI (golf_amd.dist.PeerStoreGather; one node, not yet measured across GPUs)")
    ap.add_argument("--gather-every", type=int, default=0,
                    help="N>1: stage this many steps' audio per slot and exchange them in ONE all-gather (fewer, larger "
                         "collectives: xGMI is per-link bound and a 6 MB gather per ~70 us step sits at the link rate); "
                         "0 (default) = min(8, steps per slot in one timed region)")
    ap.add_argument("--repeats", type=int, default=5,
                    help="the timed region (EXACTLY --steps steps, barrier + synchronize on both sides) is run this many "
                         "times; ms_per_step / value are the MEDIAN region, every region is listed under 'timing'")
    ap.add_argument("--settle", type=int, default=24,
                    help="setup: untimed regions (the SAME path as a timed one: --steps steps between synchronisations) run "
                         "before the timed regions.  Regions measured back to back settle over ~50 ms (77 -> 72 us/step over "
                         "the first 25 regions of 20 steps, DESIGN.md 6), and about one process in ten starts from a deeper idle "
                         "state whose first five regions all sit at 80 - 83: the timed regions are taken after that transient; "
                         "its course is listed in timing.settle_ms_per_step.  0 = rounds 1-4's behaviour.")
    ap.add_argument("--prereplay", type=int, default=64,
                    help="setup: replays of every captured hipGraph before the warm-up steps (graph upload, code objects, "
                         "clocks: instantiation is setup, not a step).  64 x 4 graphs = ~18 ms of device work: a 20-step "
                         "timed region lasts 1.6 ms, and regions measured back to back on a device that has only just left "
                         "idle fall from 81 to 73 us/step over the first 15 (round 4, DESIGN.md 6) -- with 16 replays "
                         "(rounds 2-3) the five timed regions sat on that ramp.  Reported in timing.prereplay_per_graph.")
    ap.add_argument("--lpc-mode", default="auto", choices=["auto", "serial", "chunked", "flat-scan"],
                    help="LPC-ss algorithm: time-chunked scan, batch-parallel serial recursion, or by batch size (default)")
    ap.add_argument("--issue", default="round-robin", choices=["round-robin", "idle-first"],
                    help="which slot takes the next step: round-robin (default, the headline's), or the first slot whose previous step "
                         "has finished (work-conserving; for slots of unequal speed, see --serial-slots)")
    ap.add_argument("--serial-slots", type=int, default=0,
                    help="diagnostic (VERDICT r5 #9): the LAST this-many of the in-flight slots run the sample-wise filter on the "
                         "batch-parallel serial kernels whatever --lpc-mode says -- a hybrid of chunked and serial batches in flight")
    ap.add_argument("--lpc-chain", default="auto", choices=["auto", "latency", "throughput"],
                    help="launch structure of the sample-wise filter (GOLF_SS_THROUGHPUT, include/golf_amd.h): 'throughput' "
                         "costs the least chip time with several batches in flight, 'latency' finishes a lone batch soonest; "
                         "auto = throughput when --streams > 1.  Bit-identical outputs.  single_stream reports both.")
    ap.add_argument("--headline-only", action="store_true",
                    help="counter passes: run nothing but the headline's own launch chain (no latency-chain graph of slot 0)")
    ap.add_argument("--recipe-stream", type=int, default=-1,
                    help="also time a stream of this many consecutive recipe batches (seeds 2434, 2435, ...: benign and hot "
                         "ones alike, each its own captured graph) through the same S streams; -1: 64 for the default "
                         "single-GPU golf-ss-synth run, 0 otherwise")
    ap.add_argument("--refresh-inputs", type=int, default=-1,
                    help="also time the serving mode in which EVERY step first copies a new batch (one of this many resident "
                         "packed batches, round-robin) into its slot's static inputs on the slot's stream, device-to-device, "
                         "and -- side figure -- from pinned host memory; -1: 8 for the default single-GPU golf-ss-synth run, 0 otherwise")
    ap.add_argument("--no-compare-gather-modes", action="store_true",
                    help="N>1: skip the comparison passes that time the SAME regions under each exchange (RCCL all-gather per "
                         "step, staged RCCL all-gather, peer-to-peer stores) after the headline -- exchange.modes in the line")
    ap.add_argument("--fuse-source-maps", default="auto", choices=["auto", "on", "off"],
                    help="golf-ss-synth: oscillator and the filter's transition maps as ONE launch (golf_source_transitions_f32, ABI 6); "
                         "auto = where the lone-batch launch chain is used (single_stream, --streams 1), off = the composition everywhere")
    ap.add_argument("--shared-inputs", action="store_true",
                    help="diagnostic: all in-flight slots read the SAME input tensors (round 1 behaviour)")
    return ap.parse_args()


FUSE_SOURCE_MAPS = "auto"   # --fuse-source-maps
PROFILE_ROUND = "r06"       # profiles/<round>_hbm_traffic.json / _sq_counters_4stream.json: the counter passes of THIS round's kernels


def build_modules(device):
    from golf_amd.filters import LTVMinimumPhaseFilter, LTVMinimumPhaseFilterPrecise
    from golf_amd.synth import DownsampledIndexedGlottalFlowTable

    osc = DownsampledIndexedGlottalFlowTable(hop_rate=10, in_channels=64, oversampling=4, equal_energy=True,
                                             table_type="derivative", normalize_method="constant_power",
                                             align_peak=True, trainable=False, min_R_d=0.3, max_R_d=2.7, lf_v2=True,
                                             points=2048).to(device)
    ss = LTVMinimumPhaseFilterPrecise(lpc_order=22, lpc_parameterisation="rc2lpc").to(device)
    ff = LTVMinimumPhaseFilter(window="hanning", window_length=960, lpc_order=22,
                               lpc_parameterisation="rc2lpc").to(device)
    return osc, ss, ff


def make_step(workload, inp, osc, ss, ff, fast=True, overlap=False, mode=None):
    """Returns (step_fn, samples_per_step, stage_fns) working on plain tensors (module internals)."""
    from golf_amd import functional as GF

    phase, wsel, w_hop, noise, gain, a, hop = (inp[k] for k in ("phase", "wsel", "w_hop", "noise", "gain", "a", "hop"))
    taps = osc.decimater.taps
    table = osc.table
    B = phase.shape[0]

    def source():   # oscillator + noise: the sum is fused into the oscillator's decimation kernel
        return GF.glottal_osc(phase, wsel, table, taps, 1, w_hop, 4, True, add=noise)

    t_ss = min(phase.shape[1], (a.shape[1] - 1) * hop + 1)

    if workload == "golf-ss-synth":
        def step():
            # round 6: the oscillator and the filter's transition maps as ONE launch (golf_source_transitions_f32) -- what a caller
            # WITHOUT batches in flight gets (FUSE_SOURCE_MAPS "auto": the lone-batch chain, i.e. THROUGHPUT_MODE off when the
            # step is issued or captured); bit-identical to the composition, which the headline's four-in-flight chain keeps
            if fast and not overlap and (FUSE_SOURCE_MAPS == "on" or (FUSE_SOURCE_MAPS == "auto" and not GF.THROUGHPUT_MODE)):
                return GF.source_filter_ss(phase, wsel, table, taps, 1, w_hop, 4, True, gain, a, hop, add=noise, mode=mode)
            # --overlap-transitions: the filter's excitation-independent phase (transition matrices + group composites)
            # on a second stream beside the oscillator -- shortens a lone batch's latency, not the pipelined rate
            prep = GF.ltv_allpole_prepare(a, hop, t_ss, overlap=True, fast=fast, mode=mode) if overlap else None
            return GF.ltv_allpole_ss(source(), gain, a, hop, prepared=prep, fast_inference=fast, mode=mode)
    elif workload == "golf-ss-synth-have-maps":   # diagnostic: the step WITHOUT its transition kernel and pre-pass composites
        prep = GF.ltv_allpole_prepare(a, hop, t_ss, fast=fast, mode=mode)   # (prepared once, outside every timed region)

        def step():
            return GF.ltv_allpole_ss(source(), gain, a, hop, prepared=prep, fast_inference=fast, mode=mode)
    elif workload == "lpc-ss-fwd":
        def step():
            return GF.ltv_allpole_ss(noise, gain, a, hop)
    elif workload == "lpc-ss-fast":   # the inference filter alone (diagnostic: its share of the pipelined step)
        def step():
            return GF.ltv_allpole_ss(noise, gain, a, hop, fast_inference=True, mode=mode)
    elif workload == "osc-only":      # the source alone (diagnostic)
        def step():
            return source()
    elif workload in ("golf-ss-decoder", "golf-ss-decoder-train"):
        lm, rk = inp["log_mag"], inp["room_kernel"]
        fir_win = torch.hann_window(2 * (lm.shape[-1] - 1), device=phase.device)
        K = rk.numel()
        train = workload.endswith("train")
        if train:
            gain, a, wsel_g, lm = (t.clone().requires_grad_(True) for t in (gain, a, wsel, lm))
            rk = rk.clone().requires_grad_(True)
            gy = torch.randn(B, 47760, device=phase.device)
        else:
            wsel_g = wsel
            room_taps = torch.cat([rk, rk.new_ones(1), rk.new_zeros((-(K + 1)) % 4)])
        tail = torch.cat([rk.new_ones(1), rk.new_zeros((-(K + 1)) % 4)])   # the constant end of the room filter's taps

        def step():
            prep = GF.ltv_allpole_prepare(a, hop, 47760, overlap=True, fast=True) if (overlap and not train) else None
            nz = GF.zero_phase_fir_filter(noise, lm, fir_win, hop)
            # source + filtered noise, the common length taken by the filter itself (`length=`): as `src[:, :n]` the slice's
            # backward was a 6 MB fill + a 6 MB copy in front of the oscillator's backward
            if not train and prep is None and (FUSE_SOURCE_MAPS == "on" or (FUSE_SOURCE_MAPS == "auto" and not GF.THROUGHPUT_MODE)):
                # the lone-batch chain: oscillator and transition maps as one launch (what SourceFilterSynth does in inference)
                y = GF.source_filter_ss(phase, wsel_g, table, taps, 1, w_hop, 4, True, gain, a, hop, add=nz, length=nz.shape[1])
                return GF.lti_fir(y, room_taps, K)
            src = GF.glottal_osc(phase, wsel_g, table, taps, 1, w_hop, 4, True, add=nz)
            y = GF.ltv_allpole_ss(src, gain, a, hop, prepared=prep, length=nz.shape[1])
            if not train:
                return GF.lti_fir(y, room_taps, K)
            y = GF.lti_fir(y, torch.cat([rk, tail]), K)
            for t in (gain, a, wsel_g, lm, rk):
                t.grad = None
            y.backward(gy)
            return y
    elif workload == "ddsp-decoder":  # cfg/ae/decoder/ddsp.yaml: 155-harmonic additive synth + filtered noise + room
        from golf_amd.synthetic import make_harmonic_amplitudes

        lm, rk = inp["log_mag"], inp["room_kernel"]
        fir_win = torch.hann_window(2 * (lm.shape[-1] - 1), device=phase.device)
        K = rk.numel()
        room_taps = torch.cat([rk, rk.new_ones(1), rk.new_zeros((-(K + 1)) % 4)])
        amps = make_harmonic_amplitudes(B, a.shape[1] + 1, 155, device=phase.device)
        tscale = torch.rsqrt(0.5 / phase)

        def step():
            h = GF.harmonic_osc(phase, 155, 1, amps, hop, tscale, 1)
            nz = GF.zero_phase_fir_filter(noise[:, : h.shape[1]], lm, fir_win, hop)
            return GF.lti_fir(h[:, : nz.shape[1]] + nz, room_taps, K)
    elif workload == "golf-ss-decoder-logits":
        # the decoder as the autoencoder drives it: encoder output (B, 200, 343) -> control transforms (.ctrl of every
        # module: table-selection network, exp gain, logits -> reflection -> direct-form coefficients) -> golf-precise
        # decoder, through the drop-in modules (AudioTensor protocol included)
        from golf_amd.audiotensor import AudioTensor
        from golf_amd.synthetic import make_decoder

        dec = make_decoder(injected_noise=noise).to(phase.device).eval()
        with torch.no_grad():
            dec.room_filter.kernel.copy_(inp["room_kernel"])
        split_sizes, trsfms, keys = dec.split_sizes_and_trsfms
        flat = [n for grp in split_sizes for n in grp]
        h = logits_workload_encoder_output(inp, flat, phase.device)
        ph = AudioTensor(phase)

        @torch.no_grad()
        def step():
            pieces = [AudioTensor(t.squeeze(2) if t.shape[2] == 1 else t, hop) for t in torch.split(h, flat, dim=2)]
            params, i = {}, 0
            for key, grp, fn in zip(keys, split_sizes, trsfms):
                params[key] = fn(*pieces[i:i + len(grp)])
                i += len(grp)
            return dec(phase=ph, **params).as_tensor()
    elif workload == "golf-ss-train-step":
        # BASELINE configs[4] / cfg/ae/vctk.yaml: encoder(x, f0) -> golf-precise decoder -> MSSLoss(509, 1021, 2053)
        # -> backward -> clip 0.5 -> Adam(1e-4).  Steps depend on each other through the weights: one stream, eager.
        from golf_amd.ae import VoiceAutoEncoder, train_step
        from golf_amd.loss import MSSLoss
        from golf_amd.synthetic import make_decoder

        torch.manual_seed(2434)
        model = VoiceAutoEncoder(
            decoder=make_decoder(), criterion=MSSLoss([509, 1021, 2053], alpha=1.0, window="hanning", center=True),
            encoder_class_path="models.enc.VocoderParameterEncoderInterface",
            encoder_init_args=dict(f0_min=60.0, f0_max=1000.0, backbone_type="models.unet.UNetEncoder", n_fft=1024,
                                   hop_length=240, channels=[32, 64, 128, 256], strides=[4, 4, 4, 4],
                                   lstm_hidden_size=256, num_layers=3, dropout=0.1, learn_voicing=False,
                                   learn_f0=False),
            sample_rate=SR, detach_f0=True, detach_voicing=True, train_with_true_f0=True).to(phase.device)
        with torch.no_grad():  # the reference starts out_linear at exactly 0 (constant decoder parameters); a small
            model.encoder.backbone.out_linear.weight.normal_(0, 0.02)  # random head makes them vary like a trained one
        model.train()
        if int(os.environ.get("WORLD_SIZE", "1")) > 1:   # one process per GPU, gradients all-reduced by DDP over RCCL
            from golf_amd.ae import data_parallel

            model = data_parallel(model, device_ids=[phase.device.index])
        opt = torch.optim.Adam(model.parameters(), lr=1e-4, fused=True)   # same update, one kernel per parameter group
        f0 = phase * SR
        f0[:, : SR // 5] = 0  # an unvoiced stretch (driven at a random frequency, ltng/ae.py:97-101)
        ph = torch.cumsum(phase.double(), 1)
        x = sum(torch.sin(2 * math.pi * h * ph) / h for h in range(1, 9)).float() * 0.05 + 0.005 * noise
        uv = torch.empty(B, 1, device=phase.device).uniform_(50, 500)

        def step():
            return train_step(model, opt, (x, f0), clip=0.5, unvoiced_f0=uv, return_output=True)[1]
    elif workload == "golf-ff-synth":
        win = ff._window

        def step():
            return GF.lti_frames_ola(source(), gain, a, win, hop)
    elif workload == "golf-ff-train":
        win = ff._window
        gain_g, a_g, w_g = (t.clone().requires_grad_(True) for t in (gain, a, wsel))
        gy = torch.randn(B, 47760, device=phase.device)

        def step():
            y = GF.lti_frames_ola(GF.glottal_osc(phase, w_g, table, taps, 1, w_hop, 4, True, add=noise), gain_g, a_g,
                                  win, hop)
            gain_g.grad = a_g.grad = w_g.grad = None
            y.backward(gy[:, : y.shape[1]])
            return y
    else:  # golf-ss-train: forward + custom backward w.r.t. gain, a, table_select_weight
        gain_g = gain.clone().requires_grad_(True)
        a_g = a.clone().requires_grad_(True)
        w_g = wsel.clone().requires_grad_(True)
        gy = torch.randn(B, 47761, device=phase.device)

        def step():
            prep = GF.ltv_allpole_prepare(a_g, hop, t_ss, overlap=True, fast=fast, mode=mode, training=True) if overlap else None
            y = GF.ltv_allpole_ss(GF.glottal_osc(phase, w_g, table, taps, 1, w_hop, 4, True, add=noise), gain_g, a_g,
                                  hop, prepared=prep, mode=mode, fast_inference=fast)
            gain_g.grad = a_g.grad = w_g.grad = None
            y.backward(gy[:, : y.shape[1]])
            return y

    y = step()
    return step, int(y.shape[0] * y.shape[1]), y.shape[1]


def logits_workload_encoder_output(inp, flat, device):
    """Stand-in for the encoder output (B, F, 343) of golf-ss-decoder-logits: random heads for the table-selection
    network and the noise filter, and the RECIPE's smooth tracks in the end filter's channels -- log-gain and
    the LPC logits that .ctrl turns into reflection coefficients.  (Rounds 1-2 filled those with white noise per frame:
    SURVEY 8d forbids exactly that -- interpolating unrelated stable frames is unstable -- and every utterance then had
    transition-matrix entries of 1e13.)"""
    B, F = inp["gain"].shape
    g = torch.Generator(device="cpu").manual_seed(2434)
    h = (torch.randn(B, F, sum(flat), generator=g) * 0.3).to(device)
    n_lpc = inp["logits"].shape[-1]
    assert flat[-2:] == [1, n_lpc] or list(flat[-2:]) == [1, n_lpc], flat
    h[..., -n_lpc:] = inp["logits"]
    h[..., -n_lpc - 1] = inp["log_gain"]
    return h


def conditioning_of(workload, inp, device):
    """Conditioning / health words of the sample-wise filter on one slot's coefficient tracks (golf_ltv_allpole_status_u32):
    how many of its utterances had chunk maps recomputed from fp64 trajectories (hot), how many ran on the fp64 boundary
    scan (tier 3), whether the output is finite, the largest transition-matrix entry.  One extra filter call outside every
    timed region.  None for workloads without the sample-wise filter."""
    from golf_amd import functional as GF

    if "ss" not in workload and workload != "lpc-ss-fast":
        return None
    gain, a, hop = inp["gain"], inp["a"], inp["hop"]
    if workload == "golf-ss-decoder-logits":   # the coefficient tracks the decoder derives from its logits
        from golf_amd.audiotensor import AudioTensor
        from golf_amd.synthetic import make_decoder

        dec = make_decoder(injected_noise=inp["noise"]).to(device).eval()
        split_sizes, trsfms, keys = dec.split_sizes_and_trsfms
        flat = [n for grp in split_sizes for n in grp]
        h = logits_workload_encoder_output(inp, flat, device)
        pieces = [AudioTensor(t.squeeze(2) if t.shape[2] == 1 else t, hop) for t in torch.split(h, flat, dim=2)]
        i = 0
        for key, grp, fn in zip(keys, split_sizes, trsfms):
            if key == "end_filter_params":
                g_at, a_at = fn(*pieces[i:i + len(grp)])
                gain, a = g_at.as_tensor().contiguous(), a_at.as_tensor().contiguous()
            i += len(grp)
    st = torch.zeros(4, dtype=torch.int32, device=device)
    with torch.no_grad():
        GF.ltv_allpole_ss(inp["noise"], gain, a, hop, status=st)
    torch.cuda.synchronize()
    return GF.ss_status(st)


def device_kernel_times(step, n=20):
    """Per-kernel average duration measured live with the torch profiler (ROCm tracer = HIP activity records
    on the stream the kernels ran on) — cross-checked by the rocprofv3 summary in profiles/."""
    from torch.profiler import ProfilerActivity, profile

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(n):
            step()
        torch.cuda.synchronize()
    out = {}
    for ev in prof.key_averages():
        dur = getattr(ev, "device_time_total", None)
        if dur is None:
            dur = getattr(ev, "cuda_time_total", 0.0)
        if dur and ev.count:
            out[ev.key] = dur / ev.count
    return out


def event_time_us(fn, n=50):
    """HIP-event bracket on the current stream (the stream golf_amd launches on)."""
    s = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(5):
        fn()
    e0.record(s)
    for _ in range(n):
        fn()
    e1.record(s)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000.0 / n


def cpu_baseline(B, hop, M, budget_s=15.0):
    """Time the reference algorithm's CPU structure (materialised sample-rate coefficients + fp32 sequential-tap
    recursion, parallel over the batch only; oracle/golf_oracle.c) plus the oscillator restated with the
    reference's PyTorch-CPU op sequence (oracle/cpu_baseline.py).  Same synthetic workload, bounded repetitions."""
    from oracle.cpu_baseline import golf_ss_synth_cpu, num_threads
    from golf_amd.synthetic import make_inputs
    from golf_amd.synth import DownsampledIndexedGlottalFlowTable

    inp = make_inputs(B=B, device="cpu")
    osc = DownsampledIndexedGlottalFlowTable(hop_rate=10, in_channels=64, oversampling=4, equal_energy=True, lf_v2=True,
                                             points=2048)
    run = lambda: golf_ss_synth_cpu(inp, osc.table, osc.decimater.kernel)
    y = run()  # warm-up
    times = []
    t_start = time.time()
    while len(times) < 12 and (time.time() - t_start < budget_s or len(times) < 3):
        t0 = time.time()
        run()
        times.append(time.time() - t0)
    ts = sorted(times)
    core = ts[1:-1] if len(ts) > 4 else ts  # test_rtf.py:163-172: drop fastest and slowest, mean
    mean = float(np.mean(core))
    n = int(y.shape[0] * y.shape[1])
    return {"value": n / mean, "unit": "audio samples/s", "cores": num_threads(), "kind": "port",
            "rtf": mean / (B * 2.0), "ms_per_step": mean * 1e3,
            "sample": f"{len(times)} repetitions of the full B={B} x 2 s step (osc via PyTorch-CPU ops as "
                      f"models/synth.py:213-263 arranges them + C/OpenMP sample_wise_lpc port), drop min/max, mean"}


# Algorithmic bytes per output sample (fp32, module-boundary traffic): SURVEY.md §8d for the oscillator (8.0) and the
# LPC filter (8.38 = ex 4 + y 4 + (22 + 1) * 4 / 240 frame parameters); derived the same way for the
# zero-phase FIR noise filter (noise 4 + log_mag 256 * 4 / 240 in, 4 out = 12.27) and the room filter (4 in, 4 out).
STAGE_BYTES = {"osc": 8.0, "lpc": 8.38, "noise_fir": 12.27, "room": 8.0}
PATH_BYTES = {"golf-ss-synth": 16.4, "golf-ss-synth-have-maps": 16.4, "golf-ff-synth": 16.4, "lpc-ss-fwd": 8.38, "lpc-ss-fast": 8.38, "osc-only": 8.0,
              "golf-ss-train": 16.4 + 16.8, "golf-ss-decoder": 16.4 + 12.27 + 8.0,
              "golf-ss-decoder-logits": 16.4 + 12.27 + 8.0, "golf-ss-decoder-train": 2 * (16.4 + 12.27 + 8.0),
              "golf-ff-train": 16.4 + 16.8, "golf-ss-train-step": 2 * (16.4 + 12.27 + 8.0),
              # phase 4 + amplitudes 155*4/240 in, 4 out; + noise filter 12.27 + room 8
              "ddsp-decoder": 10.6 + 12.27 + 8.0}


def stage_bytes_per_sample(kernel_name):
    if "osc" in kernel_name or "harm" in kernel_name:
        return STAGE_BYTES["osc"]
    if "fir_frames" in kernel_name or "zp_gemm" in kernel_name:
        return STAGE_BYTES["noise_fir"]
    if "lti_fir" in kernel_name:
        return STAGE_BYTES["room"]
    return STAGE_BYTES["lpc"]


def stream_set_time(streams, issue):
    """Run ``issue()`` (which launches work on ``streams``) between HIP events: a start event on the current stream
    (the device is idle: the caller synchronised) and one end event per stream; returns (wall_s, event_s) with
    event_s = the latest end event.  torch.cuda.Event only sees the stream it is recorded on, hence one per stream."""
    e0 = torch.cuda.Event(enable_timing=True)
    ends = [torch.cuda.Event(enable_timing=True) for _ in streams]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0.record(torch.cuda.current_stream())
    for st in streams:
        st.wait_event(e0)   # orders every slot stream after the start mark (no work precedes it)
    issue()
    for st, e in zip(streams, ends):
        e.record(st)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    return wall, max(e0.elapsed_time(e) for e in ends) * 1e-3


def smi_clocks():
    """Shader / memory clock and power state of device 0 as rocm-smi reports them right now (None if the tool is missing):
    recorded before the first and after the last timed region, so that a run that comes out slow in ALL its regions (one in
    six of the driver's command) can be told from a normal one by more than its time."""
    import subprocess

    try:
        out = subprocess.run(["rocm-smi", "-d", "0", "--showclocks", "--showperflevel", "--showpower", "--json"],
                             capture_output=True, text=True, timeout=10).stdout
        card = next(iter(json.loads(out).values()))
        keep = {}
        for k, v in card.items():
            kl = k.lower()
            if "sclk" in kl or "mclk" in kl or "fclk" in kl or "performance level" in kl or "power" in kl:
                keep[k] = v
        return keep
    except Exception:
        return None


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves, exactly as the driver's documented command does
    (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py ...`), and
    pass rank 0's JSON line and the launcher's exit code through.  (the flag used to be parsed and ignored.)"""
    import socket
    import subprocess

    with socket.socket() as so:   # a free port on the loopback interface
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL and the peer-store exchange need it across processes
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    global FUSE_SOURCE_MAPS
    args = parse()
    FUSE_SOURCE_MAPS = args.fuse_source_maps
    if args.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        have = torch.cuda.device_count()
        if have < args.gpus and not args.single_device:
            sys.exit(f"bench.py: --gpus {args.gpus} but this node shows {have} GPU(s) (one rank per GPU; --single-device maps "
                     f"every rank to cuda:0 for a control-flow dry run)")
        sys.exit(spawn_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        # a launcher that started a different number of ranks than the command line names: the line would carry the wrong n_gpus
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if args.single_device:
        local_rank = 0
    elif local_rank >= torch.cuda.device_count():
        sys.exit(f"bench.py: rank {rank} has LOCAL_RANK={local_rank} but only {torch.cuda.device_count()} GPU(s) are visible")
    torch.cuda.set_device(local_rank)   # before the communicator exists: RCCL's first collective binds to the current device
    device = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=args.dist_backend)  # "nccl" is RCCL on ROCm
        if dist.get_world_size() != args.gpus:
            sys.exit(f"bench.py: {dist.get_world_size()} ranks joined, --gpus {args.gpus} were asked for")

    if args.fork_transitions:
        import golf_amd.functional as _GF

        _GF.FORK_TRANSITIONS = True
    if args.split_p1:
        import golf_amd.functional as _GF

        _GF.SPLIT_P1 = True
    from golf_amd.dist import shard_inputs, gather_audio, gather_audio_async
    from golf_amd.synthetic import make_inputs

    B = args.batch
    if args.workload == "golf-ss-train-step":  # the optimiser couples consecutive steps: no batches in flight
        args.streams, args.no_graphs, args.no_cpu_baseline = 1, True, True
        args.no_gather = True   # N > 1: the exchange of this workload is DDP's gradient all-reduce, not an audio gather
    if args.workload in ("golf-ss-decoder-logits", "ddsp-decoder"):
        assert world == 1, f"{args.workload} is a single-GPU side benchmark"
    S = max(1, args.streams)
    osc, ss, ff = build_modules(device)
    import golf_amd.functional as _GFm

    throughput_chain = args.lpc_chain == "throughput" or (args.lpc_chain == "auto" and S > 1)
    _GFm.THROUGHPUT_MODE = throughput_chain   # read when a step is issued or captured: the headline's graphs carry it

    # ---- every in-flight slot owns its inputs (seed 2434 + slot; slot 0 = the SURVEY §8d tensors) and its output
    PACK_KEYS = ("phase", "wsel", "noise", "gain", "a")   # what the synthesis step reads: one flat buffer per slot, so that a
    slot_flat = []                                        # serving loop refreshes a slot's inputs with ONE copy (--refresh-inputs)

    def pack_inputs(inp, dev):
        off, lay = 0, {}
        for k in PACK_KEYS:
            lay[k] = off
            off += (inp[k].numel() + 63) // 64 * 64       # 256-byte aligned pieces
        flat = torch.zeros(off, dtype=torch.float32, device=dev)
        for k in PACK_KEYS:
            view = flat[lay[k]:lay[k] + inp[k].numel()].view(inp[k].shape)
            view.copy_(inp[k])
            inp[k] = view
        return flat

    def slot_inputs(slot, dev=None, pack=True):
        dev = device if dev is None else dev
        inp_all = make_inputs(B=B * world, device="cpu", with_noise_filter="decoder" in args.workload,
                              seed=2434 + slot)
        inp = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v)
               for k, v in shard_inputs(inp_all, rank, world).items()}
        flat = pack_inputs(inp, dev) if pack else None
        return inp, flat

    steps_fn, samples, t_out = [], None, None
    slot_cond = []   # per in-flight slot: the filter's conditioning words on that slot's coefficient tracks
    slot_inp = []    # per in-flight slot: its input tensors (views of slot_flat[i] where packed)
    for i in range(S):
        if i > 0 and args.shared_inputs:
            steps_fn.append(steps_fn[0])
            slot_cond.append(slot_cond[0])
            continue
        inp_i, flat_i = slot_inputs(i)
        slot_flat.append(flat_i)
        slot_inp.append(inp_i)
        fn, samples, t_out = make_step(args.workload, inp_i, osc, ss, ff, fast=not args.fp64_transitions,
                                       overlap=args.overlap_transitions,
                                       mode="serial" if i >= S - args.serial_slots else args.lpc_mode)
        steps_fn.append(fn)
        slot_cond.append(conditioning_of(args.workload, inp_i, device))
    step = steps_fn[0]

    do_gather = world > 1 and not args.no_gather
    GE = (args.gather_every if args.gather_every > 0 else max(1, min(8, args.steps // S))) if do_gather else 1
    pipelined = args.gather_mode == "pipelined"
    peer_store = do_gather and args.gather_mode == "peer-store"
    if peer_store:   # per slot: a ring of receive buffers on every rank, filled by the peers' stores
        from golf_amd.dist import PeerStoreGather

        GE = 1
        stagers = [PeerStoreGather(B, t_out, depth=4, device=device) for _ in range(S)]
    elif do_gather:  # per slot: GE staged steps -> one collective of GE*B rows per rank (golf_amd.dist.StagedGather)
        from golf_amd.dist import StagedGather

        stagers = [StagedGather(B, t_out, GE, device, world=world) for _ in range(S)]
    # ---- execution mode: S independent batches in flight.  The serial phases of the filter occupy a few dozen
    # waves for tens of microseconds (the boundary scan: B waves), so one batch leaves most of the chip idle;
    # a serving loop keeps several batches in flight on separate HIP streams, each step replayed as ONE hipGraph
    # (8 kernels + allocator traffic -> one launch).  Every step does the full work and writes its own output.
    use_graphs = not args.no_graphs  # training steps (forward + custom backward) are captured whole, like inference
    graphs, outs = [], []
    g_lat = None
    if use_graphs:
        for i in range(S):
            warm = torch.cuda.Stream(device=device)
            warm.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(warm):
                for _ in range(2):
                    steps_fn[i]()
            torch.cuda.current_stream().wait_stream(warm)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                yg = steps_fn[i]()
            graphs.append(g)
            outs.append(yg)
        torch.cuda.synchronize()
        for i in range(S):   # every slot's replay equals an eager call on that slot's inputs, bit for bit
            ref = steps_fn[i]()
            graphs[i].replay()
            torch.cuda.synchronize()
            assert torch.equal(outs[i], ref), f"hipGraph replay of slot {i} differs from eager execution"
        if S > 1 and not args.shared_inputs:
            assert not torch.equal(outs[0], outs[1]), "slots were expected to hold different batches"
        # the latency view in the chain a caller WITHOUT batches in flight would ask for (GOLF_SS_THROUGHPUT off): its own
        # graph of slot 0, captured and warmed like the others, checked against the headline's output (the two chains are
        # bit-identical by construction)
        if throughput_chain and "ss" in args.workload and not args.headline_only:
            _GFm.THROUGHPUT_MODE = False
            warm = torch.cuda.Stream(device=device)
            warm.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(warm):
                for _ in range(2):
                    steps_fn[0]()
            torch.cuda.current_stream().wait_stream(warm)
            g_lat = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g_lat):
                y_lat = steps_fn[0]()
            g_lat.replay()
            graphs[0].replay()
            torch.cuda.synchronize()
            if _GFm.SS_THROUGHPUT_SERIAL_MIN <= B < 2048 and args.lpc_mode == "auto":
                # with batches in flight the filter takes the serial kernels from B = 512, a lone batch the chunked scan up to
                # 2048: two algorithms, equal to rounding (each within the tests' bound of the float64 oracle), not bit for bit
                # (per utterance: on the recipe's ill-conditioned utterances two correct fp32 evaluations differ by more than
                #  1e-4 -- the tests bound each by 2 - 3 x the sequential recursion's own error against float64.  The 2e-2 for the
                #  worst row is that bound twice over for the recipe's worst-conditioned utterances, whose sequential fp32
                #  recursion is 3e-3 - 5e-3 from float64 (tier-3 rows of tests/test_gpu_lpc_ss.py::test_recipe_fuzz_every_utterance);
                #  the median and 90 % figures are what a plan-level bug would move.)
                rows = (y_lat - outs[0]).abs().amax(1) / outs[0].abs().amax(1).clamp_min(1e-30)
                med, q90, worst = (float(v) for v in (rows.median(), rows.quantile(0.9), rows.max()))
                assert med < 1e-4 and q90 < 3e-4 and worst < 2e-2, \
                    f"latency (chunked) and throughput (serial) plans differ: median {med:.2e}, 90 % {q90:.2e}, worst row {worst:.2e}"
            else:
                assert torch.equal(y_lat, outs[0]), "latency and throughput chains differ"
            _GFm.THROUGHPUT_MODE = throughput_chain
    # replay streams are created after capture: ROCm maps streams round-robin onto a few hardware queues, and
    # streams that alias one queue serialise (measured: 131 vs 105 us/step at S=4 depending on creation order)
    streams = [torch.cuda.Stream(device=device) for _ in range(S)]
    # rocm-smi's figures BEFORE the set-up replays and the warm-up: the tool takes most of a second, and a second of idle device
    # in front of the first timed regions put them at the top of the post-idle transient (unsettled 79 - 84 us/step instead of the
    # 72 - 75 the same protocol read in round 4)
    clocks_before = smi_clocks() if rank == 0 else None
    # no cyclic garbage collection from the set-up replays to the last timed region: a collection is a host pause of milliseconds,
    # the device drains, and the regions after it ride the post-idle transient (profiles/r05_bench_golf_ss_synth_driver_cmd_stalled
    # .json: one settling region at 543 us/step, every region after it 2 - 5 us/step slow -- round 4's "80 - 83 us in all regions"
    # runs).  Collected once here, where the set-up replays that follow bring the device back up.
    import gc

    gc.collect()
    gc.disable()
    if use_graphs:   # setup, not steps: upload every executable graph and bring the device out of its idle state
        for _ in range(max(0, args.prereplay)):
            for i in range(S):
                with torch.cuda.stream(streams[i]):
                    graphs[i].replay()
            if g_lat is not None:
                with torch.cuda.stream(streams[0]):
                    g_lat.replay()
        torch.cuda.synchronize()

    step_no = [0]
    refresh_pool = [None]   # --refresh-inputs: packed batches (device or pinned host) copied into the slot before every replay
    gather_on = [True]      # N > 1: the same regions once more without the exchange (exchange.ms_per_step_no_gather)

    slot_done = [None] * S   # --issue idle-first: an event behind each slot's last step

    def full_step():
        if args.issue == "idle-first":
            # work-conserving: the next batch goes to a slot whose previous step has finished (round-robin from the last one used);
            # with slots of unequal speed -- chunked and serial plans in flight together -- round-robin makes every slot wait for
            # the slowest
            i, spins = step_no[0] % S, 0
            while slot_done[i] is not None and not slot_done[i].query():
                i = (i + 1) % S
                spins += 1
                if spins % S == 0:
                    time.sleep(0)
        else:
            i = step_no[0] % S
        step_no[0] += 1
        with torch.cuda.stream(streams[i]):
            if refresh_pool[0] is not None:   # ordered on the slot's stream: after its previous replay, before this one
                slot_flat[i].copy_(refresh_pool[0][step_no[0] % len(refresh_pool[0])], non_blocking=True)
            if use_graphs:
                graphs[i].replay()
                y = outs[i]
            else:
                y = steps_fn[i]()
            if do_gather and gather_on[0]:   # copy into the slot's staging buffer; every GE-th step starts one all-gather (async)
                handle = stagers[i].push(y.detach())
                if handle is not None and not pipelined:
                    handle.wait()
            if args.issue == "idle-first":
                if slot_done[i] is None:
                    slot_done[i] = torch.cuda.Event()
                slot_done[i].record(streams[i])
        return y

    def drain():
        if do_gather and gather_on[0]:
            for i in range(S):
                with torch.cuda.stream(streams[i]):
                    stagers[i].flush()   # a partial group at the end of a region is exchanged too, then all are waited for

    def barrier():
        if world > 1:
            import torch.distributed as dist

            dist.barrier()

    def issue_region():
        for _ in range(args.steps):
            full_step()
        drain()  # every gather issued inside the timed region completes inside it

    def timed_regions(n):
        """n regions, each EXACTLY --steps steps between barrier + synchronize; (wall, HIP events) per region, max over ranks"""
        out = []
        for _ in range(max(1, n)):
            barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            _, ev_s = stream_set_time(streams, issue_region)
            barrier()
            wall = time.perf_counter() - t0
            if world > 1:
                import torch.distributed as dist

                t = torch.tensor([wall, ev_s], device=device, dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                wall, ev_s = float(t[0].item()), float(t[1].item())
            out.append((wall, ev_s))
        return out

    def median_ms(regs):
        w = sorted(x for x, _ in regs)
        return w[len(w) // 2] / args.steps * 1e3

    for _ in range(args.warmup):
        full_step()
    drain()
    # the protocol of rounds 1 - 3: the timed regions straight after --warmup (reported as ms_per_step_unsettled); the headline's
    # regions follow the settling regions below.  Both are in the line so that rounds stay comparable.
    regions_unsettled = timed_regions(args.repeats)
    settle = []
    for _ in range(max(0, args.settle)):   # untimed regions through the timed path (see --settle)
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        stream_set_time(streams, issue_region)   # exactly what a timed region does (HIP events on every stream included)
        settle.append((time.perf_counter() - t0) / args.steps * 1e3)
        spent = sum(settle) * args.steps / 1e3   # seconds; bounded at 0.3 s (long steps: a training step with its encoder)
        if world > 1:   # every rank takes the same number of regions
            import torch.distributed as dist

            t = torch.tensor([spent], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            spent = float(t.item())
        if spent > 0.3:
            break
    regions = timed_regions(args.repeats)
    gc.enable()
    clocks_after = smi_clocks() if rank == 0 else None
    walls = sorted(w for w, _ in regions)
    elapsed = walls[len(walls) // 2]   # median region (max over ranks inside each region)
    ms_per_step = elapsed / args.steps * 1e3
    value = world * samples / (elapsed / args.steps)
    ms_unsettled = median_ms(regions_unsettled)
    # steps that ran before the first headline region (the JSON's "warmup" echoes the flag only)
    effective_warmup = (args.warmup + (args.prereplay * S if use_graphs else 0)
                        + len(regions_unsettled) * args.steps + len(settle) * args.steps)

    # ---- N > 1: the same regions without the exchange, so that one line shows what the gather costs
    ms_no_gather = None
    if do_gather:
        gather_on[0] = False
        ms_no_gather = median_ms(timed_regions(args.repeats))
        gather_on[0] = True

    gather_modes = None   # (N > 1: filled by compare_gather_modes() at the very end, under a watchdog -- see there)

    def compare_gather_modes(modes):
        """N > 1: the same regions under every exchange this package has, so that ONE scaling lease yields the comparison DESIGN.md
        section 7 predicts: an RCCL all-gather per step, the staged all-gather (GE steps per collective), and
        peer-to-peer stores over xGMI.  Setup failures and timeouts are agreed on by all ranks (a mode that fails is reported,
        not fatal).  Runs LAST, after rank 0 has everything else of the line, and under a watchdog: peer-to-peer stores have never
        run across real GPUs, and a hang there must not cost the scaling run its line."""
        nonlocal stagers, pipelined
        import torch.distributed as dist
        from golf_amd.dist import PeerStoreGather, StagedGather

        def all_ok(ok):
            t = torch.tensor([1 if ok else 0], device=device, dtype=torch.int32)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            return bool(t.item())

        headline_stagers, headline_pipelined = stagers, pipelined
        ge_staged = max(1, min(8, args.steps // S))
        for name, make in (("rccl_all_gather_per_step", lambda: [StagedGather(B, t_out, 1, device, world=world) for _ in range(S)]),
                           (f"rccl_all_gather_staged_{ge_staged}", lambda: [StagedGather(B, t_out, ge_staged, device, world=world) for _ in range(S)]),
                           ("peer_store", lambda: [PeerStoreGather(B, t_out, depth=4, device=device) for _ in range(S)])):
            err, made = None, None
            try:
                made = make()
            except Exception as e:   # noqa: BLE001 -- reported in the line
                err = repr(e)
            if not all_ok(err is None):
                modes[name] = {"error": err or "failed on another rank"}
                continue
            stagers, pipelined = made, True
            try:
                timed_regions(1)
                ms = median_ms(timed_regions(args.repeats))
            except Exception as e:   # noqa: BLE001
                err, ms = repr(e), None
            if all_ok(err is None):
                inb = (world - 1) * B * t_out * 4
                modes[name] = {"ms_per_step": round(ms, 5), "value": world * samples / (ms * 1e-3),
                               "inbound_GBps_per_rank": round(inb / (ms * 1e-3) / 1e9, 1)}
            else:
                modes[name] = {"error": err or "failed on another rank"}
            if name == "peer_store":
                try:
                    for st_ in made:
                        st_.close()
                except Exception:   # noqa: BLE001
                    pass
        stagers, pipelined = headline_stagers, headline_pipelined

    # ---- serving mode with refreshed inputs: every step copies a new packed batch into its slot first
    refreshed = None
    n_ref = args.refresh_inputs if args.refresh_inputs >= 0 else (8 if (world == 1 and args.workload == "golf-ss-synth"
                                                                       and use_graphs and not args.shared_inputs) else 0)
    if n_ref > 0 and use_graphs and world == 1 and all(f is not None for f in slot_flat):
        pool_dev = [slot_inputs(S + k)[1] for k in range(n_ref)]            # resident in HBM: an encoder on the same GPU
        pool_host = [f.cpu().pin_memory() for f in pool_dev[:min(4, n_ref)]]  # side figure: batches arriving over PCIe
        keep = [f.clone() for f in slot_flat]
        refresh_pool[0] = pool_dev
        timed_regions(2)
        ms_d2d = median_ms(timed_regions(args.repeats))
        # the last batch each slot received, replayed: its output must be the eager result on exactly that batch
        torch.cuda.synchronize()
        last = [outs[i].clone() for i in range(S)]
        for i in range(S):
            ref_i = steps_fn[i]()
            torch.cuda.synchronize()
            assert torch.equal(last[i], ref_i), "refreshed slot %d: replay differs from eager on the refreshed inputs" % i
            assert not torch.equal(slot_flat[i], keep[i]), "slot %d was not refreshed" % i
        refresh_pool[0] = pool_host
        timed_regions(1)
        ms_h2d = median_ms(timed_regions(3))
        refresh_pool[0] = None
        # ---- the zero-copy route: an on-device PRODUCER writes the slot's inputs inside the slot's graph -- what
        # the decoder's control stage does with the encoder's output (models/filters.py:90-97, models/synth.py:320-332): logits ->
        # tanh -> step-up recursion -> a (golf_rc2lpc_fwd), log-gain -> exp -> gain, f0 track -> upsampled phase increments,
        # table-selection logits -> sigmoid, a fresh N(0,1) noise draw -- each written straight into the slot's static inputs
        # (views of the packed buffer), then the synthesis step.  No copy of a finished batch anywhere.
        def measure_producer():
            """(ms per step, error): the slots' graphs with an on-device producer in front of the synthesis step"""
            from golf_amd import functional as GFp

            enc, pgraphs, pouts = [], [], []
            for i in range(S):
                d = slot_inp[i]
                hop_i = int(d["hop"])
                enc.append({"logits": d["logits"].clone(), "log_gain": d["log_gain"].clone(),
                            "f0": d["phase"][:, ::hop_i].clone(), "wlogit": torch.logit(d["wsel"].clamp(1e-6, 1 - 1e-6))})

            def produce(i):
                d, e = slot_inp[i], enc[i]
                d["a"].copy_(GFp.rc2lpc_logits(e["logits"]))
                torch.exp(e["log_gain"], out=d["gain"])
                torch.sigmoid(e["wlogit"], out=d["wsel"])
                up = GFp.linear_upsample(e["f0"], int(d["hop"]))
                n = min(up.shape[1], d["phase"].shape[1])
                d["phase"][:, :n].copy_(up[:, :n])
                d["noise"].normal_()

            for i in range(S):
                warm = torch.cuda.Stream(device=device)
                warm.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(warm):
                    for _ in range(2):
                        produce(i)
                        steps_fn[i]()
                torch.cuda.current_stream().wait_stream(warm)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    produce(i)
                    yp = steps_fn[i]()
                pgraphs.append(g)
                pouts.append(yp)
            torch.cuda.synchronize()
            held, held_outs = list(graphs), list(outs)
            graphs[:], outs[:] = pgraphs, pouts
            timed_regions(2)
            ms_prod = median_ms(timed_regions(args.repeats))
            torch.cuda.synchronize()
            ok_prod = all(bool(torch.isfinite(o).all()) for o in pouts) and not torch.equal(pouts[0], held_outs[0])
            graphs[:], outs[:] = held, held_outs
            if not ok_prod:
                return None, "producer mode: non-finite output, or the produced batch equals the fixed one"
            return ms_prod, None

        ms_prod, prod_err = None, None
        if args.workload == "golf-ss-synth" and len(slot_inp) == S:
            held_graphs, held_outputs = list(graphs), list(outs)
            try:   # (a side figure: whatever goes wrong here is reported in the line, it must not cost the run its headline)
                ms_prod, prod_err = measure_producer()
            except Exception as e:   # noqa: BLE001
                ms_prod, prod_err = None, repr(e)
                graphs[:], outs[:] = held_graphs, held_outputs
        for f, k in zip(slot_flat, keep):
            f.copy_(k)
        torch.cuda.synchronize()
        nbytes = int(slot_flat[0].numel() * 4)
        refreshed = {"pool_batches": n_ref, "bytes_copied_per_step": nbytes, "copies_per_step": 1,
                     "ms_per_step_d2d": round(ms_d2d, 5), "value_d2d": samples / (ms_d2d * 1e-3),
                     "vs_fixed_inputs": round(ms_per_step / ms_d2d, 4),
                     # what the copy cannot go below: its own HBM traffic (read + write) at the achievable ~5 TB/s
                     "hbm_floor_us_per_step": round(2 * nbytes / 5e12 * 1e6, 2),
                     "ms_per_step_h2d_pinned": round(ms_h2d, 5), "h2d_GBps": round(nbytes / (ms_h2d * 1e-3) / 1e9, 2),
                     # no copy at all: the slot's graph holds an on-device producer (control transforms + noise draw) that writes
                     # the static inputs in place, then the synthesis step; the producer's own kernels are inside the figure
                     "ms_per_step_produced": None if ms_prod is None else round(ms_prod, 5),
                     "produced_vs_fixed_inputs": None if ms_prod is None else round(ms_per_step / ms_prod, 4),
                     "produced_error": prod_err,
                     "note": "each step: one device-to-device copy (hipMemcpyDtoDAsync) of the packed batch (phase, wsel, noise, gain, a) into the slot's "
                             "static inputs on the slot's stream, then the graph replay; outputs asserted equal to eager runs on the "
                             "refreshed inputs.  golf_amd.pipeline.ReplayPipeline.submit(batch) is this mode."}

    # ---- conditioning of the slots' filters, per rank (N > 1: the first SCALE run must be readable)
    cond_ranks = None
    if slot_cond and slot_cond[0] is not None:
        mine = torch.tensor([[c["hot_utterances"], c["tier3_utterances"], int(c["nonfinite"])] for c in slot_cond],
                            device=device, dtype=torch.int32)
        if world > 1:
            import torch.distributed as dist

            allr = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(allr, mine)
            cond_ranks = [t.cpu().tolist() for t in allr]
        else:
            cond_ranks = [mine.cpu().tolist()]

    # ---- a stream of consecutive recipe batches: the headline's S slots are S fixed batches; a serving
    # loop sees every batch of the recipe, the hot ones included.  N batches, each its own graph, replayed round-robin on
    # the same S streams.
    recipe_stream = None
    n_rs = args.recipe_stream if args.recipe_stream >= 0 else (64 if (world == 1 and args.workload == "golf-ss-synth"
                                                                      and B == 32 and use_graphs) else 0)
    if n_rs > 0 and use_graphs and world == 1:
        rs_graphs, rs_cond = [], []
        for k in range(n_rs):
            inp_k = make_inputs(B=B, device=device, with_noise_filter="decoder" in args.workload, seed=2434 + k)
            fn_k, _, _ = make_step(args.workload, inp_k, osc, ss, ff, fast=not args.fp64_transitions, mode=args.lpc_mode)
            warm = torch.cuda.Stream(device=device)
            warm.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(warm):
                fn_k()
            torch.cuda.current_stream().wait_stream(warm)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                yk = fn_k()
            rs_graphs.append((g, yk, fn_k, inp_k))   # the closure owns the inputs the graph reads: keep it alive
            rs_cond.append(conditioning_of(args.workload, inp_k, device))
        torch.cuda.synchronize()

        def rs_region():
            for rep_i in range(max(1, args.steps // n_rs)):
                for k in range(n_rs):
                    with torch.cuda.stream(streams[k % S]):
                        rs_graphs[k][0].replay()

        for _ in range(2):
            rs_region()
        torch.cuda.synchronize()
        rs_times = []
        for _ in range(max(3, args.repeats)):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            rs_region()
            torch.cuda.synchronize()
            rs_times.append((time.perf_counter() - t0) / (max(1, args.steps // n_rs) * n_rs))
        rs_med = sorted(rs_times)[len(rs_times) // 2]
        hot_b = [k for k, c in enumerate(rs_cond) if c and c["hot_utterances"] > 0]
        recipe_stream = {
            "batches": n_rs, "seeds": [2434, 2434 + n_rs - 1], "us_per_step": round(rs_med * 1e6, 2),
            "value": samples / rs_med, "unit": "audio samples/s",
            "hot_utterances": int(sum(c["hot_utterances"] for c in rs_cond if c)),
            "tier3_utterances": int(sum(c["tier3_utterances"] for c in rs_cond if c)),
            "batches_with_hot_utterances": len(hot_b), "nonfinite_batches": int(sum(c["nonfinite"] for c in rs_cond if c)),
            "note": "latency of each batch alone (cold / hot / tier 3): tools/recipe_latency.py -> profiles/"}
        del rs_graphs

    result = None
    collective_step = world > 1 and args.workload == "golf-ss-train-step"   # DDP: every rank must take every step
    if rank != 0 and collective_step:
        device_kernel_times(step)
        event_time_us(step)
    if rank == 0:
        # ---- roofline of the dominant kernel, measured live
        ktimes = device_kernel_times(step)
        ours = {k: v for k, v in ktimes.items() if "golf::" in k}
        dom, dom_us = max(ours.items(), key=lambda kv: kv[1]) if ours else ("n/a", float("nan"))
        bytes_per_sample = stage_bytes_per_sample(dom)
        alg_bytes = bytes_per_sample * samples
        achieved = alg_bytes / (dom_us * 1e-6) / 1e9
        step_us = event_time_us(step)                       # one batch alone, eager, one stream (the headline's chain)
        graph_us = event_time_us(graphs[0].replay) if use_graphs else None   # the same as one hipGraph launch
        lat_graph_us, lat_step_us = graph_us, step_us
        if g_lat is not None:
            lat_graph_us = event_time_us(g_lat.replay)
            _GFm.THROUGHPUT_MODE = False
            lat_step_us = event_time_us(step)
            _GFm.THROUGHPUT_MODE = throughput_chain
        # HBM bytes per launch and issued VALU instructions per step come from separate rocprofv3 --pmc passes (counters
        # cannot be read in this run): this round's committed summaries only -- no fallback to an older round's file
        traffic, valu, step_total = None, None, None
        try:
            fn = os.path.join(ROOT, "profiles", PROFILE_ROUND + "_hbm_traffic.json")
            if os.path.exists(fn):
                tr = json.load(open(fn))
                for kname, v in tr.get("kernels", {}).items():
                    if kname in dom and tr.get("batch") == B:
                        traffic = v["hbm_bytes_per_launch"]
                if tr.get("batch") == B and args.workload == "golf-ss-synth":
                    # counted HBM bytes of ONE step, summed over its kernels, for the chain this line ran
                    step_total = (tr.get("step_total_bytes") or {}).get("throughput_chain" if throughput_chain else "latency_chain")
            fn = os.path.join(ROOT, "profiles", PROFILE_ROUND + "_sq_counters_4stream.json")
            if os.path.exists(fn):
                sq = json.load(open(fn))
                if sq.get("batch") == B and sq.get("workload") == args.workload:
                    valu = sq
        except Exception:
            pass
        # stage level: the dominant kernel's stage bytes against the SUM of that stage's kernels (charging the
        # whole stage's bytes to one of its kernels flatters it)
        stage_key = "osc" if ("osc" in dom or "harm" in dom) else ("lpc" if "lpc_" in dom else None)
        stage_us = sum(v for k, v in ours.items() if stage_key and (("osc" in k or "harm" in k) if stage_key == "osc"
                                                                    else "lpc_" in k)) or dom_us
        roofline = {"bound": "hbm", "kernel": dom.split("(")[0][-60:], "kernel_us": round(dom_us, 2),
                    "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                    "traffic_source": None if traffic is None else "profiles/" + PROFILE_ROUND + "_hbm_traffic.json (separate rocprofv3 --pmc passes, not measured in this run)",
                    "step_total_bytes": step_total,
                    "algorithmic_bytes_per_launch": int(alg_bytes),
                    "stage_frac": round(alg_bytes / (stage_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 5),
                    "stage_us": round(stage_us, 2),
                    "path_frac": round(PATH_BYTES[args.workload] * samples / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS, 5),
                    "path_frac_single_stream": round(PATH_BYTES[args.workload] * samples / (step_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 5),
                    "note": f"B={B}: bound by the serial T=47761 recursion (dependency/issue latency), not by HBM; "
                            "frac = the dominant kernel's stage bytes / its duration, path_frac = the whole step's "
                            "algorithmic bytes / the step time (pipelined and single stream); DESIGN.md §5"}
        if valu is not None:
            # VALU-issue roofline of the run the headline is quoted on (S batches in flight): wave-instructions issued per
            # step (SQ_INSTS_VALU summed over the step's kernels, PMC pass over the same command) x 4 cycles / (SIMDs x
            # clock x the step time measured HERE).  4 cycles per wave-instruction is the convention of the SQ counters
            # (SQ_ACTIVE_INST_VALU counts one quad-cycle per instruction); a plain wave64 VALU instruction occupies a
            # SIMD-32 for 2 cycles and a packed one for 4, so the true pipe occupancy is lower: valu_pipe_frac prices every
            # packed instruction (SQ_INSTS_VALU of the transition waves ~ all v_pk_fma_f32) at 4 and the rest at 2.
            n_simd, clk = 1024, 2.4e9
            step_s = elapsed / args.steps
            insts = float(valu["SQ_INSTS_VALU_per_step"])
            roofline["valu_issue_frac"] = round(insts * 4.0 / (n_simd * clk * step_s), 4)
            if "packed_insts_per_step" in valu:
                pk = float(valu["packed_insts_per_step"])
                roofline["valu_pipe_frac"] = round((pk * 4.0 + (insts - pk) * 2.0) / (n_simd * clk * step_s), 4)
            roofline["valu_insts_per_step"] = int(insts)
            roofline["valu_source"] = "profiles/" + PROFILE_ROUND + "_sq_counters_4stream.json (rocprofv3 --pmc SQ_INSTS_VALU over the %d-stream run)" % valu.get("streams", S)
        stages = {k.split("(")[0].replace("void ", "")[-48:]: round(v, 2) for k, v in sorted(ours.items(), key=lambda kv: -kv[1])}
        single_us = lat_graph_us if lat_graph_us is not None else lat_step_us
        result = {
            "metric": ("audio samples/sec (24 kHz) GOLF-ss synth, batch=32x2 s" if args.workload == "golf-ss-synth" and B == 32
                       else f"audio samples/sec (24 kHz) {args.workload}, batch={B}x2 s")
                      + (f", {S} batches in flight" if S > 1 else ""),
            "value": value, "unit": "audio samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": args.workload, "batch_per_gpu": B, "seconds": 2.0, "sample_rate": SR,
                       "lpc_order": 22, "hop": 240, "frames": 200, "table": "100x2048 LF-v2", "oversampling": 4,
                       "samples_out_per_utterance": t_out,
                       "parallelism": f"dp{world}" + (f"+allgather({args.gather_mode}, every {GE} step(s))" if do_gather else ""),
                       "lpc_mode": args.lpc_mode, "lpc_chain": "throughput" if throughput_chain else "latency",
                       "fuse_source_maps": args.fuse_source_maps, "serial_slots": args.serial_slots, "issue": args.issue,
                       "batches_in_flight": S, "slot_inputs": "shared" if args.shared_inputs else "distinct per slot",
                       "hipgraph_replay": bool(use_graphs)},
            "rtf": (elapsed / args.steps) / (B * 2.0),
            "ms_per_step_unsettled": ms_unsettled,   # the same regions straight after --warmup: the protocol of rounds 1 - 3
            "timing": {"regions": len(regions), "statistic": "median region wall time (max over ranks per region)",
                       "effective_warmup_steps": int(effective_warmup),
                       "smi_before_setup_replays": clocks_before, "smi_after_last_region": clocks_after,
                       "ms_per_step_regions_unsettled_wall": [round(w / args.steps * 1e3, 5) for w, _ in regions_unsettled],
                       "ms_per_step_regions_wall": [round(w / args.steps * 1e3, 5) for w, _ in regions],
                       "ms_per_step_regions_hip_events": [round(e / args.steps * 1e3, 5) for _, e in regions],
                       "prereplay_per_graph": args.prereplay if use_graphs else 0,
                       "settle_regions": len(settle), "settle_ms_per_step": [round(x, 5) for x in settle]},
            # one batch at a time on ONE stream (no batches in flight): the latency view of the same step
            "single_stream": {"value": samples / (single_us * 1e-6), "unit": "audio samples/s",
                              "us_per_step_graph": None if lat_graph_us is None else round(lat_graph_us, 2),
                              "us_per_step_eager": round(lat_step_us, 2),
                              "lpc_chain": "latency",
                              # the headline's own graph of slot 0 replayed alone (same launch structure as the headline)
                              "headline_chain": "throughput" if throughput_chain else "latency",
                              "us_per_step_graph_headline_chain": None if graph_us is None else round(graph_us, 2),
                              "us_per_step_eager_headline_chain": round(step_us, 2)},
            # one batch alone, eager, HIP events on the launch stream: in the HEADLINE's launch chain (what a caller of the
            # headline configuration gets for a lone batch), and in the latency chain (a caller without batches in flight)
            "single_batch_latency_us": round(step_us, 2),
            "single_batch_latency_us_latency_chain": round(lat_step_us, 2),
            "roofline": roofline,
            "stages_us": stages,
        }
        if cond_ranks is not None:
            # conditioning of the sample-wise filter on the slots' inputs: [hot utterances, tier-3 utterances, non-finite]
            # per slot, per rank (golf_ltv_allpole_status_u32; csrc/lpc_ss.hip "conditioning tiers")
            result["conditioning"] = {
                "flagged_utterances_per_slot": [[c[0] for c in r] for r in cond_ranks],
                "tier3_utterances_per_slot": [[c[1] for c in r] for r in cond_ranks],
                "nonfinite_per_slot": [[c[2] for c in r] for r in cond_ranks],
                "max_phi_per_slot_rank0": [None if c is None else round(c["max_phi"], 2) for c in slot_cond],
                "thresholds": {"hot_utterance_G1": float(os.environ.get("GOLF_SS_PHI_GUARD", 30)),
                               "hot_chunk_G2": float(os.environ.get("GOLF_SS_PHI_GUARD2", 10)),
                               "tier3_G3": float(os.environ.get("GOLF_SS_PHI_GUARD3", 256)),
                               "tier3_group_log2": float(os.environ.get("GOLF_SS_GROUP_LOG2", 96))}}
        if recipe_stream is not None:
            result["recipe_stream"] = recipe_stream
        if refreshed is not None:
            result["refreshed_inputs"] = refreshed
        knobs = {k: v for k, v in os.environ.items() if k.startswith("GOLF_")}
        if knobs:   # a library override or build flags change what was measured: say so in the line
            result["env"] = knobs
        if world > 1:
            result["exchange"] = {"backend": args.dist_backend, "world_size": world,
                                  "bytes_in_per_rank_per_step": int((world - 1) * B * t_out * 4) if do_gather else 0,
                                  "bytes_per_collective_per_rank": int(GE * B * t_out * 4) if do_gather and not peer_store else 0,
                                  "collectives_per_step": (1.0 / GE) if do_gather and not peer_store else 0.0,
                                  "mode": args.gather_mode if do_gather else "none",
                                  # the same timed regions without the exchange: what the gather costs, in one line
                                  "ms_per_step_no_gather": None if ms_no_gather is None else round(ms_no_gather, 5),
                                  "value_no_gather": None if ms_no_gather is None else world * samples / (ms_no_gather * 1e-3),
                                  # the same regions under each exchange (all-gather per step / staged / peer-to-peer stores)
                                  "modes": gather_modes,
                                  # hot utterances cost their rank ~13 us more per step: the ranks a synchronous gather waits for
                                  "flagged_utterances_per_rank": None if cond_ranks is None else [sum(c[0] for c in r) for r in cond_ranks],
                                  "tier3_utterances_per_rank": None if cond_ranks is None else [sum(c[1] for c in r) for r in cond_ranks]}
        if world == 1 and not args.no_cpu_baseline:
            try:
                result["cpu_baseline"] = cpu_baseline(B, 240, 22)
                result["speedup_vs_cpu_port"] = value / result["cpu_baseline"]["value"]
            except Exception as e:  # the checker is optional for the measurement itself
                result["cpu_baseline"] = {"error": repr(e)}
        if not (do_gather and not args.no_compare_gather_modes):
            print(json.dumps(result), flush=True)
    if world > 1:
        import torch.distributed as dist

        dist.barrier()  # ranks > 0 wait for rank 0's profiling pass before going on
        if do_gather and not args.no_compare_gather_modes:
            import threading

            gather_modes = {}
            limit = float(os.environ.get("GOLF_BENCH_COMPARE_LIMIT_S", "180"))

            def give_up():   # the comparison hung: the line goes out with what was measured, and the job ends
                if rank == 0:
                    gather_modes["aborted"] = "comparison passes exceeded %.0f s (a mode hung); modes listed so far are complete" % limit
                    result["exchange"]["modes"] = gather_modes
                    print(json.dumps(result), flush=True)
                os._exit(0)

            dog = threading.Timer(limit if rank == 0 else limit + 10.0, give_up)
            dog.daemon = True
            dog.start()
            compare_gather_modes(gather_modes)
            dog.cancel()
            if rank == 0:
                result["exchange"]["modes"] = gather_modes
                print(json.dumps(result), flush=True)
        if peer_store:
            for st in stagers:
                st.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
