"""dev probe: P1f duration inside different step compositions"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from golf_amd import functional as GF
from golf_amd.synthetic import make_inputs
from golf_amd.synth import DownsampledIndexedGlottalFlowTable

inp = make_inputs(B=32, device="cuda", with_noise_filter=True)
osc = DownsampledIndexedGlottalFlowTable(hop_rate=10, in_channels=64, oversampling=4, equal_energy=True, lf_v2=True, points=2048).cuda()
taps, table = osc.decimater.taps, osc.table
phase, wsel, w_hop, noise, gain, a = (inp[k] for k in ("phase", "wsel", "w_hop", "noise", "gain", "a"))
lm, rk = inp["log_mag"], inp["room_kernel"]
win = torch.hann_window(510, device="cuda")
room = torch.cat([rk, rk.new_ones(1)])
def prof(fn, tag):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as p:
        for _ in range(10): fn()
        torch.cuda.synchronize()
    for e in p.key_averages():
        if "p1f" in e.key:
            print(f"{tag:40s} {e.device_time_total/e.count:8.1f} us")
def full(use_osc=True, use_nf=True, use_room=True, keep=None):
    def f():
        o = GF.glottal_osc(phase, wsel, table, taps, 1, w_hop, 4, True) if use_osc else noise
        if use_nf:
            nz = GF.zero_phase_fir_filter(noise[:, : o.shape[1]], lm, win, 240)
            src = o[:, : nz.shape[1]] + nz
        else:
            src = o + noise[:, : o.shape[1]]
        y = GF.ltv_allpole_ss(src, gain, a, 240)
        if use_room:
            y = GF.lti_fir(y, room, 127)
        return y
    return f
prof(full(), "full decoder")
prof(full(use_room=False), "no room")
prof(full(use_osc=False), "no osc")
prof(full(use_nf=False), "no noise filter (synth + room)")
prof(full(use_nf=False, use_room=False), "synth")
def gemm_only():
    k = GF.zero_phase_fir_kernels(lm, win)
    return GF.ltv_allpole_ss(noise, gain, a, 240)
prof(gemm_only, "gemm then lpc")
def fir_only():
    y0 = GF.lti_fir(noise, room, 127)
    return GF.ltv_allpole_ss(noise, gain, a, 240)
prof(fir_only, "lti_fir then lpc")
print("---- ws placement")
orig_ws = GF._workspace
pools = [torch.empty(160 << 20, dtype=torch.uint8, device="cuda") for _ in range(3)]
for pi, pool in enumerate(pools):
    for off in (0, 256, 4096, 65536, 1 << 20, 2 << 20, (2 << 20) + 4096, 8 << 20, 33 << 20, 64 << 20):
        GF._workspace = lambda n, d, pool=pool, off=off: pool[off: off + n]
        prof(lambda: GF.ltv_allpole_ss(noise, gain, a, 240), f"pool{pi} {hex(pool.data_ptr())} off {off}")
GF._workspace = orig_ws
