"""dev: time the glottal oscillator forward for the fused-kernel geometries (GOLF_OSCF_GEOM) and batch sizes.
usage: python tools/time_osc.py            -> sweeps geometries x batches in subprocesses
       python tools/time_osc.py one B      -> one measurement in this process (env selects the variant)"""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def one(B, iters=20):
    import torch
    from golf_amd import functional as GF
    from golf_amd.synth import DownsampledIndexedGlottalFlowTable
    from golf_amd.synthetic import make_inputs

    inp = make_inputs(B=min(B, 64), device="cuda")
    rep = (B + 63) // 64
    phase = inp["phase"].repeat(rep, 1)[:B].contiguous()
    wsel = inp["wsel"].repeat(rep, 1)[:B].contiguous()
    noise = inp["noise"].repeat(rep, 1)[:B].contiguous()
    osc = DownsampledIndexedGlottalFlowTable(hop_rate=10, in_channels=64, oversampling=4, equal_energy=True, lf_v2=True,
                                             points=2048).cuda()
    f = lambda: GF.glottal_osc(phase, wsel, osc.table, osc.decimater.taps, 1, inp["w_hop"], 4, True, add=noise)
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        f()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    print(f"geom={os.environ.get('GOLF_OSCF_GEOM', '0')} unfused={os.environ.get('GOLF_OSC_UNFUSED', '0')} B={B}: "
          f"{us:9.1f} us/call  {us / B:7.3f} us/utterance", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "one":
        one(int(sys.argv[2]))
    else:
        batches = [int(v) for v in sys.argv[1:]] or [32, 512]
        for env in ({"GOLF_OSC_UNFUSED": "1"}, {"GOLF_OSCF_GEOM": "0"}, {"GOLF_OSCF_GEOM": "1"}, {"GOLF_OSCF_GEOM": "2"},
                    {"GOLF_OSCF_GEOM": "3"}):
            for B in batches:
                subprocess.run([sys.executable, __file__, "one", str(B)], env={**os.environ, **env})
