"""Build profiles/<round>_* from gpurun_out/<round> (written by tools/refresh_profiles.sh on the GPU box).
usage: python tools/collect_profiles.py [r02]"""
import collections, csv, glob, json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RND = sys.argv[1] if len(sys.argv) > 1 else "r06"
SRC = os.path.join(ROOT, "gpurun_out", RND)
DST = os.path.join(ROOT, "profiles")
sys.path.insert(0, os.path.join(ROOT, "tools"))
# gpurun MERGES a call's outputs into gpurun_out/: counter CSVs of an earlier refresh (other process ids, kernels that have since
# been renamed) would be summed in beside the new ones.  When the last call was the refresh, drop what it did not bring back.
try:
    last = json.load(open(os.path.join(ROOT, "gpurun_out", ".last_call.json")))
    if any(f.startswith(RND + "/sq4_synth") for f in last.get("pulled_files", [])):
        pulled = set(last["pulled_files"])
        for dp, _, fns in os.walk(SRC):
            for fn in fns:
                full = os.path.join(dp, fn)
                if os.path.relpath(full, os.path.join(ROOT, "gpurun_out")) not in pulled:
                    os.remove(full)
except (OSError, ValueError, KeyError):
    pass
import rocpd_summary

for f in glob.glob(os.path.join(SRC, "bench_*.json")):
    shutil.copy(f, os.path.join(DST, RND + "_" + os.path.basename(f)))
if os.path.exists(os.path.join(SRC, "recipe_latency.txt")):
    shutil.copy(os.path.join(SRC, "recipe_latency.txt"), os.path.join(DST, RND + "_recipe_latency.txt"))
if os.path.exists(os.path.join(SRC, "osc_pmc_b2048.txt")):
    shutil.copy(os.path.join(SRC, "osc_pmc_b2048.txt"), os.path.join(DST, RND + "_osc_pmc_b2048.txt"))
if os.path.exists(os.path.join(SRC, "train_step_profile.json")):
    shutil.copy(os.path.join(SRC, "train_step_profile.json"), os.path.join(DST, RND + "_train_step_profile.json"))
for f in glob.glob(os.path.join(SRC, "*_kernel_stats.csv")):   # summarised on the GPU box by refresh_profiles.sh
    shutil.copy(f, os.path.join(DST, RND + "_" + os.path.basename(f)))

def counters(tag, per_step_only=False):
    """per_step_only: only the kernels every step launches (dispatch count >= half the most frequent kernel's): a pass also sees
    kernels that ran once -- the conditioning probe's transition launch, the tap-fragment layout -- and round 5's step totals
    counted those in."""
    out = collections.defaultdict(dict)
    ndisp = collections.defaultdict(int)
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for f in glob.glob(os.path.join(SRC, f"pmc_{c}_{tag}", "**", "*counter_collection.csv"), recursive=True):
            acc = collections.defaultdict(list)
            for r in csv.DictReader(open(f)):
                if "golf::" in r["Kernel_Name"]:
                    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
                    acc[k].append(float(r["Counter_Value"]))
                    ndisp[k] = max(ndisp[k], int(float(r.get("Dispatches") or 1)))
            for k, v in acc.items():
                out[k][c + "_KB"] = round(sum(v) / len(v))
    if per_step_only and ndisp:
        top = max(ndisp.values())
        out = {k: d for k, d in out.items() if ndisp[k] * 2 >= top}
    for k, d in out.items():
        if "FETCH_SIZE_KB" in d and "WRITE_SIZE_KB" in d:
            d["hbm_bytes_per_launch"] = int((2 * d["FETCH_SIZE_KB"] + d["WRITE_SIZE_KB"]) * 1024)
    return dict(out)

kern = counters("synth")
kern.update({k: v for k, v in counters("synthtp").items() if k not in kern})   # the throughput chain's own kernels
kern.update({k: v for k, v in counters("decoder").items() if k not in kern})
kern.update({k: v for k, v in counters("train").items() if k not in kern})
json.dump({"batch": 32, "workload": "golf-ss-synth (headline), golf-ss-decoder (inference) + golf-ss-decoder-train kernels",
           "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/prof_pmc.sh), "
                     "per-launch averages; hbm bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (FETCH doubled per "
                     "MI355X_MICROARCH.md gfx950 note: exact for wide coalesced reads, an upper bound for narrow ones)",
           "kernels": kern,
           # one step = one launch of each of its kernels: the two launch chains of the sample-wise filter, summed
           "step_total_bytes": {
               "latency_chain": sum(v.get("hbm_bytes_per_launch", 0) for k, v in counters("synth", True).items()),
               "throughput_chain": sum(v.get("hbm_bytes_per_launch", 0) for k, v in counters("synthtp", True).items()),
               "latency_chain_kernels": sorted(counters("synth", True)), "throughput_chain_kernels": sorted(counters("synthtp", True))}},
          open(os.path.join(DST, RND + "_hbm_traffic.json"), "w"), indent=1)

# ---- SQ counters -> per-kernel utilisation figures
sq = collections.defaultdict(dict)
for f in glob.glob(os.path.join(SRC, "sq_*", "**", "*counter_collection.csv"), recursive=True):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "golf::" in r["Kernel_Name"]:
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            acc[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in acc.items():
        sq[k][c] = round(sum(v) / len(v))
N_XCD, N_SIMD = 8, 1024
for k, d in sq.items():
    if "GRBM_GUI_ACTIVE" in d and "SQ_ACTIVE_INST_VALU" in d:
        cyc = d["GRBM_GUI_ACTIVE"] / N_XCD                      # elapsed shader-clock cycles of the launch
        d["valu_busy_frac"] = round(d["SQ_ACTIVE_INST_VALU"] * 4 / (cyc * N_SIMD), 4)   # 4 cycles per wave64 VALU op
    if "SQ_WAVE_CYCLES" in d and "SQ_WAIT_INST_ANY" in d and d["SQ_WAVE_CYCLES"]:
        d["wave_wait_frac"] = round(d["SQ_WAIT_INST_ANY"] / d["SQ_WAVE_CYCLES"], 4)
    if "SQ_LDS_IDX_ACTIVE" in d and "GRBM_GUI_ACTIVE" in d:
        d["lds_busy_frac"] = round(d["SQ_LDS_IDX_ACTIVE"] / (d["GRBM_GUI_ACTIVE"] / N_XCD * 256), 4)   # 256 CUs
if sq:
    json.dump({"batch": 32, "method": "rocprofv3 --kernel-trace --pmc <4 counters> (separate passes, tools/prof_pmc.sh) on the "
               "eager single-stream golf-ss-decoder and golf-ss-decoder-train steps; per-launch averages. valu_busy_frac = "
               "4*SQ_ACTIVE_INST_VALU / (GRBM_GUI_ACTIVE/8 XCDs * 1024 SIMDs); wave_wait_frac = SQ_WAIT_INST_ANY / "
               "SQ_WAVE_CYCLES; lds_busy_frac = SQ_LDS_IDX_ACTIVE / (cycles * 256 CUs)",
               "kernels": dict(sq)}, open(os.path.join(DST, RND + "_sq_counters.json"), "w"), indent=1)
# ---- the headline run (4 batches in flight): issued VALU wave-instructions per step, for bench.py's roofline.valu_issue_frac
sq4 = collections.defaultdict(lambda: collections.defaultdict(list))
sq4n = {}
for f in glob.glob(os.path.join(SRC, "sq4_synth", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "golf::" in r["Kernel_Name"]:
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            sq4[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if "Dispatches" in r:   # rows reduced on the GPU box: one per (kernel, counter), the mean over its dispatches
                sq4n[k] = int(float(r["Dispatches"]))
if sq4:
    kern4 = {k: {c: round(sum(v) / len(v)) for c, v in d.items()} for k, d in sq4.items()}
    for k, d in sq4.items():
        kern4[k]["launches"] = sq4n.get(k, len(next(iter(d.values()))))
    # every kernel of the step is launched once per step: the per-launch averages add up to the step
    per_step = sum(d.get("SQ_INSTS_VALU", 0) for d in kern4.values())
    json.dump({"batch": 32, "workload": "golf-ss-synth", "streams": 4,
               "method": "rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES SQ_BUSY_CYCLES over "
                         "`bench.py --steps 20 --warmup 5` (4 batches in flight, hipGraph replay; tools/refresh_profiles.sh); "
                         "per-launch averages per kernel, summed over the kernels of one step",
               "SQ_INSTS_VALU_per_step": int(per_step), "kernels": kern4},
              open(os.path.join(DST, RND + "_sq_counters_4stream.json"), "w"), indent=1)
print(sorted(os.listdir(DST)))
