"""Build profiles/r01_* from gpurun_out/r01 (written by tools/refresh_profiles.sh on the GPU box)."""
import collections, csv, glob, json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "r01")
DST = os.path.join(ROOT, "profiles")
sys.path.insert(0, os.path.join(ROOT, "tools"))
import rocpd_summary

for f in glob.glob(os.path.join(SRC, "bench_*.json")):
    shutil.copy(f, os.path.join(DST, "r01_" + os.path.basename(f)))
for d in glob.glob(os.path.join(SRC, "trace_*")):
    if not os.path.isdir(d):
        continue
    dbs = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)
    if dbs:
        rocpd_summary.main(dbs[0], os.path.join(DST, "r01_" + os.path.basename(d).replace("trace_", "") + "_kernel_stats.csv"))

def counters(tag):
    out = collections.defaultdict(dict)
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for f in glob.glob(os.path.join(SRC, f"pmc_{c}_{tag}", "**", "*counter_collection.csv"), recursive=True):
            acc = collections.defaultdict(list)
            for r in csv.DictReader(open(f)):
                if "golf::" in r["Kernel_Name"]:
                    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
                    acc[k].append(float(r["Counter_Value"]))
            for k, v in acc.items():
                out[k][c + "_KB"] = round(sum(v) / len(v))
    for k, d in out.items():
        if "FETCH_SIZE_KB" in d and "WRITE_SIZE_KB" in d:
            d["hbm_bytes_per_launch"] = int((2 * d["FETCH_SIZE_KB"] + d["WRITE_SIZE_KB"]) * 1024)
    return dict(out)

kern = counters("decoder")
kern.update({k: v for k, v in counters("train").items() if k not in kern})
json.dump({"batch": 32, "workload": "golf-ss-decoder (inference) + golf-ss-decoder-train kernels",
           "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/prof_pmc.sh), "
                     "per-launch averages; hbm bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (FETCH doubled per "
                     "MI355X_MICROARCH.md gfx950 note: exact for wide coalesced reads, an upper bound for narrow ones)",
           "kernels": kern}, open(os.path.join(DST, "r01_hbm_traffic.json"), "w"), indent=1)
print(sorted(os.listdir(DST)))
