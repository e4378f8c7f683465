#!/bin/bash
R=$GRAFT_REPO_ROOT
for m in slow fast; do
  bash $R/tools/prof_pmc.sh $R/gpurun_out/p1f_$m GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES -- python $R/tools/p1f_case.py $m > /dev/null 2>&1
  bash $R/tools/prof_pmc.sh $R/gpurun_out/p1f_${m}_t SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -- python $R/tools/p1f_case.py $m > /dev/null 2>&1
done
cd $R && python - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/p1f_*")):
    tr = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
    ct = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    dur = collections.defaultdict(list)
    for f in tr:
        for r in csv.DictReader(open(f)):
            if "p1f" in r["Kernel_Name"]:
                dur["p1f"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    cs = collections.defaultdict(list)
    for f in ct:
        for r in csv.DictReader(open(f)):
            if "p1f" in r["Kernel_Name"]:
                cs[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(d, "dur us", [round(x, 1) for x in dur["p1f"]][-6:], {k: sum(v[-6:]) / 6 for k, v in cs.items()})
PY
