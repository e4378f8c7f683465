#!/bin/bash
# Runs ON THE GPU BOX: the driver's command N times in fresh processes; one line per run (settled / unsettled / regions / clocks)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$1; N=${2:-8}; mkdir -p $O; export TMPDIR=/tmp; cd $R
for i in $(seq 1 $N); do
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --recipe-stream 0 --refresh-inputs 0 2>/dev/null | tail -1 > $O/run_$i.json
  python - <<PY
import json
a = json.load(open("$O/run_$i.json")); t = a["timing"]
print("run %2d  settled %6.2f  unsettled %6.2f  regions %s  sclk before/after %s / %s" % ($i, a["ms_per_step"] * 1e3, a["ms_per_step_unsettled"] * 1e3,
      [round(x * 1e3, 1) for x in t["ms_per_step_regions_wall"]],
      {k: v for k, v in (t.get("smi_before_setup_replays") or {}).items() if "sclk" in k.lower() or "ower" in k},
      {k: v for k, v in (t.get("smi_after_last_region") or {}).items() if "sclk" in k.lower() or "ower" in k}))
PY
done 2>&1 | tee $O/summary.txt
