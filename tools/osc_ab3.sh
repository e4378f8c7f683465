#!/bin/bash
# Runs ON THE GPU BOX: oscillator tile choice by environment (GOLF_OSCF_TILE = 2048 | 1536 | auto = by batch) over batch sizes
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$1; shift; mkdir -p $O; export TMPDIR=/tmp; cd $R
for v in "$@"; do
  if [ "$v" == "auto" ]; then unset GOLF_OSCF_TILE; else export GOLF_OSCF_TILE=$v; fi
  timeout 300 python bench.py --no-cpu-baseline --recipe-stream 0 --workload osc-only 2>$O/err_$v.txt | tail -1 > $O/osc32_$v.json
  python - <<PY
import json
try:
    a = json.load(open("$O/osc32_$v.json"))
    print("TILE=%-5s B=32: 4 in flight %6.2f us/step, alone graph %6.2f  stages %s" % ("$v", a["ms_per_step"] * 1e3, a["single_stream"]["us_per_step_graph"],
          {k.replace("golf::", "")[:36]: round(x, 1) for k, x in a.get("stages_us", {}).items()}))
except Exception as e:
    print("$v FAILED", e)
PY
  for B in 256 4096 16384; do timeout 200 python tools/time_osc.py one $B 2>&1 | tail -1; done
done 2>&1 | tee $O/summary.txt
