#!/bin/bash
# Runs ON THE GPU BOX: A/B of in-tree library builds (GOLF_HIP_LIBRARY=golf_amd/lib/libgolf_<tag>.so) on the headline bench:
# 200-step pipelined rate, the driver's 20-step command, single-stream latency.  usage: bash tools/ab2.sh OUTDIR tag1 tag2 ...
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$1; shift
mkdir -p $O
export TMPDIR=/tmp
cd $R
for rep in 1 2; do
for t in "$@"; do
  export GOLF_HIP_LIBRARY=$R/golf_amd/lib/libgolf_$t.so
  [ "$t" == "shipped" ] && unset GOLF_HIP_LIBRARY
  timeout 300 python bench.py --no-cpu-baseline --recipe-stream 0 $AB_ARGS 2>/dev/null | tail -1 > $O/b200_${t}_$rep.json
  timeout 300 python bench.py --no-cpu-baseline --recipe-stream 0 --steps 20 --warmup 5 $AB_ARGS 2>/dev/null | tail -1 > $O/b20_${t}_$rep.json
  python - <<PY
import json
a = json.load(open("$O/b200_${t}_$rep.json")); b = json.load(open("$O/b20_${t}_$rep.json"))
print("%-10s rep $rep  200-step %6.1f  20-step %6.1f %s  single graph %6.1f eager %6.1f  %s" % ("$t", a["ms_per_step"] * 1e3, b["ms_per_step"] * 1e3,
      [round(x * 1e3, 1) for x in b["timing"]["ms_per_step_regions_wall"]], a["single_stream"]["us_per_step_graph"], a["single_stream"]["us_per_step_eager"],
      {k.replace("golf::", "").split("<")[0][-14:] + k[k.find("<"):][-4:]: v for k, v in a["stages_us"].items()}))
PY
done
done 2>&1 | tee $O/summary.txt
