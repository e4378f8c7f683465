#!/bin/bash
# Runs ON THE GPU BOX: A/B of in-tree library builds (GOLF_HIP_LIBRARY) on the default bench + a kernel trace of the
# eager single-stream step.  usage: bash tools/ab_libs.sh OUTDIR lib1.so lib2.so ...
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$1; shift
mkdir -p $O
export TMPDIR=/tmp
for l in "$@"; do
  export GOLF_HIP_LIBRARY=$R/golf_amd/lib/$l
  timeout 300 python $R/bench.py --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_$l.json
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace -d /tmp/p_$l -- python $R/bench.py --no-cpu-baseline --streams 1 --no-graphs --steps 50 --warmup 10 > /dev/null 2>&1)
  db=$(find /tmp/p_$l -name "*.db" | head -1)
  [ -n "$db" ] && python $R/tools/rocpd_summary.py $db $O/stats_$l.csv
  python - <<PY
import json
d = json.load(open("$O/bench_$l.json"))
print("$l", round(d["ms_per_step"] * 1e3, 1), d["single_stream"])
PY
  [ -f $O/stats_$l.csv ] && head -9 $O/stats_$l.csv | cut -c1-160
done
