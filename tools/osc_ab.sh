#!/bin/bash
# Runs ON THE GPU BOX: A/B of oscillator builds (golf_amd/lib/libgolf_<tag>.so; "shipped" = libgolf_hip.so; "old" = the shipped library
# with GOLF_OSCF_OLD=1) on bench.py --workload osc-only: 4 batches in flight + one batch alone at B = 32, and B = 16384.
# usage: bash tools/osc_ab.sh OUTDIR tag1 tag2 ...
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$1; shift
mkdir -p $O
export TMPDIR=/tmp
cd $R
for t in "$@"; do
  unset GOLF_HIP_LIBRARY GOLF_OSCF_OLD
  case $t in
    shipped) ;;
    old) export GOLF_OSCF_OLD=1 ;;
    *) export GOLF_HIP_LIBRARY=$R/golf_amd/lib/libgolf_$t.so ;;
  esac
  timeout 300 python bench.py --no-cpu-baseline --recipe-stream 0 --workload osc-only 2>$O/err_${t}_32.txt | tail -1 > $O/osc32_$t.json
  timeout 300 python bench.py --no-cpu-baseline --recipe-stream 0 --workload osc-only --batch 16384 --streams 1 --steps 10 --warmup 2 2>$O/err_${t}_16k.txt | tail -1 > $O/osc16k_$t.json
  python - <<PY
import json
try:
    a = json.load(open("$O/osc32_$t.json")); b = json.load(open("$O/osc16k_$t.json"))
    st = {k.replace("golf::", "")[:24]: round(v, 1) for k, v in a.get("stages_us", {}).items()}
    print("%-10s B=32: 4 in flight %6.2f us/step, alone graph %6.2f  stages %s | B=16384: %8.1f us/step = %6.1f G samples/s" % (
        "$t", a["ms_per_step"] * 1e3, a["single_stream"]["us_per_step_graph"], st, b["ms_per_step"] * 1e3, b["value"] / 1e9))
except Exception as e:
    print("$t", "FAILED", e)
PY
done 2>&1 | tee $O/summary.txt
