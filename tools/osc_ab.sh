#!/bin/bash
# Runs ON THE GPU BOX: A/B of oscillator builds (golf_amd/lib/libgolf_v*.so, GOLF_HIP_LIBRARY): parity test of the fused
# kernel, the source alone at B = 32 / 16384, the headline step.  usage: bash tools/osc_ab.sh v0 v1 ...
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for v in "$@"; do
  export GOLF_HIP_LIBRARY=$R/golf_amd/lib/libgolf_$v.so
  echo "== $v"
  timeout 300 python -m pytest tests/test_gpu_osc.py -m gpu -x -q 2>&1 | tail -2
  timeout 200 python tools/time_osc.py one 32
  timeout 200 python tools/time_osc.py one 16384
  timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('headline us/step', round(d['ms_per_step']*1e3,1), 'single', d['single_stream']['us_per_step_graph'])"
done
