#!/bin/bash
# dev: per-stream kernel timelines of the pipelined headline run: steady state (120 steps) and the driver's command (20 steps)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/tl2; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for cfg in "120 20" "20 5"; do
  set -- $cfg
  D=$O/s$1; mkdir -p $D
  rocprofv3 --kernel-trace --output-format csv -d $D -- python $R/bench.py --no-cpu-baseline --recipe-stream 0 --steps $1 --warmup $2 --repeats 3 > $D/log.txt 2>&1
  f=$(find $D -name "*kernel_trace.csv" | head -1)
  (cd $R && python tools/timeline2.py $f osc_tile_totals 2 3 > $O/timeline_s$1.txt 2>&1)
  find $D -name "*.csv" -size +8M -delete
done
cd $R
python bench.py --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err
python bench.py --no-cpu-baseline --recipe-stream 0 > $O/bench_200.json 2> $O/bench_200.err
