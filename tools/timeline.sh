#!/bin/bash
# dev: kernel timeline of the pipelined headline run (csv), analysed by tools/timeline.py
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/tl; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O -- python $R/bench.py --no-cpu-baseline --steps 120 --warmup 20 > $O/log.txt 2>&1
cd $R && python tools/timeline.py $(find $O -name "*kernel_trace.csv" | head -1)
find $O -name "*.csv" -size +20M -delete
