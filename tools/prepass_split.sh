#!/bin/bash
# Runs ON THE GPU BOX: durations of the composite-only and the scan-only launch of lpc_group_prepass_kernel
# (bench.py --overlap-transitions issues them separately), from a rocprofv3 kernel trace.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pp
timeout 300 rocprofv3 --kernel-trace -d /tmp/pp -- python $R/bench.py --no-cpu-baseline --streams 1 --no-graphs --steps 30 --warmup 5 --overlap-transitions > /dev/null 2>&1
db=$(find /tmp/pp -name "*.db" | head -1)
python - <<PY
import sqlite3, collections
con = sqlite3.connect("$db"); cur = con.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]; ks = [t for t in tabs if "kernel_symbol" in t][0]
rows = cur.execute(f"select s.kernel_name, d.end - d.start, d.grid_size_x from {kd} d join {ks} s on d.kernel_id = s.id").fetchall()
agg = collections.defaultdict(list)
for n, dur, gx in rows:
    if "prepass" in n: agg[gx].append(dur)
for k, v in sorted(agg.items()): print("prepass grid", k, "launches", len(v), "avg ns", round(sum(v) / len(v)))
PY
