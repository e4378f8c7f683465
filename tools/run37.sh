cd $GRAFT_REPO_ROOT
O=gpurun_out/r37; mkdir -p $O; rm -f $O/*
python bench.py --no-cpu-baseline --recipe-stream 0 --steps 20 --warmup 5 --repeats 1500 --headline-only 2>/dev/null | tail -1 | python -c "
import json,sys; a=json.loads(sys.stdin.read()); r=a['timing']['ms_per_step_regions_wall']; n=len(r)
print('regions', n, 'median', a['ms_per_step']*1e3)
for i in range(0,n,50): print(i, [round(x*1e3,1) for x in r[i:i+5]])
" > $O/ramp.txt
python bench.py --no-cpu-baseline --recipe-stream 0 --steps 20 --warmup 5 --prereplay 4000 2>/dev/null | tail -1 | python -c "
import json,sys; a=json.loads(sys.stdin.read()); print('prereplay 4000:', a['ms_per_step']*1e3, [round(x*1e3,1) for x in a['timing']['ms_per_step_regions_wall']], a['single_stream']['us_per_step_graph'])" >> $O/ramp.txt
