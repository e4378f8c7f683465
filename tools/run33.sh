cd $GRAFT_REPO_ROOT
O=gpurun_out/r33; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > $O/pytest.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1
for w in golf-ss-train golf-ss-decoder-train; do
python bench.py --no-cpu-baseline --workload $w 2>/dev/null | tail -1 | python -c "
import json,sys; a=json.loads(sys.stdin.read()); print('$w', a['ms_per_step']*1e3, a['single_stream'])" >> $O/bench.txt
done
