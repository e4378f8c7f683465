cd $GRAFT_REPO_ROOT
O=gpurun_out/r30; mkdir -p $O
for t in fqt fqp; do
export GOLF_HIP_LIBRARY=$PWD/golf_amd/lib/libgolf_$t.so
python tools/fwdq2_phases.py > $O/phases_$t.txt 2>&1
python bench.py --no-cpu-baseline --recipe-stream 0 --streams 1 --lpc-chain latency 2>/dev/null | tail -1 | python -c "
import json,sys; a=json.loads(sys.stdin.read()); print('$t 1-stream latency chain', a['ms_per_step']*1e3, a['stages_us'])" >> $O/phases_$t.txt
done
