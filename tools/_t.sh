timeout 1200 python -m pytest tests/test_gpu_osc.py -x -q > gpurun_out/t_osc.txt 2>&1; tail -5 gpurun_out/t_osc.txt
python bench.py --workload osc-only --steps 200 --warmup 20 --no-cpu-baseline --recipe-stream 0 --refresh-inputs 0 > gpurun_out/b_osc_only.json 2> gpurun_out/b_osc_only.err; tail -c 1500 gpurun_out/b_osc_only.json
python bench.py --workload osc-only --batch 16384 --streams 1 --steps 5 --warmup 2 --repeats 3 --settle 0 --no-cpu-baseline --recipe-stream 0 --refresh-inputs 0 > gpurun_out/b_osc_16384.json 2> gpurun_out/b_osc_16384.err; python -c "
import json; r=json.loads(open('gpurun_out/b_osc_16384.json').read().strip().splitlines()[-1]); print('B=16384 osc-only', r['value']/1e9, 'G/s', r['ms_per_step'], r['stages_us'])"
python bench.py --steps 20 --warmup 5 > gpurun_out/b_main.json 2> gpurun_out/b_main.err; python -c "
import json; r=json.loads(open('gpurun_out/b_main.json').read().strip().splitlines()[-1]); print('headline', r['ms_per_step'], r['ms_per_step_unsettled'], r['single_stream'], r['stages_us'], r.get('refreshed_inputs',{}).get('vs_fixed_inputs'))"
