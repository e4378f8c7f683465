for kt in 2 3 1; do
for s in 4; do
  GOLF_P1H_KT=$kt python bench.py --steps 400 --warmup 40 --no-cpu-baseline --streams $s 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('KT $kt streams $s', 'ms_per_step', round(d['ms_per_step'],4), 'G samples/s', round(d['value']/1e9,2))"
done
done
