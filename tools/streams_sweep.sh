for s in 3 4 5 6 8 12; do
  python bench.py --steps 480 --warmup 48 --no-cpu-baseline --streams $s 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('streams $s', 'ms_per_step', round(d['ms_per_step'],4), 'G samples/s', round(d['value']/1e9,2))"
done
