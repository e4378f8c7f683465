set -x
for s in 31 909 606 808; do timeout 400 python tools/fuzz_tiers.py 120 $s > gpurun_out/fuzz_$s.txt 2>&1; tail -1 gpurun_out/fuzz_$s.txt; grep FAIL gpurun_out/fuzz_$s.txt; done
