timeout 2400 python -m pytest tests/ -x -q -m gpu > gpurun_out/val2_tests.txt 2>&1; tail -4 gpurun_out/val2_tests.txt
for s in 31 606 808 909 101; do
  timeout 300 python tools/fuzz_tiers.py 120 $s 2>&1 | grep -E "FAIL|failures" | awk -v n="s$s" '{print n " | " $0}'
done > gpurun_out/val2_soak.txt 2>&1
cat gpurun_out/val2_soak.txt
bash tools/refresh_profiles.sh > gpurun_out/refresh_r06.log 2>&1
tail -3 gpurun_out/refresh_r06.log
