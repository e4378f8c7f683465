timeout 1500 python -m pytest tests/test_gpu_lpc_ss.py tests/test_gpu_modules.py -x -q > gpurun_out/val1_tests.txt 2>&1
tail -5 gpurun_out/val1_tests.txt
for s in 7 31 101 202 303 404 505 606 707 808 909 1001 1002 1003; do
  timeout 300 python tools/fuzz_tiers.py 120 $s 2>&1 | grep -E "FAIL|failures" | awk -v n="s$s" '{print n " | " $0}'
done > gpurun_out/val1_soak.txt 2>&1
cat gpurun_out/val1_soak.txt
