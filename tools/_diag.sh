python tools/fuzz_diag.py 31 11 > gpurun_out/diag_31_11.txt 2>&1
python tools/fuzz_diag.py 31 115 > gpurun_out/diag_31_115.txt 2>&1
python tools/fuzz_diag.py 909 55 > gpurun_out/diag_909_55.txt 2>&1
python tools/fuzz_diag.py 606 90 bwd > gpurun_out/diag_606_90.txt 2>&1
python tools/fuzz_diag.py 808 57 bwd > gpurun_out/diag_808_57.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_modules.py -x -q -k "replay_pipeline" > gpurun_out/pipe_tests.txt 2>&1
tail -5 gpurun_out/pipe_tests.txt
