cd $GRAFT_REPO_ROOT
O=gpurun_out/r41; mkdir -p $O; rm -f $O/*
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > $O/pytest.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_driver.json
