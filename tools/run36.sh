cd $GRAFT_REPO_ROOT
O=gpurun_out/r36; mkdir -p $O; rm -f $O/*
for b in 16 32 51 64 102 153 204; do python tools/time_maps.py $b 2>&1 | tail -1 >> $O/summary.txt; done
