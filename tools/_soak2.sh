cfg() { name=$1; shift
  for s in 7 31 101 202 303 404 505 606 707 808 909; do
    env "$@" timeout 300 python tools/fuzz_tiers.py 120 $s 2>&1 | grep -E "FAIL|failures" | awk -v n="$name s$s" '{print n " | " $0}'
  done
  env "$@" python tools/recipe_latency.py 32 2>&1 | grep seed | awk -v n="$name" '{t+=$3; h+=$6; t3+=$8; print n " | " $0} END {print n " | MEAN " t/NR " hot " h " t3 " t3}'
}
cfg C1 GOLF_SS_PHI_GUARD=20 GOLF_SS_PHI_GUARD2=8 > gpurun_out/soak_C1.txt 2>&1
cfg C2 GOLF_SS_PHI_GUARD2=8 GOLF_SS_HOT_COUNT=128 GOLF_SS_HOT_ALL_16THS=15 > gpurun_out/soak_C2.txt 2>&1
cfg C0 A=1 > gpurun_out/soak_C0.txt 2>&1
