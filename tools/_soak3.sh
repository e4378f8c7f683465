cfg() { name=$1; shift
  for s in 31 606 808 909; do
    env "$@" timeout 300 python tools/fuzz_tiers.py 120 $s 2>&1 | grep -E "FAIL|failures" | awk -v n="$name s$s" '{print n " | " $0}'
  done
  env "$@" python tools/recipe_latency.py 32 2>&1 | grep seed | awk -v n="$name" '{t+=$3; h+=$6; t3+=$8; print n " | " $0} END {print n " | MEAN " t/NR " hot " h " t3 " t3}'
}
cfg V1 GOLF_SS_PHI_GUARD2=8 GOLF_SS_HOT_COUNT=128 GOLF_SS_HOT_ALL_16THS=16 > gpurun_out/soak_V1.txt 2>&1
cfg V2 GOLF_SS_PHI_GUARD2=8 GOLF_SS_HOT_COUNT=160 GOLF_SS_HOT_ALL_16THS=16 > gpurun_out/soak_V2.txt 2>&1
cfg V3 GOLF_SS_PHI_GUARD2=8 GOLF_SS_HOT_COUNT=0 GOLF_SS_HOT_ALL_16THS=0 > gpurun_out/soak_V3.txt 2>&1
