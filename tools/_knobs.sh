run() { # name env...
  name=$1; shift
  for c in "31 11" "31 115" "909 55" "606 90 bwd" "808 57 bwd"; do
    env "$@" python tools/fuzz_diag.py $c 2>&1 | grep -v amdgpu.ids | awk -v n="$name" '{print n " | " $0}' | grep -E "seed|\(1\.[0-9]+\)|\([2-9]\.[0-9]+\)|bwd.*(1\.[0-9]|[2-9]\.)" 
  done
}
{
run base A=1
run g1_20 GOLF_SS_PHI_GUARD=20
run g1_16 GOLF_SS_PHI_GUARD=16
run g2_8 GOLF_SS_PHI_GUARD2=8
run g2_6 GOLF_SS_PHI_GUARD2=6
run g1_20_g2_6 GOLF_SS_PHI_GUARD=20 GOLF_SS_PHI_GUARD2=6
run glog90 GOLF_SS_GROUP_LOG2=90
run glog84 GOLF_SS_GROUP_LOG2=84
run all GOLF_SS_PHI_GUARD=20 GOLF_SS_PHI_GUARD2=6 GOLF_SS_GROUP_LOG2=90
} > gpurun_out/knobs.txt 2>&1
