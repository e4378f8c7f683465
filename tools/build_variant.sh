#!/bin/bash
# dev: build golf_amd/lib/libgolf_<tag>.so from the current sources with extra hipcc flags for ONE or more translation units;
# the other objects are taken from golf_amd/lib/*.hip.o (run `python -c "import golf_amd._lib as l; l.build()"` first).
# usage: tools/build_variant.sh TAG "lpc_ss.hip glottal_osc.hip" -DGOLF_GP_DC=3 ...     (GOLF_FULL=1: all (W, NT) instantiations)
set -e
R=$(cd $(dirname $0)/.. && pwd); L=$R/golf_amd/lib; C=$R/golf_amd/csrc
tag=$1; units=$2; shift 2
fast="-DGOLF_SS_ONLY_24_22"; [ -n "$GOLF_FULL" ] && fast=""
objs=""
for s in abi.hip lpc_ss.hip lpc_ff.hip glottal_osc.hip noise_fir.hip ctrl.hip noise_band.hip peer.hip; do
  if [[ " $units " == *" $s "* ]]; then
    o=/tmp/var_${tag}_$s.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -falign-loops=64 -I$R/include -I$C $fast "$@" -c $C/$s -o $o &
    objs="$objs $o"
  else
    objs="$objs $L/$s.o"
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $L/libgolf_$tag.so $objs
ls -la $L/libgolf_$tag.so
