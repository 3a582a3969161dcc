#!/bin/bash
# Ring depth of the long straight-line GEMV (lab build in sglang_awq_amd/lib_ab: -DAWQ_LAB, AWQ_RP2_D = loads in flight per wave)
out=${1:-gpurun_out/ab_depth.log}
: > $out
run() { echo -n "[$1] " >> $out; env LD_LIBRARY_PATH=sglang_awq_amd/lib_ab $1 tools/kbench rgemm $2 $3 $4 128 ${5:-16} 1600 >> $out 2>&1 || exit 1; }
for rep in 1 2; do
for shape in "1 8192 28672 8" "1 28672 8192 8" "1 14336 4096"; do
  set -- $shape
  for d in 2 3 4 6; do run "AWQ_RP2_D=$d" $1 $2 $3 $4; done
done
done
