#!/bin/bash
# the knobs below are read by the laboratory build only: make -C sglang_awq_amd/csrc lab  (-DAWQ_LAB -> sglang_awq_amd/lib_ab)
export LD_LIBRARY_PATH=sglang_awq_amd/lib_ab:$LD_LIBRARY_PATH
# Deep / wide one-row shapes: the long straight-line form of gemv_rp2_kernel (up to 32 units per wave) against the round-1 loop kernel
# (AWQ_RP2=0), tools/kbench rgemm, rotating weight sets (> 2 x the Infinity Cache), graph replay
out=${1:-gpurun_out/ab_long.log}
: > $out
run() { echo -n "[$1] " >> $out; env $1 tools/kbench rgemm $2 $3 $4 128 ${5:-16} 1600 >> $out 2>&1 || exit 1; }
for rep in 1 2; do
for shape in "1 8192 28672 8" "1 28672 8192 8" "1 8192 10240 8" "1 8192 7168" "1 4096 11008" "1 14336 4096"; do
  set -- $shape
  for cfg in "AWQ_RP2=0" "AWQ_RP2=1"; do
    run "$cfg" $1 $2 $3 $4
  done
done
done
