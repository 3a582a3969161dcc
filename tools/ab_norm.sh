#!/bin/bash
# the knobs below are read by the laboratory build only: make -C sglang_awq_amd/csrc lab  (-DAWQ_LAB -> sglang_awq_amd/lib_ab)
export LD_LIBRARY_PATH=sglang_awq_amd/lib_ab:$LD_LIBRARY_PATH
out=${1:-gpurun_out/ab_norm.log}
: > $out
for shape in "4096 12288 1" "4096 22016 3"; do
set -- $shape
for M in 1 2 4 8; do
  for cfg in "AWQ_RP2_NORM=0" "AWQ_RP2_NORM=1"; do
    echo -n "[$cfg] " >> $out
    env $cfg tools/kbench rgemm $M $1 $2 128 16 1600 1 $3 >> $out 2>&1 || echo "(no kernel)" >> $out
  done
done
done
