#!/bin/bash
# the knobs below are read by the laboratory build only: make -C sglang_awq_amd/csrc lab  (-DAWQ_LAB -> sglang_awq_amd/lib_ab)
export LD_LIBRARY_PATH=sglang_awq_amd/lib_ab:$LD_LIBRARY_PATH
out=${1:-gpurun_out/ab_small.log}
: > $out
for shape in "4096 4096" "8192 1280" "1024 8192" "5120 5120"; do
set -- $shape
for M in 1 8; do
  for cfg in "AWQ_RP2=1" "AWQ_RP_NT=1 AWQ_RP_WAVES=16" "AWQ_RP_NT=0 AWQ_RP_WAVES=16"; do
    echo -n "[$cfg] " >> $out
    env $cfg tools/kbench rgemm $M $1 $2 128 16 1600 >> $out 2>&1 || echo "(no kernel)" >> $out
  done
done
done
