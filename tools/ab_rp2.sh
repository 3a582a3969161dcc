#!/bin/bash
# the knobs below are read by the laboratory build only: make -C sglang_awq_amd/csrc lab  (-DAWQ_LAB -> sglang_awq_amd/lib_ab)
export LD_LIBRARY_PATH=sglang_awq_amd/lib_ab:$LD_LIBRARY_PATH
# A/B of the decode GEMV variants over the shapes of the 7B / 70B linears (tools/kbench rgemm; 16 rotating weight sets, graph replay)
out=${1:-gpurun_out/ab_rp2.log}
: > $out
for shape in "4096 11008" "4096 22016" "11008 4096" "8192 7168" "4096 12288"; do
  set -- $shape
  for M in 1 4 16; do
    for cfg in "AWQ_RP2=0" "AWQ_RP2=1 AWQ_RP2_D=2" "AWQ_RP2=1 AWQ_RP2_D=0"; do
      echo -n "[$cfg] " >> $out
      env $cfg tools/kbench rgemm $M $1 $2 128 16 1600 >> $out 2>&1 || exit 1
    done
  done
done
