#!/bin/bash
# the knobs below are read by the laboratory build only: make -C sglang_awq_amd/csrc lab  (-DAWQ_LAB -> sglang_awq_amd/lib_ab)
export LD_LIBRARY_PATH=sglang_awq_amd/lib_ab:$LD_LIBRARY_PATH
out=${1:-gpurun_out/ab_pf.log}
: > $out
for shape in "2048 4096 11008" "2048 11008 4096" "2048 4096 12288" "2048 4096 22016" "1024 4096 11008" "4096 4096 11008" "512 4096 11008" "8192 4096 11008"; do
set -- $shape
  for cfg in "AWQ_PF_SPLIT=0" "AWQ_PF_SPLIT=1"; do
    echo -n "[$cfg] " >> $out
    env $cfg tools/kbench rgemm $1 $2 $3 128 8 96 >> $out 2>&1 || echo "(failed)" >> $out
  done
done
