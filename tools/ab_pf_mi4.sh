#!/bin/bash
# A/B on the dense prefill GEMM: 128 x 256 tiles (one workgroup per CU) against 64 x 256 tiles (two per CU), laboratory build
# (make -C sglang_awq_amd/csrc lab).  AWQ_PF_MI4: 0 product tiles, 1 one launch of 64 x 256 tiles, 2 64-row tiles with the wide / narrow split.
export LD_LIBRARY_PATH=sglang_awq_amd/lib_ab:$LD_LIBRARY_PATH
out=${1:-gpurun_out/ab_pf_mi4.log}
: > $out
for shape in "2048 4096 11008" "2048 11008 4096" "4096 4096 11008" "1024 4096 11008" "512 4096 11008" "8192 4096 11008" "2048 4096 12288" "2048 8192 28672"; do
set -- $shape
  for cfg in "AWQ_PF_MI4=0" "AWQ_PF_MI4=1" "AWQ_PF_MI4=2"; do
    echo -n "[$cfg] " >> $out
    env $cfg tools/kbench rgemm $1 $2 $3 128 8 96 >> $out 2>&1 || echo "(failed)" >> $out
  done
done
