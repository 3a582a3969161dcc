#!/bin/bash
# 33 .. 512 rows: the split-K tile route (under-filled MFMA tiles with K split over workgroups + a reduce launch; product when the
# dispatcher picks it) against the same library with the route off (laboratory build: AWQ_PF_SK=0)
export LD_LIBRARY_PATH=sglang_awq_amd/lib_ab:$LD_LIBRARY_PATH
out=${1:-gpurun_out/ab_split_tiles.log}
: > $out
for shape in "4096 4096" "4096 11008" "11008 4096" "4096 22016" "8192 8192" "8192 28672" "4096 12288"; do
  set -- $shape
  for m in 48 64 96 128 192 256 384 512; do
    for cfg in ${CFGS:-"AWQ_PF_SK=0" "AWQ_PF_SK=1"}; do
      echo -n "[$cfg] " >> $out
      env $cfg tools/kbench rgemm $m $1 $2 128 8 200 >> $out 2>&1 || echo "(failed)" >> $out
    done
  done
done
