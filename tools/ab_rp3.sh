#!/bin/bash
# the knobs below are read by the laboratory build only: make -C sglang_awq_amd/csrc lab  (-DAWQ_LAB -> sglang_awq_amd/lib_ab)
export LD_LIBRARY_PATH=sglang_awq_amd/lib_ab:$LD_LIBRARY_PATH
# A/B of the loop-form decode GEMV (gemv_rp3_kernel, AWQ_RP3=1) against the round-1 loop kernel (AWQ_RP3=0) with tools/kbench rgemm:
# 16 rotating weight sets, graph replay.  KBENCH_WS=0: no split-K scratch, so 9..16 rows take the one-strip route on both sides.
out=${1:-gpurun_out/ab_rp3.log}
: > $out
run() { echo -n "[$1] " >> $out; env $1 tools/kbench rgemm $2 $3 $4 128 ${5:-16} 1600 >> $out 2>&1 || exit 1; }
for shape in "1 8192 28672 8" "1 28672 8192 8" "1 11008 4096" "6 11008 4096" "8 11008 4096" "12 11008 4096" "16 11008 4096" "16 4096 11008" "12 8192 7168" "16 8192 1280" "8 28672 8192 8"; do
  set -- $shape
  for cfg in "AWQ_RP3=0 KBENCH_WS=0" "AWQ_RP3=1 KBENCH_WS=0" "AWQ_RP3=1 KBENCH_WS=1"; do
    run "$cfg" $1 $2 $3 $4
  done
done
