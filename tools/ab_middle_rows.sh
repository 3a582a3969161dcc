#!/bin/bash
# 33 .. 256 rows: product dispatch (passes of the 32-row GEMV while route::gemv_passes_max allows, then K-split or MFMA tiles) against
# round 2's rule (passes up to 160 rows: laboratory build, AWQ_PASSES_MAX=160) and against tiles from 33 rows on (AWQ_PASSES_MAX=32 AWQ_MID=0)
out=${1:-gpurun_out/ab_middle_rows.log}
: > $out
for shape in "4096 11008" "4096 4096" "11008 4096" "4096 22016" "8192 8192" "8192 28672"; do
  set -- $shape
  for m in 48 64 96 128 160 192 256; do
    echo -n "[product] " >> $out; env LD_LIBRARY_PATH=sglang_awq_amd/lib tools/kbench rgemm $m $1 $2 128 8 200 >> $out 2>&1 || echo "(failed)" >> $out
    echo -n "[passes to 160 rows] " >> $out; env LD_LIBRARY_PATH=sglang_awq_amd/lib_ab AWQ_PASSES_MAX=160 tools/kbench rgemm $m $1 $2 128 8 200 >> $out 2>&1 || echo "(failed)" >> $out
    echo -n "[tiles only] " >> $out; env LD_LIBRARY_PATH=sglang_awq_amd/lib_ab AWQ_PASSES_MAX=32 AWQ_MID=0 tools/kbench rgemm $m $1 $2 128 8 200 >> $out 2>&1 || echo "(failed)" >> $out
  done
done
