#!/bin/bash
# Product library: the decode GEMV family at 16 / 17 / 24 / 32 rows over the 7B shapes (is one two-row-tile launch better than two
# passes of the 16-row kernel?); KBENCH_WS=1 lets narrow deep matrices take the split-K route as the Python shim does
out=${1:-gpurun_out/ab_rows.log}
: > $out
run() { echo -n "[$1] " >> $out; env $1 tools/kbench rgemm $2 $3 $4 128 ${5:-16} 1600 >> $out 2>&1 || exit 1; }
for shape in "4096 11008" "4096 12288" "4096 22016" "4096 4096" "11008 4096"; do
  set -- $shape
  for M in 8 16 17 24 32; do run "KBENCH_WS=1" $M $1 $2; done
done
