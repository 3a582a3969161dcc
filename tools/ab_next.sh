#!/bin/bash
# A/B of the next-weights hint (awq_hip.h: awq_next_hint) and of the unit order of gemv_rp2_kernel (c-major vs t-major build in
# sglang_awq_amd/lib_ab) with tools/kbench rgemm: 16 rotating weight sets, graph replay, each launch hints the next set.
out=${1:-gpurun_out/ab_next.log}
: > $out
run() { echo -n "[$1] " >> $out; env $1 tools/kbench rgemm $2 $3 $4 128 16 3200 >> $out 2>&1 || exit 1; }
for rep in 1 2; do
  for shape in "1 4096 11008"; do
    set -- $shape
    echo -n "[t-major lib] " >> $out
    LD_LIBRARY_PATH=sglang_awq_amd/lib_ab:$LD_LIBRARY_PATH tools/kbench rgemm $1 $2 $3 128 16 3200 >> $out 2>&1 || exit 1
    run "KBENCH_NEXT=0" $1 $2 $3
    for kib in 1024 2048 4096 8192 12288; do
      for wmin in 0 8 12; do
        run "KBENCH_NEXT=$kib KBENCH_NEXT_WMIN=$wmin" $1 $2 $3
      done
    done
  done
done
for shape in "1 4096 22016" "1 11008 4096" "1 4096 4096" "1 8192 7168" "4 4096 11008"; do
  set -- $shape
  echo -n "[t-major lib] " >> $out
  LD_LIBRARY_PATH=sglang_awq_amd/lib_ab:$LD_LIBRARY_PATH tools/kbench rgemm $1 $2 $3 128 16 3200 >> $out 2>&1 || exit 1
  run "KBENCH_NEXT=0" $1 $2 $3
  for kib in 2048 6144; do
    for wmin in 0 8; do
      run "KBENCH_NEXT=$kib KBENCH_NEXT_WMIN=$wmin" $1 $2 $3
    done
  done
done
