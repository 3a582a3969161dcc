#!/bin/bash
# dense prefill GEMM: tile order.  [lib] = product (row-fastest inside super-rows of 4 row tiles); [lib_rbN] = -DPF_ROW_BLOCK=N (0: column tiles of
# one row tile consecutive, the order up to round 3):  make -C sglang_awq_amd/csrc OBJDIR=_obj_rbN OUT=../lib_rbN/libawq_hip.so EXTRA=-DPF_ROW_BLOCK=N
out=${1:-gpurun_out/ab_pf_order.log}
: > $out
for rep in 1 2; do
for shape in "2048 4096 11008" "4096 4096 11008" "2048 11008 4096" "8192 4096 11008" "1024 4096 11008"; do
  set -- $shape
  for lib in ${LIBS:-lib_rb0 lib}; do
    echo -n "[$lib] " >> $out
    env LD_LIBRARY_PATH=sglang_awq_amd/$lib tools/kbench rgemm $1 $2 $3 128 8 96 >> $out 2>&1 || echo "(failed)" >> $out
  done
done
done
