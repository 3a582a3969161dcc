#!/bin/bash
# dense prefill GEMM: output tile stored through LDS in 16-byte lanes (product) against straight from the accumulators
#   make -C sglang_awq_amd/csrc OBJDIR=_obj_e0 OUT=../lib_e0/libawq_hip.so EXTRA=-DPF_LDS_EPILOGUE=0
out=${1:-gpurun_out/ab_pf_epilogue.log}
: > $out
for rep in 1 2 3; do
for shape in "2048 4096 11008" "4096 4096 11008" "2048 11008 4096" "512 4096 11008" "8192 4096 11008" "2048 8192 28672"; do
  set -- $shape
  for lib in lib_e0 lib; do
    echo -n "[$lib] " >> $out
    env LD_LIBRARY_PATH=sglang_awq_amd/$lib tools/kbench rgemm $1 $2 $3 128 8 96 >> $out 2>&1 || echo "(failed)" >> $out
  done
done
done
