#!/bin/bash
# A/B of progress-tied wave priorities in the other decode GEMV kernels (gemv_repacked_kernel, gemv_rp3_kernel, gemv_rps_kernel):
# product against a build with -DRP_DYNPRIO=0:  make -C sglang_awq_amd/csrc OBJDIR=_obj_p2 OUT=../lib_q0/libawq_hip.so EXTRA=-DRP_DYNPRIO=0
out=${1:-gpurun_out/ab_prio2.log}
: > $out
for rep in 1 2; do
  for shape in "16 4096 11008" "16 11008 4096" "12 11008 4096" "32 4096 11008" "32 11008 4096" "32 4096 12288" "24 4096 22016" "16 8192 28672" "1 8192 57344"; do
    set -- $shape
    for lib in lib_q0 lib; do
      echo -n "[$lib] " >> $out
      env LD_LIBRARY_PATH=sglang_awq_amd/$lib tools/kbench rgemm $1 $2 $3 128 16 1600 >> $out 2>&1 || echo "(failed)" >> $out
    done
  done
done
for lib in lib_q0 lib; do
  echo "[$lib] time_moe" >> $out
  AWQ_LAB_LIB=sglang_awq_amd/$lib/libawq_hip.so python3 tools/time_moe.py 8 4096 14336 2 8,16,64 2>&1 | grep "T=" >> $out
done
