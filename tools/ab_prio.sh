#!/bin/bash
# A/B of the wave priorities inside the decode GEMV (gemv_rp2_kernel): product (s_setprio falls with a wave's progress) against a build
# with the static form of round 2:   make -C sglang_awq_amd/csrc OBJDIR=_obj_p0 OUT=../lib_p0/libawq_hip.so EXTRA=-DRP2_DYNPRIO=0
# (profiles/r03_kbench_wave_priority_ab.txt also holds three variants that were tried and dropped)
out=${1:-gpurun_out/ab_prio.log}
: > $out
for rep in 1 2; do
  for shape in "1 4096 11008" "1 4096 4096" "1 11008 4096" "1 4096 12288" "1 4096 22016" "1 8192 28672" "4 4096 11008" "8 4096 11008"; do
    set -- $shape
    for lib in ${LIBS:-lib_p0 lib}; do
      echo -n "[$lib] " >> $out
      env LD_LIBRARY_PATH=sglang_awq_amd/$lib tools/kbench rgemm $1 $2 $3 128 16 3200 >> $out 2>&1 || echo "(failed)" >> $out
    done
  done
done
