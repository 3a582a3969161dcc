#!/bin/bash
# A/B of launch-side settings on the headline decode GEMV (tools/kbench rgemm, 16 rotating weight sets unless noted, graph replay):
#   HIP_FORCE_DEV_KERNARG = 0 / 1   kernel arguments fetched from host-coherent or device memory
#   lib_ab                        the library built with -mllvm -amdgpu-kernarg-preload-count=16 (arguments preloaded into SGPRs)
#   sets = 1                      every launch on the same 23 MB (cache-resident; 16 launches per graph)
out=${1:-gpurun_out/ab_launch.log}
: > $out
run() { echo -n "[$1] " >> $out; env $1 tools/kbench rgemm $2 $3 $4 128 ${5:-16} 3200 >> $out 2>&1 || exit 1; }
for rep in 1 2; do
  for shape in "1 4096 11008" "1 4096 4096" "1 4096 22016" "4 4096 11008"; do
    set -- $shape
    run "HIP_FORCE_DEV_KERNARG=0" $1 $2 $3
    run "HIP_FORCE_DEV_KERNARG=1" $1 $2 $3
    run "HIP_FORCE_DEV_KERNARG=1 LD_LIBRARY_PATH=sglang_awq_amd/lib_ab" $1 $2 $3
    run "HIP_FORCE_DEV_KERNARG=0 LD_LIBRARY_PATH=sglang_awq_amd/lib_ab" $1 $2 $3
  done
done
for s in 1 2 4; do run "A=1" 1 4096 11008 $s; done
run "AWQ_RP_NT=0" 1 4096 11008 1
run "A=1" 1 4096 4096 1
run "A=1" 1 4096 4096 2
