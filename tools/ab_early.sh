#!/bin/bash
# the knobs below are read by the laboratory build only: make -C sglang_awq_amd/csrc lab  (-DAWQ_LAB -> sglang_awq_amd/lib_ab)
export LD_LIBRARY_PATH=sglang_awq_amd/lib_ab:$LD_LIBRARY_PATH
# gemv_rp2_kernel: replacement load issued before (sglang_awq_amd/lib) vs behind (tools/_abl/late) the unit's compute
R=$GRAFT_REPO_ROOT
for shape in "1 4096 11008" "1 11008 4096" "1 4096 12288" "1 4096 22016" "1 4096 4096" "4 4096 11008" "8 4096 12288" "16 4096 11008" "1 8192 10240" "1 8192 1280"; do
  for L in tools/_abl/late sglang_awq_amd/lib; do
    export LD_LIBRARY_PATH=$R/$L
    echo -n "[$L] "; $R/tools/kbench rgemm $shape 128 16 800
  done
done
