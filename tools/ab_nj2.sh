#!/bin/bash
# the knobs below are read by the laboratory build only: make -C sglang_awq_amd/csrc lab  (-DAWQ_LAB -> sglang_awq_amd/lib_ab)
export LD_LIBRARY_PATH=sglang_awq_amd/lib_ab:$LD_LIBRARY_PATH
# 128 x 128 prefill tiles (NJ = 2) against the wide / 192-wide split at mid M
for shape in "192 4096 11008" "256 4096 11008" "320 4096 11008" "384 4096 11008" "512 4096 11008" "256 4096 22016" "384 4096 12288" "256 8192 28672" "2048 4096 11008"; do
  for e in 0 1; do echo -n "[AWQ_PF_NJ2=$e] "; AWQ_PF_NJ2=$e tools/kbench rgemm $shape 128 8 96; done
done
