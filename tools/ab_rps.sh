#!/bin/bash
# the knobs below are read by the laboratory build only: make -C sglang_awq_amd/csrc lab  (-DAWQ_LAB -> sglang_awq_amd/lib_ab)
export LD_LIBRARY_PATH=sglang_awq_amd/lib_ab:$LD_LIBRARY_PATH
# split-K GEMV for 9..32 rows on narrow matrices (awq_repacked_splitk.hip; KBENCH_WS=0: the one-strip kernels)
for s in "11008 4096" "4096 4096" "8192 1280" "3584 8192" "8192 8192" "4096 2560"; do
  for m in 12 16 17 24 32; do
    for w in 0 1; do echo -n "[ws=$w] "; KBENCH_WS=$w tools/kbench rgemm $m $s 128 16 400; done
  done
done
