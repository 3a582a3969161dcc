#!/bin/bash
# Producer / consumer prefill kernel: which role bounds a k-step?  (tools/_abl/pc<bits>: -DPC_ABL=<bits>)
R=$GRAFT_REPO_ROOT
for A in 0 1 2 3 4 7; do
  if [ $A = 0 ]; then export LD_LIBRARY_PATH=$R/sglang_awq_amd/lib; else export LD_LIBRARY_PATH=$R/tools/_abl/pc$A; fi
  echo -n "PC_ABL=$A  "
  AWQ_PF_PC=1 AWQ_PF_SPLIT=0 timeout -k 10 60 $R/tools/kbench rgemm 2048 11008 4096 128 8 96
done
