#!/bin/bash
# Prefill kernel ablations (tools/_abl/abl<bits>/libawq_hip.so built with -DPF_ABL=<bits>): time and SQ counters at the one-round
# shape, to separate "matrix pipe idle" from "clock lower under load".
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pf_abl
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for A in 0 15 1 4 5; do
  if [ $A = 0 ]; then export LD_LIBRARY_PATH=$R/sglang_awq_amd/lib; else export LD_LIBRARY_PATH=$R/tools/_abl/abl$A; fi
  echo "== PF_ABL=$A" >> $O/time.log
  AWQ_PF_SPLIT=0 $R/tools/kbench rgemm 2048 11008 4096 128 8 96 >> $O/time.log 2>&1
  AWQ_PF_SPLIT=0 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $O/pmc_$A -o sq -- $R/tools/kbench rgemm 2048 11008 4096 128 4 8 0 > $O/pmc_$A.log 2>&1
done
echo done
