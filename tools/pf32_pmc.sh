#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pf32_pmc
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export AWQ_PF_SPLIT=0
for e in 0 1; do
  export AWQ_PF32=$e
  i=0
  for grp in "SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" \
             "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS" \
             "SQ_WAVE_CYCLES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU"; do
    i=$((i+1))
    rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/e${e}_g$i -o sq -- $R/tools/kbench rgemm 2048 11008 4096 128 4 8 0 > $O/e${e}_g$i.log 2>&1 || echo "e$e group $i failed" >> $O/fail.log
  done
done
echo done
