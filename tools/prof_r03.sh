#!/bin/bash
# Round-3 evidence run on the GPU box: the driver's bench command, then rocprofv3 passes behind each record of its JSON line.
# Outputs under gpurun_out/r03/; the summaries that are cited get copied to profiles/ by hand (tools/prof_r03_collect.sh).
# PMC passes are separate runs with --kernel-trace only (FETCH_SIZE and WRITE_SIZE do not fit one pass: MI355X_MICROARCH.md).
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03
mkdir -p $O
cd $R
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_m1 -o m1 -- python3 $R/bench.py --steps 800 --warmup 100 --cpu-seconds 0 --sections '' > $O/prof_m1.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_sections -o sec -- python3 $R/bench.py --steps 20 --warmup 5 --cpu-seconds 0 --sections prefill,dequantize,op,large > $O/prof_sections.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_dec1 -o d1 -- python3 $R/bench_decode.py --batches 1 --steps 32 --cpu-seconds 0 > $O/prof_dec1.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_dec32 -o d32 -- python3 $R/bench_decode.py --batches 32 --steps 16 --cpu-seconds 0 > $O/prof_dec32.log 2>&1
# HBM traffic of the three HBM / MFMA records: decode GEMV (M = 1), awq_dequantize, prefill (M = 2048: two launches per call)
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o f -- $R/tools/kbench rgemm 1 4096 11008 128 16 64 0 > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o w -- $R/tools/kbench rgemm 1 4096 11008 128 16 64 0 > $O/pmc_write.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_dq -o f -- $R/tools/kbench dequant 4096 11008 128 0 8 32 > $O/pmc_fetch_dq.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write_dq -o w -- $R/tools/kbench dequant 4096 11008 128 0 8 32 > $O/pmc_write_dq.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_pf -o f -- $R/tools/kbench rgemm 2048 4096 11008 128 16 32 0 > $O/pmc_fetch_pf.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write_pf -o w -- $R/tools/kbench rgemm 2048 4096 11008 128 16 32 0 > $O/pmc_write_pf.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $O/pmc_pf_sq -o sq -- $R/tools/kbench rgemm 2048 4096 11008 128 4 8 0 > $O/pmc_pf_sq.log 2>&1 || true
echo done
