#!/bin/bash
# per-kernel shares of a Llama-2-7B-AWQ decode step (batch 1 and 8), rocprofv3 kernel-trace stats
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_dec1 -o d1 -- python3 $R/bench_decode.py --batches 1 --steps 32 --cpu-seconds 0 > $R/gpurun_out/prof_dec1.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_dec8 -o d8 -- python3 $R/bench_decode.py --batches 8 --steps 32 --cpu-seconds 0 > $R/gpurun_out/prof_dec8.log 2>&1
