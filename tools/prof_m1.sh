#!/bin/bash
# rocprofv3 passes behind the M = 1 roofline record: kernel-trace stats of bench.py's headline, and two PMC passes of the kernel
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_m1 -o m1 -- python3 $R/bench.py --steps 800 --warmup 100 --cpu-seconds 0 --sections '' > $R/gpurun_out/prof_m1_bench.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_fetch -o f -- $R/tools/kbench rgemm 1 4096 11008 128 16 64 0 > $R/gpurun_out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_write -o w -- $R/tools/kbench rgemm 1 4096 11008 128 16 64 0 > $R/gpurun_out/pmc_write.log 2>&1
find $R/gpurun_out/prof_m1 $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write -name "*.csv" | head -20
