#!/bin/bash
# Copy the round-3 summaries that DESIGN.md / bench.py cite from gpurun_out/r03 (scratch) into profiles/ (tracked).
set -e
cd "$(dirname "$0")/.."
O=gpurun_out/r03
P=profiles
cp $O/bench_driver_cmd.json $P/r03_bench_driver_cmd.json
cp $O/bench_default.json $P/r03_bench_default.json
f() { find "$1" -name "$2" | head -1; }
cp "$(f $O/prof_m1 '*kernel_stats.csv')" $P/r03_bench_m1_kernel_stats.csv
cp "$(f $O/prof_sections '*kernel_stats.csv')" $P/r03_bench_sections_kernel_stats.csv
cp "$(f $O/prof_dec1 '*kernel_stats.csv')" $P/r03_decode_7b_bs1_kernel_stats.csv
d32="$(f $O/prof_dec32 '*kernel_stats.csv')"; [ -n "$d32" ] && cp "$d32" $P/r03_decode_7b_bs32_kernel_stats.csv
# PMC passes: keep only the rows of the kernel of interest (the counter_collection csv lists every dispatch, repack included)
keep() { head -1 "$1" > "$3"; grep "$2" "$1" >> "$3" || true; }
keep "$(f $O/pmc_fetch '*counter_collection.csv')" gemv_rp $P/r03_pmc_fetch_size_gemv_rp2_m1.csv
keep "$(f $O/pmc_write '*counter_collection.csv')" gemv_rp $P/r03_pmc_write_size_gemv_rp2_m1.csv
keep "$(f $O/pmc_fetch_dq '*counter_collection.csv')" dequant_kernel $P/r03_pmc_fetch_size_dequant.csv
keep "$(f $O/pmc_write_dq '*counter_collection.csv')" dequant_kernel $P/r03_pmc_write_size_dequant.csv
keep "$(f $O/pmc_fetch_pf '*counter_collection.csv')" gemm_repacked_pipelined_kernel $P/r03_pmc_fetch_size_prefill.csv
keep "$(f $O/pmc_write_pf '*counter_collection.csv')" gemm_repacked_pipelined_kernel $P/r03_pmc_write_size_prefill.csv
sq="$(f $O/pmc_pf_sq '*counter_collection.csv')"; [ -n "$sq" ] && keep "$sq" gemm_repacked_pipelined_kernel $P/r03_pmc_prefill_sq_counters.csv
ls -la $P/r03_*
