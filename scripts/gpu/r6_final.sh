#!/bin/bash
# round 6, final artifacts at HEAD: full suite + smoke, the driver's bench command (default line with secondary configs + CPU baseline), the forced-sync
# line, kernel tables + two-queue timelines and PMC traffic of the three BASELINE configurations.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
bash scripts/gpu/tests.sh r6_final > gpurun_out/r6_final_tests_tail.txt 2>&1; tail -5 gpurun_out/r6_final_tests_tail.txt
cp gpurun_out/tol_report.jsonl gpurun_out/r6_final_tol_report.jsonl
timeout 1500 python bench.py > gpurun_out/r6_final_bench_default.json 2> gpurun_out/r6_final_bench_default.err; echo "bench rc=$?"; cut -c1-300 gpurun_out/r6_final_bench_default.json
PPF_FORCE_GRADSYNC=1 timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r6_final_bench_gradsync.json 2>/dev/null; cut -c1-200 gpurun_out/r6_final_bench_gradsync.json
PPF_FORCE_GRADSYNC=1 PPF_GRADSYNC_BF16=1 timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r6_final_bench_gradsync_bf16.json 2>/dev/null; cut -c1-200 gpurun_out/r6_final_bench_gradsync_bf16.json
bash scripts/gpu/prof.sh r6_final > /dev/null 2>&1
bash scripts/gpu/prof.sh r6_final_deit_tiny --config deit_tiny > /dev/null 2>&1
bash scripts/gpu/prof.sh r6_final_cait --config cait_xxs24 > /dev/null 2>&1
bash scripts/gpu/pmc.sh r6_final > /dev/null 2>&1
bash scripts/gpu/pmc.sh r6_final_deit_tiny --config deit_tiny > /dev/null 2>&1
bash scripts/gpu/pmc.sh r6_final_cait --config cait_xxs24 > /dev/null 2>&1
head -12 gpurun_out/r6_final_kernel_stats.txt | cut -c1-160; head -4 gpurun_out/r6_final_pmc_traffic.txt
ls gpurun_out | grep r6_final
