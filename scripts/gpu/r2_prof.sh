#!/bin/bash
# bench (default, eager) + rocprofv3 kernel trace of a short run; summaries via scripts/rocpd_stats.py / rocpd_timeline.py
mkdir -p gpurun_out
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" > gpurun_out/r2_bench.json 2> gpurun_out/r2_bench.err
echo "bench rc=$?"; cut -c1-420 gpurun_out/r2_bench.json
cd /tmp && export TMPDIR=/tmp && rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_r2 && mkdir -p $GRAFT_REPO_ROOT/gpurun_out/prof_r2
timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r2 -o r2 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" > $GRAFT_REPO_ROOT/gpurun_out/prof_r2/bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/prof_r2/err.txt
cd $GRAFT_REPO_ROOT
python3 scripts/rocpd_stats.py gpurun_out/prof_r2/r2_results.db > gpurun_out/r2_kernel_stats.txt 2>&1
python3 scripts/rocpd_timeline.py gpurun_out/prof_r2/r2_results.db >> gpurun_out/r2_kernel_stats.txt 2>&1
rm -rf gpurun_out/prof_r2
head -40 gpurun_out/r2_kernel_stats.txt | cut -c1-150
