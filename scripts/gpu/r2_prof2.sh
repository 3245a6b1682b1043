#!/bin/bash
# rocprofv3 kernel trace over a longer run: the LAST step is analysed (the host is far ahead of the GPU by then)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_r2 && mkdir -p $GRAFT_REPO_ROOT/gpurun_out/prof_r2
timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r2 -o r2 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 16 --warmup 3 --no-cpu-baseline "$@" > $GRAFT_REPO_ROOT/gpurun_out/prof_r2/bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/prof_r2/err.txt
cd $GRAFT_REPO_ROOT
cut -c1-400 gpurun_out/prof_r2/bench.json
python3 scripts/rocpd_timeline.py gpurun_out/prof_r2/r2_results.db > gpurun_out/r2_timeline_last.txt 2>&1
python3 scripts/rocpd_window.py gpurun_out/prof_r2/r2_results.db > gpurun_out/r2_window.txt 2>&1
rm -rf gpurun_out/prof_r2
cat gpurun_out/r2_timeline_last.txt
