#!/bin/bash
# bench (eager) + rocprofv3 kernel trace of a short run: bash scripts/gpu/prof.sh TAG [bench args...]
# -> gpurun_out/TAG_bench.json, gpurun_out/TAG_kernel_stats.txt (per-kernel stats + two-queue timeline)
TAG=${1:-r3}; shift
mkdir -p gpurun_out
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
echo "bench rc=$?"; cut -c1-420 gpurun_out/${TAG}_bench.json
cd /tmp && export TMPDIR=/tmp && rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG && mkdir -p $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" > $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG/bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG/err.txt
cd $GRAFT_REPO_ROOT
python3 scripts/rocpd_stats.py gpurun_out/prof_$TAG/p_results.db 60 > gpurun_out/${TAG}_kernel_stats.txt 2>&1
python3 scripts/rocpd_timeline.py gpurun_out/prof_$TAG/p_results.db >> gpurun_out/${TAG}_kernel_stats.txt 2>&1
rm -rf gpurun_out/prof_$TAG
head -45 gpurun_out/${TAG}_kernel_stats.txt | cut -c1-150
