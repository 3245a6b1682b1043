#!/bin/bash
mkdir -p gpurun_out
rm -f gpurun_out/tol_report.jsonl
timeout 1500 python -X faulthandler -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/r2_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r2_tests.log
tail -5 gpurun_out/r2_tests.log
PPF_BENCH_GRAPH_PROBE=0 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2_bench_graph_noprobe.json 2> gpurun_out/r2_bench_graph.err
echo "bench graph (no probe nodes) rc=$?"; cut -c1-330 gpurun_out/r2_bench_graph_noprobe.json
cd /tmp && export TMPDIR=/tmp
PPF_BENCH_GRAPH_PROBE=0 timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_graph -o graph -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 4 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_graph.log 2>&1
echo "rocprof rc=$?"
ls $GRAFT_REPO_ROOT/gpurun_out/prof_graph | head
