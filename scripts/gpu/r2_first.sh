#!/bin/bash
# round 2, first GPU call: the whole -m gpu suite (all failures, short tracebacks), then the bench with and without the captured graph
mkdir -p gpurun_out
rm -f gpurun_out/tol_report.jsonl
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/r2_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r2_tests.log
tail -5 gpurun_out/r2_tests.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2_bench_graph.json 2> gpurun_out/r2_bench_graph.err
echo "bench graph rc=$?"; cat gpurun_out/r2_bench_graph.json | cut -c1-600
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-graph > gpurun_out/r2_bench_eager.json 2> gpurun_out/r2_bench_eager.err
echo "bench eager rc=$?"; cat gpurun_out/r2_bench_eager.json | cut -c1-600
