#!/bin/bash
mkdir -p gpurun_out; rm -f gpurun_out/tol_report.jsonl
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/r2_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2_tests.log; tail -3 gpurun_out/r2_tests.log
echo "== single-rank RCCL path (PPF_FORCE_GRADSYNC=1)"
PPF_FORCE_GRADSYNC=1 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>gpurun_out/r2_force.err | cut -c1-330; tail -2 gpurun_out/r2_force.err
echo "== self-launch with --gpus 1 children semantics (N=1 direct)"
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r2_bench_full.json 2> gpurun_out/r2_bench_full.err; echo "rc=$?"; python3 -c "
import json; d=json.load(open('gpurun_out/r2_bench_full.json')); print(round(d['value']), d['ms_per_step'], d['roofline']['frac'], d['cpu_baseline']['value'], d['cpu_baseline']['cpu'], d['cpu_baseline']['config1_deit_tiny_bs32']['value'])"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
