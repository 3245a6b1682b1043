#!/bin/bash
# round 6, GPU call 2: re-run of the two tests that failed + the new ones, gradient-exchange readiness timeline, reservation precision study,
# SQ counters of the attention forward before / after the packed softmax.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_baseline_configs.py tests/test_gpu_precise.py tests/test_gpu_attention.py -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/r6b_tests.log 2>&1; tail -5 gpurun_out/r6b_tests.log
timeout 900 python scripts/gpu/gradsync_timeline.py deit_small > gpurun_out/r6b_gradsync_timeline.txt 2>&1; grep -v amdgpu.ids gpurun_out/r6b_gradsync_timeline.txt | grep -v GRADSYNC_TIMELINE | tail -60
timeout 900 python scripts/gpu/reserve_precision_study.py 4 > gpurun_out/r6b_reserve_precision.txt 2>&1; grep -v amdgpu.ids gpurun_out/r6b_reserve_precision.txt | tail -70
{ echo "== packed"; bash scripts/gpu/pmc_sq.sh attn_fwd16 scripts/gpu/attn_fwd_bench.py; echo "== PPF_ATTN_FWD_PACKED=0"; PPF_ATTN_FWD_PACKED=0 bash scripts/gpu/pmc_sq.sh attn_fwd16 scripts/gpu/attn_fwd_bench.py; } > gpurun_out/r6b_attn_fwd_sq.txt 2>&1; grep -v amdgpu.ids gpurun_out/r6b_attn_fwd_sq.txt | tail -20
