#!/bin/bash
# round 6, GPU call 7: completion events attached to the producer launches (ppf_stream_arm) vs event-record packets (PPF_X_ARM=0)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_baseline_configs.py tests/test_gpu_train_state.py -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/r6g_tests.log 2>&1; tail -3 gpurun_out/r6g_tests.log
python scripts/gpu/ab_step.py 3 "armed:" "records:PPF_X_ARM=0" > gpurun_out/r6g_ab.txt 2>&1; cat gpurun_out/r6g_ab.txt
python scripts/gpu/ab_step.py 2 "armed:" "records:PPF_X_ARM=0" -- --config deit_tiny > gpurun_out/r6g_ab_tiny.txt 2>&1; cat gpurun_out/r6g_ab_tiny.txt
python scripts/gpu/ab_step.py 2 "armed:" "records:PPF_X_ARM=0" -- --config cait_xxs24 > gpurun_out/r6g_ab_cait.txt 2>&1; cat gpurun_out/r6g_ab_cait.txt
bash scripts/gpu/prof.sh r6g_armed > gpurun_out/r6g_prof_tail.txt 2>&1
grep -n "step wall\|union\|queue\|idle between" gpurun_out/r6g_armed_kernel_stats.txt
