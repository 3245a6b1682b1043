#!/bin/bash
# round 6, GPU call 3: full suite after the switchboard prune, the retired kernels' own tests, gradient-exchange timeline (repeat), kernel
# tables + PMC traffic of the two secondary configurations, ln_bwd2 without the side stream (contention knock-out)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
bash scripts/gpu/tests.sh r6c > gpurun_out/r6c_tests_tail.txt 2>&1; tail -8 gpurun_out/r6c_tests_tail.txt
{ bash scripts/gpu/experiments/mlpfwd/build.sh && bash scripts/gpu/experiments/gemm_nt256/build.sh && timeout 900 python -m pytest scripts/gpu/experiments -q -p no:cacheprovider -m "gpu or not gpu" 2>&1 | tail -5; } > gpurun_out/r6c_experiments.txt 2>&1; tail -4 gpurun_out/r6c_experiments.txt
timeout 900 python scripts/gpu/gradsync_timeline.py deit_small > gpurun_out/r6c_gradsync_timeline.txt 2>&1; grep "^==\|exposed" gpurun_out/r6c_gradsync_timeline.txt
bash scripts/gpu/prof.sh r6c_deit_tiny --config deit_tiny > gpurun_out/r6c_prof_tiny_tail.txt 2>&1; head -3 gpurun_out/r6c_prof_tiny_tail.txt | cut -c1-300
bash scripts/gpu/prof.sh r6c_cait --config cait_xxs24 > gpurun_out/r6c_prof_cait_tail.txt 2>&1; head -3 gpurun_out/r6c_prof_cait_tail.txt | cut -c1-300
bash scripts/gpu/pmc.sh r6c_deit_tiny --config deit_tiny > /dev/null 2>&1
bash scripts/gpu/pmc.sh r6c_cait --config cait_xxs24 > /dev/null 2>&1
PPF_WGRAD_STREAM=0 bash scripts/gpu/prof.sh r6c_onestream > gpurun_out/r6c_prof_onestream_tail.txt 2>&1; head -3 gpurun_out/r6c_prof_onestream_tail.txt | cut -c1-300
grep "ln_bwd2\|wgrad8\|splitk" gpurun_out/r6c_onestream_kernel_stats.txt | cut -c1-170
ls gpurun_out | grep r6c
