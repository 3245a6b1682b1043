#!/bin/bash
# round 6, GPU call 18: ln_bwd2 with its column-sum accumulators in LDS (74 VGPRs: two waves per SIMD next to wgrad8) -- temporary switch PPF_X_LN_LACC
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
PPF_X_LN_LACC=1 timeout 900 python -m pytest tests/test_gpu_norm_elementwise.py tests/test_gpu_e2e.py tests/test_gpu_train_state.py -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/r6r_tests.log 2>&1; tail -4 gpurun_out/r6r_tests.log
{ echo "== registers"; python scripts/gpu/ln_bench.py; echo "== LDS accumulators"; PPF_X_LN_LACC=1 python scripts/gpu/ln_bench.py; } 2>&1 | grep -v amdgpu > gpurun_out/r6r_ln.txt; cat gpurun_out/r6r_ln.txt
python scripts/gpu/ab_step.py 3 "ln_regs:" "ln_lacc:PPF_X_LN_LACC=1" > gpurun_out/r6r_ab.txt 2>&1; cat gpurun_out/r6r_ab.txt
PPF_X_LN_LACC=1 bash scripts/gpu/prof.sh r6r_lacc > /dev/null 2>&1; grep "ln_bwd2\|wgrad8\|step wall\|queue" gpurun_out/r6r_lacc_kernel_stats.txt | head -8 | cut -c1-170
