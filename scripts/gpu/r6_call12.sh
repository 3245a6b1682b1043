#!/bin/bash
# round 6, GPU call 12: attn_fwd4_kernel (two 4-wave workgroups per CU) -- tests, stand-alone, step A/B
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_attention.py tests/test_gpu_e2e.py tests/test_gpu_baseline_configs.py -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/r6l_tests.log 2>&1; tail -5 gpurun_out/r6l_tests.log
{ echo "== fwd4"; python scripts/gpu/attn_fwd_bench.py | grep hm; python scripts/gpu/attn_fwd_bench.py 256 6 82 384 | grep hm; echo "== fwd16 (PPF_X_FWD4=0)"; PPF_X_FWD4=0 python scripts/gpu/attn_fwd_bench.py | grep hm;  PPF_X_FWD4=0 python scripts/gpu/attn_fwd_bench.py 256 6 82 384 | grep hm; echo "== deit_tiny shape B128 H3 fwd4 / fwd16"; python scripts/gpu/attn_fwd_bench.py 128 3 197 192 | grep hm; PPF_X_FWD4=0 python scripts/gpu/attn_fwd_bench.py 128 3 197 192 | grep hm; } > gpurun_out/r6l_attn_fwd4.txt 2>&1; grep -v amdgpu gpurun_out/r6l_attn_fwd4.txt
python scripts/gpu/ab_step.py 3 "fwd4:" "fwd16:PPF_X_FWD4=0" > gpurun_out/r6l_ab.txt 2>&1; cat gpurun_out/r6l_ab.txt
