#!/bin/bash
# round 6, GPU call 8: full suite; th_grads with hidden LDS-DMA (CaiT step A/B against the previous cait.hip)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
bash scripts/gpu/tests.sh r6h > gpurun_out/r6h_tests_tail.txt 2>&1; tail -6 gpurun_out/r6h_tests_tail.txt
python scripts/gpu/ab_step.py 3 "hidden_dma:" "prev_cait:PPF_LIB_PATH=$GRAFT_REPO_ROOT/protopformer_amd/lib/libppf_hip_prev.so" -- --config cait_xxs24 > gpurun_out/r6h_ab_cait.txt 2>&1; cat gpurun_out/r6h_ab_cait.txt
python scripts/gpu/cait_bench.py > gpurun_out/r6h_cait_bench.txt 2>&1; grep -v amdgpu gpurun_out/r6h_cait_bench.txt | tail -12
