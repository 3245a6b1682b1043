#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python scripts/gpu/wgrad_check.py > gpurun_out/r6k_wgrad_check.txt 2>&1; PPF_LIB_PATH=$GRAFT_REPO_ROOT/protopformer_amd/lib/libppf_hip_wg128.so python scripts/gpu/wgrad_check.py > gpurun_out/r6k_wgrad_check_128.txt 2>&1
grep "dW\|WORST" gpurun_out/r6k_wgrad_check.txt; echo "--- 128 VGPR build"; grep "dW\|WORST" gpurun_out/r6k_wgrad_check_128.txt
python scripts/gpu/ab_step.py 3 "wgrad8_160vgpr:" "wgrad8_128vgpr:PPF_LIB_PATH=$GRAFT_REPO_ROOT/protopformer_amd/lib/libppf_hip_wg128.so" > gpurun_out/r6k_ab.txt 2>&1; cat gpurun_out/r6k_ab.txt
