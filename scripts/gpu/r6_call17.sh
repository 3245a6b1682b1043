#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
{ for v in "" _lnlb5 _lnlb6; do echo "== lib$v"; PPF_LIB_PATH=$GRAFT_REPO_ROOT/protopformer_amd/lib/libppf_hip$v.so python scripts/gpu/ln_bench.py; done; } 2>&1 | grep -v amdgpu > gpurun_out/r6q_ln_regs.txt; cat gpurun_out/r6q_ln_regs.txt
python scripts/gpu/ab_step.py 3 "ln_112vgpr:" "ln_96vgpr:PPF_LIB_PATH=$GRAFT_REPO_ROOT/protopformer_amd/lib/libppf_hip_lnlb5.so" "ln_80vgpr:PPF_LIB_PATH=$GRAFT_REPO_ROOT/protopformer_amd/lib/libppf_hip_lnlb6.so" > gpurun_out/r6q_ab.txt 2>&1; cat gpurun_out/r6q_ab.txt
