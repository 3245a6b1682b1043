#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
{ for v in "" _ahead0 _ahead4; do echo "== lib$v"; PPF_LIB_PATH=$GRAFT_REPO_ROOT/protopformer_amd/lib/libppf_hip$v.so python scripts/gpu/attn_fwd_bench.py | grep "hm\|two pass"; done; for v in "" _ahead0 _ahead4; do echo "== lib$v (repeat)"; PPF_LIB_PATH=$GRAFT_REPO_ROOT/protopformer_amd/lib/libppf_hip$v.so python scripts/gpu/attn_fwd_bench.py | grep "hm"; done; } > gpurun_out/r6j_attn_fwd_ahead.txt 2>&1
grep -v amdgpu gpurun_out/r6j_attn_fwd_ahead.txt
timeout 600 python -m pytest tests/test_gpu_attention.py -q -p no:cacheprovider 2>&1 | tail -2
python scripts/gpu/ab_step.py 3 "ahead2:" "ahead0:PPF_LIB_PATH=$GRAFT_REPO_ROOT/protopformer_amd/lib/libppf_hip_ahead0.so" > gpurun_out/r6j_ab.txt 2>&1; cat gpurun_out/r6j_ab.txt
