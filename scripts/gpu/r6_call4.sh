#!/bin/bash
# round 6, GPU call 4: LDS-DMA issued through inline assembly in the attention kernels (tests, stand-alone, step A/B against HEAD~ library),
# gradient-exchange anomaly (6 pending fp32 works: 20.7 ms steps) with immediate waits
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_attention.py tests/test_gpu_train_state.py tests/test_gpu_e2e.py tests/test_gpu_switches.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/r6d_tests.log 2>&1; tail -4 gpurun_out/r6d_tests.log
{ echo "== attn_fwd_bench (hidden DMA)"; python scripts/gpu/attn_fwd_bench.py; echo "== attn_fwd_bench, library of the previous commit"; PPF_LIB_PATH=$GRAFT_REPO_ROOT/protopformer_amd/lib/libppf_hip_prev.so python scripts/gpu/attn_fwd_bench.py
  echo "== attn_bench (hidden DMA)"; python scripts/gpu/attn_bench.py; echo "== attn_bench, previous library"; PPF_LIB_PATH=$GRAFT_REPO_ROOT/protopformer_amd/lib/libppf_hip_prev.so python scripts/gpu/attn_bench.py; } > gpurun_out/r6d_attn_micro.txt 2>&1
grep -v amdgpu.ids gpurun_out/r6d_attn_micro.txt
python scripts/gpu/ab_step.py 3 "hidden_dma:" "prev_lib:PPF_LIB_PATH=$GRAFT_REPO_ROOT/protopformer_amd/lib/libppf_hip_prev.so" > gpurun_out/r6d_ab.txt 2>&1; cat gpurun_out/r6d_ab.txt
timeout 900 python scripts/gpu/gradsync_timeline.py deit_small > gpurun_out/r6d_gradsync_timeline_waitnow.txt 2>&1; grep "^==\|exposed" gpurun_out/r6d_gradsync_timeline_waitnow.txt
PPF_GS_PENDING=1 timeout 900 python scripts/gpu/gradsync_timeline.py deit_small > gpurun_out/r6d_gradsync_timeline_pending.txt 2>&1; grep "^==\|exposed" gpurun_out/r6d_gradsync_timeline_pending.txt
