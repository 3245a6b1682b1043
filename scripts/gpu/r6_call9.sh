#!/bin/bash
# round 6, GPU call 9: knock-out table of attn_fwd16_kernel<64,13,false,true> (measurement builds, WRONG results by construction: /tmp patches, not in the tree)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
{ echo "== full kernel"; python scripts/gpu/attn_fwd_bench.py | grep hm; for v in nodma noexp noqk nopv nomean nobarrier; do echo "== knock-out: $v"; PPF_LIB_PATH=$GRAFT_REPO_ROOT/protopformer_amd/lib/libppf_hip_ko_$v.so python scripts/gpu/attn_fwd_bench.py | grep hm; done; } > gpurun_out/r6i_attn_fwd_knockout.txt 2>&1
grep -v amdgpu gpurun_out/r6i_attn_fwd_knockout.txt
timeout 600 python -m pytest tests/test_gpu_attention.py -q -p no:cacheprovider 2>&1 | tail -2
