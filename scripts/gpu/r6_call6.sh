#!/bin/bash
# round 6, GPU call 6: gradient-exchange readiness timeline, one partition per process; prototype-gradient ordering A/B
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for cfg in "equal fp32" "default fp32" "1+2+4+8 fp32" "default bf16" "equal bf16"; do timeout 600 python scripts/gpu/gradsync_timeline.py $cfg 2>&1 | grep -v "amdgpu.ids\|GRADSYNC_TIMELINE\|socket.cpp\|RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl"; done > gpurun_out/r6f_gradsync_timeline.txt 2>&1
cat gpurun_out/r6f_gradsync_timeline.txt
timeout 600 python scripts/gpu/gradsync_timeline.py default fp32 cait_xxs24 2>&1 | grep "^==\|exposed\|^ " > gpurun_out/r6f_gradsync_timeline_cait.txt; cat gpurun_out/r6f_gradsync_timeline_cait.txt
python scripts/gpu/ab_step.py 3 "side_first:" "side_after:PPF_X_PROTO_SIDE_AFTER=1" > gpurun_out/r6f_ab_proto_order.txt 2>&1; cat gpurun_out/r6f_ab_proto_order.txt
