#!/bin/bash
for v in 4; do echo "== PPF_ATTN_BWD_FUSED=$v"; PPF_ATTN_BWD_FUSED=$v timeout 120 python scripts/gpu/attn_bench.py 2>&1 | grep "fwd \|headmean\|bwd"; done
timeout 600 python -m pytest tests/test_gpu_attention.py tests/test_gpu_e2e.py tests/test_gpu_baseline_configs.py tests/test_gpu_train_state.py -q -x 2>&1 | tail -3
timeout 400 python scripts/gpu/ab_step.py 2 "onepass:" "twophase:PPF_ATTN_BWD_FUSED=1" 2>&1 | tail -3
