#!/bin/bash
timeout 300 python scripts/gpu/attn_bench.py 2>&1 | grep "fwd \|headmean\|bwd"
timeout 900 python -m pytest tests/test_gpu_attention.py tests/test_gpu_e2e.py tests/test_gpu_cait.py -q 2>&1 | tail -3
timeout 900 python scripts/gpu/ab_step.py 2 "base:" "twok:PPF_ATTN_BWD_FUSED=0" 2>&1 | tail -7
