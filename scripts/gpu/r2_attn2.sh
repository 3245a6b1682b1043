#!/bin/bash
timeout 120 python scripts/gpu/attn_bench.py 2>&1 | grep "fwd \|headmean\|bwd"
timeout 600 python -m pytest tests/test_gpu_attention.py tests/test_gpu_e2e.py tests/test_gpu_baseline_configs.py -q -x 2>&1 | tail -3
timeout 400 python scripts/gpu/ab_step.py 2 "base:" 2>&1 | tail -2
