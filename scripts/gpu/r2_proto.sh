#!/bin/bash
timeout 900 python scripts/gpu/ab_step.py 3 "x6:" "fp32:PPF_PROTO_FP32=1" 2>&1 | tail -3
timeout 600 python -m pytest tests/test_gpu_baseline_configs.py tests/test_gpu_cait.py tests/test_gpu_train_state.py -q 2>&1 | tail -2
