#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_norm_elementwise.py tests/test_gpu_e2e.py tests/test_gpu_cait.py -q 2>&1 | tail -2
timeout 900 python scripts/gpu/ab_step.py 2 "base:"
