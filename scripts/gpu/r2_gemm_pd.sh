#!/bin/bash
timeout 300 python scripts/bench_gemm.py 2>&1 | grep -v amdgpu | head -9
timeout 600 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_e2e.py -q 2>&1 | tail -2
