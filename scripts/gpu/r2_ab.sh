#!/bin/bash
timeout 300 python scripts/bench_gemm.py 2>&1 | grep "fc1 \|dgelu"
timeout 900 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_e2e.py tests/test_gpu_cait.py tests/test_gpu_baseline_configs.py -q 2>&1 | tail -3
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | cut -c1-200
