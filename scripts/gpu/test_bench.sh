timeout 800 python -m pytest tests -q -m gpu -x 2>&1 | grep -E "^E |passed|failed|Error|error|^tests" | cut -c1-400
PPF_GEMM_NT256=1 timeout 300 python scripts/gpu/gemm_check.py 2>&1 | tail -2
timeout 300 python scripts/gpu/gemm_big.py 2>&1 | tail -6 | head -3
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | cut -c90-220
