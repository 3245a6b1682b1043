timeout 800 python -m pytest tests -q -m gpu -x 2>&1 | grep -E "^E |passed|failed|Error|error|^tests" | cut -c1-400
timeout 300 python scripts/gpu/ln_bench.py 2>&1 | tail -1
timeout 300 python scripts/bench_gemm.py 2>&1 | tail -13 | head -5
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | cut -c90-220
