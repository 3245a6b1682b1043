timeout 600 python -m pytest tests/test_gpu_gemm.py -q -x 2>&1 | tail -2
timeout 300 python scripts/gpu/gemm_check.py 2>&1 | grep -E "N=384|WORST|repeat" | head -12
echo "--- rows on"; timeout 300 python scripts/bench_gemm.py 2>&1 | tail -14 | head -8
echo "--- rows off"; PPF_GEMM_ROWS=0 timeout 300 python scripts/bench_gemm.py 2>&1 | tail -14 | head -8
