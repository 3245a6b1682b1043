PPF_GEMM_TT_DEEP=1 timeout 300 python scripts/gpu/wgrad_dbg.py 2>&1 | tail -1 | cut -c1-100
PPF_GEMM_TT_DEEP=1 timeout 300 python scripts/gpu/wgrad_check.py 2>&1 | tail -1
PPF_GEMM_NT256=1 timeout 300 python scripts/gpu/gemm_check.py 2>&1 | tail -2
echo "--- nt256"; PPF_GEMM_NT256=1 timeout 300 python scripts/gpu/gemm_big.py 2>&1 | tail -3 | head -2
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | cut -c90-200
