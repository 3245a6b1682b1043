echo "--- check (nt256 forced)"; PPF_GEMM_NT256=1 timeout 300 python scripts/gpu/gemm_check.py 2>&1 | tail -4
echo "--- nt256 forced"; PPF_GEMM_NT256=1 timeout 300 python scripts/bench_gemm.py 2>&1 | tail -14 | head -8
echo "--- nt256 off"; PPF_GEMM_NT256=0 timeout 300 python scripts/bench_gemm.py 2>&1 | tail -14
