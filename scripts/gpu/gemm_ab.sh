timeout 300 python scripts/gpu/gemm_check.py 2>&1 | tail -34
echo "--- nt256 on"; timeout 300 python scripts/bench_gemm.py 2>&1 | tail -14
echo "--- nt256 off"; PPF_GEMM_NT256=0 timeout 300 python scripts/bench_gemm.py 2>&1 | tail -14
