echo "--- KW=2"; PPF_GEMM_KW=2 timeout 300 python scripts/bench_gemm.py 2>&1 | tail -14
echo "--- KW=1"; timeout 300 python scripts/bench_gemm.py 2>&1 | tail -14
