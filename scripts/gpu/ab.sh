echo "--- nt256 wgrad"; timeout 300 python scripts/gpu/wgrad_check.py 2>&1 | tail -7
echo "--- 128x128"; PPF_GEMM_NT256_WGRAD=0 timeout 300 python scripts/gpu/wgrad_check.py 2>&1 | tail -7
