for d in 0 2 4 6; do echo "--- dbg $d"; PPF_NT256_DBG=$d timeout 300 python scripts/gpu/gemm_big.py 2>&1 | tail -4; done
