timeout 300 python scripts/gpu/ln_bench.py 2>&1 | tail -1
timeout 300 python scripts/gpu/ln_bench.py 2>&1 | tail -1
