timeout 600 python bench.py --steps 10 --warmup 3 2>&1 | tail -1
