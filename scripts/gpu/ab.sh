timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -2
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | cut -c90-200
