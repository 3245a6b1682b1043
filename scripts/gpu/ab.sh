timeout 900 python -m pytest tests/test_gpu_e2e.py -q -x 2>&1 | tail -2
for i in 1 2; do
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | cut -c90-200
PPF_WGRAD_SCHED=0 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | cut -c90-200
done
