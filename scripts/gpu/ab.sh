timeout 800 python -m pytest tests -q -m gpu -x 2>&1 | grep -E "^E |passed|failed|Error|error|^tests" | cut -c1-300
for i in 1 2; do
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | cut -c90-200
PPF_GEMM_G4=0 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | cut -c90-200
done
