timeout 300 python scripts/gpu/attn_bench.py 2>&1 | tail -5
timeout 600 python -m pytest tests/test_gpu_attention.py -q -x 2>&1 | tail -3
