timeout 600 python -m pytest tests/test_gpu_norm_elementwise.py -q -x 2>&1 | tail -8 | cut -c1-300
