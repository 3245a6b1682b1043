#!/bin/bash
for tgt in 432 512 768 432; do echo "== PPF_SPLITK_TARGET=$tgt"; PPF_SPLITK_TARGET=$tgt timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],2), round(d['roofline']['avg_launch_ms']*1e3,1))"; done
timeout 900 python -m pytest tests/test_gpu_norm_elementwise.py tests/test_gpu_e2e.py tests/test_gpu_attention.py -q 2>&1 | tail -2
