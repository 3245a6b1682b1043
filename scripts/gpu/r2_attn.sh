#!/bin/bash
for v in 0 1; do echo "== PPF_ATTN_BWD_FUSED=$v"; PPF_ATTN_BWD_FUSED=$v timeout 300 python scripts/gpu/attn_bench.py 2>&1 | grep "fwd \|headmean\|bwd"; done
for v in 0 1; do PPF_ATTN_BWD_FUSED=$v timeout 600 python -m pytest tests/test_gpu_attention.py -q 2>&1 | tail -1; done
