#!/bin/bash
for v in 0 1 2 3; do echo "== PPF_GEMM_STAGGER=$v"; PPF_GEMM_STAGGER=$v timeout 300 python scripts/bench_gemm.py 2>&1 | grep "qkv\|fc1 \|dgelu" | head -3; done
