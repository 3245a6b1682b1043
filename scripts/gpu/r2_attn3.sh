#!/bin/bash
for v in 0 1 2 3 4 5 6 7; do echo "== PPF_ATTN_DBG=$v"; PPF_ATTN_DBG=$v timeout 300 python scripts/gpu/attn_bench.py 2>&1 | grep "bwd"; done
