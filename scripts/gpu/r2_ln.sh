#!/bin/bash
for g in 2048 1024 768 512; do echo "== grid cap $g"; PPF_LN_BWD_GRID=$g timeout 200 python scripts/gpu/ln_bench.py 2>&1 | tail -4; done
timeout 900 python scripts/gpu/ab_step.py 2 "g2048:" "g1024:PPF_LN_BWD_GRID=1024" "g768:PPF_LN_BWD_GRID=768" "g512:PPF_LN_BWD_GRID=512" 2>&1 | tail -5
