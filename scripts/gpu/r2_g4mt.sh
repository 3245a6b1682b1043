#!/bin/bash
for v in 2 4 5; do echo "== PPF_GEMM_G4_MT=$v"; PPF_GEMM_G4_MT=$v timeout 300 python scripts/bench_gemm.py 2>&1 | tail -14; done
for v in 4 5; do PPF_GEMM_G4_MT=$v timeout 600 python -m pytest tests/test_gpu_gemm.py -q -x 2>&1 | tail -2; done
timeout 900 python scripts/gpu/ab_step.py 2 "base:" "mt4:PPF_GEMM_G4_MT=4" "mt4o2:PPF_GEMM_G4_MT=5" 2>&1 | tail -4
