#!/bin/bash
for pd in 1 3; do for tgt in 432 512; do echo "== PPF_GEMM_PD=$pd PPF_SPLITK_TARGET=$tgt"; PPF_GEMM_PD=$pd PPF_SPLITK_TARGET=$tgt timeout 300 python scripts/gpu/wgrad_check.py 2>&1 | grep "dW\|WORST"; done; done
