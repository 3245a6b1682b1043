#!/bin/bash
for k in 0 1 2 4 3 5 6 7; do PPF_GEMM_KNOCK=$k timeout 120 python scripts/gpu/knock.py 2>&1 | grep knock; done
