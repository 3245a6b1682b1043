#!/bin/bash
timeout 900 python scripts/gpu/ab_step.py 3 "ordered:" "atomic:PPF_WGRAD_ATOMIC=1" 2>&1 | tail -3
