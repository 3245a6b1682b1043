#!/bin/bash
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x 2>&1 | tail -4
timeout 1500 python scripts/gpu/ab_step.py 2 "base:" "nolane:PPF_WGRAD_STREAM=0"
