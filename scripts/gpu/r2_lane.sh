#!/bin/bash
timeout 900 python scripts/gpu/ab_step.py 3 "one:" "two:PPF_LANES=2" "three:PPF_LANES=3" 2>&1 | tail -4
PPF_LANES=2 timeout 600 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_train_state.py -q -x 2>&1 | tail -3
