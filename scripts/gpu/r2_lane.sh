#!/bin/bash
timeout 600 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_cait.py tests/test_gpu_train_state.py -q -x 2>&1 | tail -3
timeout 900 python scripts/gpu/ab_step.py 3 "defer2:" "nodefer:PPF_LANE_DEFER_WGRAD=0" 2>&1 | tail -3
