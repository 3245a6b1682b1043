#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_cait.py tests/test_gpu_train_state.py tests/test_gpu_baseline_configs.py -q 2>&1 | tail -3
timeout 1200 python scripts/gpu/ab_step.py 3 "base:" "nodefer:PPF_LANE_DEFER=0" "markall:PPF_LANE_MARK_ALL=1" "old:PPF_LANE_DEFER=0,PPF_LANE_MARK_ALL=1" 2>&1 | tail -5
