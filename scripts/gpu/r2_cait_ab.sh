#!/bin/bash
timeout 1500 python scripts/gpu/ab_step.py 3 "head:" "nodefer:PPF_LN_DEFER=0" "markall:PPF_LANE_MARK_ALL=1" "both:PPF_LN_DEFER=0,PPF_LANE_MARK_ALL=1" -- --config cait_xxs24 2>&1 | tail -5
cd _old && timeout 600 python scripts/gpu/ab_step.py 3 "old:" -- --config cait_xxs24 2>&1 | tail -2
