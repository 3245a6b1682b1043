#!/bin/bash
# round 6, GPU call 13: side stream restricted to a subset of the CUs (hipExtStreamCreateWithCUMask) -- temporary experiment
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python scripts/gpu/ab_step.py 2 "all_cus:" "p77777777:PPF_X_SIDE_CUMASK=p77777777" "p55555555:PPF_X_SIDE_CUMASK=p55555555" "n192:PPF_X_SIDE_CUMASK=n192" "n128:PPF_X_SIDE_CUMASK=n128" "pf0f0f0f0:PPF_X_SIDE_CUMASK=pf0f0f0f0" "p7f7f7f7f:PPF_X_SIDE_CUMASK=p7f7f7f7f" > gpurun_out/r6m_ab_cumask.txt 2>&1; cat gpurun_out/r6m_ab_cumask.txt
