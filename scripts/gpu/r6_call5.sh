#!/bin/bash
# round 6, GPU call 5: which chunk partition makes the forced single-rank fp32 exchange slow (19.5 vs 14.7 ms)?
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python scripts/gpu/ab_step.py 1 "nosync:" "cuts_default:PPF_FORCE_GRADSYNC=1" "cuts_8_4:PPF_FORCE_GRADSYNC=1,PPF_GRADSYNC_CUTS=8+4" "cuts_8_4_2:PPF_FORCE_GRADSYNC=1,PPF_GRADSYNC_CUTS=8+4+2" "cuts_4_2_1:PPF_FORCE_GRADSYNC=1,PPF_GRADSYNC_CUTS=4+2+1" "cuts_1:PPF_FORCE_GRADSYNC=1,PPF_GRADSYNC_CUTS=1" "cuts_6:PPF_FORCE_GRADSYNC=1,PPF_GRADSYNC_CUTS=6" "cuts_default_bf16:PPF_FORCE_GRADSYNC=1,PPF_GRADSYNC_BF16=1" > gpurun_out/r6e_ab.txt 2>&1; cat gpurun_out/r6e_ab.txt
PPF_FORCE_GRADSYNC=1 bash scripts/gpu/prof.sh r6e_gradsync_cuts > gpurun_out/r6e_prof_tail.txt 2>&1
grep -v "^ *gap\|->" gpurun_out/r6e_gradsync_cuts_kernel_stats.txt | sed -n 1,25p | cut -c1-170
grep -n "step wall\|queue\|union" gpurun_out/r6e_gradsync_cuts_kernel_stats.txt
