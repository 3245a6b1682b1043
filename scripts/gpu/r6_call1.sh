#!/bin/bash
# round 6, GPU call 1: tests at HEAD, attention-forward packed softmax A/B (stand-alone + in the step), LayerNorm-backward non-temporal
# variants, the forced-gradient-sync bench line and its two-queue timeline.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
bash scripts/gpu/tests.sh r6a > gpurun_out/r6a_tests_tail.txt 2>&1
tail -6 gpurun_out/r6a_tests_tail.txt
{
echo "== attn_fwd_bench packed (default)"; python scripts/gpu/attn_fwd_bench.py
echo "== attn_fwd_bench PPF_ATTN_FWD_PACKED=0"; PPF_ATTN_FWD_PACKED=0 python scripts/gpu/attn_fwd_bench.py
echo "== attn_fwd_bench packed, N=82 B=256"; python scripts/gpu/attn_fwd_bench.py 256 6 82 384
echo "== attn_fwd_bench PPF_ATTN_FWD_PACKED=0, N=82"; PPF_ATTN_FWD_PACKED=0 python scripts/gpu/attn_fwd_bench.py 256 6 82 384
echo "== ln_bench NT masks"; for m in 0 1 2 4 6 7 31; do echo "PPF_LN_NT=$m"; PPF_LN_NT=$m python scripts/gpu/ln_bench.py; done
} > gpurun_out/r6a_micro.txt 2>&1
cat gpurun_out/r6a_micro.txt
python scripts/gpu/ab_step.py 3 "base:" "attn_unpacked:PPF_ATTN_FWD_PACKED=0" "ln_nt1:PPF_LN_NT=1" "ln_nt6:PPF_LN_NT=6" "ln_nt7:PPF_LN_NT=7" "gradsync:PPF_FORCE_GRADSYNC=1" > gpurun_out/r6a_ab.txt 2>&1
cat gpurun_out/r6a_ab.txt
PPF_FORCE_GRADSYNC=1 bash scripts/gpu/prof.sh r6a_gradsync > gpurun_out/r6a_gradsync_prof_tail.txt 2>&1
tail -30 gpurun_out/r6a_gradsync_prof_tail.txt
