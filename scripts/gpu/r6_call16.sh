#!/bin/bash
# round 6, GPU call 16: `bench.py --gpus 2` end to end on one GPU (both ranks on cuda:0 over gloo): the multi-rank control flow with the new chunk plan,
# for DeiT and CaiT, fp32 and bf16 wire format
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export PPF_BENCH_ONE_GPU=1
{ echo "== deit_tiny fp32"; timeout 900 python bench.py --gpus 2 --steps 3 --warmup 3 --batch 16 --config deit_tiny --no-cpu-baseline --no-secondary 2>&1 | grep "^{" | cut -c1-330
  echo "== cait_xxs24 fp32"; timeout 900 python bench.py --gpus 2 --steps 3 --warmup 3 --batch 8 --config cait_xxs24 --no-cpu-baseline --no-secondary 2>&1 | grep "^{" | cut -c1-330
  echo "== deit_tiny bf16 wire"; PPF_GRADSYNC_BF16=1 timeout 900 python bench.py --gpus 2 --steps 3 --warmup 3 --batch 16 --config deit_tiny --no-cpu-baseline --no-secondary 2>&1 | grep "^{" | cut -c1-330; } > gpurun_out/r6p_two_ranks.txt 2>&1
cat gpurun_out/r6p_two_ranks.txt
