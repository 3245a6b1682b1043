#!/bin/bash
# `python bench.py --gpus 2` end to end on a ONE-GPU box: both ranks on cuda:0, process group over gloo (test-only switches of bench.py).
# Exercises the launcher, the rank-0 broadcast, the chunked gradient all-reduce + loss guard inside the replayed step and every extra leg
# behind the timed region on both ranks (a leg only rank 0 ran would hang here).  ~10 s per step through gloo: 3-4 minutes.
# Round 5: passed (one JSON line, n_gpus 2, rccl_world 2, named_path present) after the path-probe leg was moved from rank 0 to all ranks.
cd $GRAFT_REPO_ROOT
export PPF_BENCH_ONE_GPU=1
timeout 900 python bench.py --gpus 2 --steps 3 --warmup 3 --batch 16 --config deit_tiny --no-cpu-baseline --no-secondary | cut -c1-400
