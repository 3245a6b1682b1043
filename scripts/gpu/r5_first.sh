#!/bin/bash
# round-5 first GPU pass: micro-benchmark follow-ups, full GPU test suite + smoke, a short bench line with named_path
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(cd scripts/gpu/micro && timeout 200 ./l2tile scale > $GRAFT_REPO_ROOT/gpurun_out/r5_l2tile_scale.txt 2>&1; timeout 200 ./l2tile mix >> $GRAFT_REPO_ROOT/gpurun_out/r5_l2tile_scale.txt 2>&1)
bash scripts/gpu/tests.sh r5a
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > gpurun_out/r5a_bench.json 2> gpurun_out/r5a_bench.err
echo "bench rc=$?"; cut -c1-600 gpurun_out/r5a_bench.json
