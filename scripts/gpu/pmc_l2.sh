#!/bin/bash
# L2 (TCC) load per kernel over a bench run: bash scripts/gpu/pmc_l2.sh TAG [env K=V ...]  -> gpurun_out/TAG_l2.txt
# one --pmc pass (TCC_REQ_sum TCC_BUSY_sum TCC_CYCLE_sum TCC_MISS_sum), program directly after `--`
TAG=$1; shift
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && rm -rf gpurun_out/pmc_l2 && mkdir -p gpurun_out/pmc_l2
timeout 300 rocprofv3 --pmc TCC_REQ_sum TCC_BUSY_sum TCC_CYCLE_sum TCC_MISS_sum -d gpurun_out/pmc_l2 -o s -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_l2/out.txt 2> gpurun_out/pmc_l2/err.txt
python3 scripts/pmc_l2_table.py gpurun_out/pmc_l2/s_results.db 4 > gpurun_out/${TAG}_l2.txt; head -45 gpurun_out/${TAG}_l2.txt
rm -rf gpurun_out/pmc_l2
