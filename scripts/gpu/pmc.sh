#!/bin/bash
# HBM traffic per kernel: separate --pmc passes (FETCH_SIZE, WRITE_SIZE), program directly after `--`: bash scripts/gpu/pmc.sh TAG
# -> gpurun_out/TAG_pmc_traffic.json + TAG_pmc_traffic.txt (per kernel per step table)
TAG=${1:-r3}; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write && mkdir -p gpurun_out/pmc_fetch gpurun_out/pmc_write
timeout 600 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_fetch -o f -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > gpurun_out/pmc_fetch/out.txt 2> gpurun_out/pmc_fetch/err.txt
timeout 600 rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmc_write -o w -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > gpurun_out/pmc_write/out.txt 2> gpurun_out/pmc_write/err.txt
python3 scripts/rocpd_pmc.py gpurun_out/pmc_fetch/f_results.db gpurun_out/pmc_write/w_results.db gpurun_out/${TAG}_pmc_traffic.json
python3 scripts/pmc_table.py gpurun_out/${TAG}_pmc_traffic.json 4 > gpurun_out/${TAG}_pmc_traffic.txt
head -40 gpurun_out/${TAG}_pmc_traffic.txt
rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write
