#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write && mkdir -p gpurun_out/pmc_fetch gpurun_out/pmc_write
timeout 600 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_fetch -o f -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_fetch/out.txt 2> gpurun_out/pmc_fetch/err.txt
timeout 600 rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmc_write -o w -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_write/out.txt 2> gpurun_out/pmc_write/err.txt
python3 scripts/rocpd_pmc.py gpurun_out/pmc_fetch/f_results.db gpurun_out/pmc_write/w_results.db gpurun_out/r2_pmc_traffic.json
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r2_pmc_traffic.json"))["kernels"]
tot = sum((v["hbm_bytes_per_launch"] or 0) * v["launches"] for v in d.values())
print("total HBM bytes over the profiled run (4 steps incl. warm-up): %.2f GB -> %.2f GB/step" % (tot / 1e9, tot / 1e9 / 4))
PY
rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write
