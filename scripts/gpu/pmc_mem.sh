#!/bin/bash
# memory-path counters of one kernel family: bash scripts/gpu/pmc_mem.sh <kernel-name filter> <python script> [args...]
FLT=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && rm -rf gpurun_out/pmc_m? && mkdir -p gpurun_out/pmc_m1 gpurun_out/pmc_m2 gpurun_out/pmc_m3 gpurun_out/pmc_m4
timeout 600 rocprofv3 --pmc TCC_REQ_sum TCC_READ_sum TCC_BUSY_sum TCC_CYCLE_sum GRBM_GUI_ACTIVE -d gpurun_out/pmc_m1 -o s -- python3 "$@" > gpurun_out/pmc_m1/out.txt 2> gpurun_out/pmc_m1/err.txt
timeout 600 rocprofv3 --pmc TCC_TAG_STALL_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum -d gpurun_out/pmc_m2 -o s -- python3 "$@" > gpurun_out/pmc_m2/out.txt 2> gpurun_out/pmc_m2/err.txt
timeout 600 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum -d gpurun_out/pmc_m3 -o s -- python3 "$@" > gpurun_out/pmc_m3/out.txt 2> gpurun_out/pmc_m3/err.txt
timeout 600 rocprofv3 --pmc TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum -d gpurun_out/pmc_m4 -o s -- python3 "$@" > gpurun_out/pmc_m4/out.txt 2> gpurun_out/pmc_m4/err.txt
for i in 1 2 3 4; do python3 scripts/rocpd_sq.py gpurun_out/pmc_m$i/s_results.db "$FLT" 2>&1 | cut -c1-330; tail -1 gpurun_out/pmc_m$i/err.txt | cut -c1-200; done
rm -rf gpurun_out/pmc_m?
