#!/bin/bash
# SQ / TCC counters of one kernel family: bash scripts/gpu/pmc_sq.sh <kernel-name filter> <python script> [args...]
# three separate --pmc passes (program directly after `--`), per-kernel averages printed by scripts/rocpd_sq.py
FLT=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && rm -rf gpurun_out/pmc_s1 gpurun_out/pmc_s2 gpurun_out/pmc_s3 && mkdir -p gpurun_out/pmc_s1 gpurun_out/pmc_s2 gpurun_out/pmc_s3
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -d gpurun_out/pmc_s1 -o s -- python3 "$@" > gpurun_out/pmc_s1/out.txt 2> gpurun_out/pmc_s1/err.txt
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM -d gpurun_out/pmc_s2 -o s -- python3 "$@" > gpurun_out/pmc_s2/out.txt 2> gpurun_out/pmc_s2/err.txt
timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE -d gpurun_out/pmc_s3 -o s -- python3 "$@" > gpurun_out/pmc_s3/out.txt 2> gpurun_out/pmc_s3/err.txt
for i in 1 2 3; do python3 scripts/rocpd_sq.py gpurun_out/pmc_s$i/s_results.db "$FLT"; done
tail -2 gpurun_out/pmc_s3/err.txt
rm -rf gpurun_out/pmc_s1 gpurun_out/pmc_s2 gpurun_out/pmc_s3
