#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && rm -rf gpurun_out/pmc_w1 gpurun_out/pmc_w2 gpurun_out/pmc_w3 && mkdir -p gpurun_out/pmc_w1 gpurun_out/pmc_w2 gpurun_out/pmc_w3
export PPF_GEMM_PD=${PPF_GEMM_PD:-1}
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -d gpurun_out/pmc_w1 -o s -- python3 scripts/gpu/wgrad_check.py > gpurun_out/pmc_w1/out.txt 2> gpurun_out/pmc_w1/err.txt
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM -d gpurun_out/pmc_w2 -o s -- python3 scripts/gpu/wgrad_check.py > gpurun_out/pmc_w2/out.txt 2> gpurun_out/pmc_w2/err.txt
timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE GRBM_COUNT -d gpurun_out/pmc_w3 -o s -- python3 scripts/gpu/wgrad_check.py > gpurun_out/pmc_w3/out.txt 2> gpurun_out/pmc_w3/err.txt
python3 scripts/rocpd_sq.py gpurun_out/pmc_w1/s_results.db gemm_kernel
python3 scripts/rocpd_sq.py gpurun_out/pmc_w2/s_results.db gemm_kernel
python3 scripts/rocpd_sq.py gpurun_out/pmc_w3/s_results.db gemm_kernel
tail -3 gpurun_out/pmc_w3/err.txt
