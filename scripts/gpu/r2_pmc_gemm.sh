#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && rm -rf gpurun_out/pmc_g1 gpurun_out/pmc_g2 && mkdir -p gpurun_out/pmc_g1 gpurun_out/pmc_g2
timeout 300 python scripts/bench_gemm.py 2>&1 | grep -v amdgpu
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS -d gpurun_out/pmc_g1 -o s -- python3 scripts/bench_gemm.py > gpurun_out/pmc_g1/out.txt 2> gpurun_out/pmc_g1/err.txt
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum -d gpurun_out/pmc_g2 -o s -- python3 scripts/bench_gemm.py > gpurun_out/pmc_g2/out.txt 2> gpurun_out/pmc_g2/err.txt
python3 scripts/rocpd_sq.py gpurun_out/pmc_g1/s_results.db gemm
python3 scripts/rocpd_sq.py gpurun_out/pmc_g2/s_results.db gemm
rm -rf gpurun_out/pmc_g1 gpurun_out/pmc_g2
