#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && rm -rf gpurun_out/pmc_sq gpurun_out/pmc_sq2 && mkdir -p gpurun_out/pmc_sq gpurun_out/pmc_sq2
export PPF_ATTN_BWD_FUSED=${PPF_ATTN_BWD_FUSED:-1}
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -d gpurun_out/pmc_sq -o s -- python3 scripts/gpu/attn_bench.py > gpurun_out/pmc_sq/out.txt 2> gpurun_out/pmc_sq/err.txt
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU -d gpurun_out/pmc_sq2 -o s -- python3 scripts/gpu/attn_bench.py > gpurun_out/pmc_sq2/out.txt 2> gpurun_out/pmc_sq2/err.txt
python3 scripts/rocpd_sq.py gpurun_out/pmc_sq/s_results.db attn
python3 scripts/rocpd_sq.py gpurun_out/pmc_sq2/s_results.db attn
tail -2 gpurun_out/pmc_sq2/err.txt
