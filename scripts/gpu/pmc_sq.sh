cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && rm -rf gpurun_out/pmc_sq && mkdir -p gpurun_out/pmc_sq
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -d gpurun_out/pmc_sq -o s -- python3 scripts/gpu/attn_bench.py > gpurun_out/pmc_sq/out.txt 2> gpurun_out/pmc_sq/err.txt
tail -3 gpurun_out/pmc_sq/out.txt; ls gpurun_out/pmc_sq
