cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write && mkdir -p gpurun_out/pmc_fetch gpurun_out/pmc_write
timeout 600 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_fetch -o f -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_fetch/out.txt 2> gpurun_out/pmc_fetch/err.txt
timeout 600 rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmc_write -o w -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_write/out.txt 2> gpurun_out/pmc_write/err.txt
ls -la gpurun_out/pmc_fetch gpurun_out/pmc_write | head; tail -c 300 gpurun_out/pmc_write/out.txt
