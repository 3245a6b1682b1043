cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && rm -rf gpurun_out/prof_ddp && mkdir -p gpurun_out/prof_ddp
export PPF_FORCE_GRADSYNC=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29566
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_ddp -o d -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/prof_ddp/bench.json 2> gpurun_out/prof_ddp/err.txt; cut -c1-200 gpurun_out/prof_ddp/bench.json; tail -3 gpurun_out/prof_ddp/err.txt
