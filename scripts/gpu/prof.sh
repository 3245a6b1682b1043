cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/prof_r1f && rm -f gpurun_out/prof_r1f/*.db
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r1f -o r1 -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/prof_r1f/bench.json 2> gpurun_out/prof_r1f/err.txt; cat gpurun_out/prof_r1f/bench.json | cut -c1-200
