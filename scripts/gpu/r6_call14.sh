#!/bin/bash
# round 6, GPU call 14: step time with the reference's default add-on head ('bottleneck'; advice r5: its fp32 tail had never been timed)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
{ for a in "--addon regular" "--addon bottleneck" "--addon bottleneck --proto-dim 64" "--addon regular --proto-dim 64"; do echo "== bench.py --no-cpu-baseline --no-secondary $a"; python bench.py --no-cpu-baseline --no-secondary $a | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(round(d['value']), 'img/s', round(d['ms_per_step'],3), 'ms')"; done; } > gpurun_out/r6n_bottleneck_step.txt 2>&1; cat gpurun_out/r6n_bottleneck_step.txt
