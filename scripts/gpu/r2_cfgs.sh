#!/bin/bash
for c in deit_small cait_xxs24 deit_tiny; do timeout 600 python bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$c', round(d['value']), 'img/s', round(d['ms_per_step'],2), 'ms host', round(d['host_enqueue_ms_per_step'],2), 'mfma', round(d['step_mfma_frac'],4))"; done
timeout 300 python scripts/gpu/host_profile.py deit_tiny 2>&1 | grep -v amdgpu | head -24
