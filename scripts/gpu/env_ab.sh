#!/bin/bash
# Same-box A/B of runtime environment switches on one bench configuration: bash scripts/gpu/env_ab.sh <config> VAR=VAL [VAR=VAL ...]
cfg=$1; shift
run() { python bench.py --config $cfg --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), d['ms_per_step'])"; }
echo "base: $(run)"
for kv in "$@"; do echo "$kv: $(env $kv bash -c "$(declare -f run); cfg=$cfg; run")"; done
echo "base: $(run)"
