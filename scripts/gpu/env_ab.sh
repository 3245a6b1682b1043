#!/bin/bash
# Same-box A/B of environment switches on one bench configuration: bash scripts/gpu/env_ab.sh <config> VAR=VAL [VAR=VAL ...]
# (every setting is run twice, interleaved; prints images/s and ms per step)
cfg=$1; shift
run() { python bench.py --config $cfg --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'], 3))"; }
for round in 1 2; do
  echo "base: $(run)"
  for kv in "$@"; do echo "$kv: $(export $kv; run)"; done
done
