#!/bin/bash
# same-box comparison of HEAD with an older checkout under _old/ (git worktree), secondary configs included
for c in deit_tiny cait_xxs24 deit_small; do
  for d in . _old; do
    (cd $d && timeout 600 python bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$c', '$d', round(d['value']), 'img/s', round(d['ms_per_step'],2), 'ms host', round(d['host_enqueue_ms_per_step'],2))")
  done
done
