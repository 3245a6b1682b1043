#!/bin/bash
# kernel-time totals of HEAD vs _old for one config (rocprofv3 --kernel-trace --stats), top 14 kernels each
cfg=${1:-cait_xxs24}
for d in . _old; do
  root=$GRAFT_REPO_ROOT/$d
  cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_cmp && mkdir -p /tmp/prof_cmp
  timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_cmp -o r2 -- python3 $root/bench.py --config $cfg --steps 5 --warmup 2 --no-cpu-baseline > /tmp/prof_cmp/bench.json 2> /tmp/prof_cmp/err.txt
  cd $GRAFT_REPO_ROOT
  echo "=== $d"
  python3 scripts/rocpd_stats.py /tmp/prof_cmp/r2_results.db 2>&1 | head -16 | cut -c1-60,112-160
  python3 scripts/rocpd_stats.py /tmp/prof_cmp/r2_results.db 2>&1 | grep TOTAL
done
