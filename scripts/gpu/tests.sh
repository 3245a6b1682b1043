#!/bin/bash
# GPU test pass + smoke: bash scripts/gpu/tests.sh [TAG] [pytest args...]  -> gpurun_out/TAG_tests.log, gpurun_out/tol_report.jsonl
TAG=${1:-r3}; shift
mkdir -p gpurun_out; rm -f gpurun_out/tol_report.jsonl
timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider "$@" > gpurun_out/${TAG}_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/${TAG}_tests.log
tail -15 gpurun_out/${TAG}_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
