#!/bin/bash
mkdir -p gpurun_out
rm -f gpurun_out/tol_report.jsonl
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider "$@" > gpurun_out/r2_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r2_tests.log
tail -15 gpurun_out/r2_tests.log
