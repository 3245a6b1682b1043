timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" 2>&1 | tail -3
timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -3
timeout 600 python bench.py 2>&1 | tail -1 > gpurun_out/bench_final.json; cut -c1-400 gpurun_out/bench_final.json
