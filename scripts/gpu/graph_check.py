"""Captured-graph train step vs the eager step at a BASELINE configuration's full size (run as a child process by
tests/test_gpu_baseline_configs.py so that a runtime abort inside hipGraph cannot take the test session down).
Prints one line: GRAPH_CHECK <config> eager=[...] graphed=[...] steps=<optimizer step count>."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch  # noqa: E402

from test_gpu_baseline_configs import CFG, _run_steps  # noqa: E402

name = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4
_, _, eager = _run_steps(CFG[name], n, graph=False)
torch.cuda.empty_cache()
m, opt, graphed = _run_steps(CFG[name], n, graph=True)
print("GRAPH_CHECK", json.dumps(dict(config=name, eager=eager, graphed=graphed, steps=opt.step_count)), flush=True)
