"""Same-box A/B of environment-variable variants of the train step: runs `bench.py --no-cpu-baseline` for each variant in turn,
ROUNDS times, and prints img/s per run plus the per-variant mean (box-to-box spread of the pool is +-3 %, so only numbers from one
gpurun call are comparable).  usage: ab_step.py ROUNDS "NAME:K=V,K=V" "NAME2:" ... [-- extra bench args]"""
import json
import os
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
args = sys.argv[1:]
extra = []
if "--" in args:
    i = args.index("--"); extra = args[i + 1:]; args = args[:i]
rounds = int(args[0])
variants = []
for spec in args[1:]:
    name, _, kv = spec.partition(":")
    variants.append((name, dict(x.split("=", 1) for x in kv.split(",") if x)))
res = {n: [] for n, _ in variants}
for r in range(rounds):
    for name, env in variants:
        e = dict(os.environ, **env)
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "20", "--warmup", "5", "--no-cpu-baseline"] + extra,
                             capture_output=True, text=True, env=e, cwd=root, timeout=900)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if not line:
            print(name, "FAILED", out.stderr[-500:]); continue
        d = json.loads(line[0])
        res[name].append(d["value"])
        print(f"round {r} {name:24s} {d['value']:9.0f} img/s  {d['ms_per_step']:7.3f} ms  host {d['host_enqueue_ms_per_step']:6.2f} ms  wgrad {1e3 * d['roofline']['avg_launch_ms']:6.1f} us", flush=True)
for name, v in res.items():
    if v:
        print(f"MEAN {name:24s} {sum(v) / len(v):9.0f} img/s over {len(v)} runs (min {min(v):.0f} max {max(v):.0f})")
