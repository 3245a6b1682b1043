"""Dump the recorded command list of one train step: per stream, how many kernels / event records / waits / memsets it holds.
python scripts/gpu/cmdlist.py [config]  (every non-kernel entry is a packet boundary of its queue, 4-9 us of queue time)"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
import bench
from protopformer_amd import _lib
from protopformer_amd.engine import ReplayedTrainStep

cfg = dict(bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "deit_small"])
dev = torch.device("cuda", 0)
model, opt, crit, sync = bench.build(cfg, dev, 1028)
g = torch.Generator(device=dev).manual_seed(1)
img = torch.randn(cfg["batch"], 3, 224, 224, device=dev, generator=g)
label = torch.randint(0, cfg["C"], (cfg["batch"],), device=dev, generator=g)
step = ReplayedTrainStep(model, crit, opt, epoch=20, warmup=2)
for _ in range(4):
    step(img, label)
torch.cuda.synchronize()
rec = step.rec
main = rec.main_stream
names = {main: "main"}
seq = []
for c in rec.cmds:
    if c[0] == 0:
        name, a = c[3], c[2]
        if name == "ppf_stream_wait_stream":
            seq.append((a[1], "RECORD(for other stream)")); seq.append((a[0], "WAIT(event)"))
        else:
            seq.append((a[-1], name))
    elif c[0] == 1:
        seq.append((c[1], "MARK(record)"))
    elif c[0] == 2:
        seq.append((c[2], "WAIT(mark)"))
    else:
        seq.append((main, "LIVE(python)"))
per = collections.defaultdict(collections.Counter)
for s, n in seq:
    per[names.setdefault(s, f"side{len(names)}")][n] += 1
for s, cnt in per.items():
    tot = sum(cnt.values())
    sync_n = sum(v for k, v in cnt.items() if k.startswith(("RECORD", "WAIT", "MARK")))
    print(f"== stream {s}: {tot} entries, {sync_n} ordering packets, {cnt.get('ppf_memset_zero', 0)} memsets")
    for k, v in cnt.most_common(40):
        print(f"   {v:4d}  {k}")
