"""Which ATen operators (and how many) does one train step still call?  Everything that launches a kernel or a copy outside the library
shows up here (torch.profiler, CPU-side operator events with the innermost Python frame)."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
import bench
from protopformer_amd.engine import train_one_step

cfg = dict(bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "deit_small"])
dev = torch.device("cuda", 0)
model, opt, crit, sync = bench.build(cfg, dev, 1028)
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
g = torch.Generator(device=dev).manual_seed(1)
img = torch.randn(B, 3, 224, 224, device=dev, generator=g); label = torch.randint(0, cfg["C"], (B,), device=dev, generator=g)
for _ in range(3):
    train_one_step(model, crit, img, label, opt, epoch=20)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    train_one_step(model, crit, img, label, opt, epoch=20)
    torch.cuda.synchronize()
cnt = collections.Counter()
where = collections.defaultdict(collections.Counter)
for ev in prof.events():
    if ev.name.startswith("aten::") and ev.name not in ("aten::empty", "aten::empty_like", "aten::empty_strided", "aten::view", "aten::reshape", "aten::as_strided",
                                                        "aten::_unsafe_view", "aten::alias", "aten::detach", "aten::select", "aten::slice", "aten::_reshape_alias",
                                                        "aten::expand", "aten::t", "aten::transpose", "aten::unsqueeze", "aten::squeeze", "aten::contiguous", "aten::to",
                                                        "aten::result_type", "aten::is_nonzero", "aten::item", "aten::_local_scalar_dense", "aten::lift_fresh", "aten::view_as"):
        cnt[ev.name] += 1
        st = [f for f in (ev.stack or []) if "protopformer_amd" in f or "bench.py" in f]
        where[ev.name][st[0] if st else "?"] += 1
for name, n in cnt.most_common(30):
    print(f"{n:5d} {name:32s} " + "; ".join(f"{k.split('/')[-1][:60]} x{v}" for k, v in where[name].most_common(4)))
