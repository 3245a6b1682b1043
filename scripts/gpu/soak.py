"""Soak run: 300 train steps at the headline configuration; checks that the loss stays finite, that step time and allocated memory do not
drift (the side-stream lane holds references until each join), and prints a few loss values."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
import bench
from protopformer_amd.engine import train_one_step

cfg = dict(bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "deit_small"])
dev = torch.device("cuda", 0)
model, opt, crit, sync = bench.build(cfg, dev, 1028)
g = torch.Generator(device=dev).manual_seed(1)
img = torch.randn(cfg["batch"], 3, 224, 224, device=dev, generator=g)
label = torch.randint(0, cfg["C"], (cfg["batch"],), device=dev, generator=g)
losses, mem, t_blocks = [], [], []
for blk in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50):
        loss, cov, mean = train_one_step(model, crit, img, label, opt, epoch=20)
    torch.cuda.synchronize(); t_blocks.append((time.perf_counter() - t0) / 50 * 1e3)
    losses.append((float(loss), float(cov), float(mean)))
    mem.append(torch.cuda.memory_allocated() / 2**20)
print("ms/step per 50-step block:", [round(t, 2) for t in t_blocks])
print("allocated MiB after each block:", [round(m) for m in mem], " peak", round(torch.cuda.max_memory_allocated() / 2**20))
print("loss, cov, mean:", [tuple(round(v, 4) for v in l) for l in losses])
assert all(all(v == v and abs(v) < 1e6 for v in l) for l in losses), "non-finite loss"
assert max(mem) - min(mem) < 64, "allocated memory drifts"
assert losses[-1][0] < losses[0][0], "the loss on a fixed batch should fall"
print("soak OK")
