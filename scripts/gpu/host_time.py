"""Host-side enqueue time of one train step from an idle GPU vs its GPU time (is the step launch-bound?)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from protopformer_amd.engine import train_one_step
dev = torch.device("cuda", 0)
model, opt, crit, sync = bench.build(dev, seed=1028)
img = torch.randn(bench.BATCH, 3, 224, 224, device=dev); label = torch.randint(0, bench.C, (bench.BATCH,), device=dev)
for _ in range(3): train_one_step(model, crit, img, label, opt, epoch=20)
for i in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    train_one_step(model, crit, img, label, opt, epoch=20)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"enqueue {1e3*(t1-t0):.2f} ms, total {1e3*(t2-t0):.2f} ms")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(3): train_one_step(model, crit, img, label, opt, epoch=20)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
