"""cProfile of the host side of the train step (which Python functions the ~400 launches per step spend their time in)."""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
import bench
from protopformer_amd.engine import train_one_step
cfg = dict(bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "deit_tiny"])
dev = torch.device("cuda", 0)
model, opt, crit, sync = bench.build(cfg, dev, 1028)
g = torch.Generator(device=dev).manual_seed(1)
img = torch.randn(cfg["batch"], 3, 224, 224, device=dev, generator=g); label = torch.randint(0, cfg["C"], (cfg["batch"],), device=dev, generator=g)
for _ in range(5): train_one_step(model, crit, img, label, opt, epoch=20)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(10): train_one_step(model, crit, img, label, opt, epoch=20)
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(28)
