"""How much do the N = 384 GEMMs lose to the partly filled last round of workgroups?  Time per 128x128 tile at row counts that give
768 (one full round at 3 workgroups / CU), 1182 (the train step: 50 432 rows) and 1536 tiles."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from protopformer_amd import ops


def timeit(fn, iters=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


D = 384
for name, K, kw in (("dgrad fc1 (NN, K=1536)", 1536, dict(trans_b=True)), ("dgrad qkv (NN, K=1152)", 1152, dict(trans_b=True)), ("fwd proj-like (NT bf16, K=384)", 384, dict())):
    for panels in (256, 320, 384, 394, 448, 512):
        M = panels * 128
        a = (torch.randn(M, K, device="cuda") * 0.5).bfloat16()
        w = (torch.randn(K, D, device="cuda") * 0.5).bfloat16() if kw else (torch.randn(D, K, device="cuda") * 0.5).bfloat16()
        t = timeit(lambda: ops.gemm(a, w, epi=ops.EPI_BF16, **kw))
        tiles = panels * 3
        print(f"{name:32s} tiles {tiles:5d} ({tiles / 768:4.2f} rounds)  {t:7.1f} us   {t / tiles * 768:7.1f} us per 768 tiles")
