"""Stand-alone timing of the rollout chain kernel (deit_small: 11 layers of [256, 197, NP] head-mean maps, thresholds precomputed)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from protopformer_amd import ops

L, B, N = 11, 256, 197
NP = 200
g = torch.Generator().manual_seed(0)
hm = torch.softmax(torch.randn(L, B, N, NP, generator=g) * 2, dim=-1).cuda()
thr = torch.empty(L, B, dtype=torch.int32, device="cuda")
for l in range(L):
    ops.rollout_threshold(hm[l], thr[l], N)


def timeit(fn, iters=20):
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


t = timeit(lambda: ops.rollout(hm, L, B, N, 81, thr=thr))
c, i, pol = ops.rollout(hm, L, B, N, 81, thr=thr)
print(f"rollout chain {t:7.1f} us  ({L * B * N * NP * 4 / 1e6:.0f} MB)   checksum {float(c.double().sum()):.9f} idx {int(i.long().sum())}")
