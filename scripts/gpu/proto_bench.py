"""Stand-alone timing of the prototype-layer kernels at the deit_small train-step shape (B=256, 1+81 tokens, Dp=384, P=2000)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from protopformer_amd import ops

B, T, Dp, P = 256, 81, 384, 2000
g = torch.Generator().manual_seed(0)
tok = torch.rand(B, T + 1, Dp, generator=g).cuda()
pro = torch.rand(P, Dp, generator=g).cuda()


def timeit(fn, iters=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


t_full = timeit(lambda: ops.proto_fwd(tok, 1, T, pro))
t_max = timeit(lambda: ops.proto_fwd(tok, 1, T, pro, want_dist=False, want_act=False))
print(f"proto_fwd full maps {t_full:7.1f} us | max/argmax only {t_max:7.1f} us   (fp32 MFMA floor 240 us, maps 332 MB)")
