import os, sys
sys.path.insert(0, "/root/repo")
import torch
from protopformer_amd import ops
B, T, Dp, P = 256, 81, 384, 2000
g = torch.Generator().manual_seed(0)
tok = torch.rand(B, T + 1, Dp, generator=g).cuda()
pro = torch.rand(P, Dp, generator=g).cuda()
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
a = ops.proto_fwd(tok, 1, T, pro)
print("xcd", os.environ.get("PPF_PROTO_XCD", "1"),
      "both maps %.1f" % timeit(lambda: ops.proto_fwd(tok, 1, T, pro)),
      "act only %.1f" % timeit(lambda: ops.proto_fwd(tok, 1, T, pro, want_dist=False, want_act=True)),
      "max only %.1f" % timeit(lambda: ops.proto_fwd(tok, 1, T, pro, want_dist=False, want_act=False)))

