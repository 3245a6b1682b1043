"""Input-gradient GEMMs of a deit_small layer (dy [50432, K] x W [K, 384]): the transposed-read 128 x 128 kernel vs the 224 x 128 NT kernel."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from protopformer_amd import ops
M = 50432
def timeit(fn, iters=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for K in (1536, 1152, 384):
    dy = (torch.randn(M, K, device="cuda") * 0.1).bfloat16()
    w = (torch.randn(K, 384, device="cuda") * 0.05).bfloat16()          # [out][in] weight of a Linear(in = 384, out = K)
    wt = w.t().contiguous()                                               # [in][out]: the transposed shadow
    t_tb = timeit(lambda: ops.gemm(dy, w, trans_b=True, epi=ops.EPI_BF16))
    t_nt = timeit(lambda: ops.gemm(dy, wt, epi=ops.EPI_BF16))
    ref = ops.gemm(dy, w, trans_b=True, epi=ops.EPI_BF16).float(); got = ops.gemm(dy, wt, epi=ops.EPI_BF16).float()
    print(f"K={K:5d}  transposed-read 128x128: {t_tb:7.1f} us   NT path: {t_nt:7.1f} us   max |diff| {float((ref - got).abs().max()):.3e}  ({2 * M * 384 * K / t_nt / 1e6:.0f} TF/s)")
