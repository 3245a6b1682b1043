"""Weight-gradient GEMM (both operands contraction-strided, split-K + ordered reduce) vs an fp32 torch reference, at the train
step's shapes, with bit-identical repeats."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from protopformer_amd import ops
torch.manual_seed(0)
dev = "cuda"
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
worst = 0
for (Mo, No, K) in [(1536, 384, 50432), (384, 1536, 50432), (1152, 384, 50432), (384, 384, 50432), (520, 392, 8192), (256, 256, 4096)]:
    dy = (torch.randn(K, Mo, device=dev) * 0.5).bfloat16(); x = (torch.randn(K, No, device=dev) * 0.5).bfloat16()
    gw = torch.zeros(Mo, No, device=dev); gb = torch.zeros(Mo, device=dev)
    ops.gemm(dy, x, trans_a=True, trans_b=True, epi=ops.EPI_ATOMIC, out=gw, colsum=gb)
    ref = dy.float().t() @ x.float(); refb = dy.float().sum(0)
    e1 = ((gw - ref).abs().max() / ref.abs().max()).item(); e2 = ((gb - refb).abs().max() / refb.abs().max()).item()
    gw2 = torch.zeros_like(gw); gb2 = torch.zeros_like(gb)
    ops.gemm(dy, x, trans_a=True, trans_b=True, epi=ops.EPI_ATOMIC, out=gw2, colsum=gb2)
    same = torch.equal(gw, gw2) and torch.equal(gb, gb2)
    t = timeit(lambda: ops.gemm(dy, x, trans_a=True, trans_b=True, epi=ops.EPI_ATOMIC, out=gw, colsum=gb))
    print(f"dW {Mo}x{No} K={K}: rel err {e1:.2e} colsum rel err {e2:.2e} repeatable {same}  {t:7.1f} us {2*Mo*No*K/t/1e6:7.1f} TFLOP/s", flush=True)
    worst = max(worst, e1, e2)
print("WORST", worst)
