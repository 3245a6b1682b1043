import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from protopformer_amd import ops
dev = "cuda"
def timeit(fn, iters=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3
for (M, N, K) in [(8192, 8192, 8192), (4096, 4096, 4096), (50432, 1536, 1536), (50432, 1536, 384), (50432, 1536, 768), (50432, 1536, 3072)]:
    x = (torch.rand(M, K, device=dev) * 2 - 1).bfloat16(); w = (torch.rand(N, K, device=dev) * 2 - 1).bfloat16()
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    t = timeit(lambda: ops.gemm(x, w, epi=ops.EPI_BF16, out=out))
    print(f"M={M} N={N} K={K}: {t*1e6:9.1f} us {2*M*N*K/t/1e12:8.1f} TF", flush=True)
