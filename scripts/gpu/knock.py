"""Knock-out timing of the 4-workgroup 128x128 GEMM (qkv and fc1+GELU shapes): which of loads / MFMAs / stores bounds it."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from protopformer_amd import ops
M, D = 256 * 197, 384
x = (torch.randn(M, D, device="cuda") * 0.5).bfloat16(); wq = (torch.randn(3 * D, D, device="cuda") * 0.5).bfloat16(); w1 = (torch.randn(4 * D, D, device="cuda") * 0.5).bfloat16()
bias = torch.randn(4 * D, device="cuda"); h = torch.empty(M, 4 * D, dtype=torch.uint8, device="cuda")
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
print(f"knock={os.environ.get('PPF_GEMM_KNOCK', '0')}  qkv {timeit(lambda: ops.gemm(x, wq, epi=ops.EPI_BF16, bias=bias[:3 * D])):7.1f} us   fc1+gelu {timeit(lambda: ops.gemm(x, w1, epi=ops.EPI_GELU, bias=bias, aux_out=h)):7.1f} us")
