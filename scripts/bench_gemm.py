"""Micro-benchmark of the GEMM family at the train step's shapes (deit_small, B=256). Prints TFLOP/s and effective GB/s."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from protopformer_amd import ops

M, D = 256 * 197, 384
dev = "cuda"


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
    return e0.elapsed_time(e1) / iters * 1e-3


def rnd(*shape):
    return (torch.randn(*shape, device=dev) * 0.5).bfloat16()


cases = []
x = rnd(M, D); w_qkv = rnd(3 * D, D); w_proj = rnd(D, D); w1 = rnd(4 * D, D); w2 = rnd(D, 4 * D)
res = torch.randn(M, D, device=dev); bias = torch.randn(4 * D, device=dev)
h = torch.empty(M, 4 * D, dtype=torch.uint8, device=dev); g = rnd(M, 4 * D)       # gelu' codes (one byte per element); dy = rnd(M, D); dqkv = rnd(M, 3 * D); dh = rnd(M, 4 * D)
out_res = torch.empty(M, D, device=dev)
cases.append(("fwd qkv   NT bf16  N=1152 K=384", lambda: ops.gemm(x, w_qkv, epi=ops.EPI_BF16, bias=bias[:3 * D]), 2 * M * 3 * D * D, (M * D + M * 3 * D) * 2))
cases.append(("fwd proj  NT resid N=384  K=384", lambda: ops.gemm(x, w_proj, epi=ops.EPI_RESID, bias=bias[:D], res=res, out=out_res), 2 * M * D * D, M * D * 2 + M * D * 8))
cases.append(("fwd fc1   NT gelu  N=1536 K=384", lambda: ops.gemm(x, w1, epi=ops.EPI_GELU, bias=bias, aux_out=h), 2 * M * 4 * D * D, M * D * 2 + 2 * M * 4 * D * 2))
cases.append(("fwd fc2   NT resid N=384  K=1536", lambda: ops.gemm(g, w2, epi=ops.EPI_RESID, bias=bias[:D], res=res, out=out_res), 2 * M * 4 * D * D, M * 4 * D * 2 + M * D * 8))
cases.append(("dgrad fc2 NN dgelu N=1536 K=384", lambda: ops.gemm(dy, w2, trans_b=True, epi=ops.EPI_DGELU, aux_in=h), 2 * M * 4 * D * D, M * D * 2 + 2 * M * 4 * D * 2))
cases.append(("dgrad fc1 NN bf16  N=384  K=1536", lambda: ops.gemm(dh, w1, trans_b=True, epi=ops.EPI_BF16), 2 * M * 4 * D * D, M * 4 * D * 2 + M * D * 2))
cases.append(("dgrad qkv NN bf16  N=384  K=1152", lambda: ops.gemm(dqkv, w_qkv, trans_b=True, epi=ops.EPI_BF16), 2 * M * 3 * D * D, M * 3 * D * 2 + M * D * 2))
cases.append(("dgrad prj NN bf16  N=384  K=384", lambda: ops.gemm(dy, w_proj, trans_b=True, epi=ops.EPI_BF16), 2 * M * D * D, 2 * M * D * 2))
gw1 = torch.zeros(4 * D, D, device=dev); gb1 = torch.zeros(4 * D, device=dev); gw2 = torch.zeros(D, 4 * D, device=dev); gwq = torch.zeros(3 * D, D, device=dev); gwp = torch.zeros(D, D, device=dev)
cases.append(("wgrad fc1 TN 1536x384  K=50432", lambda: ops.gemm(dh, x, trans_a=True, trans_b=True, epi=ops.EPI_ATOMIC, out=gw1, colsum=gb1), 2 * M * 4 * D * D, (M * 4 * D + M * D) * 2))
cases.append(("wgrad fc2 TN 384x1536  K=50432", lambda: ops.gemm(dy, g, trans_a=True, trans_b=True, epi=ops.EPI_ATOMIC, out=gw2), 2 * M * 4 * D * D, (M * 4 * D + M * D) * 2))
cases.append(("wgrad qkv TN 1152x384  K=50432", lambda: ops.gemm(dqkv, x, trans_a=True, trans_b=True, epi=ops.EPI_ATOMIC, out=gwq), 2 * M * 3 * D * D, (M * 3 * D + M * D) * 2))
cases.append(("wgrad prj TN 384x384   K=50432", lambda: ops.gemm(dy, x, trans_a=True, trans_b=True, epi=ops.EPI_ATOMIC, out=gwp), 2 * M * D * D, 2 * M * D * 2))
tot_t = 0
for name, fn, flops, bytes_ in cases:
    t = timeit(fn)
    tot_t += t
    print(f"{name:34s} {t * 1e6:8.1f} us  {flops / t / 1e12:7.1f} TFLOP/s  {bytes_ / t / 1e9:7.0f} GB/s(alg)")
print(f"sum of the 12 per-layer GEMMs: {tot_t * 1e3:.3f} ms -> x12 layers = {tot_t * 12e3:.2f} ms")
