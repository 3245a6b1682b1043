"""fc1 + GELU forward (M = 256 x 197, K = 384, N = 1536; writes g and gelu', 310 MB) timed with ONE pair of output buffers reused
by every launch against a ring of 12 pairs (as in the train step, where every layer writes fresh memory): does the memory-side
cache absorb the stores of the micro-benchmark?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from protopformer_amd import ops

M, K, N = 256 * 197, 384, 1536
g = torch.Generator(device="cuda").manual_seed(1)
ring = int(sys.argv[1]) if len(sys.argv) > 1 else 12
a = [torch.randn(M, K, device="cuda", generator=g).bfloat16() for _ in range(ring)]
w = (0.05 * torch.randn(N, K, device="cuda", generator=g)).bfloat16()
bias = torch.zeros(N, device="cuda")
outs = [(torch.empty((M, N), dtype=torch.bfloat16, device="cuda"), torch.empty((M, N), dtype=torch.bfloat16, device="cuda")) for _ in range(ring)]


def run(i, rotate_out, rotate_in):
    o, h = outs[i % ring] if rotate_out else outs[0]
    ops.gemm(a[i % ring] if rotate_in else a[0], w, epi=ops.EPI_GELU, bias=bias, aux_out=h, out=o)


def t(rotate_out, rotate_in, n=48):
    for i in range(6):
        run(i, rotate_out, rotate_in)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        run(i, rotate_out, rotate_in)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for ro, ri in ((False, False), (True, False), (False, True), (True, True)):
    print(f"outputs {'ring of %d' % ring if ro else 'one pair   '}  inputs {'ring' if ri else 'one '}: {t(ro, ri):7.1f} us")
