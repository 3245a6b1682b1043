"""Stand-alone times of the attention forward kernels at one shape: the 32-row two-pass kernel + the head-mean kernel against the
one-launch 16-row kernel (with / without the map).  python scripts/gpu/attn_fwd_bench.py [B H N D]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from protopformer_amd import ops

B, H, N, D = (int(v) for v in sys.argv[1:5]) if len(sys.argv) >= 5 else (256, 6, 197, 384)
g = torch.Generator(device="cuda").manual_seed(1)
qkv = torch.randn(B * N, 3 * D, device="cuda", generator=g).bfloat16()
NP = (N + 3) // 4 * 4
hm = torch.empty((B, N, NP), device="cuda")
out, rowmax, zinv = ops.attn_fwd(qkv, B, H, N, D)


def t(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def old_fwd():
    ops._lib.call("ppf_attn_fwd", qkv, out, None, rowmax, zinv, B, H, N, D, 1, 0)


res = {"attn_fwd (32-row, two pass)": t(old_fwd),
       "attn_headmean": t(lambda: ops.attn_headmean(qkv, rowmax, zinv, B, H, N, D, out=hm)),
       "attn_fwd_hm with map": t(lambda: ops._lib.call("ppf_attn_fwd_hm", qkv, out, None, rowmax, zinv, hm, NP, B, H, N, D, 1, 0)),
       "attn_fwd_hm no map": t(lambda: ops._lib.call("ppf_attn_fwd_hm", qkv, out, None, rowmax, zinv, None, NP, B, H, N, D, 1, 0))}
for k, v in res.items():
    print(f"{k:32s} {v:8.1f} us")
