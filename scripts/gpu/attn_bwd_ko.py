"""Knock-out timing of the streaming attention backward (PPF_ATTN_KO bit mask: 1 pair loop, 2 result stores, 4 next-item loads)."""
import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
code = r'''
import sys, os, torch
sys.path.insert(0, %r)
from protopformer_amd import ops
B, H, N, D = 256, 6, 197, 384
qkv = (torch.randn(B * N, 3 * D, device="cuda") * 0.5).bfloat16()
dout = (torch.randn(B * N, D, device="cuda") * 0.1).bfloat16()
out, rowmax, zinv = ops.attn_fwd(qkv, B, H, N, D)
def timeit(fn, iters=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
print("%%8.1f us" %% timeit(lambda: ops.attn_bwd(qkv, out, dout, rowmax, zinv, B, H, N, D)))
''' % root
for ko in sys.argv[1:] or ["0", "1", "2", "4", "3", "5", "6", "7"]:
    env = dict(os.environ, PPF_ATTN_KO=ko)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env)
    print("KO", ko, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:], flush=True)
