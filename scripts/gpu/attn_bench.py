"""Attention kernels at the train step's shape (deit_small, B=256): per-kernel time via rocprof-free CUDA events."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from protopformer_amd import ops
B, H, N, D = 256, 6, 197, 384
dev = "cuda"
torch.manual_seed(0)
qkv = (torch.randn(B * N, 3 * D, device=dev) * 0.5).bfloat16()
dout = (torch.randn(B * N, D, device=dev) * 0.1).bfloat16()
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
out, rowmax, zinv = ops.attn_fwd(qkv, B, H, N, D)
hm = torch.empty((B, N, 200), dtype=torch.float32, device=dev)
print(f"fwd      {timeit(lambda: ops.attn_fwd(qkv, B, H, N, D)):8.1f} us")
print(f"headmean {timeit(lambda: ops.attn_headmean(qkv, rowmax, zinv, B, H, N, D, out=hm)):8.1f} us")
print(f"bwd      {timeit(lambda: ops.attn_bwd(qkv, out, dout, rowmax, zinv, B, H, N, D)):8.1f} us (dq + dkv)")
# reference check (fp32 torch) on a few samples
q, k, v = [t.reshape(B, N, H, D // H).permute(0, 2, 1, 3).float() for t in qkv.split(D, dim=1)]
s = (q[:4] @ k[:4].transpose(-1, -2)) / (D // H) ** 0.5
pr = torch.softmax(s, -1)
ref = (pr @ v[:4]).permute(0, 2, 1, 3).reshape(4 * N, D)
err = (out[:4 * N].float() - ref).abs().max().item()
print("fwd max abs err vs fp32 softmax (4 samples):", err)
