import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from protopformer_amd import ops
dev = "cuda"
Mo, No, K = 256, 256, 4096
# 1) which kc positions of a 32-stage pair up: dy[k, m] = onehot(k%32 == a), x[k, n] = onehot(k%32 == b)
res = torch.zeros(32, 32)
for a in range(32):
    dy = torch.zeros(K, Mo, device=dev); dy[a::32] = 1
    for b in (a, (a + 4) % 32, (a + 1) % 32, (a+8)%32, (a+16)%32):
        x = torch.zeros(K, No, device=dev); x[b::32] = 1
        gw = torch.zeros(Mo, No, device=dev)
        ops.gemm(dy.bfloat16(), x.bfloat16(), trans_a=True, trans_b=True, epi=ops.EPI_ATOMIC, out=gw)
        res[a, b] = gw.mean().item()
print("diag (expect K/32 = 128):", [int(res[a, a]) for a in range(32)])
print("off-diag nonzero:", [(a, b, int(res[a, b])) for a in range(32) for b in range(32) if a != b and res[a, b] != 0][:20])
# 2) row/col mapping: dy[k, m] = m, x[k, n] = 1 -> out[m, n] = K * m ; and transposed
dy = torch.arange(Mo, device=dev).float()[None, :].expand(K, Mo).contiguous() / 64
x = torch.ones(K, No, device=dev)
gw = torch.zeros(Mo, No, device=dev); ops.gemm(dy.bfloat16(), x.bfloat16(), trans_a=True, trans_b=True, epi=ops.EPI_ATOMIC, out=gw)
ref = dy.bfloat16().float().t() @ x
print("row map err", ((gw - ref).abs().max() / ref.abs().max()).item())
x = torch.arange(No, device=dev).float()[None, :].expand(K, No).contiguous() / 64
dy = torch.ones(K, Mo, device=dev)
gw = torch.zeros(Mo, No, device=dev); ops.gemm(dy.bfloat16(), x.bfloat16(), trans_a=True, trans_b=True, epi=ops.EPI_ATOMIC, out=gw)
ref = dy.t() @ x.bfloat16().float()
print("col map err", ((gw - ref).abs().max() / ref.abs().max()).item())
dy = torch.arange(Mo, device=dev).float()[None, :].expand(K, Mo).contiguous() / 64
x = torch.ones(K, No, device=dev)
gw = torch.zeros(Mo, No, device=dev); ops.gemm(dy.bfloat16(), x.bfloat16(), trans_a=True, trans_b=True, epi=ops.EPI_ATOMIC, out=gw)
rows = (gw[:, 0] * 64 / K).round().int().tolist()
print("row values:", rows[:72])
bad = [(i, r) for i, r in enumerate(rows) if i != r]
print("bad rows:", bad[:40], len(bad))
