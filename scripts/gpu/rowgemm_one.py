"""A few launches of one csrc/rowgemm.hip product for profiling: rowgemm_one.py <bf16|resid_ln|lnbwd> <K> [iters]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from protopformer_amd import ops

kind, K = sys.argv[1], int(sys.argv[2])
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 6
B, N, D = 256, 197, 384
M = B * N
a = (torch.randn(M, K, device="cuda") * 0.5).bfloat16()
w = (torch.randn(D, K, device="cuda") * 0.05).bfloat16()
x = torch.randn(M, D, device="cuda"); dx = torch.randn(M, D, device="cuda")
bias = torch.zeros(D, device="cuda"); lw = torch.ones(D, device="cuda"); lb = torch.zeros(D, device="cuda")
scale = torch.ones(B, device="cuda"); mean = torch.zeros(M, device="cuda"); rstd = torch.ones(M, device="cuda")
dw = torch.zeros(D, device="cuda"); db = torch.zeros(D, device="cuda"); cast = torch.empty(M, D, dtype=torch.bfloat16, device="cuda")
for _ in range(iters):
    if kind == "bf16":
        ops.rowgemm_bf16(a, w, N)
    elif kind == "resid_ln":
        ops.rowgemm_resid_ln(a, w, x, N, bias=bias, rowscale=scale, rows_per_group=N, ln_w=lw, ln_b=lb)
    else:
        ops.rowgemm_lnbwd(a, w, x, mean, rstd, lw, dw, db, N, dres_in=dx, dx_out=dx, cast_out=cast, rowscale=scale, rows_per_group=N)
torch.cuda.synchronize()
