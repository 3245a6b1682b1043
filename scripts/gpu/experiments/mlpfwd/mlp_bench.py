"""Stand-alone A/B of csrc/mlpfwd.hip (fused MLP forward) against the two launches it replaces (fc1 + GELU 128x128 GEMM, fc2 full-row GEMM +
residual + LayerNorm) at the train step's shapes.  Interleaved rounds in one process (median), random operands."""
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(_HERE)))))
sys.path.insert(0, _HERE)
import torch

from protopformer_amd import ops
import mlp_fwd as _X                                   # the retired kernel's own wrapper (build.sh first)
ops.mlp_fwd, ops.mlp_fwd_supported = _X.mlp_fwd, _X.mlp_fwd_supported

B, N, D = int(os.environ.get("RB_B", 256)), int(os.environ.get("RB_N", 197)), int(os.environ.get("RB_D", 384))
M, hid = B * N, 4 * D
dev = "cuda"


def rnd(*shape, s=0.5):
    return (torch.randn(*shape, device=dev) * s).bfloat16()


def time_all(cases, rounds=7, iters=10):
    res = {k: [] for k in cases}
    for fn in cases.values():
        for _ in range(2):
            fn()
    torch.cuda.synchronize()
    for _ in range(rounds):
        for k, fn in cases.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                fn()
            e1.record()
            torch.cuda.synchronize()
            res[k].append(e0.elapsed_time(e1) / iters * 1e3)
    return {k: sorted(v)[len(v) // 2] for k, v in res.items()}


x1 = torch.randn(M, D, device=dev)
n2 = rnd(M, D)
w1, w2 = rnd(hid, D, s=0.05), rnd(D, hid, s=0.05)
b1, b2 = torch.randn(hid, device=dev) * 0.1, torch.randn(D, device=dev) * 0.1
lw, lb = torch.ones(D, device=dev), torch.zeros(D, device=dev)
scale = torch.ones(B, device=dev)
rpt_full = ops.rowgemm_tile_rows(M, N)
dg = torch.empty((M, hid), dtype=torch.uint8, device=dev)


def unfused():
    g = ops.gemm(n2, w1, epi=ops.EPI_GELU, bias=b1, aux_out=dg)
    return ops.rowgemm_resid_ln(g, w2, x1, rpt_full, bias=b2, rowscale=scale, rows_per_group=N, ln_w=lw, ln_b=lb)


cases = {"fc1+GELU, fc2 rowgemm+LN (2 launches)": unfused}
for tile in [int(t) for t in os.environ.get("RB_TILES", "99,112,66").split(",")]:
    if ops.mlp_fwd_supported(D, hid, tile):
        cases[f"fused mlp_fwd, {tile}-row tiles ({(M + tile - 1) // tile} workgroups)"] = (
            lambda tile=tile: ops.mlp_fwd(n2, w1, b1, w2, b2, x1, tile, rowscale=scale, rows_per_group=N, ln_w=lw, ln_b=lb))
cases["fc1+GELU alone"] = lambda: ops.gemm(n2, w1, epi=ops.EPI_GELU, bias=b1, aux_out=dg)
t = time_all(cases)
fl = 4.0 * M * D * hid
for k, v in t.items():
    print(f"{k:64s} {v:8.1f} us   {fl / v / 1e6:7.1f} TFLOP/s (of the pair's 2 x 2 M D hid)")
