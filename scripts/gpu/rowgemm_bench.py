"""Stand-alone A/B of csrc/rowgemm.hip against the kernels it replaces, at the train step's shapes (deit_small, batch 256, M = 50 432):
old = 128x128 GEMM (+ epilogue) followed by the separate LayerNorm forward / backward kernel; new = one full-row GEMM launch with
the LayerNorm fused.  Interleaved rounds in one process (median of the rounds), random operands."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from protopformer_amd import ops

B, N, D = int(os.environ.get("RB_B", 256)), 197, int(os.environ.get("RB_D", 384))
M = B * N
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


x = torch.randn(M, D, device=dev)
ao, g4, dh, dqkv, dyb = rnd(M, D), rnd(M, 4 * D), rnd(M, 4 * D), rnd(M, 3 * D), rnd(M, D)
wp, w2, w1, wq = rnd(D, D, s=0.05), rnd(D, 4 * D, s=0.05), rnd(4 * D, D, s=0.05), rnd(3 * D, D, s=0.05)
w1t, wqt, wpt = w1.t().contiguous(), wq.t().contiguous(), wp.t().contiguous()
bias = torch.randn(D, device=dev) * 0.1
lw, lb = torch.ones(D, device=dev), torch.zeros(D, device=dev)
scale = torch.ones(B, device=dev)
mean, rstd = torch.zeros(M, device=dev), torch.ones(M, device=dev)
dx = torch.randn(M, D, device=dev)
dw, db = torch.zeros(D, device=dev), torch.zeros(D, device=dev)
cast = torch.empty(M, D, dtype=torch.bfloat16, device=dev)
out_res = torch.empty(M, D, device=dev)


def old_resid_ln(a, w):
    x1 = ops.gemm(a, w, epi=ops.EPI_RESID, bias=bias, res=x, rowscale=scale, rows_per_group=N, out=out_res)
    return ops.layernorm_fwd(x1, lw, lb)


def old_lnbwd(a, w):
    dn = ops.gemm(a, w, trans_b=True, epi=ops.EPI_BF16)
    ops.layernorm_bwd(dn, x, lw, mean, rstd, dw, db, dres_in=dx, dx_out=dx, cast_out=cast, rowscale=scale, rows_per_group=N, dbias_next=db)


cases = {
    "proj fwd + LN2   K=384   old (gemm resid, ln_fwd)": lambda: old_resid_ln(ao, wp),
    "proj fwd + LN2   K=384   new (rowgemm_resid_ln)": lambda: ops.rowgemm_resid_ln(ao, wp, x, N, bias=bias, rowscale=scale, rows_per_group=N, ln_w=lw, ln_b=lb),
    "fc2 fwd + LN1    K=1536  old": lambda: old_resid_ln(g4, w2),
    "fc2 fwd + LN1    K=1536  new": lambda: ops.rowgemm_resid_ln(g4, w2, x, N, bias=bias, rowscale=scale, rows_per_group=N, ln_w=lw, ln_b=lb),
    "fc2 fwd no LN    K=1536  old (gemm resid only)": lambda: ops.gemm(g4, w2, epi=ops.EPI_RESID, bias=bias, res=x, rowscale=scale, rows_per_group=N, out=out_res),
    "fc2 fwd no LN    K=1536  new": lambda: ops.rowgemm_resid_ln(g4, w2, x, N, bias=bias, rowscale=scale, rows_per_group=N),
    "dgrad fc1 + LN2' K=1536  old (gemm NN, ln_bwd, colsum reduce)": lambda: old_lnbwd(dh, w1),
    "dgrad fc1 + LN2' K=1536  new (rowgemm_lnbwd + colsum)": lambda: ops.rowgemm_lnbwd(dh, w1t, x, mean, rstd, lw, dw, db, N, dres_in=dx, dx_out=dx, cast_out=cast, rowscale=scale, rows_per_group=N),
    "dgrad qkv + LN1' K=1152  old": lambda: old_lnbwd(dqkv, wq),
    "dgrad qkv + LN1' K=1152  new": lambda: ops.rowgemm_lnbwd(dqkv, wqt, x, mean, rstd, lw, dw, db, N, dres_in=dx, dx_out=dx, cast_out=cast, rowscale=scale, rows_per_group=N),
    "dgrad proj       K=384   old (gemm NN bf16)": lambda: ops.gemm(dyb, wp, trans_b=True, epi=ops.EPI_BF16),
    "dgrad proj       K=384   new (rowgemm_bf16)": lambda: ops.rowgemm_bf16(dyb, wpt, N),
}
if os.environ.get("RB_SWEEP"):
    # T(K) = T0 + (K / 32) * t_stage: separates the main loop (per-stage time) from the fixed cost (prologue + epilogue) per epilogue kind
    big = rnd(M, 3072)
    wbig = rnd(D, 3072, s=0.05)
    sweep = {}
    for K in (64, 384, 768, 1536, 3072):
        a, w = big[:, :K].contiguous(), wbig[:, :K].contiguous()
        sweep[f"bf16 K={K}"] = (lambda a=a, w=w: ops.rowgemm_bf16(a, w, N))
        sweep[f"resid_ln K={K}"] = (lambda a=a, w=w: ops.rowgemm_resid_ln(a, w, x, N, bias=bias, rowscale=scale, rows_per_group=N, ln_w=lw, ln_b=lb))
        sweep[f"lnbwd K={K}"] = (lambda a=a, w=w: ops.rowgemm_lnbwd(a, w, x, mean, rstd, lw, dw, db, N, dres_in=dx, dx_out=dx, cast_out=cast, rowscale=scale, rows_per_group=N))
    ts = time_all(sweep, rounds=5, iters=8)
    for kind in ("bf16", "resid_ln", "lnbwd"):
        t64, t3072 = ts[f"{kind} K=64"], ts[f"{kind} K=3072"]
        per = (t3072 - t64) / ((3072 - 64) / 32)
        print(f"{kind:9s} " + "  ".join(f"K={K}: {ts[f'{kind} K={K}']:6.1f}" for K in (64, 384, 768, 1536, 3072)) + f"  -> {per:.3f} us/stage, fixed {t64 - 2 * per:.1f} us")
    raise SystemExit(0)
if D not in (192, 384):
    raise SystemExit("RB_D must be 192 or 384")
t = time_all(cases)
tot = {"old": 0.0, "new": 0.0}
for k, v in t.items():
    print(f"{k:64s} {v:8.1f} us")
    if "no LN" not in k:
        tot["new" if " new" in k else "old"] += v
print(f"sum (5 products + their LayerNorm passes): old {tot['old']:.1f} us, new {tot['new']:.1f} us per layer -> {12 * (tot['old'] - tot['new']) / 1e3:.2f} ms per 12-layer step")
