"""nt256 GEMM vs the 128x128 kernel (PPF_GEMM_NT256 is read once per process, so compare against a torch fp32 reference) + timing."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from protopformer_amd import ops
torch.manual_seed(0)
dev = "cuda"
def rnd(*s, sc=0.5): return (torch.randn(*s, device=dev) * sc).bfloat16()
def check(M, N, K, epi, tag):
    x = rnd(M, K); w = rnd(N, K, sc=0.05); bias = torch.randn(N, device=dev) * 0.1
    ref = x.float() @ w.float().t() + bias
    kw = {}
    if epi == ops.EPI_RESID:
        res = torch.randn(M, N, device=dev); kw = dict(res=res, out=torch.empty(M, N, device=dev)); want = res + ref
    elif epi == ops.EPI_GELU:
        aux = torch.empty(M, N, dtype=torch.uint8, device=dev); kw = dict(aux_out=aux); want = torch.nn.functional.gelu(ref)
    elif epi == ops.EPI_DGELU:
        hpre = rnd(M, N); kw = dict(aux_in=hpre); bias = None
        ref = x.float() @ w.float().t()
        want = ref * hpre.float()
    else:
        want = ref
    out = ops.gemm(x, w, epi=epi, bias=bias, **kw)
    torch.cuda.synchronize()
    err = (out.float() - want).abs().max().item(); sc = want.abs().max().item()
    bad = (~torch.isfinite(out.float())).sum().item()
    print(f"{tag:28s} M={M} N={N} K={K} max_abs_err={err:.4e} scale={sc:.3f} rel={err / sc:.2e} nonfinite={bad}", flush=True)
    return err / sc
worst = 0
for (M, N, K) in [(50432, 1152, 384), (50432, 384, 384), (50432, 1536, 384), (50432, 384, 1536), (50432, 384, 1152), (33000, 520, 192), (40000, 256, 128)]:
    for epi, tag in [(ops.EPI_BF16, "bf16"), (ops.EPI_RESID, "resid"), (ops.EPI_GELU, "gelu"), (ops.EPI_DGELU, "dgelu")]:
        worst = max(worst, check(M, N, K, epi, tag))
print("WORST rel", worst)
# repeatability (race screen): same inputs 20x must be bit-identical
x = rnd(50432, 1536); w = rnd(384, 1536, sc=0.05)
o0 = ops.gemm(x, w, epi=ops.EPI_BF16)
same = all(torch.equal(o0, ops.gemm(x, w, epi=ops.EPI_BF16)) for _ in range(20))
x2 = rnd(50432, 384); w2 = rnd(1536, 384, sc=0.05)
o1 = ops.gemm(x2, w2, epi=ops.EPI_BF16)
same2 = all(torch.equal(o1, ops.gemm(x2, w2, epi=ops.EPI_BF16)) for _ in range(20))
print("repeatable:", same, same2)
