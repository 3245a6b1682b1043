import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from protopformer_amd import ops
M, D = 256 * 197, 384
dev = "cuda"
torch.manual_seed(0)
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
x = torch.randn(M, D, device=dev); dy = (torch.randn(M, D, device=dev) * 0.1).bfloat16(); w = torch.randn(D, device=dev)
mean = x.mean(1); rstd = 1.0 / torch.sqrt(x.var(1, unbiased=False) + 1e-6)
dres = torch.randn(M, D, device=dev); dx = torch.empty_like(dres); cast = torch.empty(M, D, dtype=torch.bfloat16, device=dev)
dw = torch.zeros(D, device=dev); db = torch.zeros(D, device=dev); dnb = torch.zeros(D, device=dev)
rs = torch.ones(256, device=dev)
t_full = timeit(lambda: ops.layernorm_bwd(dy, x, w, mean, rstd, dw, db, dres_in=dres, dx_out=dx, cast_out=cast, rowscale=rs, rows_per_group=197, dbias_next=dnb))
t_nocol = timeit(lambda: ops.layernorm_bwd(dy, x, w, mean, rstd, None, None, dres_in=dres, dx_out=dx, cast_out=cast, rowscale=rs, rows_per_group=197))
y = torch.empty(M, D, dtype=torch.bfloat16, device=dev)
t_fwd = timeit(lambda: ops.layernorm_fwd(x, w, w, 1e-6))
print(f"ln_bwd full {t_full:.1f} us | without column sums {t_nocol:.1f} us | ln_fwd {t_fwd:.1f} us   (bytes bwd: {(M*D*(2+4+4+4+2))/1e6:.0f} MB)")
