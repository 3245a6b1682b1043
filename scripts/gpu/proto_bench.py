"""Stand-alone timing of the prototype-layer kernels at the deit_small train-step shape (B=256, 1+81 tokens, Dp=384, P=2000)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from protopformer_amd import ops

B, T, Dp, P = 256, 81, 384, 2000
g = torch.Generator().manual_seed(0)
tok = torch.rand(B, T + 1, Dp, generator=g).cuda()
pro = torch.rand(P, Dp, generator=g).cuda()


def timeit(fn, iters=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


t_full = timeit(lambda: ops.proto_fwd(tok, 1, T, pro))
t_max = timeit(lambda: ops.proto_fwd(tok, 1, T, pro, want_dist=False, want_act=False))
print(f"proto_fwd full maps {t_full:7.1f} us | max/argmax only {t_max:7.1f} us   (fp32 MFMA floor 240 us, maps 332 MB)")

# global branch (one token per sample) backward: the dense two-product form
Pg = 2000
prg = torch.rand(Pg, Dp, generator=g).cuda()
am, _, dist_g, _ = ops.proto_fwd(tok, 0, 1, prg, want_act=False)
gg = torch.randn(B, Pg, generator=g).cuda()
dtok = torch.zeros_like(tok); dpro = torch.zeros_like(prg)
for mode in ("1",):
    t_tok = timeit(lambda: ops.proto_bwd(tok, 0, 1, prg, dist_g, None, gg, None, dtok, None))
    t_pro = timeit(lambda: ops.proto_bwd(tok, 0, 1, prg, dist_g, None, gg, None, None, dpro))
    print(f"T=1 backward, dense={mode}: d tokens {t_tok:7.1f} us | d prototypes {t_pro:7.1f} us")
# local branch backward (PPC-like sparse g_full + arg-max gradient)
am, argmax, dist, act = ops.proto_fwd(tok, 1, T, pro)
gfull = torch.zeros(B, P, T, device="cuda")
lab = torch.randint(0, 200, (B,), generator=g)
for b in range(B):
    gfull[b, lab[b] * 10: lab[b] * 10 + 10] = 0.01
gmax = torch.randn(B, P, generator=g).cuda()
t_tok = timeit(lambda: ops.proto_bwd(tok, 1, T, pro, dist, gfull, gmax, argmax, dtok, None))
t_pro = timeit(lambda: ops.proto_bwd(tok, 1, T, pro, dist, gfull, gmax, argmax, None, dpro))
print(f"T=81 backward: d tokens (mark + gather) {t_tok:7.1f} us | d prototypes {t_pro:7.1f} us")
gz = torch.zeros_like(gfull)
t_z = timeit(lambda: ops.proto_bwd(tok, 1, T, pro, dist, gz, gmax, argmax, None, dpro))
t_n = timeit(lambda: ops.proto_bwd(tok, 1, T, pro, dist, None, gmax, argmax, None, dpro))
print(f"   d prototypes with an all-zero g_full {t_z:7.1f} us | without g_full {t_n:7.1f} us")
# block-row form of the same gradient (what the train step uses): nothing of shape (B,P,T) is scanned
rows = torch.full((B, 10, T), 0.01, device="cuda")
blk = (rows, lab.cuda(), 10)
t_tok = timeit(lambda: ops.proto_bwd(tok, 1, T, pro, dist, None, gmax, argmax, dtok, None, rows=blk))
t_pro = timeit(lambda: ops.proto_bwd(tok, 1, T, pro, dist, None, gmax, argmax, None, dpro, rows=blk))
print(f"T=81 backward, block rows: d tokens {t_tok:7.1f} us | d prototypes {t_pro:7.1f} us")
