"""GPU parity of LayerNorm fwd/bwd and the streaming kernels vs plain fp32 torch / the oracle."""
import pytest
import torch

from helpers import assert_close
from oracle import ppf_oracle as O

pytestmark = pytest.mark.gpu


# rows >= 4096 take the partial-sum + ordered-reduce path for the column sums (and, without LayerScale operands, the
# specialised kernel); smaller calls use fp32 atomics
@pytest.mark.parametrize("rows,D", [(394, 384), (1000, 192), (77, 96), (50, 128), (33, 64), (6304, 384), (4100, 192)])
def test_layernorm_fwd_bwd(rows, D):
    from protopformer_amd import ops
    g = torch.Generator().manual_seed(0)
    x = torch.randn(rows, D, generator=g) * 2 + 0.5
    w = 1 + 0.2 * torch.randn(D, generator=g); b = 0.1 * torch.randn(D, generator=g)
    dy = torch.randn(rows, D, generator=g).bfloat16()
    dres = torch.randn(rows, D, generator=g)
    xr = x.clone().requires_grad_(True); wr = w.clone().requires_grad_(True); br = b.clone().requires_grad_(True)
    y_ref = O.layer_norm(xr, wr, br)
    y_ref.backward(dy.float())
    y, mean, rstd = ops.layernorm_fwd(x.cuda(), w.cuda(), b.cuda())
    assert_close(y.float(), y_ref.detach().bfloat16().float(), rtol=8e-3, atol=1e-3, what="ln fwd")
    assert_close(mean, x.mean(-1), rtol=1e-4, atol=1e-5, what="mean")
    dw = torch.zeros(D, device="cuda"); db = torch.zeros(D, device="cuda")
    dx = torch.empty(rows, D, device="cuda")
    rs = torch.tensor([0.0, 1.0 / 0.9, 1.0], device="cuda")
    rpg = (rows + 2) // 3
    cs = (0.5 + torch.rand(D, generator=g)).cuda()
    cast = torch.empty(rows, D, dtype=torch.bfloat16, device="cuda")
    dbn = torch.zeros(D, device="cuda"); dcs = torch.zeros(D, device="cuda")
    branch = torch.randn(rows, D, generator=g).bfloat16().cuda()
    ops.layernorm_bwd(dy.cuda(), x.cuda(), w.cuda(), mean, rstd, dw, db, dres_in=dres.cuda(), dx_out=dx, cast_out=cast, rowscale=rs,
                      rows_per_group=rpg, colscale=cs, dbias_next=dbn, branch=branch, dcolscale=dcs)
    dx_ref = xr.grad + dres
    assert_close(dx, dx_ref, rtol=1e-3, atol=1e-4, what="ln dx")
    assert_close(dw, wr.grad, rtol=1e-3, atol=1e-3, what="ln dw")
    assert_close(db, br.grad, rtol=1e-3, atol=1e-3, what="ln db")
    ridx = torch.arange(rows) // rpg
    scaled = dx_ref * rs.cpu()[ridx][:, None]
    cast_ref = (scaled * cs.cpu()).bfloat16().float()
    assert_close(cast.float(), cast_ref, rtol=8e-3, atol=1e-4, what="fused cast")
    assert_close(dbn, cast_ref.sum(0), rtol=2e-3, atol=2e-2, what="bias grad (colsum)")
    assert_close(dcs, (scaled * branch.float().cpu()).sum(0), rtol=2e-3, atol=2e-2, what="layerscale grad")


def test_layernorm_bwd_large_plain_and_lane():
    """DeiT configuration at scale: no LayerScale operands, column sums through the partial workspace, reduction on a side stream."""
    from protopformer_amd import ops
    from protopformer_amd.backbone import WgradLane
    rows, D = 8192, 384
    g = torch.Generator().manual_seed(1)
    x = (torch.randn(rows, D, generator=g) * 2 + 0.5).cuda(); w = (1 + 0.2 * torch.randn(D, generator=g)).cuda()
    dy = torch.randn(rows, D, generator=g).bfloat16().cuda(); dres = torch.randn(rows, D, generator=g).cuda()
    xr = x.clone().requires_grad_(True); wr = w.clone().requires_grad_(True); br = torch.zeros(D, device="cuda", requires_grad=True)
    O.layer_norm(xr, wr, br).backward(dy.float())
    mean = x.mean(-1); rstd = 1.0 / torch.sqrt(x.var(-1, unbiased=False) + 1e-6)
    for lane in (None, WgradLane(x.device)):
        dw = torch.zeros(D, device="cuda"); db = torch.zeros(D, device="cuda"); dbn = torch.zeros(D, device="cuda")
        dx = torch.empty(rows, D, device="cuda"); cast = torch.empty(rows, D, dtype=torch.bfloat16, device="cuda")
        ops.layernorm_bwd(dy, x, w, mean, rstd, dw, db, dres_in=dres, dx_out=dx, cast_out=cast, dbias_next=dbn, lane=lane)
        if lane is not None:
            lane.join()
        dx_ref = xr.grad + dres
        assert_close(dx, dx_ref, rtol=1e-3, atol=1e-4, what="ln dx")
        assert_close(dw, wr.grad, rtol=1e-3, atol=2e-3, what="ln dw")
        assert_close(db, br.grad, rtol=1e-3, atol=2e-3, what="ln db")
        assert_close(dbn, dx_ref.bfloat16().float().sum(0), rtol=2e-3, atol=5e-2, what="bias grad (colsum)")
    dw2 = torch.zeros(D, device="cuda"); db2 = torch.zeros(D, device="cuda"); dbn2 = torch.zeros(D, device="cuda")
    ops.layernorm_bwd(dy, x, w, mean, rstd, dw2, db2, dres_in=dres, dx_out=dx, cast_out=cast, dbias_next=dbn2)
    assert torch.equal(dw, dw2) and torch.equal(db, db2) and torch.equal(dbn, dbn2)      # ordered reduce: bit-identical


def test_layernorm_gather_rows():
    from protopformer_amd import ops
    g = torch.Generator().manual_seed(1)
    x = torch.randn(40, 64, generator=g); w = torch.ones(64); b = torch.zeros(64)
    rm = torch.tensor([5, 0, 39, 7, 7], dtype=torch.int32)
    y, _, _ = ops.layernorm_fwd(x.cuda(), w.cuda(), b.cuda(), row_map=rm.cuda())
    assert_close(y.float(), O.layer_norm(x[rm.long()], w, b).bfloat16().float(), rtol=8e-3, atol=1e-3, what="gathered ln")


def test_im2col_assemble():
    from protopformer_amd import ops
    g = torch.Generator().manual_seed(2)
    B, D, p = 3, 64, 16
    img = torch.randn(B, 3, 64, 48, generator=g)
    cols = ops.im2col_patch(img.cuda(), p).float().cpu()
    gh, gw = 4, 3
    ref = img.reshape(B, 3, gh, p, gw, p).permute(0, 2, 4, 1, 3, 5).reshape(B * gh * gw, 3 * p * p)
    assert torch.equal(cols, ref.bfloat16().float())
    Np = gh * gw
    tok = torch.randn(B * Np, D, generator=g); cls = torch.randn(D, generator=g); pos = torch.randn(Np + 1, D, generator=g)
    x = ops.assemble_tokens(tok.cuda(), cls.cuda(), pos.cuda(), B, Np, D, 1).cpu()
    ref = torch.cat([cls.expand(B, 1, D), tok.reshape(B, Np, D)], 1) + pos
    assert torch.equal(x, ref)
    x0 = ops.assemble_tokens(tok.cuda(), cls.cuda(), pos[:Np].contiguous().cuda(), B, Np, D, 0).cpu()
    assert torch.equal(x0, tok.reshape(B, Np, D) + pos[:Np])
    dx = torch.randn(B, Np + 1, D, generator=g)
    dpos = torch.zeros(Np + 1, D, device="cuda"); dcls = torch.zeros(D, device="cuda")
    dtok = ops.assemble_tokens_bwd(dx.cuda(), dpos, dcls, B, Np, D, 1)
    assert_close(dpos, dx.sum(0), rtol=1e-5, atol=1e-5, what="dpos")
    assert_close(dcls, dx[:, 0].sum(0), rtol=1e-5, atol=1e-5, what="dcls")
    assert torch.equal(dtok.float().cpu(), dx[:, 1:].reshape(B * Np, D).bfloat16().float())


def test_adamw_matches_torch():
    from protopformer_amd import _lib
    g = torch.Generator().manual_seed(3)
    n = 4096
    p0 = torch.randn(n, generator=g); grads = [torch.randn(n, generator=g) * 0.1 for _ in range(3)]
    pr = p0.clone().requires_grad_(True)
    opt = torch.optim.AdamW([{"params": [pr], "lr": 3e-3, "weight_decay": 0.05}], eps=1e-8)
    p = p0.clone().cuda(); m = torch.zeros(n, device="cuda"); v = torch.zeros(n, device="cuda"); ema = p.clone()
    p16 = torch.empty(n, dtype=torch.bfloat16, device="cuda")
    bounds = torch.tensor([0, 1024, n], dtype=torch.int64); lr = torch.tensor([3e-3, 3e-3]); wd = torch.tensor([0.05, 0.05])
    for step, gr in enumerate(grads, 1):
        pr.grad = gr.clone(); opt.step()
        _lib.call("ppf_adamw_step", p, gr.cuda(), m, v, ema, p16, n, 2, bounds.data_ptr(), lr.data_ptr(), wd.data_ptr(), 0.9, 0.999, 1e-8, step, 0.99, 1.0)
    assert_close(p, pr.detach(), rtol=1e-5, atol=1e-6, what="adamw params")
    assert_close(p16.float(), p.bfloat16().float().cpu(), rtol=0, atol=0, what="bf16 recast")


def test_axpbypcz_and_loss_fn_backward_constants():
    """loss = ce + c1 cov + c2 mean in one launch; seeded with the cached one the backward hands out cached constants (no launches)."""
    import torch
    from protopformer_amd import ops
    from protopformer_amd.protopformer import WeightedLossFn
    x, y, z = (torch.randn(5, device="cuda") for _ in range(3))
    out = ops.axpbypcz(x, y, z, 1.0, 0.1, 0.5)
    assert torch.allclose(out, x + 0.1 * y + 0.5 * z, rtol=1e-6, atol=1e-6)
    ce, cov, mean = (torch.tensor(v, device="cuda", requires_grad=True) for v in (2.0, 3.0, 4.0))
    loss = WeightedLossFn.apply(ce, cov, mean, 0.1, 0.5)
    assert abs(float(loss) - (2.0 + 0.3 + 2.0)) < 1e-6
    loss.backward(gradient=ops.const_scalar(loss.device, 1.0))
    assert float(ce.grad) == 1.0 and abs(float(cov.grad) - 0.1) < 1e-7 and float(mean.grad) == 0.5
    ce2, cov2, mean2 = (torch.tensor(v, device="cuda", requires_grad=True) for v in (2.0, 3.0, 4.0))
    (WeightedLossFn.apply(ce2, cov2, mean2, 0.1, 0.5) * 2.0).backward()             # generic upstream gradient
    assert float(ce2.grad) == 2.0 and abs(float(cov2.grad) - 0.2) < 1e-6 and float(mean2.grad) == 1.0
