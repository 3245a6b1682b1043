"""csrc/rowgemm.hip through the C ABI vs a plain PyTorch fp32 reference of the same op (bf16-rounded operands, fp32 math):
full-row GEMM + bias, + DropPath-scaled residual + LayerNorm forward (deit:76-81), and the input-gradient GEMM fused with the
LayerNorm backward (autograd of the same lines, checked against torch.autograd).  Tolerances: fp32 accumulation of bf16 products ->
1e-3 of the output scale for fp32 outputs; bf16 outputs add their own rounding (2^-9 relative)."""
import pytest
import torch

from helpers import assert_close, rel_err, report

pytestmark = pytest.mark.gpu


def _mk(M, D, K, seed):
    g = torch.Generator().manual_seed(seed)
    a = (torch.randn(M, K, generator=g) * 0.5).bfloat16()
    b = (torch.randn(D, K, generator=g) * 0.05).bfloat16()
    return a, b, g


@pytest.mark.parametrize("M,D,K,rpt", [(4 * 197, 384, 384, 197), (3 * 197, 384, 1536, 197), (5 * 82, 384, 1152, 82), (2 * 196, 192, 768, 196),
                                       (3 * 197, 192, 192, 197), (2 * 122, 192, 192, 122), (1000, 384, 64, 208), (3 * 197, 384, 128, 197)])
def test_rowgemm_bf16_and_transposed_shadow(M, D, K, rpt):
    from protopformer_amd import ops
    a, b, g = _mk(M, D, K, 1)
    bias = torch.randn(D, generator=g) * 0.1
    ref = a.float() @ b.float().t() + bias
    out = ops.rowgemm_bf16(a.cuda(), b.cuda(), rpt, bias=bias.cuda())
    torch.cuda.synchronize()
    assert_close(out.float().cpu(), ref, rtol=1e-2, atol=1e-2 * float(ref.abs().max()), what="rowgemm bf16")
    assert rel_err(out.float().cpu(), ref) < 4e-3            # bf16 output rounding (2^-9) dominates; measured ~2e-3


@pytest.mark.parametrize("M,D,K,rpt,with_ln,layerscale", [(4 * 197, 384, 384, 197, True, False), (3 * 197, 384, 1536, 197, True, False),
                                                          (5 * 82, 384, 1536, 82, False, False), (2 * 196, 192, 768, 196, True, False),
                                                          (3 * 197, 192, 192, 197, True, False), (3 * 196, 192, 192, 196, True, True),
                                                          (2 * 196, 192, 768, 196, False, True)])
@pytest.mark.parametrize("half_tiles", [False, True], ids=["sample-tiles", "half-sample-tiles"])
def test_rowgemm_resid_ln_vs_torch(M, D, K, rpt, with_ln, layerscale, half_tiles):
    """layerscale: CaiT's per-channel gamma on the branch (cait:153-155) and the unscaled branch as a second (bf16) output."""
    from protopformer_amd import ops
    a, b, g = _mk(M, D, K, 2)
    bias = torch.randn(D, generator=g) * 0.1
    res = torch.randn(M, D, generator=g)
    B = M // rpt
    scale = torch.tensor([0.0, 1.0 / 0.9] * B)[:B].contiguous()                      # DropPath factors: dropped / kept
    lw, lb = 1.0 + 0.2 * torch.randn(D, generator=g), 0.1 * torch.randn(D, generator=g)
    gamma = (0.5 + torch.rand(D, generator=g)) if layerscale else None
    branch = a.float() @ b.float().t() + bias
    x_ref = res + scale.repeat_interleave(rpt)[:, None] * (branch * gamma if layerscale else branch)
    raw = torch.empty((M, D), dtype=torch.bfloat16, device="cuda") if layerscale else None
    # half_tiles: workgroup tiles of half a sample that do not end on sample boundaries (ops.rowgemm_tile_rows at small batches); the
    # DropPath factor stays per SAMPLE (rows_per_group)
    tile = (rpt + 1) // 2 if half_tiles else rpt
    xo, n, mean, rstd = ops.rowgemm_resid_ln(a.cuda(), b.cuda(), res.cuda(), tile, bias=bias.cuda(), rowscale=scale.cuda(), rows_per_group=rpt,
                                             ln_w=lw.cuda() if with_ln else None, ln_b=lb.cuda() if with_ln else None,
                                             colscale=gamma.cuda() if layerscale else None, aux_out=raw)
    torch.cuda.synchronize()
    e = dict(x=rel_err(xo.cpu(), x_ref))
    assert e["x"] < 1e-3
    if layerscale:
        e["raw"] = rel_err(raw.float().cpu(), branch)
        assert e["raw"] < 4e-3                                                     # bf16 output
    if with_ln:
        n_ref = torch.nn.functional.layer_norm(x_ref, (D,), lw, lb, 1e-6)
        mu, var = x_ref.mean(-1), x_ref.var(-1, unbiased=False)
        e.update(n=rel_err(n.float().cpu(), n_ref), mean=rel_err(mean.cpu(), mu), rstd=rel_err(rstd.cpu(), (var + 1e-6).rsqrt()))
        assert e["mean"] < 1e-3 and e["rstd"] < 1e-3 and e["n"] < 4e-3              # n is bf16: its rounding dominates
    else:
        assert n is None
    report(f"rowgemm_resid_ln[{M},{D},{K}]", **e)


@pytest.mark.parametrize("M,D,K,rpt", [(4 * 197, 384, 1536, 197), (3 * 197, 384, 1152, 197), (5 * 82, 384, 1536, 82), (2 * 196, 192, 768, 196),
                                       (3 * 197, 192, 576, 197)])
@pytest.mark.parametrize("half_tiles", [False, True], ids=["sample-tiles", "half-sample-tiles"])
@pytest.mark.parametrize("layerscale", [False, True], ids=["plain", "layerscale"])
def test_rowgemm_lnbwd_vs_autograd(M, D, K, rpt, half_tiles, layerscale):
    """dn = dy W (W^T passed contraction-contiguous) -> LayerNorm backward + residual gradient, against torch.autograd on
    y = LN(x) * w + b, L = sum(dn * y) + sum(dres * x)."""
    from protopformer_amd import ops
    a, b, g = _mk(M, D, K, 3)
    x = torch.randn(M, D, generator=g) * 1.5 + 0.3
    w = 1.0 + 0.2 * torch.randn(D, generator=g)
    dres = torch.randn(M, D, generator=g) * 0.2
    B = M // rpt
    scale = torch.tensor([1.0 / 0.95, 0.0, 1.0 / 0.95] * B)[:B].contiguous()
    dn = a.float() @ b.float().t()
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), torch.zeros(D, requires_grad=True)
    y = torch.nn.functional.layer_norm(xr, (D,), wr, br, 1e-6)
    (y * dn).sum().backward()
    dx_ref = dres + xr.grad
    mu, var = x.mean(-1), x.var(-1, unbiased=False)
    mean, rstd = mu.cuda(), (var + 1e-6).rsqrt().cuda()
    dw = torch.full((D,), 0.5, device="cuda"); db = torch.full((D,), -0.25, device="cuda")           # accumulate (+=) semantics
    cast = torch.empty((M, D), dtype=torch.bfloat16, device="cuda")
    tile = (rpt + 1) // 2 if half_tiles else rpt
    # layerscale (cait:153-155): the branch below the residual is gamma * branch: cast_out carries gamma, d gamma = sum_m scale * dx * branch
    gamma = (0.5 + torch.rand(D, generator=g)) if layerscale else None
    branch = torch.randn(M, D, generator=g).bfloat16() if layerscale else None
    dg = torch.full((D,), 0.125, device="cuda") if layerscale else None
    ls = dict(colscale=gamma.cuda(), branch=branch.cuda(), dcolscale=dg) if layerscale else {}
    dx = ops.rowgemm_lnbwd(a.cuda(), b.cuda(), x.cuda(), mean, rstd, w.cuda(), dw, db, tile, dres_in=dres.cuda(), cast_out=cast, rowscale=scale.cuda(),
                           rows_per_group=rpt, **ls)
    torch.cuda.synchronize()
    rows_scale = scale.repeat_interleave(rpt)[:, None]
    e = dict(dx=rel_err(dx.cpu(), dx_ref), dw=rel_err(dw.cpu() - 0.5, wr.grad), db=rel_err(db.cpu() + 0.25, br.grad),
             cast=rel_err(cast.float().cpu(), dx_ref * rows_scale * (gamma if layerscale else 1.0)))
    if layerscale:
        e["dgamma"] = rel_err(dg.cpu() - 0.125, (dx_ref * rows_scale * branch.float()).sum(0))
        assert e["dgamma"] < 1e-3
    report(f"rowgemm_lnbwd[{M},{D},{K}{',ls' if layerscale else ''}]", **e)
    assert e["dx"] < 1e-3 and e["dw"] < 1e-3 and e["db"] < 1e-3 and e["cast"] < 4e-3
    # in place on the residual gradient (dx_out aliases dres_in) and bit-identical from run to run
    dres_c = dres.cuda()
    dw2 = torch.full((D,), 0.5, device="cuda"); db2 = torch.full((D,), -0.25, device="cuda")
    if layerscale:
        ls["dcolscale"] = torch.full((D,), 0.125, device="cuda")
    dx2 = ops.rowgemm_lnbwd(a.cuda(), b.cuda(), x.cuda(), mean, rstd, w.cuda(), dw2, db2, tile, dres_in=dres_c, dx_out=dres_c, cast_out=cast, rowscale=scale.cuda(),
                            rows_per_group=rpt, **ls)
    torch.cuda.synchronize()
    assert torch.equal(dx2, dx) and torch.equal(dw2, dw) and torch.equal(db2, db)
    assert not layerscale or torch.equal(ls["dcolscale"], dg)


def test_transposed_weight_shadow():
    from protopformer_amd import ops
    g = torch.Generator().manual_seed(4)
    shapes = [(1152, 384), (384, 384), (1536, 384), (100, 72)]
    src = torch.randn(sum(r * c for r, c in shapes) + 64, generator=g).bfloat16().cuda()
    desc, off, doff, tiles = [], 8, 0, 0
    for r, c in shapes:
        desc.append((off, doff, r, c)); off += r * c; doff += r * c + 8; tiles += ((r + 63) // 64) * ((c + 63) // 64)
    dst = torch.zeros(doff, dtype=torch.bfloat16, device="cuda")
    ops.transpose_bf16_batched(src, dst, torch.tensor(desc, dtype=torch.int64).cuda(), len(desc), tiles)
    torch.cuda.synchronize()
    for so, do, r, c in desc:
        assert torch.equal(dst[do:do + r * c].view(c, r), src[so:so + r * c].view(r, c).t())
