"""Parity tests of the retired fused MLP forward (run on an MI355X: `python -m pytest scripts/gpu/experiments/mlpfwd -q` after build.sh).
Moved out of tests/test_gpu_rowgemm.py in round 6 together with the kernel; the end-to-end PPF_MLP_FUSED=1 test went with the switch
(git history: tests/test_gpu_e2e.py::test_fused_mlp_forward_matches_two_launch_path)."""
import os
import sys

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(HERE))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, HERE)
from helpers import assert_close, rel_err, report          # noqa: E402
import mlp_fwd as X                                        # noqa: E402


class ops:                                                 # the names the moved tests use
    from protopformer_amd.ops import EPI_GELU, gemm
    gemm = staticmethod(gemm)
    mlp_fwd = staticmethod(X.mlp_fwd)
    mlp_fwd_supported = staticmethod(X.mlp_fwd_supported)


@pytest.mark.parametrize("M,D,hid,rpt,tile,with_ln,layerscale", [
    (4 * 197, 384, 1536, 197, 99, True, False),        # deit_small block: half-sample tiles (the forward default)
    (3 * 197, 384, 1536, 197, 112, True, False),       # full 7 m-tiles, tiles straddle samples
    (5 * 82, 384, 1536, 82, 82, False, False),         # the compacted block behind the reservation (no following LayerNorm fused)
    (2 * 197, 384, 1536, 197, 50, True, False),        # small ragged tiles
    (3 * 197, 192, 768, 197, 99, True, False),         # deit_tiny / cait_xxs24 width
    (3 * 196, 192, 768, 196, 98, True, True),          # CaiT: LayerScale + the unscaled branch as a second output
])
def test_mlp_fwd_fused_vs_torch(M, D, hid, rpt, tile, with_ln, layerscale):
    """csrc/mlpfwd.hip (fc1 -> GELU -> fc2 -> residual + DropPath (+ LayerScale) -> next LayerNorm in one launch) vs a PyTorch fp32
    reference on the same bf16-rounded operands (timm Mlp with nn.GELU = erf form, deit:76-81), and h / gelu' against the unfused
    EPI_GELU GEMM: same formula, so equal up to the rounding of a differently ordered fp32 accumulation."""
    from helpers import gelu8_decode
    g = torch.Generator().manual_seed(11)
    a = (torch.randn(M, D, generator=g) * 0.7).bfloat16()
    w1 = (torch.randn(hid, D, generator=g) * 0.06).bfloat16()
    w2 = (torch.randn(D, hid, generator=g) * 0.04).bfloat16()
    b1, b2 = torch.randn(hid, generator=g) * 0.2, torch.randn(D, generator=g) * 0.1
    res = torch.randn(M, D, generator=g)
    B = M // rpt
    scale = torch.tensor([0.0, 1.0 / 0.9] * B)[:B].contiguous()
    lw, lb = 1.0 + 0.2 * torch.randn(D, generator=g), 0.1 * torch.randn(D, generator=g)
    gamma = (0.5 + torch.rand(D, generator=g)) if layerscale else None
    pre = a.float() @ w1.float().t() + b1
    h_ref = torch.nn.functional.gelu(pre)
    dg_ref = 0.5 * (1 + torch.erf(pre / 2 ** 0.5)) + pre * torch.exp(-0.5 * pre * pre) / (2 * torch.pi) ** 0.5
    branch = h_ref.bfloat16().float() @ w2.float().t() + b2                       # the kernel multiplies the bf16-rounded hidden layer
    x_ref = res + scale.repeat_interleave(rpt)[:, None] * (branch * gamma if layerscale else branch)
    raw = torch.empty((M, D), dtype=torch.bfloat16, device="cuda") if layerscale else None
    assert ops.mlp_fwd_supported(D, hid, tile)
    run = lambda: ops.mlp_fwd(a.cuda(), w1.cuda(), b1.cuda(), w2.cuda(), b2.cuda(), res.cuda(), tile, rowscale=scale.cuda(), rows_per_group=rpt,
                              ln_w=lw.cuda() if with_ln else None, ln_b=lb.cuda() if with_ln else None,
                              colscale=gamma.cuda() if layerscale else None, aux_out=raw)
    xo, n, mean, rstd, h, dg = run()
    torch.cuda.synchronize()
    e = dict(h=rel_err(h.float().cpu(), h_ref), x=rel_err(xo.cpu(), x_ref))
    assert_close(h.float().cpu(), h_ref, rtol=8e-3, atol=2e-3, what="h = gelu(fc1)")            # bf16 output
    assert_close(gelu8_decode(dg).cpu(), dg_ref, rtol=0.0, atol=2.7e-3, what="gelu' codes")    # half a code step = 2.55e-3
    assert e["x"] < 1.5e-3, e                                                              # fp32 residual stream; h enters in bf16 on both sides
    if layerscale:
        e["raw"] = rel_err(raw.float().cpu(), branch)
        assert e["raw"] < 5e-3
    if with_ln:
        n_ref = torch.nn.functional.layer_norm(x_ref, (D,), lw, lb, 1e-6)
        mu, var = x_ref.mean(-1), x_ref.var(-1, unbiased=False)
        e.update(n=rel_err(n.float().cpu(), n_ref), mean=rel_err(mean.cpu(), mu), rstd=rel_err(rstd.cpu(), (var + 1e-6).rsqrt()))
        assert e["mean"] < 1.5e-3 and e["rstd"] < 1.5e-3 and e["n"] < 5e-3, e
    else:
        assert n is None
    # against the unfused path: the same GELU formula on an fp32 accumulator summed in another order -> identical except where that
    # rounding crosses a bf16 / code boundary
    dg2 = torch.empty((M, hid), dtype=torch.uint8, device="cuda")
    h2 = ops.gemm(a.cuda(), w1.cuda(), epi=ops.EPI_GELU, bias=b1.cuda(), aux_out=dg2)
    diff_h = float((h != h2).float().mean()); diff_d = float((dg != dg2).float().mean())
    e.update(h_differs=diff_h, dg_differs=diff_d)
    assert diff_h < 2e-2 and diff_d < 2e-2, e
    assert int((dg.int() - dg2.int()).abs().max()) <= 1
    assert_close(h.float(), h2.float(), rtol=8e-3, atol=1e-5, what="h vs the unfused GEMM (one bf16 ulp)")
    # bit-identical repeats
    for _ in range(2):
        xo2, n2, _, _, h3, dg3 = run()
        assert torch.equal(xo, xo2) and torch.equal(h, h3) and torch.equal(dg, dg3) and (n is None or torch.equal(n, n2))
    report(f"mlp_fwd[{M},{D},{hid},{tile}]", **e)


def test_mlp_fwd_transpose_detecting():
    """An asymmetric pattern through both products: one input feature, one hidden unit per output column pattern."""
    M, D, hid = 224, 384, 1536
    a = torch.zeros(M, D); a[:, 3] = (torch.arange(M) % 13 - 6.0) / 4          # row pattern
    w1 = torch.zeros(hid, D); w1[:, 3] = (torch.arange(hid) % 7 - 3.0) / 4     # hidden-unit pattern
    w2 = torch.zeros(D, hid)
    w2[torch.arange(D), (torch.arange(D) * 5 + 1) % hid] = 1.0                  # output column n reads hidden unit 5 n + 1
    b1, b2 = torch.zeros(hid), torch.zeros(D)
    res = torch.zeros(M, D)
    xo, _, _, _, h, _ = ops.mlp_fwd(a.bfloat16().cuda(), w1.bfloat16().cuda(), b1.cuda(), w2.bfloat16().cuda(), b2.cuda(), res.cuda(), 112)
    pre = a[:, 3:4] * w1[:, 3][None, :]
    h_ref = torch.nn.functional.gelu(pre)
    assert_close(h.float().cpu(), h_ref, rtol=8e-3, atol=1e-3, what="h pattern")
    x_ref = h_ref.bfloat16().float()[:, (torch.arange(D) * 5 + 1) % hid]
    assert_close(xo.cpu(), x_ref, rtol=1e-3, atol=1e-5, what="out pattern")
