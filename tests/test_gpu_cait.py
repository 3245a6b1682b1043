"""GPU parity of the CaiT path: talking-heads attention, class attention (golden vectors from the reference), end to end."""
import numpy as np
import pytest
import torch

import golden_inputs as gi
from helpers import assert_close, assert_elementwise, load_npz, micro, rel_err, report
from oracle import ppf_oracle as O

pytestmark = pytest.mark.gpu

# gradient-direction gates of the two backbone tests below: 3 x the deficit measured on MI355X (profiles/r4_tol_report.jsonl)
GATE_MICRO_BACKBONE, GATE_REAL_BACKBONE = 0.9996, 0.9995          # measured 0.99988 / 0.99986


class _P:
    pass


def _th_block(wl, bl, ww, bw):
    """Minimal stand-in for a TalkingHeadAttn block: the two head-mixing layers."""
    blk = _P(); blk.attn = _P(); blk.attn.proj_l = _P(); blk.attn.proj_w = _P()
    blk.attn.proj_l.weight, blk.attn.proj_l.bias = wl.cuda().contiguous(), bl.cuda()
    blk.attn.proj_w.weight, blk.attn.proj_w.bias = ww.cuda().contiguous(), bw.cuda()
    return blk


def _th_forward(qkv16, wl, bl, ww, bw, B, H, N, D):
    """The product's forward helper (fused kernel or the materialising kernels, PPF_TH_FUSED): (ao, saved statistics / P, a16, head mean)."""
    from protopformer_amd.cait import _th_attention_fwd
    hm = torch.empty((B, N, (N + 3) // 4 * 4), dtype=torch.float32, device="cuda")
    ao, prob, a16 = _th_attention_fwd(_th_block(wl, bl, ww, bw), qkv16, B, H, N, D, hm)
    return ao, prob, a16, hm


FUSED = pytest.mark.parametrize("fused", ["1", "pv0", "0"], ids=["fused", "fused-separate-products", "materialised"])


def _set_th_mode(monkeypatch, fused):
    """fused: one launch incl. A.V forward, th_bwd + one launch for dQ / dK / dV backward; pv0: the fused kernels up to A / dS, the
    per-head products as batched GEMMs; 0: the materialising kernels."""
    monkeypatch.setenv("PPF_TH_FUSED", fused)


@FUSED
def test_talking_heads_forward_golden(fused, monkeypatch):
    """TalkingHeadAttn.forward on the reference's weights/inputs (ops_real.npz), qkv/proj through the GEMM kernel."""
    from protopformer_amd import ops
    _set_th_mode(monkeypatch, fused)
    assert ops.th_fused_ok(4, 196, 192) == (fused != "0")
    z = load_npz("ops_real.npz")
    c = gi.cait_inputs()
    B, H, N, D = c["B"], c["H"], c["N"], c["D"]
    w = {k: v.cuda() for k, v in c["th"].items()}
    x16 = c["x"].reshape(B * N, D).cuda().bfloat16()
    qkv = ops.gemm(x16, w["qkv.weight"].bfloat16(), epi=ops.EPI_BF16, bias=w["qkv.bias"])
    ao, prob, a16, hm = _th_forward(qkv, w["proj_l.weight"].contiguous(), w["proj_l.bias"], w["proj_w.weight"].contiguous(), w["proj_w.bias"], B, H, N, D)
    out = ops.gemm(ao, w["proj.weight"].bfloat16(), epi=ops.EPI_F32, bias=w["proj.bias"])
    # bf16 operands through qkv GEMM, A.V and proj: error relative to the tensor's scale (values reach +-60 here)
    assert rel_err(out.reshape(B, N, D), z["cait/th_out"]) < 1.5e-2, rel_err(out.reshape(B, N, D), z["cait/th_out"])
    idx = torch.from_numpy(z["cait/th_attn_idx"])
    post = a16[..., :N].float().cpu().reshape(-1)[idx]
    assert rel_err(post, z["cait/th_attn_val"]) < 2e-2, rel_err(post, z["cait/th_attn_val"])
    # head mean vs oracle on the same bf16 qkv
    y, p_ref = O.cait_talking_heads_attention({"t." + k: v for k, v in c["th"].items()}, "t.", c["x"], H)
    assert rel_err(hm[..., :N], p_ref.mean(1)) < 2e-2, rel_err(hm[..., :N], p_ref.mean(1))


@FUSED
def test_talking_heads_backward_vs_oracle(fused, monkeypatch):
    from protopformer_amd import ops
    _set_th_mode(monkeypatch, fused)
    B, H, N, D = 2, 4, 196, 192
    hd = D // H
    g = torch.Generator().manual_seed(3)
    qkv = (0.8 * torch.randn(B * N, 3 * D, generator=g)).bfloat16()
    wl = torch.eye(H) + 0.3 * torch.randn(H, H, generator=g); bl = 0.1 * torch.randn(H, generator=g)
    ww = torch.eye(H) + 0.3 * torch.randn(H, H, generator=g); bw = 0.1 * torch.randn(H, generator=g)
    dao = torch.randn(B * N, D, generator=g).bfloat16()
    # oracle on the same bf16-rounded qkv
    t = qkv.float().clone().requires_grad_(True)
    wl_r, bl_r, ww_r, bw_r = (v.clone().requires_grad_(True) for v in (wl, bl, ww, bw))
    q5 = t.reshape(B, N, 3, H, hd)
    q, k, v = q5[:, :, 0].transpose(1, 2) * hd ** -0.5, q5[:, :, 1].transpose(1, 2), q5[:, :, 2].transpose(1, 2)
    s = q @ k.transpose(-1, -2)
    s = torch.einsum("bhnm,gh->bgnm", s, wl_r) + bl_r.reshape(1, -1, 1, 1)
    p = s.softmax(-1)
    a = torch.einsum("bhnm,gh->bgnm", p, ww_r) + bw_r.reshape(1, -1, 1, 1)
    o = (a @ v).transpose(1, 2).reshape(B * N, D)
    o.backward(dao.float())
    # HIP path
    from protopformer_amd.cait import _th_attention_bwd

    class _S:                                   # minimal stand-in for the flat store the helper expects
        device = torch.device("cuda", 0)
        def __init__(self): self.g = {}
        def grad_view(self, p_): return self.g.setdefault(id(p_), torch.zeros_like(p_))

    blk = _th_block(wl, bl, ww, bw)
    qd = qkv.cuda()
    ao, prob, a16, hm = _th_forward(qd, blk.attn.proj_l.weight, blk.attn.proj_l.bias, blk.attn.proj_w.weight, blk.attn.proj_w.bias, B, H, N, D)
    assert rel_err(ao.float(), o.detach()) < 1.5e-2
    st = _S()
    dqkv = _th_attention_bwd(st, blk, dict(qkv=qd, prob=prob, a16=a16), dao.cuda(), B, H, N, D)
    from protopformer_amd.backbone import wgrad_lane
    wgrad_lane(st).join()                       # the fused path sums the parameter-gradient partials on the side stream
    torch.cuda.synchronize()
    dqkv = dqkv.float().cpu()
    scale = float(t.grad.abs().max())
    for name, sl in (("dq", slice(0, D)), ("dk", slice(D, 2 * D)), ("dv", slice(2 * D, 3 * D))):
        assert_close(dqkv[:, sl], t.grad[:, sl], rtol=3e-2, atol=2e-2 * scale, what=name)
    # (d proj_l.bias is identically zero: a per-row constant in front of a softmax -- both sides are rounding noise)
    assert float(st.g[id(blk.attn.proj_l.bias)].abs().max()) < 1e-3 * float(wl_r.grad.abs().max())
    for name, mine, ref in (("dWl", st.g[id(blk.attn.proj_l.weight)], wl_r.grad),
                            ("dWw", st.g[id(blk.attn.proj_w.weight)], ww_r.grad), ("dbw", st.g[id(blk.attn.proj_w.bias)], bw_r.grad)):
        assert_close(mine, ref, rtol=3e-2, atol=2e-2 * float(ref.abs().max()), what=name)


def test_class_attention_golden_and_backward():
    from protopformer_amd import ops
    z = load_npz("ops_real.npz")
    c = gi.cait_inputs()
    B, H, N, D = c["B"], c["H"], c["N"], c["D"]
    N1 = N + 1
    w = {k: v.cuda() for k, v in c["ca"].items()}
    u16 = c["u"].reshape(B * N1, D).cuda().bfloat16()
    kk = ops.gemm(u16, w["k.weight"].bfloat16(), epi=ops.EPI_BF16, bias=w["k.bias"])
    vv = ops.gemm(u16, w["v.weight"].bfloat16(), epi=ops.EPI_BF16, bias=w["v.bias"])
    qq = ops.gemm(u16.reshape(B, N1, D)[:, 0].contiguous(), w["q.weight"].bfloat16(), epi=ops.EPI_BF16, bias=w["q.bias"])
    for tag, pol in (("", c["policy"].cuda()), ("_ones", None)):
        out, attn, zinv, rowmean = ops.class_attn_fwd(qq, kk, vv, pol, B, H, N1, D)
        y = ops.gemm(out, w["proj.weight"].bfloat16(), epi=ops.EPI_F32, bias=w["proj.bias"])
        assert rel_err(y.reshape(B, 1, D), z[f"cait/ca_out{tag}"]) < 1.5e-2
        assert rel_err(attn.reshape(B, H, 1, N1), z[f"cait/ca_attn{tag}"]) < 3e-2
        assert rel_err(rowmean, torch.from_numpy(z[f"cait/ca_attn{tag}"]).mean(1)[:, 0]) < 3e-2
    # backward vs autograd of the oracle's policy softmax on the same bf16 q/k/v
    hd = D // H
    g = torch.Generator().manual_seed(7)
    dout = torch.randn(B, D, generator=g).bfloat16()
    qf, kf, vf = (t.float().cpu().clone().requires_grad_(True) for t in (qq, kk, vv))
    qh = qf.reshape(B, 1, H, hd).permute(0, 2, 1, 3) * hd ** -0.5
    kh = kf.reshape(B, N1, H, hd).permute(0, 2, 1, 3); vh = vf.reshape(B, N1, H, hd).permute(0, 2, 1, 3)
    pr = O.policy_softmax(qh @ kh.transpose(-1, -2), c["policy"], self_keep=False)
    (pr @ vh).transpose(1, 2).reshape(B, D).backward(dout.float())
    out, attn, zinv, _ = ops.class_attn_fwd(qq, kk, vv, c["policy"].cuda(), B, H, N1, D)
    dq, dk, dv = ops.class_attn_bwd(qq, kk, vv, attn, zinv, dout.cuda(), B, H, N1, D)
    for name, mine, ref in (("dq", dq, qf.grad), ("dk", dk, kf.grad), ("dv", dv, vf.grad)):
        assert_close(mine.float(), ref, rtol=3e-2, atol=2e-2 * float(ref.abs().max()), what=name)


def _build_micro_cait(cfg, sd):
    from protopformer_amd.cait import MyCait
    from protopformer_amd.protopformer import PPNet
    feats = MyCait(img_size=cfg["img"], patch_size=16, embed_dim=cfg["dim"], depth=cfg["depth"], num_heads=cfg["heads"], drop_path_rate=0.0)
    m = PPNet(features=feats, img_size=cfg["img"], prototype_shape=[cfg["num_prototypes"], cfg["proto_dim"], 1, 1], proto_layer_rf_info=None,
              num_classes=cfg["num_classes"], reserve_layers=[cfg["reserve_layer"]], reserve_token_nums=[cfg["reserve_k"]], use_global=True,
              use_ppc_loss=True, ppc_cov_thresh=1., ppc_mean_thresh=2., global_coe=cfg["global_coe"],
              global_proto_per_class=cfg["global_per_class"], add_on_layers_type="regular")
    m.load_state_dict(sd, strict=True)
    return m.cuda()


def test_micro_cait_against_reference_fixture():
    from protopformer_amd.protopformer import CrossEntropyLoss
    sd, cfg, z = micro("micro_cait.npz")
    m = _build_micro_cait(cfg, sd)
    img, label = torch.from_numpy(z["img"]).cuda(), torch.from_numpy(z["label"]).cuda()
    m.eval()
    logits, (cls_attn, dist, lg, ll) = m(img)
    assert rel_err(cls_attn, z["eval/cls_token_attn"]) < 6e-2, rel_err(cls_attn, z["eval/cls_token_attn"])     # measured 2.6e-2
    ref_idx = torch.from_numpy(z["eval/cls_token_attn"]).topk(cfg["reserve_k"], dim=-1)[1].sort(dim=-1)[0]
    assert torch.equal(m._tokens(img)[2].cpu().long(), ref_idx), "reserved tokens differ from the reference"
    TOL = 2.5e-3              # measured: logits 2.6e-4, distances 8.2e-4 (gates <= 3x)
    report("micro_cait_eval", logits=rel_err(logits, z["eval/logits"]), cls_attn=rel_err(cls_attn, z["eval/cls_token_attn"]), dist=rel_err(dist, z["eval/distances"]))
    assert rel_err(logits, z["eval/logits"]) < TOL, rel_err(logits, z["eval/logits"])
    assert_elementwise(logits, z["eval/logits"], TOL, "cait eval logits")
    assert rel_err(dist, z["eval/distances"]) < TOL, rel_err(dist, z["eval/distances"])
    m.train()
    logits, aux = m(img)
    ce = CrossEntropyLoss()(logits, label)
    cov, mean = m.get_PPC_loss(aux[2], aux[3], aux[4], label)
    loss = ce + 0.1 * cov + 0.5 * mean
    report("micro_cait_train", **{name: rel_err(val, z[f"train/{name}"]) for name, val in (("ce", ce), ("ppc_cov", cov), ("ppc_mean", mean), ("loss", loss))})
    for name, val in (("ce", ce), ("ppc_cov", cov), ("ppc_mean", mean), ("loss", loss)):
        assert rel_err(val, z[f"train/{name}"]) < TOL, (name, float(val), float(z[f"train/{name}"]))
    loss.backward()
    cos = {}
    for name, p in m.named_parameters():
        if not p.requires_grad:
            continue
        assert p.grad is not None, name
        gflat = p.grad.detach().float().cpu().reshape(-1)
        if f"grad/{name}" in z.files:
            ref = torch.from_numpy(z[f"grad/{name}"]).reshape(-1)
        else:
            idx = torch.from_numpy(z[f"grad_idx/{name}"]); ref = torch.from_numpy(z[f"grad_val/{name}"]); gflat = gflat[idx]
        if float(ref.abs().max()) < 1e-7 or name.endswith("proj_l.bias") or name.endswith("attn.k.bias"):   # zero true gradient
            continue
        cos[name] = float(torch.dot(gflat, ref) / (gflat.norm() * ref.norm()).clamp_min(1e-30))
    report("micro_cait_grads", worst_cos=min(cos.values()))
    # measured on MI355X: worst cosine 0.99953 (profiles/r4_tol_report.jsonl); gate = 3 x the measured deficit
    bad = {k: v for k, v in cos.items() if v < 0.9986}
    assert not bad, f"gradient direction mismatch vs reference: {bad}"


def test_micro_cait_backbone_grads_vs_oracle():
    """Max-pool-free loss on the add-on tokens: every CaiT parameter gradient (incl. LayerScale, proj_l/proj_w, class attention)."""
    sd, cfg, z = micro("micro_cait.npz")
    m = _build_micro_cait(cfg, sd).train()
    img = torch.from_numpy(z["img"])
    g = torch.Generator().manual_seed(2)
    f, _, idx = m._tokens(img.cuda())
    w = torch.randn(f.shape, generator=g)
    (f * w.cuda()).sum().backward()
    params = {k: v.clone().requires_grad_(k not in O.FROZEN_KEYS) for k, v in sd.items()}
    out = O.ppnet_forward(params, img, cfg, train=True, force_idx=idx.cpu().long())
    fo = torch.cat([out["cls_tokens"], out["tokens"]], dim=1)
    assert rel_err(f, fo) < 5e-2
    (fo * w).sum().backward()
    rows = {}
    for name, p in m.named_parameters():
        if not p.requires_grad or params[name].grad is None:
            continue
        gm, gr = p.grad.float().cpu().reshape(-1), params[name].grad.reshape(-1)
        # proj_l.bias and the class-attention key bias sit in front of a softmax as per-row constants: their true gradient is
        # zero (eps-term aside), both sides carry rounding noise only
        if float(gr.abs().max()) < 1e-10 or name.endswith("proj_l.bias") or name.endswith("attn.k.bias"):
            continue
        rows[name] = (float((gm - gr).abs().max() / gr.abs().max()), float(torch.dot(gm, gr) / (gm.norm() * gr.norm()).clamp_min(1e-30)))
    assert len(rows) > 80
    report("micro_cait_backbone_grads", worst_cos=min(v[1] for v in rows.values()))
    bad = {k: v for k, v in rows.items() if v[1] < GATE_MICRO_BACKBONE}
    assert not bad, bad


def test_real_shape_cait_xxs24_train_step_vs_oracle():
    """cait_xxs24_224 architecture (24 talking-heads + 2 class-attention blocks, D=192), B=2: losses vs the fp32 CPU oracle following
    the same reservation, then every backbone gradient on a max-pool-free loss."""
    from protopformer_amd.protopformer import CrossEntropyLoss, construct_PPNet
    arch = "cait_xxs24_224"
    cfg = O.make_cfg(arch, 200, 64, 20, 1, 121, global_per_class=5)
    sd = O.init_state_dict(cfg, seed=3)
    g = torch.Generator().manual_seed(5)
    for k_ in sd:
        if k_.endswith(".bias") and "add_on" not in k_:
            sd[k_] = 0.05 * torch.randn(sd[k_].shape, generator=g)
        if "qkv.weight" in k_:
            sd[k_] = sd[k_] * 6.0                               # peaky attention => a well-separated top-k
    m = construct_PPNet(arch, pretrained=False, prototype_shape=(200, 64, 1, 1), num_classes=20, reserve_layers=[cfg["reserve_layer"]],
                        reserve_token_nums=[cfg["reserve_k"]], use_global=True, use_ppc_loss=True, global_proto_per_class=5,
                        add_on_layers_type="regular")
    m.load_state_dict(sd, strict=True)
    m = m.cuda().train()
    for blk in m.features.blocks:
        blk.drop_path_rate = 0.0
    img = torch.randn(2, 3, 224, 224, generator=g); label = torch.tensor([3, 17])
    logits, aux = m(img.cuda())
    ce = CrossEntropyLoss()(logits, label.cuda())
    cov, mean = m.get_PPC_loss(aux[2], aux[3], aux[4], label.cuda())
    (ce + 0.1 * cov + 0.5 * mean).backward()
    assert all(torch.isfinite(p.grad).all() for p in m.parameters() if p.requires_grad and p.grad is not None)
    my_idx = m._ppc_cache[1].cpu().long()
    with torch.no_grad():
        out = O.ppnet_forward(sd, img, cfg, train=True, force_idx=my_idx)
        _, parts = O.train_loss(out, label, cfg, with_ppc=True)
    # bf16 operands through 26 blocks: tolerances as in the DeiT real-shape test
    report("real_shape_cait_peaky", logits=rel_err(logits, out["logits"]), ce=rel_err(ce, parts["ce"]), cov=rel_err(cov, parts["ppc_cov"]), mean=rel_err(mean, parts["ppc_mean"]))
    # measured: logits 1.2e-4, CE 3.4e-5, PPC 1.8e-5 / 4.8e-5
    assert rel_err(logits, out["logits"]) < 4e-4
    assert_elementwise(logits, out["logits"], 4e-4, "cait logits, real shape")
    assert rel_err(ce, parts["ce"]) < 1e-4 and rel_err(cov, parts["ppc_cov"]) < 6e-5 and rel_err(mean, parts["ppc_mean"]) < 1.5e-4
    m.flat_store().zero_grad()
    f, _, idx = m._tokens(img.cuda())
    w = torch.randn(f.shape, generator=g)
    (f * w.cuda()).sum().backward()
    params = {k: v.clone().requires_grad_(k not in O.FROZEN_KEYS) for k, v in sd.items()}
    out = O.ppnet_forward(params, img, cfg, train=True, force_idx=idx.cpu().long())
    fo = torch.cat([out["cls_tokens"], out["tokens"]], dim=1)
    assert rel_err(f, fo) < 8e-2
    (fo * w).sum().backward()
    cosines = {}
    for name, p in m.named_parameters():
        if not p.requires_grad or params[name].grad is None:
            continue
        gm, gr = p.grad.float().cpu().reshape(-1), params[name].grad.reshape(-1)
        if float(gr.abs().max()) < 1e-10 or name.endswith("proj_l.bias") or name.endswith("attn.k.bias"):
            continue
        cosines[name] = float(torch.dot(gm, gr) / (gm.norm() * gr.norm()).clamp_min(1e-30))
    assert len(cosines) > 300
    report("real_shape_cait_backbone_grads", worst_cos=min(cosines.values()))
    bad = {k: v for k, v in cosines.items() if v < GATE_REAL_BACKBONE}
    assert not bad, bad
