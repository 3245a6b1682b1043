"""End-to-end GPU parity of the drop-in PPNet (HIP kernels) against the reference-generated micro fixtures and the oracle.

The backbone GEMMs run with bf16 operands / fp32 accumulation (BASELINE.json: "1xMI355X bf16"), so end-to-end numbers
carry bf16 rounding (~4e-3 per operand); tolerances below are stated per quantity.  Op-level kernels are held to 1e-3
against the oracle on identical inputs in the other test_gpu_* files; reserved-token indices are bit-exact given
identical fp32 cls_token_attn (test_gpu_head.py)."""
import numpy as np
import pytest
import torch

from helpers import assert_close, assert_elementwise, micro, rel_err, report
from oracle import ppf_oracle as O

pytestmark = pytest.mark.gpu


def _build(cfg, sd):
    from protopformer_amd.deit import MyVisionTransformer
    from protopformer_amd.protopformer import PPNet
    assert cfg["arch"] == "deit"
    feats = MyVisionTransformer(img_size=cfg["img"], patch_size=16, embed_dim=cfg["dim"], depth=cfg["depth"], num_heads=cfg["heads"],
                                drop_path_rate=0.0)
    m = PPNet(features=feats, img_size=cfg["img"], prototype_shape=[cfg["num_prototypes"], cfg["proto_dim"], 1, 1], proto_layer_rf_info=None,
              num_classes=cfg["num_classes"], reserve_layers=[cfg["reserve_layer"]], reserve_token_nums=[cfg["reserve_k"]], use_global=True,
              use_ppc_loss=True, ppc_cov_thresh=1., ppc_mean_thresh=2., global_coe=cfg["global_coe"],
              global_proto_per_class=cfg["global_per_class"], add_on_layers_type="regular")
    missing = m.load_state_dict(sd, strict=True)          # reference state-dict keys must match exactly
    return m.cuda()


def test_micro_deit_eval_train_against_reference_fixture():
    from protopformer_amd.protopformer import CrossEntropyLoss
    sd, cfg, z = micro("micro_deit.npz")
    m = _build(cfg, sd)
    img, label = torch.from_numpy(z["img"]).cuda(), torch.from_numpy(z["label"]).cuda()
    m.eval()
    logits, (cls_attn, dist, lg, ll) = m(img)
    assert dist.shape == z["eval/distances"].shape
    assert rel_err(cls_attn, z["eval/cls_token_attn"]) < 3e-3, rel_err(cls_attn, z["eval/cls_token_attn"])     # measured 8.1e-4 of its maximum
    ref_idx = torch.from_numpy(z["eval/cls_token_attn"]).topk(cfg["reserve_k"], dim=-1)[1].sort(dim=-1)[0]
    my_idx = cls_attn.cpu().topk(cfg["reserve_k"], dim=-1)[1].sort(dim=-1)[0]
    assert torch.equal(my_idx, ref_idx), "reserved tokens differ from the reference on the micro fixture"
    # bf16 MFMA operands through the blocks, fp32 everywhere else.  Measured on MI355X (gpurun_out/tol_report.jsonl, round 2):
    # logits 3.3e-4, global / local logits 6.7e-4 / 5.3e-4, distances 9.7e-4 (of the largest), push activations 1.2e-3;
    # gates at <= 3x the measurement (the north star's 1e-3 is met on the logits; the fp32 mode of test_gpu_precise.py holds
    # every quantity to 1e-3)
    TOL = 2e-3
    report("micro_deit_eval", logits=rel_err(logits, z["eval/logits"]), lg=rel_err(lg, z["eval/logits_global"]), ll=rel_err(ll, z["eval/logits_local"]),
           cls_attn=rel_err(cls_attn, z["eval/cls_token_attn"]), dist=rel_err(dist, z["eval/distances"]))
    assert rel_err(logits, z["eval/logits"]) < TOL, rel_err(logits, z["eval/logits"])
    assert_elementwise(logits, z["eval/logits"], TOL, "eval logits")
    # eval `distances` VALUES (not only the shape): bf16 tokens against fp32 prototypes, error relative to the largest distance
    assert rel_err(dist, z["eval/distances"]) < TOL, rel_err(dist, z["eval/distances"])
    assert rel_err(lg, z["eval/logits_global"]) < TOL and rel_err(ll, z["eval/logits_local"]) < TOL
    cls2, acts = m.push_forward(img)
    report("micro_deit_push", acts=rel_err(acts, z["eval/push_proto_acts"]))
    assert rel_err(acts, z["eval/push_proto_acts"]) < 4e-3

    m.train()
    crit = CrossEntropyLoss()
    logits, aux = m(img)
    assert aux[0] is None and aux[4] == 16 and aux[2].shape == z["train/total_proto_act"].shape
    ce = crit(logits, label)
    cov, mean = m.get_PPC_loss(aux[2], aux[3], aux[4], label)
    loss = ce + 0.1 * cov + 0.5 * mean
    report("micro_deit_train", **{name: rel_err(val, z[f"train/{name}"]) for name, val in (("ce", ce), ("ppc_cov", cov), ("ppc_mean", mean), ("loss", loss))})
    gates = {"ce": 6e-5, "ppc_cov": 4e-4, "ppc_mean": 2e-4, "loss": 1e-5}           # measured 1.8e-5 / 1.1e-4 / 6.6e-5 / 1.8e-6
    for name, val in (("ce", ce), ("ppc_cov", cov), ("ppc_mean", mean), ("loss", loss)):
        assert rel_err(val, z[f"train/{name}"]) < gates[name], (name, float(val), float(z[f"train/{name}"]))
    loss.backward()
    # Gradients vs the reference fixture: every tensor's cosine with the reference gradient, measured worst 0.99996 here.
    # (With peaky attention -- next test -- the max-pool's arg-max routing can flip on a near-tie and move a prototype's whole
    #  gradient to another token: cosine 0.94 for the affected tensors there.)
    cos = {}
    for name, p in m.named_parameters():
        if not p.requires_grad:
            continue
        assert p.grad is not None, name
        g = p.grad.detach().float().cpu().reshape(-1)
        if f"grad/{name}" in z.files:
            ref = torch.from_numpy(z[f"grad/{name}"]).reshape(-1)
        else:
            idx = torch.from_numpy(z[f"grad_idx/{name}"]); ref = torch.from_numpy(z[f"grad_val/{name}"]); g = g[idx]
        if float(ref.abs().max()) < 1e-7:
            continue
        cos[name] = float(torch.dot(g, ref) / (g.norm() * ref.norm()))
    report("micro_deit_grads", worst_cos=min(cos.values()))
    bad = {k: v for k, v in cos.items() if v < 0.9998}
    assert not bad, f"gradient direction mismatch vs reference: {bad}"


@pytest.mark.parametrize("shape", ["micro", "deit_small"])
def test_bottleneck_chain_matches_fp32_chain(shape):
    """backbone.addon_fwd / addon_bwd (first convolution on the bf16 MFMA GEMM, fp32 tail) against the all-fp32 chain of the verification mode
    on a bf16-representable input: outputs and saved activations to bf16 rounding; the backward evaluated with the SAME ReLU gates
    (the product's activations) to the rounding of its bf16 result -- parameter gradients of the tail included.
    'deit_small' (round 6): the REAL shape of the reference's default head on BASELINE's backbone -- D = 384 -> 192 -> 192 -> 96 -> 96 -> 64 -> 64
    (protopformer.py:90-107 with prototype_shape (2000, 64, 1, 1)) on the 128 x (1 + 81) = 10 496 rows of a batch-128 step."""
    from helpers import build_micro
    from protopformer_amd import backbone, ops, precise
    g = torch.Generator().manual_seed(0)
    if shape == "micro":
        sd, cfg, z = micro("micro_deit_bottleneck.npz")
        m = build_micro(cfg, sd).train()
        rows, D, widths = 40, 64, [(40, 32), (40, 32), (40, 16)]
    else:
        from protopformer_amd.protopformer import construct_PPNet
        torch.manual_seed(3)
        m = construct_PPNet("deit_small_patch16_224", pretrained=False, prototype_shape=(2000, 64, 1, 1), num_classes=200, reserve_layers=[11],
                            reserve_token_nums=[81], use_global=True, use_ppc_loss=True, global_proto_per_class=10,
                            add_on_layers_type="bottleneck").cuda().train()
        rows, D = 128 * 82, 384
        widths = [(rows, 192), (rows, 192), (rows, 96), (rows, 96), (rows, 64)]
        with torch.no_grad():                                  # biases off zero so that their gradients and the ReLU gates matter
            for c in backbone.addon_convs(m):
                c.bias.copy_(0.1 * torch.randn(c.bias.shape, generator=g))
    store = m.flat_store()
    store.refresh_bf16()
    nf16 = torch.randn(rows, D, generator=g).cuda().bfloat16()
    f_p, acts_p = backbone.addon_fwd(m, store, nf16)
    f_r, acts_r = precise._addon_fwd(m, nf16.float())
    # against the all-fp32 chain the difference is the bf16 rounding of the FIRST convolution's weight (2^-9 per element: 1.6e-3 of the
    # first activation's maximum at K = 384), carried through the tail: measured f 9.4e-3 at the real shape (five more layers), 1e-3 micro
    tol_f, tol_a = (2e-3, 4e-3) if shape == "micro" else (3e-2, 8e-3)
    assert rel_err(f_p, f_r) < tol_f and all(rel_err(a, b) < tol_a for a, b in zip(acts_p, acts_r)), (rel_err(f_p, f_r), [rel_err(a, b) for a, b in zip(acts_p, acts_r)])
    assert [tuple(a.shape) for a in acts_p] == widths
    # ... and against the SAME arithmetic in fp64 (first convolution on the bf16-rounded weight the MFMA GEMM reads, fp32 weights after it): what
    # is left is fp32 accumulation order -- the kernels themselves at 1e-4
    cv = backbone.addon_convs(m)
    w2 = lambda c_: c_.weight.detach().reshape(c_.out_channels, -1)
    h64 = torch.relu(nf16.double() @ w2(cv[0]).bfloat16().double().t() + cv[0].bias.detach().double())
    for j, c_ in enumerate(cv[1:]):
        assert rel_err(acts_p[j], h64.float()) < 1e-4, (j, rel_err(acts_p[j], h64.float()))
        h64 = h64 @ w2(c_).double().t() + c_.bias.detach().double()
        h64 = torch.sigmoid(h64) if j == len(cv) - 2 else torch.relu(h64)            # protopformer.py:90-107: ReLU after every convolution but the last
    assert rel_err(f_p, h64.float()) < 1e-4, rel_err(f_p, h64.float())
    df = torch.randn(f_p.shape, generator=g).cuda()
    convs = backbone.addon_convs(m)
    assert len(convs) == len(widths) + 1
    store.attach_all_grads()
    for c in convs:
        ops.zero_(store.grad_view(c.weight)); ops.zero_(store.grad_view(c.bias))
    dz = backbone.addon_bwd(m, store, dict(chain=acts_p), f_p, df)
    got = {i: (store.grad_view(c.weight).clone(), store.grad_view(c.bias).clone()) for i, c in enumerate(convs)}
    # the fp32 chain backward on the same activations (torch, fp64 accumulation as the referee of two fp32 summation orders over `rows`)
    at = 1e-5 if shape == "micro" else 1e-4
    d = (df * f_p * (1 - f_p)).double()
    for j in reversed(range(1, len(convs))):
        w = convs[j].weight.detach().reshape(convs[j].out_channels, -1).double()
        wg = d.t() @ acts_p[j - 1].double()
        assert_close(got[j][0].reshape(w.shape).double(), wg, rtol=1e-4, atol=at * float(wg.abs().max()), what=f"tail weight gradient {j}")
        assert_close(got[j][1].double(), d.sum(0), rtol=1e-4, atol=at * float(d.sum(0).abs().max()), what=f"tail bias gradient {j}")
        d = (d @ w) * (acts_p[j - 1] > 0)
    assert_close(got[0][1].double(), d.sum(0), rtol=1e-4, atol=at * float(d.sum(0).abs().max()), what="first bias gradient")
    assert_close(dz.double(), d, rtol=4e-3, atol=4e-3 * float(d.abs().max()), what="gradient w.r.t. the first pre-activation (bf16)")
    report(f"bottleneck_chain_{shape}", f=rel_err(f_p, f_r), dz=rel_err(dz.float(), d.float()))


def test_micro_deit_bottleneck_head_product_path_against_reference_fixture():
    """The reference's DEFAULT add-on head (add_on_layers_type='bottleneck', protopformer.py:90-107: 64 -> 32 -> 32 -> 16 -> 16 here) on the
    bf16 product path against the reference-generated fixture: reservation, logits, losses, every gradient's direction, and one optimizer
    step through the flat AdamW (the tail's eight tensors live in the add-on group, lr 3e-3)."""
    from protopformer_amd.engine import FlatAdamW
    from protopformer_amd.protopformer import CrossEntropyLoss
    from helpers import build_micro
    sd, cfg, z = micro("micro_deit_bottleneck.npz")
    assert cfg["add_on"] == "bottleneck"
    m = build_micro(cfg, sd)
    assert [k for k in m.state_dict() if k.startswith("add_on_layers.")] == [k for k in sd if k.startswith("add_on_layers.")]
    img, label = torch.from_numpy(z["img"]).cuda(), torch.from_numpy(z["label"]).cuda()
    m.eval()
    logits, (cls_attn, dist, lg, ll) = m(img)
    ref_idx = torch.from_numpy(z["eval/cls_token_attn"]).topk(cfg["reserve_k"], dim=-1)[1].sort(dim=-1)[0]
    assert torch.equal(cls_attn.cpu().topk(cfg["reserve_k"], dim=-1)[1].sort(dim=-1)[0], ref_idx)
    report("micro_bottleneck_eval", logits=rel_err(logits, z["eval/logits"]), dist=rel_err(dist, z["eval/distances"]))
    assert rel_err(logits, z["eval/logits"]) < 3e-3, rel_err(logits, z["eval/logits"])
    assert_elementwise(logits, z["eval/logits"], 3e-3, "eval logits (bottleneck head)")
    assert rel_err(dist, z["eval/distances"]) < 3e-3
    m.train()
    opt = FlatAdamW(m, weight_decay=0.05)
    logits, aux = m(img)
    ce = CrossEntropyLoss()(logits, label)
    cov, mean = m.get_PPC_loss(aux[2], aux[3], aux[4], label)
    loss = ce + 0.1 * cov + 0.5 * mean
    report("micro_bottleneck_train", loss=rel_err(loss, z["train/loss"]), ce=rel_err(ce, z["train/ce"]))
    assert rel_err(loss, z["train/loss"]) < 1e-3, (float(loss), float(z["train/loss"]))
    opt.zero_grad()
    loss.backward()
    cos = {}
    for name, p in m.named_parameters():
        if not p.requires_grad:
            continue
        assert p.grad is not None, name
        g = p.grad.detach().float().cpu().reshape(-1)
        if f"grad/{name}" in z.files:
            ref = torch.from_numpy(z[f"grad/{name}"]).reshape(-1)
        else:
            g = g[torch.from_numpy(z[f"grad_idx/{name}"])]; ref = torch.from_numpy(z[f"grad_val/{name}"])
        if float(ref.abs().max()) < 1e-7:
            continue
        cos[name] = float(torch.dot(g, ref) / (g.norm() * ref.norm()))
    assert all(f"add_on_layers.{i}.{t}" in cos for i in (0, 2, 4, 6) for t in ("weight", "bias"))
    # Everything downstream of the first ReLU layers agrees as on the 'regular' fixture.  Upstream of them the bf16 backbone perturbs the
    # pre-activations by ~1e-3, so a few of the 40 x 32 ReLU units (measured: 1-4) sit on the other side of zero than in the fp32
    # reference and gate a whole unit's gradient differently: per-tensor cosines 0.93-0.98 on this 40-row fixture (a property of bf16
    # activations in front of a ReLU, not of the kernels -- test_bottleneck_chain_matches_fp32_chain holds the chain itself, and the fp32
    # mode holds every gradient of this fixture to 1e-3, test_gpu_precise.py).
    report("micro_bottleneck_grads", worst_cos=min(cos.values()), worst_downstream=min(v for k, v in cos.items() if k.startswith(("prototype", "add_on_layers.4", "add_on_layers.6"))))
    for k, v in cos.items():
        floor = 0.9998 if k.startswith(("prototype", "add_on_layers.4", "add_on_layers.6")) else 0.85      # measured worst 0.905 (deterministic)
        assert v >= floor, f"gradient direction mismatch vs reference: {k} {v}"
    opt.step()
    torch.cuda.synchronize()
    # AdamW's first step is lr * sign(g) where |g| >> eps: compare where the reference gradient is well above rounding noise
    for i in (4, 6):
        name = f"add_on_layers.{i}.weight"
        p = dict(m.named_parameters())[name].detach().float().cpu()
        big = torch.from_numpy(z[f"grad/{name}"]).abs() > 1e-4
        assert_close(p[big], torch.from_numpy(z[f"step/{name}"])[big], rtol=1e-3, atol=2e-4, what=f"step/{name}")


# gradient-cosine floor of test_real_shape_train_step_vs_oracle (same reservation, same max-pool routing; measured value in its comments)
COS_FLOOR = 0.98          # measured 0.9921


def _grad_agreement(m, params):
    rows = {}
    for name, p in m.named_parameters():
        if not p.requires_grad or params[name].grad is None:
            continue
        gm, gr = p.grad.float().cpu().reshape(-1), params[name].grad.reshape(-1)
        if float(gr.abs().max()) < 1e-12:
            continue
        rows[name] = (float((gm - gr).abs().max() / gr.abs().max()), float(torch.dot(gm, gr) / (gm.norm() * gr.norm())))
    return rows


def test_real_shape_train_step_vs_oracle():
    """deit_tiny architecture, B=4: forward losses and a few gradients vs the fp32 CPU oracle with bf16-rounded weights."""
    from protopformer_amd.protopformer import CrossEntropyLoss, construct_PPNet
    torch.manual_seed(0)
    cfg = O.make_cfg("deit_tiny_patch16_224", 200, 64, 20, 11, 81, global_per_class=5)
    sd = O.init_state_dict(cfg, seed=3)
    g = torch.Generator().manual_seed(5)
    for k_ in sd:                                               # non-trivial LN / bias values
        if k_.endswith(".bias") and "add_on" not in k_:
            sd[k_] = 0.05 * torch.randn(sd[k_].shape, generator=g)
        if "qkv.weight" in k_:
            sd[k_] = sd[k_] * 6.0                               # peaky attention => a well-separated top-k
    m = construct_PPNet("deit_tiny_patch16_224", pretrained=False, prototype_shape=(200, 64, 1, 1), num_classes=20, reserve_layers=[11],
                        reserve_token_nums=[81], use_global=True, use_ppc_loss=True, global_proto_per_class=5, add_on_layers_type="regular")
    m.load_state_dict(sd, strict=True)
    m = m.cuda()
    for blk in m.features.blocks:
        blk.drop_path_rate = 0.0
    img = torch.randn(4, 3, 224, 224, generator=g); label = torch.tensor([3, 0, 19, 3])
    m.train()
    logits, aux = m(img.cuda())
    ce = CrossEntropyLoss()(logits, label.cuda())
    cov, mean = m.get_PPC_loss(aux[2], aux[3], aux[4], label.cuda())
    (ce + 0.1 * cov + 0.5 * mean).backward()
    params = {k: v.clone().requires_grad_(k not in O.FROZEN_KEYS) for k, v in sd.items()}
    my_idx = m._ppc_cache[1].cpu().long()
    with torch.no_grad():
        free = O.ppnet_forward(sd, img, cfg, train=True)
    assert_close(aux[3], free["cls_token_attn"], rtol=5e-2, atol=2e-4, what="cls rollout")
    # bf16 perturbs cls_token_attn by up to ~5e-2 relative: the reservation must agree with the fp32 oracle on every token
    # whose score is outside that band around the k-th value (exactness is only attainable at the kernel boundary)
    ref_attn = free["cls_token_attn"]
    kth = ref_attn.topk(81, dim=-1)[0][:, -1:]
    sel = torch.zeros_like(ref_attn, dtype=torch.bool).scatter_(1, my_idx, True)
    assert bool(sel[ref_attn > kth * 1.06].all()) and not bool(sel[ref_attn < kth * 0.94].any())
    n_diff = int((sel != torch.zeros_like(sel).scatter_(1, free["reserve_idx"], True)).sum()) // 2
    print(f"reserved tokens differing from the fp32 oracle: {n_diff} of {my_idx.numel()}")
    # downstream parity with the oracle following the SAME reservation and the SAME max-pool routing: a near-tied arg-max routes one
    # prototype's gradient through another token than in the fp32 run (a discontinuity of the reference's max_pool2d, not a kernel
    # error -- 0.94 cosine on EVERY tensor when left free), so the oracle's pooling gathers at the token the HIP run selected
    my_arg = m._last_argmax.cpu().long()
    out_free = O.ppnet_forward(sd, img, cfg, train=True, force_idx=my_idx)
    n_flip = int((out_free["total_proto_act"].flatten(2).argmax(-1) != my_arg).sum())
    out = O.ppnet_forward(params, img, cfg, train=True, force_idx=my_idx, force_argmax=my_arg)
    loss_ref, parts = O.train_loss(out, label, cfg, with_ppc=True)
    loss_ref.backward()
    rows = _grad_agreement(m, params)
    report("real_shape_tiny_peaky", logits=rel_err(logits, out["logits"]), ce=rel_err(ce, parts["ce"]), cov=rel_err(cov, parts["ppc_cov"]),
           mean=rel_err(mean, parts["ppc_mean"]), worst_cos=min(c for _, c in rows.values()), argmax_flips=n_flip, argmax_total=my_arg.numel())
    # measured (attention sharpened 6x on purpose, far peakier than any trained model): logits 1.5e-3 / 1.6e-3, CE 2.7e-4 / 2.5e-4, PPC cov
    # 7e-6 / 2.4e-5, PPC mean 3e-5 / 8.7e-5 with the 32-row two-pass / the 16-row one-launch attention forward (bf16 probabilities summed
    # in a different order); gates = 3 x the larger measurement
    assert rel_err(logits, out["logits"]) < 4.5e-3
    assert_elementwise(logits, out["logits"], 4.5e-3, "logits (peaky attention)")
    assert rel_err(ce, parts["ce"]) < 8e-4 and rel_err(cov, parts["ppc_cov"]) < 7.5e-5 and rel_err(mean, parts["ppc_mean"]) < 2.6e-4
    assert n_flip <= 0.06 * my_arg.numel(), (n_flip, my_arg.numel())           # the routing itself agrees on all but near-ties (measured 22 of 800)
    assert min(c for _, c in rows.values()) > COS_FLOOR, {k: v for k, v in rows.items() if v[1] <= COS_FLOOR}

    # backbone backward in isolation: L = sum(w * f) on the add-on tokens (no max-pool routing) -> every parameter gradient
    # must agree with the oracle's autograd up to bf16 operand rounding accumulated over 12 layers
    m.flat_store().zero_grad()
    f, _, idx = m._tokens(img.cuda())
    w = torch.randn(f.shape, generator=g)
    (f * w.cuda()).sum().backward()
    params = {k: v.clone().requires_grad_(k not in O.FROZEN_KEYS) for k, v in sd.items()}
    out = O.ppnet_forward(params, img, cfg, train=True, force_idx=idx.cpu().long())
    fo = torch.cat([out["cls_tokens"], out["tokens"]], dim=1)
    (fo * w).sum().backward()
    rows = _grad_agreement(m, params)
    report("real_shape_tiny_peaky_backbone", f=rel_err(f, fo), worst_cos=min(c for _, c in rows.values()), median_rel=sorted(r for r, _ in rows.values())[len(rows) // 2],
           worst_rel=max(r for r, _ in rows.values()))
    assert rel_err(f, fo) < 8e-2                              # measured 4.4e-2 (sigmoid outputs after 12 peaky bf16 layers)
    assert len(rows) > 140
    rels = sorted(r for r, _ in rows.values())
    worst_cos = min(c for _, c in rows.values())
    # measured: cosine >= 0.997 for every tensor, rel-to-max error median 3e-2, worst 0.2 (bf16 drift over 12 peaky layers)
    assert worst_cos > 0.992 and rels[len(rels) // 2] < 6e-2 and rels[-1] < 0.4, (rels[len(rels) // 2], rels[-1], worst_cos)


@pytest.mark.parametrize("payload", ["fp32", "bf16"])
def test_two_rank_data_parallel_on_one_gpu(payload):
    """GradSync + the weight-gradient lane with two real processes (gloo over one GPU; RCCL refuses duplicate devices):
    identical parameters on both ranks after three steps, and averaged per-rank gradients == full-batch gradients.
    payload bf16 (round 6, PPF_GRADSYNC_BF16=1): the chunks travel as bf16 -- the replicas must STILL be bit-identical (every rank widens the
    same all-reduced bf16 values), the averaged gradient agrees with the full-batch one to the wire format's rounding."""
    import os, re, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29591" if payload == "fp32" else "29593", os.path.join(root, "scripts", "gpu", "ddp_gloo_check.py")]
    env = dict(os.environ, PPF_GRADSYNC_BF16="1" if payload == "bf16" else "0")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=root, env=env)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-3000:]
    m = re.search(r"ranks identical after 3 steps: (\w+); loss ([0-9.]+); cos\(.*\) = ([0-9.]+)", out)
    assert m, out[-3000:]
    assert m.group(1) == "True" and float(m.group(3)) > 0.9999


def test_single_rank_nccl_gradsync_is_bit_identical():
    """The RCCL path on the hardware the suite runs on: one rank, backend nccl, PPF_FORCE_GRADSYNC=1 -> the chunked all-reduce really
    runs (communication stream behind both compute streams) and three train steps equal three steps without any exchange, bit for bit
    (scripts/gpu/nccl_single_rank_check.py, child process: the process group must not leak into the other tests)."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29577", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "gpu", "nccl_single_rank_check.py")], capture_output=True, text=True,
                       timeout=900, cwd=root, env=env)
    line = [l for l in r.stdout.splitlines() if l.startswith("NCCL_SINGLE_RANK ")]
    assert r.returncode == 0 and line, (r.returncode, r.stdout[-2000:], r.stderr[-3000:])
    out = json.loads(line[0][len("NCCL_SINGLE_RANK "):])
    assert out["backend"] == "nccl" and out["world"] == 1
    assert out["collectives"] >= 3 * 3, out                      # >= 3 chunks per step really went through RCCL
    assert out["params_equal"] and out["moments_equal"] and out["ema_equal"] and out["losses_equal"], out
    # ... and the recorded / replayed step (bench.py's default) issues its collectives live on every replay and stays bit-identical
    assert out["replay_collectives"] >= 4 * 3 and out["replay_params_equal"] and out["replay_losses_equal"], out


def test_reserved_token_compaction_matches_masked_blocks(monkeypatch):
    """From the reservation layer on, the blocks run on the reserved rows only (backbone.deit_blocks_fwd); the masked full-length
    computation of the reference (PPF_COMPACT_RESERVED=0) must give the same tokens, logits and gradients up to bf16 rounding of the
    attention probabilities (the two softmaxes subtract different row maxima)."""
    from protopformer_amd.protopformer import CrossEntropyLoss, construct_PPNet
    torch.manual_seed(0)
    g = torch.Generator().manual_seed(11)
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("PPF_COMPACT_RESERVED", mode)
        torch.manual_seed(3)
        m = construct_PPNet("deit_tiny_patch16_224", pretrained=False, prototype_shape=(200, 64, 1, 1), num_classes=20, reserve_layers=[10],
                            reserve_token_nums=[81], use_global=True, use_ppc_loss=True, global_proto_per_class=5, add_on_layers_type="regular").cuda().train()
        for blk in m.features.blocks:
            blk.drop_path_rate = 0.0
        if "img" not in res:
            res["img"] = torch.randn(4, 3, 224, 224, generator=g).cuda(); res["label"] = torch.tensor([3, 0, 19, 7]).cuda()
        f, cls_attn, idx = m._tokens(res["img"])
        logits, aux = m(res["img"])
        ce = CrossEntropyLoss()(logits, res["label"])
        cov, mean = m.get_PPC_loss(aux[2], aux[3], aux[4], res["label"])
        (ce + 0.1 * cov + 0.5 * mean).backward()
        res[mode] = dict(f=f.detach().clone(), idx=idx.clone(), logits=logits.detach().clone(), loss=float(ce.detach()),
                         grads=m.flat_store().grads.clone())
    a, b = res["1"], res["0"]
    report("compact_vs_masked_tiny", f=rel_err(a["f"], b["f"]), logits=rel_err(a["logits"], b["logits"]), loss=abs(a["loss"] - b["loss"]) / abs(b["loss"]),
           cos=float(torch.dot(a["grads"], b["grads"]) / (a["grads"].norm() * b["grads"].norm())))
    assert torch.equal(a["idx"], b["idx"])                      # the reservation itself happens before the compaction
    # measured: tokens 2.7e-3, logits 1.3e-4, loss 3.8e-6, gradient cosine 0.99995
    assert rel_err(a["f"], b["f"]) < 8e-3 and rel_err(a["logits"], b["logits"]) < 4e-4 and abs(a["loss"] - b["loss"]) < 2e-5 * abs(b["loss"])
    cos = float(torch.dot(a["grads"], b["grads"]) / (a["grads"].norm() * b["grads"].norm()))
    assert cos > 0.9998, cos
