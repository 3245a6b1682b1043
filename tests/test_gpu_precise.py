"""The fp32 verification mode (PPNet.precise, protopformer_amd/precise.py + csrc/precise.hip) against the REFERENCE-generated
fixtures at the north-star tolerance: 1e-3 relative on logits / losses / activations, bit-exact reserved-token indices.

The product path computes its GEMMs / attention with bf16 operands (gated at bf16-level tolerances in test_gpu_e2e.py /
test_gpu_baseline_configs.py); this mode runs the same orchestration, rollout, reservation, prototype, PPC and CE kernels with
fp32 operands everywhere, so any disagreement beyond 1e-3 is a kernel or orchestration bug, not rounding."""
import pytest
import torch

from helpers import assert_close, assert_elementwise, build_micro, micro, rel_err
from oracle import ppf_oracle as O

pytestmark = pytest.mark.gpu

TOL = 1e-3          # BASELINE.json north_star: "within 1e-3 rel fp32"


@pytest.mark.parametrize("name", ["micro_deit.npz", "micro_cait.npz", "micro_deit_bottleneck.npz"])
def test_precise_forward_and_loss_match_reference_fixture(name):
    from protopformer_amd.protopformer import CrossEntropyLoss
    sd, cfg, z = micro(name)
    m = build_micro(cfg, sd)
    m.precise = True
    img, label = torch.from_numpy(z["img"]).cuda(), torch.from_numpy(z["label"]).cuda()
    with torch.no_grad():
        m.eval()
        logits, (cls_attn, dist, lg, ll) = m(img)
        assert_close(cls_attn, z["eval/cls_token_attn"], rtol=TOL, atol=1e-7, what="cls_token_attn")
        ref_idx = torch.from_numpy(z["eval/cls_token_attn"]).topk(cfg["reserve_k"], dim=-1)[1].sort(dim=-1)[0]
        assert torch.equal(m._tokens(img)[2].cpu().long(), ref_idx), "reserved-token indices must be bit-exact"
        assert rel_err(logits, z["eval/logits"]) < TOL, rel_err(logits, z["eval/logits"])
        assert_elementwise(logits, z["eval/logits"], TOL, "fp32 mode eval logits")
        assert rel_err(lg, z["eval/logits_global"]) < TOL and rel_err(ll, z["eval/logits_local"]) < TOL
        # distances: d = x2 - 2xp + p2 cancels, so compare on the scale of its terms (SURVEY 8(c) tolerances)
        d_ref = torch.from_numpy(z["eval/distances"])
        assert_close(dist, d_ref, rtol=TOL, atol=1e-5 * cfg["proto_dim"], what="eval distances")
        _, acts = m.push_forward(img)
        a_ref = torch.from_numpy(z["eval/push_proto_acts"])
        far = d_ref.reshape(a_ref.shape) >= 0.05                        # log-similarity is steep near d = 0
        assert_close(acts.cpu()[far], a_ref[far], rtol=TOL, atol=1e-5, what="push_forward activations (d >= 0.05)")
        m.train()
        logits, aux = m(img)
        assert aux[0] is None and aux[4] == 16
        assert rel_err(logits, z["train/logits"]) < TOL
        assert_elementwise(logits, z["train/logits"], TOL, "fp32 mode train logits")
        assert_close(aux[3], z["train/cls_attn_rollout"], rtol=TOL, atol=1e-7, what="cls_attn_rollout")
        ce = CrossEntropyLoss()(logits, label)
        cov, mean = m.get_PPC_loss(aux[2], aux[3], aux[4], label)
        loss = ce + 0.1 * cov + 0.5 * mean
        for nm, val in (("ce", ce), ("ppc_cov", cov), ("ppc_mean", mean), ("loss", loss)):
            assert rel_err(val, z[f"train/{nm}"]) < TOL, (nm, float(val), float(z[f"train/{nm}"]))


@pytest.mark.parametrize("name", ["micro_deit.npz", "micro_cait.npz", "micro_deit_bottleneck.npz"])
def test_precise_backward_matches_reference_gradients(name):
    """fp32 forward + fp32 backward of the DeiT / CaiT micro models against grad/* of the reference-generated fixture (autograd of the
    reference itself): EVERY parameter gradient within 1e-3 of its tensor's scale, element by element.  The bf16 step of the product
    path is gated on gradient direction (cosine, test_gpu_e2e.py); this holds the orchestration of the backward -- LayerNorm / GELU /
    policy-softmax / talking-heads / class-attention / LayerScale derivatives, the residual and DropPath routing, the reserved-row scatter, the prototype / PPC / CE gradients that
    the product path shares -- to the north-star tolerance."""
    from protopformer_amd.protopformer import CrossEntropyLoss
    sd, cfg, z = micro(name)
    m = build_micro(cfg, sd)
    m.precise = True
    m.train()
    img, label = torch.from_numpy(z["img"]).cuda(), torch.from_numpy(z["label"]).cuda()
    logits, aux = m(img)
    ce = CrossEntropyLoss()(logits, label)
    cov, mean = m.get_PPC_loss(aux[2], aux[3], aux[4], label)
    loss = ce + 0.1 * cov + 0.5 * mean
    assert rel_err(loss, z["train/loss"]) < TOL
    loss.backward()
    torch.cuda.synchronize()
    worst, n = {}, 0
    for name, p in m.named_parameters():
        if not p.requires_grad:
            continue
        assert p.grad is not None, name
        g = p.grad.detach().float().cpu().reshape(-1)
        if f"grad/{name}" in z.files:
            ref = torch.from_numpy(z[f"grad/{name}"]).reshape(-1)
        else:
            g = g[torch.from_numpy(z[f"grad_idx/{name}"])]
            ref = torch.from_numpy(z[f"grad_val/{name}"])
        if float(ref.abs().max()) < 1e-7:                 # mathematically zero (the key bias of a softmax): absolute check
            assert float(g.abs().max()) < 1e-6, name
            continue
        worst[name] = rel_err(g, ref)
        assert_elementwise(g, ref, TOL, f"grad/{name}")
        n += 1
    assert n >= 20, n
    print("fp32 backward: worst per-tensor rel err", max(worst.items(), key=lambda kv: kv[1]))
@pytest.mark.parametrize("arch,k,layer,C,gpc,P,Dp", [("deit_small_patch16_224", 81, 11, 200, 10, 2000, 384),
                                                      ("cait_xxs24_224", 121, 1, 196, 5, 1960, 192)])
def test_precise_baseline_heads_vs_oracle(arch, k, layer, C, gpc, P, Dp):
    """BASELINE configs 3 and 5 with their real heads (2000x384 / 1960x192 prototypes) at B=2: the fp32 mode must reproduce the
    CPU oracle's reservation exactly and its logits / CE / PPC terms within 1e-3."""
    from protopformer_amd.protopformer import CrossEntropyLoss, construct_PPNet
    cfg = O.make_cfg(arch, P, Dp, C, layer, k, global_per_class=gpc)
    sd = O.init_state_dict(cfg, seed=11)
    g = torch.Generator().manual_seed(12)
    for k_ in sd:
        # (the talking-heads mixers keep their zero biases: a +-0.05 proj_w bias against probabilities of 1/196 makes the attention
        #  negative and the rollout's row normalisation ill-conditioned -- a property of that input, not of either implementation)
        if k_.endswith(".bias") and "add_on" not in k_ and "proj_l" not in k_ and "proj_w" not in k_:
            sd[k_] = 0.05 * torch.randn(sd[k_].shape, generator=g)
        if "qkv.weight" in k_:
            sd[k_] = sd[k_] * 6.0                                       # peaky attention => a well-separated top-k
        if "gamma_" in k_:
            sd[k_] = 0.1 + 0.05 * torch.rand(sd[k_].shape, generator=g)  # LayerScale large enough to matter
    m = construct_PPNet(arch, pretrained=False, prototype_shape=(P, Dp, 1, 1), num_classes=C, reserve_layers=[layer], reserve_token_nums=[k],
                        use_global=True, use_ppc_loss=True, global_proto_per_class=gpc, add_on_layers_type="regular")
    m.load_state_dict(sd, strict=True)
    m = m.cuda().train()
    m.precise = True
    for blk in m.features.blocks:
        blk.drop_path_rate = 0.0
    label = torch.tensor([3, C - 1])
    # tie-free at the top-k boundary BY CONSTRUCTION (as tests/golden/make_golden.py does for the fixtures): of 32 drawn images take the one whose
    # k-th and (k+1)-th largest cls_token_attn entries (oracle) are separated most (> 2e-5 of the map's maximum required): the index assert is unconditional
    with torch.no_grad():
        best = None
        for _try in range(32):                                    # the best-separated of 32 draws (0.1-0.2 s of oracle each)
            img_t = torch.randn(2, 3, 224, 224, generator=g)
            out_t = O.ppnet_forward(sd, img_t, cfg, train=True)
            srt = out_t["cls_token_attn"].sort(dim=-1, descending=True)[0]
            gap_t = float((srt[:, k - 1] - srt[:, k]).min()) / float(srt.max())
            if best is None or gap_t > best[0]:
                best = (gap_t, img_t, out_t)
        gap, img, out = best
        # (fp32 mode vs oracle agree to ~1e-6 per kernel; CaiT's boundary sits in the dense part of the map -- 121 of 196 tokens -- so its
        #  best gap is ~5e-5, DeiT's ~1e-3)
        assert gap > 2e-5, f"no tie-free input in 32 draws (best boundary gap {gap:.2e} of the maximum)"
        logits, aux = m(img.cuda())
        ce = CrossEntropyLoss()(logits, label.cuda())
        cov, mean = m.get_PPC_loss(aux[2], aux[3], aux[4], label.cuda())
        _, parts = O.train_loss(out, label, cfg, with_ppc=True)
    # the rollout zeroes the 90 % smallest entries of every layer's map: an entry within fp32 rounding of that threshold is kept by one
    # implementation and dropped by the other, so after 11 / 25 layers a handful of outputs differ at the 1e-3 level (relative to the
    # row maximum) although every kernel is exact to 1e-6 -- gate the map at 5e-3 of its maximum, everything downstream at 1e-3
    assert rel_err(aux[3], out["cls_token_attn"]) < 5e-3, rel_err(aux[3], out["cls_token_attn"])
    assert torch.equal(m._ppc_cache[1].cpu().long(), out["reserve_idx"])            # bit-exact reservation (north_star), unconditional
    assert rel_err(logits, out["logits"]) < TOL, rel_err(logits, out["logits"])
    assert_elementwise(logits, out["logits"], TOL, "fp32 mode logits, real head")
    assert rel_err(ce, parts["ce"]) < TOL and rel_err(cov, parts["ppc_cov"]) < TOL and rel_err(mean, parts["ppc_mean"]) < TOL
    far = out["distances"] >= 0.05
    assert_close(aux[2].cpu()[far], out["total_proto_act"][far], rtol=TOL, atol=1e-5, what="total_proto_act (d >= 0.05)")


def test_single_reserved_token_trains():
    """reserve_token_nums=[1] (k = 1 passes the reference's perfect-square assert): the pooled branch degenerates to ONE token, whose backward
    reads the distance map -- round-5 advice: ProtoLayerFn saved only the activation map and this configuration stopped training.  fp32 mode
    forward + backward against autograd of the oracle at 1e-3 on the micro DeiT weights."""
    from protopformer_amd.protopformer import CrossEntropyLoss
    sd, cfg, z = micro("micro_deit.npz")
    cfg = dict(cfg, reserve_k=1)
    m = build_micro(cfg, sd)
    m.precise = True
    m.train()
    img, label = torch.from_numpy(z["img"]), torch.from_numpy(z["label"])
    logits, aux = m(img.cuda())
    ce = CrossEntropyLoss()(logits, label.cuda())
    cov, mean = m.get_PPC_loss(aux[2], aux[3], aux[4], label.cuda())
    (ce + 0.1 * cov + 0.5 * mean).backward()
    params = {k: v.clone().requires_grad_(k not in O.FROZEN_KEYS) for k, v in sd.items()}
    out = O.ppnet_forward(params, img, cfg, train=True)
    assert torch.equal(m._ppc_cache[1].cpu().long(), out["reserve_idx"])
    loss_ref, parts = O.train_loss(out, label, cfg, with_ppc=True)
    (parts["ce"] + 0.1 * parts["ppc_cov"] + 0.5 * parts["ppc_mean"]).backward()
    assert rel_err(logits, out["logits"]) < TOL and rel_err(ce, parts["ce"]) < TOL
    n = 0
    for name, p in m.named_parameters():
        ref = params[name].grad
        if not p.requires_grad or ref is None or float(ref.abs().max()) < 1e-7:
            continue
        assert p.grad is not None, name
        assert_elementwise(p.grad.detach().float().cpu().reshape(-1), ref.reshape(-1), TOL, f"k=1 grad/{name}")
        n += 1
    assert n >= 20, n
