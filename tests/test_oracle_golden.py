"""Pins oracle/ppf_oracle.py against the reference's own outputs (tests/golden/*.npz). CPU only."""
import numpy as np
import pytest
import torch

import golden_inputs as gi
from helpers import assert_close, check_param_tensors, load_npz, micro
from oracle import ppf_oracle as O

torch.set_num_threads(4)
TOL = dict(rtol=1e-4, atol=1e-5)     # oracle vs reference, fp32 CPU both (observed ~1e-6)


@pytest.mark.parametrize("name", ["micro_deit.npz", "micro_cait.npz", "micro_deit_bottleneck.npz"])
def test_micro_eval_and_train_forward(name):
    sd, cfg, z = micro(name)
    img, label = torch.from_numpy(z["img"]), torch.from_numpy(z["label"])
    with torch.no_grad():
        out = O.ppnet_forward(sd, img, cfg, train=False)
    assert_close(out["logits"], z["eval/logits"], what="eval logits", **TOL)
    assert_close(out["cls_token_attn"], z["eval/cls_token_attn"], what="cls_token_attn", rtol=1e-4, atol=1e-7)
    assert_close(out["logits_global"], z["eval/logits_global"], what="logits_global", **TOL)
    assert_close(out["logits_local"], z["eval/logits_local"], what="logits_local", **TOL)
    assert_close(out["distances"], z["eval/distances"], what="distances", rtol=1e-4, atol=2e-5)
    assert_close(out["total_proto_act"], z["eval/push_proto_acts"], what="push proto acts", rtol=2e-4, atol=2e-4)
    # train branch returns the same arithmetic (DropPath frozen)
    assert_close(out["logits"], z["train/logits"], what="train logits", **TOL)
    assert_close(out["total_proto_act"], z["train/total_proto_act"], what="total_proto_act", rtol=2e-4, atol=2e-4)
    assert_close(out["cls_token_attn"], z["train/cls_attn_rollout"], what="cls_attn_rollout", rtol=1e-4, atol=1e-7)
    # exact reserved-token indices
    ref_idx = torch.from_numpy(z["eval/cls_token_attn"]).topk(cfg["reserve_k"], dim=-1)[1].sort(dim=-1)[0]
    assert torch.equal(out["reserve_idx"], ref_idx)


@pytest.mark.parametrize("name", ["micro_deit.npz", "micro_cait.npz", "micro_deit_bottleneck.npz"])
def test_micro_loss_grads_and_adamw_step(name):
    sd, cfg, z = micro(name)
    img, label = torch.from_numpy(z["img"]), torch.from_numpy(z["label"])
    params = {k: (v.clone().requires_grad_(k not in O.FROZEN_KEYS)) for k, v in sd.items()}
    out = O.ppnet_forward(params, img, cfg, train=True)
    loss, parts = O.train_loss(out, label, cfg, with_ppc=True)
    assert_close(parts["ce"], z["train/ce"], what="ce", **TOL)
    assert_close(parts["ppc_cov"], z["train/ppc_cov"], what="ppc_cov", **TOL)
    assert_close(parts["ppc_mean"], z["train/ppc_mean"], what="ppc_mean", **TOL)
    assert_close(loss, z["train/loss"], what="loss", **TOL)
    opt = torch.optim.AdamW(O.adamw_groups(params), weight_decay=0.05, eps=1e-8)
    opt.zero_grad()
    loss.backward()
    trainable = {k: v for k, v in params.items() if v.requires_grad}
    grads = {k: v.grad for k, v in trainable.items()}
    assert all(g is not None for g in grads.values())
    n = check_param_tensors(z, "grad", grads, rtol=2e-3, atol=2e-6)
    assert n == len(trainable)
    opt.step()
    check_param_tensors(z, "step", trainable, rtol=1e-5, atol=2e-6, grad_floor=1e-5)


def test_op_attention_real_shape():
    z = load_npz("ops_real.npz")
    a = gi.attn_inputs()
    sd = {"a." + k: v for k, v in a["w"].items()}
    for tag, pol in (("ones", a["policy_ones"]), ("topk", a["policy_topk"])):
        y, p = O.deit_attention(sd, "a.", a["x"], a["H"], pol)
        assert_close(y, z[f"attn/out_{tag}"], what=f"attn out {tag}", rtol=1e-4, atol=1e-5)
        idx = torch.from_numpy(z[f"attn/probs_{tag}_idx"])
        assert_close(p.reshape(-1)[idx], z[f"attn/probs_{tag}_val"], what=f"probs {tag}", rtol=1e-4, atol=1e-9)
        assert abs(float(p.double().sum()) - float(z[f"attn/probs_{tag}_sum"])) < 1e-3


def test_op_rollout_real_shape():
    z = load_npz("ops_real.npz")
    R = O.deit_rollout(gi.rollout_inputs())
    cls_attn = R[:, 0, 1:]
    assert_close(cls_attn, z["rollout/cls_token_attn"], what="cls_token_attn", rtol=1e-4, atol=1e-8)
    assert_close(R[:, 5], z["rollout/R_row5"], what="rollout row 5", rtol=1e-4, atol=1e-8)
    assert np.array_equal(O.topk_sorted(cls_attn, 81).numpy(), z["rollout/idx"])


def test_op_prototype_layer_real_shape():
    z = load_npz("ops_real.npz")
    tok, protos = gi.proto_inputs()
    B, Dp = tok.shape[0], tok.shape[1]
    tokens = tok.flatten(2).transpose(1, 2)                      # (B,81,Dp)
    act_max, dist, act = O.proto_activations(tokens, protos)
    # distances compared with abs 1e-6*(x2+p2) (~1e-4 at Dp=192) or rel 1e-3
    idx = torch.from_numpy(z["proto/dist_idx"])
    assert_close(dist.reshape(-1)[idx], z["proto/dist_val"], what="distances", rtol=1e-3, atol=2e-4)
    assert abs(float(dist.double().sum()) - float(z["proto/dist_sum"])) < 1e-6 * float(z["proto/dist_sum"]) + 1.0
    assert float(dist[1, 7].reshape(-1)[0]) == 0.0 and float(z["proto/dist_b1_p7"].reshape(-1)[0]) == 0.0
    assert float(dist[0, 5].reshape(9, 9)[2, 3]) < 1e-4
    ref_max = torch.from_numpy(z["proto/act_max"])
    far = ref_max < 2.9                                          # act < 2.9  <=>  d > 0.05: steep region excluded
    assert_close(act_max[far], ref_max[far], what="act_max (d>=0.05)", rtol=1e-3)
    assert_close(act[0, :20].reshape(20, 9, 9), z["proto/act_b0_p0_20"], what="act sample", rtol=1e-3, atol=1e-3)


def test_op_ppc_loss_real_shape():
    z = load_npz("ops_real.npz")
    tpa, roll, lab = gi.ppc_inputs()
    cov, mean = O.ppc_loss(tpa, roll, 196, lab, 10, 1.0, 2.0)
    assert_close(cov, z["ppc/cov"], what="ppc cov", **TOL)
    assert_close(mean, z["ppc/mean"], what="ppc mean", **TOL)


def test_op_cait_real_shape():
    z = load_npz("ops_real.npz")
    c = gi.cait_inputs()
    y, p = O.cait_talking_heads_attention({"t." + k: v for k, v in c["th"].items()}, "t.", c["x"], c["H"])
    assert_close(y, z["cait/th_out"], what="talking heads out", rtol=1e-4, atol=1e-5)
    idx = torch.from_numpy(z["cait/th_attn_idx"])
    assert_close(p.reshape(-1)[idx], z["cait/th_attn_val"], what="talking heads attn", rtol=1e-4, atol=1e-7)
    sdc = {"c." + k: v for k, v in c["ca"].items()}
    y, p = O.cait_class_attention(sdc, "c.", c["u"], c["H"], c["policy"])
    assert_close(y, z["cait/ca_out"], what="class attn out", rtol=1e-4, atol=1e-5)
    assert_close(p, z["cait/ca_attn"], what="class attn probs", rtol=1e-4, atol=1e-10)
    y, p = O.cait_class_attention(sdc, "c.", c["u"], c["H"], torch.ones(c["B"], c["N"] + 1))
    assert_close(y, z["cait/ca_out_ones"], what="class attn out (ones)", rtol=1e-4, atol=1e-5)
    res = O.cait_rollout(c["sa"], c["cas"])
    assert_close(res[:, 0], z["cait/rollout_cls"], what="cait rollout", rtol=1e-4, atol=1e-8)


# ---- known-answer tests that need no fixture -------------------------------------
def test_kat_token_equals_prototype():
    tok = torch.full((1, 4, 8), 0.5)
    protos = torch.full((3, 8, 1, 1), 0.5)
    mx, d, act = O.proto_activations(tok, protos)
    assert float(d.abs().max()) == 0.0
    assert abs(float(mx[0, 0]) - 9.21034) < 1e-4


def test_kat_policy_softmax():
    g = torch.Generator().manual_seed(0)
    s = torch.randn(2, 3, 7, 7, generator=g)
    p = O.policy_softmax(s, torch.ones(2, 7), self_keep=True)
    assert float((p - s.softmax(-1)).abs().max()) < 1e-6   # eps/N with N=7
    pol = torch.zeros(2, 7); pol[:, 0] = 1
    p = O.policy_softmax(s, pol, self_keep=True)
    mass = p[..., 0] + torch.diagonal(p, dim1=-2, dim2=-1)
    mass[..., 0] = p[..., 0, 0]
    assert float((mass - 1).abs().max()) < 1e-5


def test_kat_grid_moments_and_ppc():
    mean, cov = O.weighted_grid_moments(torch.ones(1, 196), 14)
    assert torch.allclose(mean, torch.full((1, 2), 6.5))
    assert abs(float(cov[0, 0, 0]) - 16.25 * 196 / 195) < 1e-4 and abs(float(cov[0, 0, 1])) < 1e-5
    act = torch.ones(2, 200, 14, 14)
    g = torch.Generator().manual_seed(1)
    cov_l, mean_l = O.ppc_loss(act, torch.rand(2, 196, generator=g), 196, torch.tensor([0, 5]), 10, 1.0, 2.0)
    assert abs(float(cov_l) - (16.25 * 196 / 195 - 1)) < 1e-4
    assert abs(float(mean_l) - 1.8) < 1e-5


def test_kat_last_layer_pattern():
    cfg = O.make_cfg("deit_tiny_patch16_224", 2000, 192, 200, 11, 81)
    cfg.update(depth=1)
    sd = O.init_state_dict(cfg, seed=0)
    w = sd["last_layer.weight"]
    assert w.shape == (200, 2000)
    assert float(w[3, 30:40].min()) == 1.0 and float(w[3, :30].max()) == -0.5 and float(w[3, 40:].max()) == -0.5
