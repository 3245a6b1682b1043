"""GPU parity of rollout / prototype layer / PPC / CE / frozen-head kernels vs the CPU oracle and the golden vectors."""
import numpy as np
import pytest
import torch

import golden_inputs as gi
from helpers import assert_close, load_npz
from oracle import ppf_oracle as O

pytestmark = pytest.mark.gpu


def _pad_hm(fused):
    """[L,B,N,N] -> kernel layout [L,B,N,NP] with NP = N rounded up to 4."""
    L, B, N, _ = fused.shape
    NP = (N + 3) // 4 * 4
    out = torch.zeros(L, B, N, NP)
    out[..., :N] = fused
    return out.cuda()


def _check_topk(idx, ref, k):
    """Tie-robust top-k check (torch.topk's tie order is implementation-defined): every index strictly
    above the k-th value must be selected, none strictly below it, indices ascending and unique."""
    kth = ref.topk(k, dim=-1)[0][:, -1:]
    sel = torch.zeros_like(ref, dtype=torch.bool)
    sel.scatter_(1, idx, True)
    assert int(sel.sum()) == idx.numel(), "duplicate indices"
    assert bool((idx[:, 1:] > idx[:, :-1]).all()), "indices must be ascending"
    assert bool(sel[ref > kth].all()) and not bool(sel[ref < kth].any())


def test_rollout_deit_golden():
    from protopformer_amd import ops
    z = load_npz("ops_real.npz")
    probs = gi.rollout_inputs()
    fused = torch.stack([p.mean(1) for p in probs])                      # head fusion is the attention kernel's job
    cls_attn, idx, policy = ops.rollout(_pad_hm(fused), 11, 2, 197, 81, lead=1)
    assert_close(cls_attn, z["rollout/cls_token_attn"], rtol=1e-3, atol=1e-8, what="cls_token_attn vs reference")
    assert np.array_equal(idx.cpu().numpy(), z["rollout/idx"]), "reserved-token indices must be bit-exact"
    pol = torch.zeros(2, 197); pol[:, 0] = 1
    pol.scatter_(1, torch.from_numpy(z["rollout/idx"]).long() + 1, 1.0)
    assert torch.equal(policy.cpu(), pol)


@pytest.mark.parametrize("B,N,L,k", [(3, 17, 2, 9), (2, 65, 3, 16), (5, 197, 1, 81)])
def test_rollout_deit_vs_oracle(B, N, L, k):
    from protopformer_amd import ops
    g = torch.Generator().manual_seed(N)
    probs = [torch.softmax(2.0 * torch.randn(B, 2, N, N, generator=g), dim=-1) for _ in range(L)]
    R = O.deit_rollout(probs)
    ref = R[:, 0, 1:]
    fused = torch.stack([p.mean(1) for p in probs])
    cls_attn, idx, _ = ops.rollout(_pad_hm(fused), L, B, N, k, lead=1)
    assert_close(cls_attn, ref, rtol=1e-3, atol=1e-8, what="cls_token_attn")
    _check_topk(idx.cpu().long(), ref, k)


def test_rollout_with_precomputed_thresholds_is_identical():
    """ppf_rollout_threshold per layer (side stream in the model) + ppf_rollout(thr=...) == the monolithic rollout, bit for bit."""
    from protopformer_amd import ops
    g = torch.Generator().manual_seed(21)
    L, B, N, k = 5, 6, 197, 81
    fused = torch.softmax(3.0 * torch.randn(L, B, N, N, generator=g), dim=-1)
    hm = _pad_hm(fused)
    ref = ops.rollout(hm, L, B, N, k, lead=1)
    thr = torch.empty((L, B), dtype=torch.int32, device="cuda")
    for l in range(L):
        ops.rollout_threshold(hm[l], thr[l], N)
    got = ops.rollout(hm, L, B, N, k, lead=1, thr=thr)
    for a, b in zip(ref, got):
        assert torch.equal(a, b)


def test_rollout_cait_golden():
    from protopformer_amd import ops
    z = load_npz("ops_real.npz")
    c = gi.cait_inputs()
    fused = torch.stack([p.mean(1) for p in c["sa"]])
    init = torch.stack([p.mean(1)[:, 0] for p in c["cas"]])                 # [n_init, B, N+1]
    cls_attn, idx, policy = ops.rollout(_pad_hm(fused), 4, c["B"], c["N"], 121, lead=0, init_rows=init.cuda().contiguous())
    ref = torch.from_numpy(z["cait/rollout_cls"])
    assert_close(cls_attn, ref, rtol=1e-3, atol=1e-8, what="cait cls rollout vs reference")
    assert torch.equal(idx.cpu().long(), O.topk_sorted(ref, 121))


def test_proto_fwd_golden_and_kat():
    from protopformer_amd import ops
    z = load_npz("ops_real.npz")
    tok, protos = gi.proto_inputs()
    tokens = tok.flatten(2).transpose(1, 2).contiguous()                    # (B,81,Dp)
    act_max, argmax, dist, act = ops.proto_fwd(tokens.cuda(), 0, 81, protos.reshape(2000, -1).cuda())
    idx = torch.from_numpy(z["proto/dist_idx"])
    # d compared with abs 1e-6*(x2+p2) (~1e-4) or rel 1e-3
    assert_close(dist.reshape(-1).cpu()[idx], z["proto/dist_val"], rtol=1e-3, atol=2e-4, what="distances vs reference")
    assert float(dist[1, 7, 0]) == 0.0                                       # exactly representable token == prototype
    assert abs(float(act[1, 7, 0]) - 9.21034) < 1e-4
    assert float(dist[0, 5].reshape(9, 9)[2, 3]) < 1e-4
    ref_max = torch.from_numpy(z["proto/act_max"])
    far = ref_max < 2.9                                                      # d >= 0.05: outside the steep region
    assert_close(act_max.cpu()[far], ref_max[far], rtol=1e-3, what="max-pooled activation (d>=0.05)")
    # internal consistency: max/argmax agree with the materialised map
    mx, am = act.max(dim=-1)
    assert torch.equal(mx, act_max)
    assert torch.equal(torch.gather(act, 2, argmax.long()[..., None])[..., 0], act_max)


@pytest.mark.parametrize("B,T,Dp,P", [(3, 9, 32, 20), (2, 121, 192, 200), (130, 1, 64, 50), (4, 81, 384, 300), (300, 9, 32, 20), (2, 121, 384, 40)])
def test_proto_fwd_bwd_vs_oracle(B, T, Dp, P):
    from protopformer_amd import ops
    g = torch.Generator().manual_seed(T)
    tokens = torch.rand(B, T + 1, Dp, generator=g)
    protos = torch.rand(P, Dp, generator=g)
    t0 = 1 if T > 1 else 0
    xt = tokens.clone().requires_grad_(True); pr = protos.clone().requires_grad_(True)
    mx, d_ref, a_ref = O.proto_activations(xt[:, t0:t0 + T], pr)
    gmax = torch.randn(B, P, generator=g)
    gfull = torch.zeros(B, P, T)
    gfull[:, : max(1, P // 10)] = torch.randn(B, max(1, P // 10), T, generator=g)     # PPC-like sparse rows
    ((mx * gmax).sum() + (a_ref * gfull).sum()).backward()
    act_max, argmax, dist, act = ops.proto_fwd(tokens.cuda(), t0, T, protos.cuda())
    assert_close(dist, d_ref.detach(), rtol=1e-3, atol=1e-6 * Dp, what="dist")
    ok = d_ref.detach() > 0.05
    assert_close(act.cpu()[ok], a_ref.detach()[ok], rtol=1e-3, what="act")
    dtok = torch.zeros(B, T + 1, Dp, device="cuda"); dpro = torch.zeros(P, Dp, device="cuda")
    ops.proto_bwd(tokens.cuda(), t0, T, protos.cuda(), dist, gfull.cuda(), gmax.cuda(), argmax, dtok, dpro)
    assert_close(dtok, xt.grad, rtol=2e-3, atol=2e-3 * float(xt.grad.abs().max()), what="d tokens")
    assert_close(dpro, pr.grad, rtol=2e-3, atol=2e-3 * float(pr.grad.abs().max()), what="d prototypes")
    if T > 1:
        # the training form: d act / d dist taken from the ACTIVATION map (the only (B,P,T) map a training forward writes)
        dtok_a = torch.zeros_like(dtok); dpro_a = torch.zeros_like(dpro)
        ops.proto_bwd(tokens.cuda(), t0, T, protos.cuda(), act, gfull.cuda(), gmax.cuda(), argmax, dtok_a, dpro_a, from_act=True)
        assert_close(dtok_a, xt.grad, rtol=2e-3, atol=2e-3 * float(xt.grad.abs().max()), what="d tokens (from activations)")
        assert_close(dpro_a, pr.grad, rtol=2e-3, atol=2e-3 * float(pr.grad.abs().max()), what="d prototypes (from activations)")
        assert_close(dtok_a, dtok, rtol=1e-4, atol=2e-5 * float(dtok.abs().max()), what="from activations vs from distances, tokens")
        assert_close(dpro_a, dpro, rtol=1e-4, atol=2e-5 * float(dpro.abs().max()), what="from activations vs from distances, prototypes")


@pytest.mark.parametrize("B,T,Dp,P,ppc", [(5, 9, 32, 20, 2), (3, 121, 192, 200, 10), (6, 81, 384, 300, 10), (300, 9, 64, 40, 4), (2, 121, 384, 48, 16),
                                          (4, 16, 100, 60, 20)])
def test_proto_bwd_block_rows_vs_oracle(B, T, Dp, P, ppc):
    """ppf_proto_bwd_rows: the activation-map gradient given as the PPC loss's [B, ppc, T] block rows (prototypes label*ppc .. +ppc-1 of each
    sample) against autograd of the oracle fed the equivalent dense (B,P,T) gradient -- token and prototype gradients, together and in
    the two separate calls the train step makes (token gradients on the main stream, prototype gradients on the side stream)."""
    from protopformer_amd import ops
    g = torch.Generator().manual_seed(T + ppc)
    tokens = torch.rand(B, T + 1, Dp, generator=g)
    protos = torch.rand(P, Dp, generator=g)
    label = torch.randint(0, P // ppc, (B,), generator=g)
    if B >= 5:
        label[3] = label[1]                                                          # two samples of one class: ordered accumulation
    rows = torch.randn(B, ppc, T, generator=g)
    rows[:, :, ::3] = 0.0                                                            # exact zeros inside the block, as the hinge terms leave
    gmax = torch.randn(B, P, generator=g)
    gmax[:, 1] = 0.0
    gfull = torch.zeros(B, P, T)
    for b in range(B):
        gfull[b, int(label[b]) * ppc: int(label[b]) * ppc + ppc] = rows[b]
    xt = tokens.clone().requires_grad_(True); pr = protos.clone().requires_grad_(True)
    mx, d_ref, a_ref = O.proto_activations(xt[:, 1:1 + T], pr)
    ((mx * gmax).sum() + (a_ref * gfull).sum()).backward()
    act_max, argmax, dist, act = ops.proto_fwd(tokens.cuda(), 1, T, protos.cuda())
    cu = lambda t: t.cuda()
    blk = (cu(rows), cu(label), ppc)
    dtok = torch.zeros(B, T + 1, Dp, device="cuda"); dpro = torch.zeros(P, Dp, device="cuda")
    ops.proto_bwd(cu(tokens), 1, T, cu(protos), dist, None, cu(gmax), argmax, dtok, dpro, rows=blk)
    assert_close(dtok, xt.grad, rtol=2e-3, atol=2e-3 * float(xt.grad.abs().max()), what="d tokens (block rows)")
    assert_close(dpro, pr.grad, rtol=2e-3, atol=2e-3 * float(pr.grad.abs().max()), what="d prototypes (block rows)")
    dtok2 = torch.zeros_like(dtok); dpro2 = torch.zeros_like(dpro)
    ops.proto_bwd(cu(tokens), 1, T, cu(protos), dist, None, cu(gmax), argmax, dtok2, None, rows=blk)
    ops.proto_bwd(cu(tokens), 1, T, cu(protos), dist, None, cu(gmax), argmax, None, dpro2, rows=blk)
    assert torch.equal(dtok2, dtok) and torch.equal(dpro2, dpro), "separate calls differ from the combined one"
    # the training form (derivative from the activation map) through the block-row kernels
    dtok_a = torch.zeros_like(dtok); dpro_a = torch.zeros_like(dpro)
    ops.proto_bwd(cu(tokens), 1, T, cu(protos), act, None, cu(gmax), argmax, dtok_a, dpro_a, rows=blk, from_act=True)
    assert_close(dtok_a, xt.grad, rtol=2e-3, atol=2e-3 * float(xt.grad.abs().max()), what="d tokens (block rows, from activations)")
    assert_close(dpro_a, pr.grad, rtol=2e-3, atol=2e-3 * float(pr.grad.abs().max()), what="d prototypes (block rows, from activations)")
    assert_close(dtok_a, dtok, rtol=1e-4, atol=2e-5 * float(dtok.abs().max()), what="from activations vs from distances, tokens")
    assert_close(dpro_a, dpro, rtol=1e-4, atol=2e-5 * float(dpro.abs().max()), what="from activations vs from distances, prototypes")
    # the dense form of the same gradient agrees to rounding (different summation order)
    dtok3 = torch.zeros_like(dtok); dpro3 = torch.zeros_like(dpro)
    ops.proto_bwd(cu(tokens), 1, T, cu(protos), dist, cu(gfull), cu(gmax), argmax, dtok3, dpro3)
    assert_close(dtok3, dtok, rtol=1e-4, atol=1e-5 * float(dtok.abs().max()), what="dense vs block form, tokens")
    assert_close(dpro3, dpro, rtol=1e-3, atol=2e-5 * float(dpro.abs().max()), what="dense vs block form, prototypes")
    # rows only (no max-pool gradient)
    dpro4 = torch.zeros_like(dpro); dpro5 = torch.zeros_like(dpro)
    ops.proto_bwd(cu(tokens), 1, T, cu(protos), dist, None, None, None, None, dpro4, rows=blk)
    ops.proto_bwd(cu(tokens), 1, T, cu(protos), dist, cu(gfull), None, None, None, dpro5)
    assert_close(dpro4, dpro5, rtol=1e-3, atol=2e-5 * float(dpro5.abs().max()), what="rows only")


@pytest.mark.parametrize("from_act", [False, True])
def test_proto_linear_activation_pooled_backward(from_act):
    """prototype_activation_function='linear' (act = -d, protopformer.py:222-225) through the pooled branch: forward maps and the backward
    from the distance map and from the activation map (d act / d dist = -1 where d > 0) against autograd of the oracle."""
    from protopformer_amd import ops
    B, T, Dp, P = 3, 9, 32, 20
    g = torch.Generator().manual_seed(21)
    tokens = torch.rand(B, T + 1, Dp, generator=g)
    protos = torch.rand(P, Dp, generator=g)
    xt = tokens.clone().requires_grad_(True); pr = protos.clone().requires_grad_(True)
    mx, d_ref, a_ref = O.proto_activations(xt[:, 1:1 + T], pr, activation="linear")
    gmax = torch.randn(B, P, generator=g)
    gfull = torch.zeros(B, P, T); gfull[:, :3] = torch.randn(B, 3, T, generator=g)
    ((mx * gmax).sum() + (a_ref * gfull).sum()).backward()
    act_max, argmax, dist, act = ops.proto_fwd(tokens.cuda(), 1, T, protos.cuda(), act_kind=1)
    assert_close(act, a_ref.detach(), rtol=1e-3, atol=1e-5 * Dp, what="linear activation map")
    assert_close(act_max, mx.detach(), rtol=1e-3, atol=1e-5 * Dp, what="linear max-pooled activation")
    dtok = torch.zeros(B, T + 1, Dp, device="cuda"); dpro = torch.zeros(P, Dp, device="cuda")
    ops.proto_bwd(tokens.cuda(), 1, T, protos.cuda(), act if from_act else dist, gfull.cuda(), gmax.cuda(), argmax, dtok, dpro, act_kind=1, from_act=from_act)
    assert_close(dtok, xt.grad, rtol=2e-3, atol=2e-3 * float(xt.grad.abs().max()), what="d tokens (linear)")
    assert_close(dpro, pr.grad, rtol=2e-3, atol=2e-3 * float(pr.grad.abs().max()), what="d prototypes (linear)")


def test_proto_bwd_from_activations_clipped_distance_known_answer():
    """token == prototype: d = 0 exactly, the reference's relu clips there and autograd gives a ZERO gradient through that pair
    (protopformer.py:216); the activation-map form must recognise the forward's own value at d = 0 and do the same -- and give the
    closed-form derivative 1/(d+1) - 1/(d+eps) elsewhere."""
    from protopformer_amd import ops
    g = torch.Generator().manual_seed(5)
    B, T, Dp, P = 2, 9, 32, 12
    tokens = torch.randint(0, 16, (B, T + 1, Dp), generator=g).float() / 16          # exactly representable: d == 0 comes out exactly
    protos = torch.randint(0, 16, (P, Dp), generator=g).float() / 16
    protos[3] = tokens[0, 1 + 4]
    protos[7] = tokens[1, 1 + 0]
    act_max, argmax, dist, act = ops.proto_fwd(tokens.cuda(), 1, T, protos.cuda())
    assert float(dist[0, 3, 4]) == 0.0 and float(dist[1, 7, 0]) == 0.0
    assert int(argmax[0, 3]) == 4 and int(argmax[1, 7]) == 0
    gmax = torch.zeros(B, P); gmax[0, 3] = 1.0; gmax[1, 7] = -2.0; gmax[0, 5] = 0.5
    for from_act, m in ((False, dist), (True, act)):
        dtok = torch.zeros(B, T + 1, Dp, device="cuda"); dpro = torch.zeros(P, Dp, device="cuda")
        ops.proto_bwd(tokens.cuda(), 1, T, protos.cuda(), m, None, gmax.cuda(), argmax, dtok, dpro, from_act=from_act)
        assert float(dpro[3].abs().max()) == 0.0 and float(dpro[7].abs().max()) == 0.0, from_act
        t5 = int(argmax[0, 5]); d5 = float(dist[0, 5, t5])
        coef = 0.5 * (1.0 / (d5 + 1.0) - 1.0 / (d5 + 1e-4))
        want = coef * 2.0 * (protos[5] - tokens[0, 1 + t5])
        assert_close(dpro[5], want, rtol=1e-4, atol=1e-5 * float(want.abs().max()), what=f"closed form (from_act={from_act})")
        assert float(dtok[1].abs().max()) == 0.0, "sample 1 only touches the clipped pair"


def test_ppc_loss_golden_and_grad():
    from protopformer_amd import ops
    z = load_npz("ops_real.npz")
    tpa, roll, lab = gi.ppc_inputs()
    idx = O.topk_sorted(roll, 81).int()
    act = tpa.flatten(2).contiguous()
    loss, gcov, gmean = ops.ppc_loss(act.cuda(), idx.cuda(), lab.cuda(), 10, 14, 1.0, 2.0)
    assert_close(loss[0], z["ppc/cov"], rtol=1e-4, atol=1e-6, what="ppc cov vs reference")
    assert_close(loss[1], z["ppc/mean"], rtol=1e-4, atol=1e-6, what="ppc mean vs reference")
    a = act.clone().requires_grad_(True)
    cov, mean = O.ppc_loss(a.reshape(4, 200, 9, 9), roll, 196, lab, 10, 1.0, 2.0)
    (0.3 * cov + 0.7 * mean).backward()
    up = torch.tensor([0.3, 0.7], device="cuda")
    g_full = ops.ppc_loss_bwd(gcov, gmean, up[0:1], up[1:2], lab.cuda(), 200)
    assert_close(g_full, a.grad, rtol=1e-3, atol=1e-7, what="ppc grad")


def test_ppc_known_answer():
    from protopformer_amd import ops
    act = torch.ones(2, 200, 196, device="cuda")
    idx = torch.arange(196, dtype=torch.int32, device="cuda").repeat(2, 1).contiguous()
    loss, _, _ = ops.ppc_loss(act, idx, torch.tensor([0, 5], device="cuda"), 10, 14, 1.0, 2.0)
    assert abs(float(loss[0]) - (16.25 * 196 / 195 - 1)) < 1e-4 and abs(float(loss[1]) - 1.8) < 1e-5


def test_cross_entropy_and_frozen_head():
    from protopformer_amd import ops
    g = torch.Generator().manual_seed(4)
    B, C, P = 37, 200, 2000
    logits = torch.randn(B, C, generator=g) * 3
    lab = torch.randint(0, C, (B,), generator=g)
    lr = logits.clone().requires_grad_(True)
    ref = torch.nn.functional.cross_entropy(lr, lab); ref.backward()
    loss, dl = ops.cross_entropy(logits.cuda(), lab.cuda())
    assert_close(loss[0], ref.detach(), rtol=1e-5, atol=1e-6, what="ce")
    assert_close(dl, lr.grad, rtol=1e-4, atol=1e-7, what="dlogits")
    act = torch.rand(B, P, generator=g) * 5
    sd = O.init_state_dict(dict(O.make_cfg("deit_tiny_patch16_224", P, 192, C, 11, 81), depth=1), seed=0)
    w = sd["last_layer.weight"]
    out = torch.empty(B, C, device="cuda")
    ops.sgemm(act.cuda(), w.cuda(), out, B, C, P, P, 1, P, 1, alpha=0.5)
    assert_close(out, 0.5 * act @ w.t(), rtol=1e-4, atol=1e-3, what="frozen head logits")
    dact = torch.empty(B, P, device="cuda")
    ops.sgemm(dl, w.cuda(), dact, B, P, C, C, 1, 1, P, alpha=0.5)            # dA = dlogits @ W
    assert_close(dact, 0.5 * lr.grad @ w, rtol=1e-4, atol=1e-6, what="frozen head dgrad")


@pytest.mark.parametrize("B,Dp,P,kind", [(130, 64, 50, 0), (256, 384, 2000, 0), (7, 192, 333, 1)])
def test_proto_bwd_single_token_dense_form(B, Dp, P, kind):
    """T == 1 (global branch): the dense two-product backward (ppf_proto_bwd_single) against autograd of the oracle, with dtok and
    dprotos requested in separate calls (as the train step does: main stream / side stream) and dprotos accumulated."""
    from protopformer_amd import ops
    g = torch.Generator().manual_seed(B + P)
    tokens = torch.rand(B, 3, Dp, generator=g)
    protos = torch.rand(P, Dp, generator=g)
    xt = tokens.clone().requires_grad_(True); pr = protos.clone().requires_grad_(True)
    mx, d_ref, a_ref = O.proto_activations(xt[:, 1:2], pr, "log" if kind == 0 else "linear")
    gmax = torch.randn(B, P, generator=g)
    (mx * gmax).sum().backward()
    act_max, _, dist, _ = ops.proto_fwd(tokens.cuda(), 1, 1, protos.cuda(), act_kind=kind)
    dtok = torch.zeros(B, 3, Dp, device="cuda")
    dpro = torch.full((P, Dp), 0.25, device="cuda")
    ops.proto_bwd(tokens.cuda(), 1, 1, protos.cuda(), dist, None, gmax.cuda(), None, dtok, None, act_kind=kind)
    ops.proto_bwd(tokens.cuda(), 1, 1, protos.cuda(), dist, None, gmax.cuda(), None, None, dpro, act_kind=kind)
    assert_close(dtok, xt.grad, rtol=2e-3, atol=2e-3 * float(xt.grad.abs().max()), what="d tokens (dense form)")
    assert float(dtok[:, 0].abs().max()) == 0.0 and float(dtok[:, 2].abs().max()) == 0.0        # only the token row is written
    assert_close(dpro - 0.25, pr.grad, rtol=2e-3, atol=2e-3 * float(pr.grad.abs().max()), what="d prototypes (dense form, accumulated)")


def test_sgemm_pair_matches_two_products():
    """ppf_sgemm_pair: both class-connection products, their scaled copies and the weighted sum in one launch + one reduction."""
    from protopformer_amd import ops
    g = torch.Generator().manual_seed(5)
    B, C, P0, P1 = 37, 200, 980, 1960
    a0, a1 = torch.rand(B, P0, generator=g).cuda(), torch.rand(B, P1, generator=g).cuda()
    w0, w1 = torch.randn(C, P0, generator=g).cuda(), torch.randn(C, P1, generator=g).cuda()
    o0, o1, tot = (torch.empty(B, C, device="cuda") for _ in range(3))
    ops.sgemm_pair(a0, w0, o0, C, P0, (P0, 1), (P0, 1), 1.0, a1, w1, o1, C, P1, (P1, 1), (P1, 1), 1.0, B, total=tot, c0=0.3, c1=0.7)
    r0, r1 = a0.double() @ w0.double().t(), a1.double() @ w1.double().t()
    assert_close(o0, r0.float(), rtol=1e-5, atol=1e-3, what="first product")
    assert_close(o1, r1.float(), rtol=1e-5, atol=1e-3, what="second product")
    assert_close(tot, (0.3 * r0 + 0.7 * r1).float(), rtol=1e-5, atol=1e-3, what="weighted sum")
    # input gradients: different N per problem, no sum
    d = torch.randn(B, C, generator=g).cuda()
    g0, g1 = torch.empty(B, P0, device="cuda"), torch.empty(B, P1, device="cuda")
    ops.sgemm_pair(d, w0, g0, P0, C, (C, 1), (1, P0), 0.3, d, w1, g1, P1, C, (C, 1), (1, P1), 0.7, B)
    assert_close(g0, (0.3 * d.double() @ w0.double()).float(), rtol=1e-5, atol=1e-4, what="d act global")
    assert_close(g1, (0.7 * d.double() @ w1.double()).float(), rtol=1e-5, atol=1e-4, what="d act local")
