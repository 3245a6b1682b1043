"""BASELINE.json configurations themselves on the GPU:

* configs[2] deit_small_patch16_224 + 2000x384 prototypes / 200 classes / k=81: parity with the fp32 CPU oracle at B=2 (same
  reservation), then the FULL batch-256 train step as size-independent properties (finite, every gradient live, bit-identical
  repeat, loss decreases, compacted == masked blocks, captured graph == eager);
* configs[1] deit_tiny 2000x192 bs128 and configs[4] cait_xxs24 1960x192 (196 classes, k=121) bs128: the same properties.
Tolerances: bf16 MFMA operands, fp32 accumulate -- gates are <= 3x the errors measured on MI355X (helpers.report -> gpurun_out)."""
import pytest
import torch

from helpers import assert_close, assert_elementwise, rel_err, report
from oracle import ppf_oracle as O

pytestmark = pytest.mark.gpu

CFG = {
    # not a BASELINE configuration: the third DeiT the reference registers (deit:315-328); K = 768 / 3072 GEMMs take the 256x256 kernel
    "deit_base": dict(arch="deit_base_patch16_224", P=2000, Dp=192, C=200, k=81, layer=11, gpc=10, batch=64),
    "deit_small": dict(arch="deit_small_patch16_224", P=2000, Dp=384, C=200, k=81, layer=11, gpc=10, batch=256),
    "deit_tiny": dict(arch="deit_tiny_patch16_224", P=2000, Dp=192, C=200, k=81, layer=11, gpc=10, batch=128),
    "cait_xxs24": dict(arch="cait_xxs24_224", P=1960, Dp=192, C=196, k=121, layer=1, gpc=5, batch=128),
}


def _construct(c, seed, sd=None):
    from protopformer_amd import backbone
    from protopformer_amd.protopformer import construct_PPNet
    backbone._KEEP_CACHE.clear()                       # DropPath stream restarts: (seed, step counter) -> identical draws
    torch.manual_seed(seed)
    m = construct_PPNet(c["arch"], pretrained=False, img_size=224, prototype_shape=(c["P"], c["Dp"], 1, 1), num_classes=c["C"],
                        reserve_layers=[c["layer"]], reserve_token_nums=[c["k"]], use_global=True, use_ppc_loss=True, ppc_cov_thresh=1.,
                        ppc_mean_thresh=2., global_coe=0.5, global_proto_per_class=c["gpc"], add_on_layers_type="regular")
    if sd is not None:
        m.load_state_dict(sd, strict=True)
    return m.cuda().train()


def _batch(c, B, seed=1028):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return (torch.randn(B, 3, 224, 224, device="cuda", generator=g), torch.randint(0, c["C"], (B,), device="cuda", generator=g))


@pytest.mark.parametrize("name", ["deit_small", "deit_tiny", "cait_xxs24", "deit_base"])
def test_baseline_config_small_batch_vs_oracle(name):
    """Real architecture, real head, the reference's initialisers, B=2: logits / CE / PPC terms / sampled gradients vs the fp32 oracle
    following the reservation the bf16 run selected (index exactness is only attainable at the kernel boundary)."""
    from protopformer_amd.protopformer import CrossEntropyLoss
    c = CFG[name]
    cfg = O.make_cfg(c["arch"], c["P"], c["Dp"], c["C"], c["layer"], c["k"], global_per_class=c["gpc"])
    sd = O.init_state_dict(cfg, seed=1028)
    m = _construct(c, 0, sd)
    for blk in m.features.blocks:
        blk.drop_path_rate = 0.0
    g = torch.Generator().manual_seed(77)
    img = torch.randn(2, 3, 224, 224, generator=g); label = torch.tensor([5, c["C"] - 2])
    logits, aux = m(img.cuda())
    ce = CrossEntropyLoss()(logits, label.cuda())
    cov, mean = m.get_PPC_loss(aux[2], aux[3], aux[4], label.cuda())
    loss = ce + 0.1 * cov + 0.5 * mean
    loss.backward()
    my_idx = m._ppc_cache[1].cpu().long()
    params = {k_: v.clone().requires_grad_(k_ not in O.FROZEN_KEYS) for k_, v in sd.items()}
    # ... and the max-pool routing it selected: a near-tied arg-max sends one prototype's whole gradient through another token than in the
    # fp32 run (a discontinuity of the reference's max_pool2d, as in test_gpu_e2e.py; at B = 2 ONE such flip drops every tensor's cosine to
    # 0.97-0.985, which round 5 absorbed with a 0.97 floor for deit_base and round 6's forward tripped at 0.969) -- the routing itself is gated below
    my_arg = m._last_argmax.cpu().long()
    out = O.ppnet_forward(params, img, cfg, train=True, force_idx=my_idx, force_argmax=my_arg)
    loss_ref, parts = O.train_loss(out, label, cfg, with_ppc=True)
    loss_ref.backward()
    with torch.no_grad():
        free = O.ppnet_forward(sd, img, cfg, train=True)
        same_idx = O.ppnet_forward(sd, img, cfg, train=True, force_idx=my_idx)
    n_flip = int((same_idx["total_proto_act"].flatten(2).argmax(-1) != my_arg).sum())
    n_diff = int((torch.zeros_like(free["cls_token_attn"], dtype=torch.bool).scatter_(1, my_idx, True)
                  != torch.zeros_like(free["cls_token_attn"], dtype=torch.bool).scatter_(1, free["reserve_idx"], True)).sum()) // 2
    e = dict(logits=rel_err(logits, out["logits"]), ce=rel_err(ce, parts["ce"]), cov=rel_err(cov, parts["ppc_cov"]),
             mean=rel_err(mean, parts["ppc_mean"]), loss=rel_err(loss, loss_ref), cls_attn=rel_err(aux[3], free["cls_token_attn"]),
             act=rel_err(aux[2], out["total_proto_act"]), reserved_tokens_differing=n_diff, argmax_flips=n_flip, argmax_total=my_arg.numel(),
             # what a user switching from the reference sees: the fp32 oracle following ITS OWN reservation (no force_idx)
             logits_own_reservation=rel_err(logits, free["logits"]))
    cos = {}
    for nm, p in m.named_parameters():
        if p.requires_grad and params[nm].grad is not None and float(params[nm].grad.abs().max()) > 1e-12:
            gm, gr = p.grad.float().cpu().reshape(-1), params[nm].grad.reshape(-1)
            cos[nm] = float(torch.dot(gm, gr) / (gm.norm() * gr.norm()).clamp_min(1e-30))
    skip = ("proj_l.bias", "attn.k.bias")                     # mathematically zero gradients (softmax shift invariance)
    worst = min(v for k_, v in cos.items() if not k_.endswith(skip))
    e["worst_grad_cos"] = worst
    report(f"baseline_small_batch[{name}]", **e)
    # measured on MI355X (deit_small / deit_tiny / cait_xxs24): logits 1.9e-4 / 1.1e-4 / 8.6e-5, CE 1.3e-6 / 8.9e-6 / 9.9e-7, PPC terms <= 1.7e-5,
    # activations 1.2e-3 / 1.5e-3 / 1.1e-3, worst gradient cosine 0.99994 / 0.99994 / 0.99976 -> the north star's 1e-3 holds for logits and
    # losses on the bf16 product path at the BASELINE shapes; gates at <= 3x the measurements
    # (deit_base, D = 768: logits 3.3e-4, loss 1.7e-5, PPC 2.5e-5 / 9.0e-5, activations 2.1e-3)
    assert_elementwise(logits, out["logits"], 1e-3, "logits vs the oracle on the same reservation")
    assert e["logits_own_reservation"] < 3.5e-3, e          # measured 6.7e-4 ... 1.1e-3: a handful of swapped near-tied tokens move the logits
    assert e["logits"] < 1e-3 and e["ce"] < 3e-5 and e["loss"] < 5e-5 and e["cov"] < 8e-5 and e["mean"] < 3e-4 and e["act"] < 6e-3, e
    # the rollout map multiplies 11 (24) bf16-derived attention maps: measured 5.3e-2 / 5.8e-2 / 2.0e-3 of its maximum
    assert e["cls_attn"] < 0.1, e
    # reservation: the bf16 rollout may swap tokens whose fp32 scores sit within its error band of the k-th value; measured 7 of 162
    # (deit_small), 3 / 4 / 3 (deit_tiny / cait_xxs24 / deit_base) -- gate at 10 % of the reserved tokens
    assert n_diff <= 0.1 * my_idx.numel(), (n_diff, my_idx.numel())
    # max-pool routing: all but near-ties agree with the fp32 oracle on the same tokens (random-init prototypes are nearly equidistant from every
    # token: measured 52 of 4 000 at deit_base, fewer elsewhere; 22 of 800 on the peaky model of test_gpu_e2e.py) -- gate 3 %
    assert n_flip <= 0.03 * my_arg.numel(), (n_flip, my_arg.numel())
    floor = 0.9992
    assert len(cos) > 100 and worst > floor, {k_: v for k_, v in cos.items() if v <= floor}


def _run_steps(c, n_steps, graph, seed=5, B=None):
    from protopformer_amd.engine import FlatAdamW, GraphedTrainStep, train_one_step
    from protopformer_amd.protopformer import CrossEntropyLoss
    B = B or c["batch"]
    m = _construct(c, seed)
    opt = FlatAdamW(m, weight_decay=0.05, ema_decay=0.99996)
    img, label = _batch(c, B)
    crit = CrossEntropyLoss()
    step = GraphedTrainStep(m, crit, opt, epoch=20, warmup=1) if graph else (lambda a, b: train_one_step(m, crit, a, b, opt, epoch=20))
    losses = [float(step(img, label)[0]) for _ in range(n_steps)]
    torch.cuda.synchronize()
    return m, opt, losses


@pytest.mark.parametrize("name", ["deit_small", "deit_tiny", "cait_xxs24"])
def test_baseline_config_full_batch_properties(name):
    """The bench workload at FULL size (configs[2]: batch 256; configs[1] / [4]: batch 128), DropPath 0.1 active."""
    c = CFG[name]
    m, opt, losses = _run_steps(c, 5, graph=False)
    assert all(torch.isfinite(torch.tensor(losses))), losses
    assert losses[-1] < losses[0], losses                         # same batch, AdamW: the loss must go down
    st = m.flat_store()
    assert bool(torch.isfinite(st.grads).all()) and bool(torch.isfinite(st.params).all()) and bool(torch.isfinite(opt.ema).all())
    dead = [n for n, p, o, k_ in st.entries if float(st.grads[o:o + k_].abs().max()) == 0.0]
    # zero-gradient tensors by construction: none for DeiT; CaiT's key bias of the class attention has a mathematically zero gradient
    assert not [n for n in dead if not n.endswith(("attn.k.bias", "proj_l.bias"))], dead
    params_a = st.params.clone()
    del m, opt
    torch.cuda.empty_cache()
    m2, opt2, losses2 = _run_steps(c, 5, graph=False)
    # no float atomics on either path (CaiT's proj_l / proj_w gradients: per-workgroup partial rows + ordered sum since round 3): bit-identical repeat
    from protopformer_amd import ops
    if name.startswith("deit") or ops.th_fused_ok(4, 196, 192):
        assert losses2 == losses, (losses, losses2)
        assert torch.equal(m2.flat_store().params, params_a)
    else:                                                         # PPF_TH_FUSED=0: the materialising talking-heads kernels sum proj_l / proj_w gradients with fp32 atomics
        assert max(abs(a - b) / abs(a) for a, b in zip(losses, losses2)) < 1e-3, (losses, losses2)
    report(f"baseline_full_batch[{name}]", loss0=losses[0], loss4=losses[-1])


@pytest.mark.parametrize("name", ["deit_small", "cait_xxs24"])
def test_baseline_config_graph_replay_equals_eager(name):
    """engine.GraphedTrainStep at full size, in a child process (scripts/gpu/graph_check.py): replayed losses == eager losses."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "gpu", "graph_check.py"), name, "4"], capture_output=True, text=True,
                       timeout=900, cwd=root)
    line = [l for l in r.stdout.splitlines() if l.startswith("GRAPH_CHECK ")]
    # a child that dies on a signal inside the HIP graph runtime is a FAILURE of the captured-step path, not a skip
    assert r.returncode == 0 and line, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
    out = json.loads(line[0][len("GRAPH_CHECK "):])
    eager, graphed = out["eager"], out["graphed"]
    assert out["steps"] == 4
    from protopformer_amd import ops
    if name.startswith("deit") or ops.th_fused_ok(4, 196, 192):
        assert graphed == eager, (eager, graphed)
    else:                                                         # PPF_TH_FUSED=0 (fp32 atomics in the fallback kernels)
        assert max(abs(a - b) / abs(a) for a, b in zip(eager, graphed)) < 1e-3, (eager, graphed)


def test_deit_small_bs256_compacted_equals_masked_blocks(monkeypatch):
    """configs[2] at batch 256: blocks after the reservation on the 1+k reserved rows vs the reference's masked full-length blocks."""
    from protopformer_amd.protopformer import CrossEntropyLoss
    c = CFG["deit_small"]
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("PPF_COMPACT_RESERVED", mode)
        m = _construct(c, 9)
        for blk in m.features.blocks:
            blk.drop_path_rate = 0.0
        img, label = _batch(c, 256)
        logits, aux = m(img)
        ce = CrossEntropyLoss()(logits, label)
        cov, mean = m.get_PPC_loss(aux[2], aux[3], aux[4], label)
        (ce + 0.1 * cov + 0.5 * mean).backward()
        torch.cuda.synchronize()
        res[mode] = dict(idx=m._ppc_cache[1].clone(), logits=logits.detach().clone(), ce=float(ce), cov=float(cov), mean=float(mean),
                         grads=m.flat_store().grads.clone())
        del m
        torch.cuda.empty_cache()
    a, b = res["1"], res["0"]
    assert torch.equal(a["idx"], b["idx"])
    e = dict(logits=rel_err(a["logits"], b["logits"]), ce=abs(a["ce"] - b["ce"]) / abs(b["ce"]), cov=abs(a["cov"] - b["cov"]) / abs(b["cov"]),
             cos=float(torch.dot(a["grads"], b["grads"]) / (a["grads"].norm() * b["grads"].norm())))
    report("compact_vs_masked_bs256", **e)
    assert e["logits"] < 1e-2 and e["ce"] < 1e-3 and e["cov"] < 1e-2 and e["cos"] > 0.995, e
