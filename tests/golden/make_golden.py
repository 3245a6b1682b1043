"""Generate the golden fixtures in tests/golden/ by running the REFERENCE implementation.

Runs only in the build container (needs /root/reference):  python tests/golden/make_golden.py
It imports /root/reference/protopformer.py through the stand-ins in _timm_standins.py, feeds it
seeded inputs and stores inputs + the reference's outputs as .npz data files.  No reference source
is copied; the fixtures are numbers only.  The oracle (oracle/ppf_oracle.py) is then pinned
against these files by tests/test_oracle_golden.py, which runs anywhere.
"""
import os
import sys
from functools import partial

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _timm_standins as standins  # noqa: E402

ref = standins.install()
import tools.deit_models_attn as ref_deit  # noqa: E402
import tools.cait_models_attn as ref_cait  # noqa: E402


def np32(t):
    return t.detach().cpu().numpy().astype(np.float32) if t.is_floating_point() else t.detach().cpu().numpy()


def randomize(model, seed):
    """Replace the near-trivial default init (zero biases, unit LN, 1e-5 LayerScale) by seeded values
    that exercise every term; frozen tensors (ones / last layers) are left alone."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if not p.requires_grad:
                continue
            if name.startswith('prototype_vectors'):
                p.copy_(torch.rand(p.shape, generator=g))
            elif 'gamma_' in name:
                p.copy_(0.05 + 0.25 * torch.rand(p.shape, generator=g))
            elif name.endswith('norm1.weight') or name.endswith('norm2.weight') or name.endswith('norm.weight'):
                p.copy_(1.0 + 0.2 * torch.randn(p.shape, generator=g))
            elif name.endswith('.bias'):
                p.copy_(0.1 * torch.randn(p.shape, generator=g))
            elif 'qkv.weight' in name or name.endswith('attn.q.weight') or name.endswith('attn.k.weight'):
                p.copy_(0.10 * torch.randn(p.shape, generator=g))
            elif 'proj_l.weight' in name or 'proj_w.weight' in name:
                p.copy_(torch.eye(p.shape[0]) + 0.3 * torch.randn(p.shape, generator=g))
            elif 'cls_token' in name or 'pos_embed' in name:
                p.copy_(0.2 * torch.randn(p.shape, generator=g))
            elif 'patch_embed.proj.weight' in name:
                p.copy_(0.03 * torch.randn(p.shape, generator=g))
            else:
                p.copy_(0.05 * torch.randn(p.shape, generator=g))


def split_groups(model):
    """tools/create_optimizer.py:27-39 (weight_decay=None branch) with main.py:364-366 lrs."""
    return [
        {'params': model.features.parameters(), 'lr': 1e-4, 'weight_decay': 1e-3},
        {'params': model.add_on_layers.parameters(), 'lr': 3e-3, 'weight_decay': 1e-3},
        {'params': model.prototype_vectors, 'lr': 3e-3},
        {'params': model.prototype_vectors_global, 'lr': 3e-3},
    ]


def micro_fixture(kind, path, seed, add_on='regular', proto_dim=32):
    torch.manual_seed(seed)
    if kind == 'deit':
        feats = ref_deit.MyVisionTransformer(img_size=64, patch_size=16, embed_dim=64, depth=6, num_heads=2, mlp_ratio=4,
                                             qkv_bias=True, norm_layer=partial(nn.LayerNorm, eps=1e-6), num_classes=10,
                                             drop_rate=0., drop_path_rate=0.)
        del_head = False
        reserve_layer, meta = 5, dict(arch='deit', dim=64, depth=6, heads=2)
    else:
        feats = ref_cait.MyCait(img_size=64, patch_size=16, embed_dim=96, depth=5, num_heads=2, init_scale=1e-5,
                                num_classes=10, drop_rate=0., drop_path_rate=0.)
        del feats.head
        reserve_layer, meta = 1, dict(arch='cait', dim=96, depth=5, heads=2)
    model = ref.PPNet(features=feats, img_size=64, prototype_shape=[20, proto_dim, 1, 1], proto_layer_rf_info=None,
                      num_classes=10, reserve_layers=[reserve_layer], reserve_token_nums=[9], use_global=True,
                      use_ppc_loss=True, ppc_cov_thresh=1., ppc_mean_thresh=2., global_coe=0.5,
                      global_proto_per_class=2, init_weights=True, prototype_activation_function='log',
                      add_on_layers_type=add_on)
    randomize(model, seed + 1)
    g = torch.Generator().manual_seed(seed + 2)
    img = torch.randn(4, 3, 64, 64, generator=g)
    label = torch.tensor([3, 0, 9, 3])

    out = {f'sd/{k}': np32(v) for k, v in model.state_dict().items()}
    out.update(img=np32(img), label=label.numpy())
    for k, v in meta.items():
        out[f'meta/{k}'] = np.array(v)
    if add_on != 'regular':
        out['meta/add_on'] = np.array(add_on)
    out.update({'meta/reserve_layer': np.array(reserve_layer), 'meta/reserve_k': np.array(9), 'meta/img': np.array(64),
                'meta/num_prototypes': np.array(20), 'meta/proto_dim': np.array(proto_dim), 'meta/num_classes': np.array(10),
                'meta/global_per_class': np.array(2), 'meta/global_coe': np.array(0.5)})

    model.eval()
    with torch.no_grad():
        logits, (cls_attn, distances, lg, ll) = model(img)
        cls_attn_p, proto_acts = model.push_forward(img)
    # the fixture must be tie-free at the top-k boundary (torch.topk's tie order is implementation-defined)
    srt = cls_attn.sort(dim=-1, descending=True)[0]
    assert float((srt[:, 8] - srt[:, 9]).min()) > 1e-5 * float(srt.max()), 'top-k boundary tie in the fixture: change the seed'
    out.update({'eval/logits': np32(logits), 'eval/cls_token_attn': np32(cls_attn), 'eval/distances': np32(distances),
                'eval/logits_global': np32(lg), 'eval/logits_local': np32(ll), 'eval/push_proto_acts': np32(proto_acts)})

    standins.freeze_droppath(model)
    opt = torch.optim.AdamW(split_groups(model), weight_decay=0.05, eps=1e-8)
    logits, aux = model(img)
    assert aux[0] is None and aux[4] == 16
    ce = nn.CrossEntropyLoss()(logits, label)
    cov, mean = model.get_PPC_loss(aux[2], aux[3], aux[4], label)
    loss = ce + 0.1 * cov + 0.5 * mean
    opt.zero_grad()
    loss.backward()
    out.update({'train/logits': np32(logits), 'train/total_proto_act': np32(aux[2]), 'train/cls_attn_rollout': np32(aux[3]),
                'train/ce': np32(ce), 'train/ppc_cov': np32(cov), 'train/ppc_mean': np32(mean), 'train/loss': np32(loss)})
    def put(tag, name, t):
        # small tensors in full; large ones as a seeded sample + checksum (keeps the fixture small)
        if t.numel() <= 8192:
            out[f'{tag}/{name}'] = np32(t)
        else:
            i, v, s = sample(t, n=2048, seed=len(name))
            out[f'{tag}_idx/{name}'] = i; out[f'{tag}_val/{name}'] = v; out[f'{tag}_sum/{name}'] = s
    for name, p in model.named_parameters():
        if p.requires_grad:
            assert p.grad is not None, name
            put('grad', name, p.grad)
    opt.step()
    for name, p in model.named_parameters():
        if p.requires_grad:
            put('step', name, p)
    np.savez_compressed(path, **out)
    print('wrote', path, f'{os.path.getsize(path) / 1e6:.2f} MB')


def sample(t, n=4096, seed=0):
    """Deterministic sample + checksum of a large tensor."""
    from golden_inputs import sample_indices
    flat = t.detach().reshape(-1)
    idx = sample_indices(flat.numel(), n, seed)
    return idx.numpy(), np32(flat[idx]), np.float64(flat.double().sum().item())


def ops_fixture(path):
    """Op-level vectors at real shapes; inputs come from golden_inputs.py (seeded),
    only outputs (or samples + checksums of large ones) are stored."""
    import golden_inputs as gi
    out = {}
    a = gi.attn_inputs()
    attn = ref_deit.Attention(a['D'], num_heads=a['H'], qkv_bias=True)
    attn.load_state_dict(a['w'])
    with torch.no_grad():
        y1, p1 = attn(a['x'], a['policy_ones'][:, :, None])
        y2, p2 = attn(a['x'], a['policy_topk'][:, :, None])
    out.update({'attn/out_ones': np32(y1), 'attn/out_topk': np32(y2)})
    for tag, p in (('ones', p1), ('topk', p2)):
        i, v, s = sample(p)
        out.update({f'attn/probs_{tag}_idx': i, f'attn/probs_{tag}_val': v, f'attn/probs_{tag}_sum': s})

    probs = gi.rollout_inputs()
    vit = ref_deit.MyVisionTransformer(img_size=32, patch_size=16, embed_dim=32, depth=1, num_heads=1, mlp_ratio=1, qkv_bias=True,
                                       norm_layer=partial(nn.LayerNorm, eps=1e-6), num_classes=2, drop_rate=0., drop_path_rate=0.)
    with torch.no_grad():
        R = vit.attn_rollout([p.clone() for p in probs])
    cls_attn = R[:, 0, 1:]
    idx = torch.topk(cls_attn, k=81, dim=-1)[1].sort(dim=-1)[0]
    out.update({'rollout/cls_token_attn': np32(cls_attn), 'rollout/idx': idx.numpy(), 'rollout/R_row5': np32(R[:, 5])})

    tok, protos = gi.proto_inputs()
    pp = ref.PPNet.__new__(ref.PPNet)
    nn.Module.__init__(pp)
    pp.epsilon = 1e-4; pp.prototype_activation_function = 'log'; pp.use_global = True
    with torch.no_grad():
        act_max, (dist, act) = pp.get_activations(tok, protos)
    i, v, s = sample(dist)
    out.update({'proto/act_max': np32(act_max), 'proto/dist_idx': i, 'proto/dist_val': v, 'proto/dist_sum': s,
                'proto/dist_b0_p5': np32(dist[0, 5]), 'proto/dist_b1_p7': np32(dist[1, 7]),
                'proto/act_b0_p0_20': np32(act[0, :20])})

    tpa, roll, lab = gi.ppc_inputs()
    pp.num_prototypes_per_class = 10; pp.ppc_cov_thresh = 1.; pp.ppc_mean_thresh = 2.
    with torch.no_grad():
        cov, mean = pp.get_PPC_loss(tpa, roll, 196, lab)
    out.update({'ppc/cov': np32(cov), 'ppc/mean': np32(mean)})

    c = gi.cait_inputs()
    th = ref_cait.TalkingHeadAttn(c['D'], num_heads=c['H'], qkv_bias=True); th.load_state_dict(c['th'])
    ca = ref_cait.ClassAttn(c['D'], num_heads=c['H'], qkv_bias=True); ca.load_state_dict(c['ca'])
    with torch.no_grad():
        yt, pt = th(c['x'])
        yc, pc = ca(c['u'], c['policy'][:, :, None])
        yc1, pc1 = ca(c['u'], torch.ones(c['B'], c['N'] + 1, 1))
    i, v, s = sample(pt)
    out.update({'cait/th_out': np32(yt), 'cait/th_attn_idx': i, 'cait/th_attn_val': v, 'cait/th_attn_sum': s,
                'cait/ca_out': np32(yc), 'cait/ca_attn': np32(pc), 'cait/ca_out_ones': np32(yc1), 'cait/ca_attn_ones': np32(pc1)})
    mc = ref_cait.MyCait(img_size=32, patch_size=16, embed_dim=32, depth=1, num_heads=1, init_scale=1e-5, num_classes=2,
                         drop_rate=0., drop_path_rate=0.)
    with torch.no_grad():
        _, cls_res = mc.attn_rollout_cait([p.clone() for p in c['sa']] + [p.clone() for p in c['cas']], discard_ratio=0.9,
                                          head_fusion='mean', layer_nums=[4, 1])
    out.update({'cait/rollout_cls': np32(cls_res[:, 0])})
    np.savez_compressed(path, **out)
    print('wrote', path, f'{os.path.getsize(path) / 1e6:.2f} MB')


if __name__ == '__main__':
    torch.set_num_threads(4)
    micro_fixture('deit', os.path.join(HERE, 'micro_deit.npz'), seed=101)
    micro_fixture('cait', os.path.join(HERE, 'micro_cait.npz'), seed=202)
    # the reference's DEFAULT add-on head (protopformer.py:90-107): 64 -> 32 ReLU 32 -> 32 ReLU, 32 -> 16 ReLU 16 -> 16 Sigmoid
    micro_fixture('deit', os.path.join(HERE, 'micro_deit_bottleneck.npz'), seed=303, add_on='bottleneck', proto_dim=16)
    ops_fixture(os.path.join(HERE, 'ops_real.npz'))
