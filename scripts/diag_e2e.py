"""Diagnostic (not a test): per-parameter gradient agreement of the HIP train step vs the fp32 oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import ppf_oracle as O
from protopformer_amd.protopformer import CrossEntropyLoss, construct_PPNet

arch = sys.argv[1] if len(sys.argv) > 1 else "deit_tiny_patch16_224"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
cfg = O.make_cfg(arch, 200, 64, 20, 11 if "deit" in arch else 1, 81 if "deit" in arch else 121, global_per_class=5)
sd = O.init_state_dict(cfg, seed=3)
g = torch.Generator().manual_seed(5)
for k_ in sd:
    if k_.endswith(".bias") and "add_on" not in k_:
        sd[k_] = 0.05 * torch.randn(sd[k_].shape, generator=g)
    if "qkv.weight" in k_:
        sd[k_] = sd[k_] * 6.0
m = construct_PPNet(arch, pretrained=False, prototype_shape=(200, 64, 1, 1), num_classes=20, reserve_layers=[cfg["reserve_layer"]],
                    reserve_token_nums=[cfg["reserve_k"]], use_global=True, use_ppc_loss=True, global_proto_per_class=5, add_on_layers_type="regular")
m.load_state_dict(sd, strict=True)
m = m.cuda().train()
for blk in m.features.blocks:
    blk.drop_path_rate = 0.0
img = torch.randn(B, 3, 224, 224, generator=g); label = torch.randint(0, 20, (B,), generator=g)
logits, aux = m(img.cuda())
ce = CrossEntropyLoss()(logits, label.cuda())
cov, mean = m.get_PPC_loss(aux[2], aux[3], aux[4], label.cuda())
(ce + 0.1 * cov + 0.5 * mean).backward()
my_idx = m._ppc_cache[1].cpu().long()
params = {k: v.clone().requires_grad_(k not in O.FROZEN_KEYS) for k, v in sd.items()}
out = O.ppnet_forward(params, img, cfg, train=True, force_idx=my_idx)
loss_ref, parts = O.train_loss(out, label, cfg, with_ppc=True)
loss_ref.backward()
print("loss", float(ce), float(parts["ce"]), float(cov), float(parts["ppc_cov"]), float(mean), float(parts["ppc_mean"]))
print("logits relerr", float((logits.cpu() - out["logits"]).abs().max() / out["logits"].abs().max()))
tp = aux[2].cpu(); to = out["total_proto_act"]
print("act relerr", float((tp - to).abs().max() / to.abs().max()), "argmax agree", float((tp.flatten(2).argmax(-1) == to.flatten(2).argmax(-1)).float().mean()))
rows = []
for name, p in m.named_parameters():
    if not p.requires_grad:
        continue
    gm = p.grad.float().cpu().reshape(-1); gr = params[name].grad.reshape(-1)
    rel = float((gm - gr).abs().max() / gr.abs().max().clamp_min(1e-30))
    cos = float(torch.dot(gm, gr) / (gm.norm() * gr.norm()).clamp_min(1e-30))
    rows.append((rel, cos, name, float(gr.abs().max())))
rows.sort(reverse=True)
for rel, cos, name, mx in rows[:25]:
    print(f"{rel:9.3e} cos={cos:.5f} max|g|={mx:.2e} {name}")
print("median rel", sorted(r[0] for r in rows)[len(rows) // 2], "min cos", min(r[1] for r in rows))

# ---- backbone-only gradient check: L = sum(w * f), no max-pool routing ambiguity
m.flat_store().zero_grad()
f, cls_attn, idx = m._tokens(img.cuda())
w = torch.randn(f.shape, generator=g)
(f * w.cuda()).sum().backward()
params = {k: v.clone().requires_grad_(k not in O.FROZEN_KEYS) for k, v in sd.items()}
out = O.ppnet_forward(params, img, cfg, train=True, force_idx=idx.cpu().long())
fo = torch.cat([out["cls_tokens"], out["tokens"]], dim=1)
print("f relerr", float((f.detach().cpu() - fo).abs().max() / fo.abs().max()))
(fo * w).sum().backward()
rows = []
for name, p in m.named_parameters():
    if not p.requires_grad or params[name].grad is None:
        continue
    gm = p.grad.float().cpu().reshape(-1); gr = params[name].grad.reshape(-1)
    rel = float((gm - gr).abs().max() / gr.abs().max().clamp_min(1e-30))
    cos = float(torch.dot(gm, gr) / (gm.norm() * gr.norm()).clamp_min(1e-30))
    rows.append((rel, cos, name, float(gr.abs().max())))
rows.sort(reverse=True)
print("---- backbone-only")
for rel, cos, name, mx in rows[:15]:
    print(f"{rel:9.3e} cos={cos:.5f} max|g|={mx:.2e} {name}")
print("median rel", sorted(r[0] for r in rows)[len(rows) // 2], "min cos", min(r[1] for r in rows))
