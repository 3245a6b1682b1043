import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch
from oracle import ppf_oracle as O
from test_gpu_baseline_configs import CFG, _construct
from protopformer_amd.protopformer import CrossEntropyLoss
c = CFG[sys.argv[1] if len(sys.argv) > 1 else "deit_base"]
cfg = O.make_cfg(c["arch"], c["P"], c["Dp"], c["C"], c["layer"], c["k"], global_per_class=c["gpc"])
sd = O.init_state_dict(cfg, seed=1028)
m = _construct(c, 0, sd)
for blk in m.features.blocks: blk.drop_path_rate = 0.0
g = torch.Generator().manual_seed(77)
img = torch.randn(2, 3, 224, 224, generator=g); label = torch.tensor([5, c["C"] - 2])
logits, aux = m(img.cuda())
ce = CrossEntropyLoss()(logits, label.cuda()); cov, mean = m.get_PPC_loss(aux[2], aux[3], aux[4], label.cuda())
(ce + 0.1 * cov + 0.5 * mean).backward()
my_idx = m._ppc_cache[1].cpu().long()
params = {k_: v.clone().requires_grad_(k_ not in O.FROZEN_KEYS) for k_, v in sd.items()}
out = O.ppnet_forward(params, img, cfg, train=True, force_idx=my_idx)
loss_ref, parts = O.train_loss(out, label, cfg, with_ppc=True); loss_ref.backward()
rows = []
for nm, p in m.named_parameters():
    if p.requires_grad and params[nm].grad is not None and float(params[nm].grad.abs().max()) > 1e-12:
        gm, gr = p.grad.float().cpu().reshape(-1), params[nm].grad.reshape(-1)
        rows.append((float(torch.dot(gm, gr) / (gm.norm() * gr.norm()).clamp_min(1e-30)), nm, float(gr.norm()), float(gm.norm())))
rows.sort()
for r in rows[:25]: print(f"{r[0]:.5f} {r[1]:50s} |ref|={r[2]:.3e} |hip|={r[3]:.3e}")
