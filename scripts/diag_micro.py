import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from helpers import micro
from oracle import ppf_oracle as O
from test_gpu_e2e import _build
sd, cfg, z = micro("micro_deit.npz")
m = _build(cfg, sd).eval()
img = torch.from_numpy(z["img"])
with torch.no_grad():
    f, cls_attn, idx = m._tokens(img.cuda())
    out = O.ppnet_forward(sd, img, cfg, train=False)
fo = torch.cat([out["cls_tokens"], out["tokens"]], 1)
print("idx equal", torch.equal(idx.cpu().long(), out["reserve_idx"]))
print("f err", float((f.cpu() - fo).abs().max()), "f max", float(fo.abs().max()))
logits, (ca, dist, lg, ll) = m(img.cuda())
d_ref = out["distances"]
print("dist err", float((dist.cpu() - d_ref).abs().max()), "dist max", float(d_ref.max()), "dist min", float(d_ref.min()))
_, acts = m.push_forward(img.cuda())
a_ref = out["total_proto_act"]
print("act err", float((acts.cpu() - a_ref).abs().max()), "act max", float(a_ref.max()))
# oracle act from MY f tokens: isolates the proto kernel
mx, d2, a2 = O.proto_activations(f.cpu()[:, 1:], sd["prototype_vectors"])
print("kernel-vs-oracle on same tokens: dist", float((dist.cpu().flatten(2) - d2).abs().max()), "act", float((acts.cpu().flatten(2) - a2).abs().max()))
x = out["x"]
