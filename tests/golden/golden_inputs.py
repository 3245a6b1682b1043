"""Seeded inputs of the op-level golden vectors (tests/golden/ops_real.npz).

Shared by make_golden.py (which feeds them to the reference) and by the tests (which feed them to the
oracle / the HIP path), so the fixture only has to store outputs.  Pure torch CPU generator code."""
import torch


def attn_inputs():
    D, H, N, B = 192, 3, 197, 2
    g = torch.Generator().manual_seed(11)
    w = {
        "qkv.weight": 0.2 * torch.randn(3 * D, D, generator=g), "qkv.bias": 0.1 * torch.randn(3 * D, generator=g),
        "proj.weight": 0.1 * torch.randn(D, D, generator=g), "proj.bias": 0.1 * torch.randn(D, generator=g),
    }
    x = torch.randn(B, N, D, generator=g)
    keep = torch.stack([torch.randperm(N - 1, generator=g)[:81].sort()[0] + 1 for _ in range(B)])
    pol = torch.zeros(B, N)
    pol[:, 0] = 1
    pol.scatter_(1, keep, 1.0)
    return dict(D=D, H=H, N=N, B=B, w=w, x=x, keep=keep, policy_topk=pol, policy_ones=torch.ones(B, N))


def rollout_inputs():
    B, H, N = 2, 3, 197
    g = torch.Generator().manual_seed(13)
    return [torch.softmax(3.0 * torch.randn(B, H, N, N, generator=g), dim=-1) for _ in range(11)]


def proto_inputs():
    B, k, Dp, P = 2, 81, 192, 2000
    g = torch.Generator().manual_seed(17)
    tok = torch.rand(B, Dp, 9, 9, generator=g)
    protos = torch.rand(P, Dp, 1, 1, generator=g)
    tok[0, :, 2, 3] = protos[5, :, 0, 0]      # token == prototype (near-zero distance)
    tok[1, :, 0, 0] = 0.5
    protos[7] = 0.5                           # exactly representable => d == 0
    return tok, protos


def ppc_inputs():
    g = torch.Generator().manual_seed(19)
    tpa = 4.0 * torch.rand(4, 200, 9, 9, generator=g)
    roll = torch.rand(4, 196, generator=g)
    lab = torch.tensor([3, 19, 0, 3])
    return tpa, roll, lab


def cait_inputs():
    Dc, Hc, Nc, B = 192, 4, 196, 2
    g = torch.Generator().manual_seed(23)

    def mk(names_shapes):
        out = {}
        for n_, shp in names_shapes:
            if "proj_l.weight" in n_ or "proj_w.weight" in n_:
                out[n_] = torch.eye(Hc) + 0.3 * torch.randn(Hc, Hc, generator=g)
            elif len(shp) == 2:
                out[n_] = 0.15 * torch.randn(shp, generator=g)
            else:
                out[n_] = 0.1 * torch.randn(shp, generator=g)
        return out
    th = mk([("qkv.weight", (3 * Dc, Dc)), ("qkv.bias", (3 * Dc,)), ("proj.weight", (Dc, Dc)), ("proj.bias", (Dc,)),
             ("proj_l.weight", (Hc, Hc)), ("proj_l.bias", (Hc,)), ("proj_w.weight", (Hc, Hc)), ("proj_w.bias", (Hc,))])
    ca = mk([("q.weight", (Dc, Dc)), ("q.bias", (Dc,)), ("k.weight", (Dc, Dc)), ("k.bias", (Dc,)),
             ("v.weight", (Dc, Dc)), ("v.bias", (Dc,)), ("proj.weight", (Dc, Dc)), ("proj.bias", (Dc,))])
    xc = torch.randn(B, Nc, Dc, generator=g)
    uc = torch.randn(B, Nc + 1, Dc, generator=g)
    keepc = torch.stack([torch.randperm(Nc, generator=g)[:121].sort()[0] + 1 for _ in range(B)])
    polc = torch.zeros(B, Nc + 1)
    polc[:, 0] = 1
    polc.scatter_(1, keepc, 1.0)
    g2 = torch.Generator().manual_seed(29)
    sa = [torch.softmax(3.0 * torch.randn(B, Hc, Nc, Nc, generator=g2), dim=-1) + 0.02 * torch.randn(B, Hc, Nc, Nc, generator=g2)
          for _ in range(4)]
    cas = [torch.softmax(2.0 * torch.randn(B, Hc, 1, Nc + 1, generator=g2), dim=-1)]
    return dict(D=Dc, H=Hc, N=Nc, B=B, th=th, ca=ca, x=xc, u=uc, keep=keepc, policy=polc, sa=sa, cas=cas)


def sample_indices(numel, n=4096, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.randperm(numel, generator=g)[:n].sort()[0]
