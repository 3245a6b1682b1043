"""CPU oracle for the ProtoPFormer hot path -- TEST INFRASTRUCTURE, NOT PRODUCT.

A plain fp32 PyTorch-CPU restatement of the reference's forward / loss arithmetic, written
functionally over a flat ``state_dict`` (reference key names) so that it shares no code or
structure with either the reference classes or the shipped HIP-backed modules.  Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.

Pinning status
--------------
* Pinned against outputs of the reference itself: ``tests/golden/make_golden.py`` imports
  ``/root/reference/protopformer.py`` in the build container and stores inputs/outputs under
  ``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks every function here against them.
* **Parity unpinned at the timm boundary**: the reference depends on ``timm==0.5.4``
  (README.md:59; not vendored, not installable here).  ``PatchEmbed`` / ``Mlp`` / ``DropPath`` /
  the ``VisionTransformer`` and ``Cait`` constructors used while generating the fixtures are
  stand-ins written from the timm 0.5.4 public API (tests/golden/_timm_standins.py), i.e. the
  oracle is pinned to reference code under /root/reference, and to torch.nn primitives for
  the timm pieces (conv2d k=s=16, Linear-GELU(erf)-Linear, per-sample stochastic depth).

Every function cites the reference lines it restates (paths relative to /root/reference).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]

LN_EPS = 1e-6          # tools/deit_models_attn.py:289, tools/cait_models_attn.py:191
ROLLOUT_DISCARD = 0.9  # tools/deit_models_attn.py:99
ROLLOUT_IDENTITY = 0.2  # tools/deit_models_attn.py:118
PROTO_EPS = 1e-4       # protopformer.py:41


# --------------------------------------------------------------------------------------
# small pieces
# --------------------------------------------------------------------------------------
def layer_norm(x: Tensor, w: Tensor, b: Tensor) -> Tensor:
    """nn.LayerNorm(eps=1e-6) over the last dim (tools/deit_models_attn.py:67,72,289)."""
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) * torch.rsqrt(var + LN_EPS) * w + b


def linear(x: Tensor, w: Tensor, b: Optional[Tensor]) -> Tensor:
    y = x @ w.t()
    return y if b is None else y + b


def gelu_erf(x: Tensor) -> Tensor:
    """nn.GELU() (exact erf form) used by timm Mlp (tools/deit_models_attn.py:87)."""
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


def patch_embed(sd: SD, img: Tensor, pre: str = "features.") -> Tensor:
    """timm PatchEmbed: Conv2d(k=s=patch) -> flatten(2).transpose(1,2)  (deit:174, cait:305).

    Restated as an explicit im2col + matmul (the form the HIP path uses)."""
    w = sd[pre + "patch_embed.proj.weight"]          # (D, 3, p, p)
    b = sd[pre + "patch_embed.proj.bias"]
    D, C, p, _ = w.shape
    B, _, Hh, Ww = img.shape
    gh, gw = Hh // p, Ww // p
    cols = img.reshape(B, C, gh, p, gw, p).permute(0, 2, 4, 1, 3, 5).reshape(B, gh * gw, C * p * p)
    return cols @ w.reshape(D, -1).t() + b


def policy_softmax(scores: Tensor, policy: Tensor, self_keep: bool, eps: float = 1e-6) -> Tensor:
    """softmax_with_policy (deit:29-43 with the identity term; cait:50-69 without it).

    scores (B,H,M,N) fp32, policy (B,N) in {0,1}."""
    B, H, M, N = scores.shape
    keep = policy.reshape(B, 1, 1, N).to(torch.float32)
    if self_keep:
        eye = torch.eye(N, dtype=torch.float32).reshape(1, 1, N, N)
        keep = keep + (1.0 - keep) * eye
    mx = scores.max(dim=-1, keepdim=True)[0]
    e = torch.exp(scores - mx) * keep
    return (e + eps / N) / (e.sum(dim=-1, keepdim=True) + eps)


# --------------------------------------------------------------------------------------
# DeiT backbone (tools/deit_models_attn.py)
# --------------------------------------------------------------------------------------
def deit_attention(sd: SD, pre: str, x: Tensor, heads: int, policy: Tensor) -> Tuple[Tensor, Tensor]:
    """Attention.forward (deit:45-60). Returns (out, probs (B,H,N,N))."""
    B, N, C = x.shape
    hd = C // heads
    qkv = linear(x, sd[pre + "qkv.weight"], sd[pre + "qkv.bias"]).reshape(B, N, 3, heads, hd)
    q = qkv[:, :, 0].transpose(1, 2)
    k = qkv[:, :, 1].transpose(1, 2)
    v = qkv[:, :, 2].transpose(1, 2)
    s = (q @ k.transpose(-1, -2)) * (hd ** -0.5)
    p = policy_softmax(s, policy, self_keep=True)
    o = (p @ v).transpose(1, 2).reshape(B, N, C)
    return linear(o, sd[pre + "proj.weight"], sd[pre + "proj.bias"]), p


def mlp(sd: SD, pre: str, x: Tensor) -> Tensor:
    h = gelu_erf(linear(x, sd[pre + "fc1.weight"], sd[pre + "fc1.bias"]))
    return linear(h, sd[pre + "fc2.weight"], sd[pre + "fc2.bias"])


def deit_block(sd: SD, pre: str, x: Tensor, heads: int, policy: Tensor,
               keep1: Optional[Tensor] = None, keep2: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
    """Block.forward (deit:76-81). keep1/keep2: optional per-sample DropPath scale (B,) = mask/keep_prob."""
    a, p = deit_attention(sd, pre + "attn.", layer_norm(x, sd[pre + "norm1.weight"], sd[pre + "norm1.bias"]), heads, policy)
    if keep1 is not None:
        a = a * keep1.reshape(-1, 1, 1)
    x = x + a
    m = mlp(sd, pre + "mlp.", layer_norm(x, sd[pre + "norm2.weight"], sd[pre + "norm2.bias"]))
    if keep2 is not None:
        m = m * keep2.reshape(-1, 1, 1)
    return x + m, p


def rollout_discard_normalize(fused: Tensor, discard_ratio: float = ROLLOUT_DISCARD) -> Tensor:
    """One layer of attn_rollout before the chain product (deit:110-121 / cait:236-248).

    fused (B,R,N): head-fused attention (R = N for self-attn, 1 for a class-attn row).  The
    ``int(R*N*ratio)`` smallest entries of each sample are zeroed, ``0.2*I[:R]`` added, /1.2,
    rows normalised."""
    B, R, N = fused.shape
    flat = fused.reshape(B, R * N).clone()
    kdrop = int(R * N * discard_ratio)
    if kdrop > 0:
        idx = flat.topk(kdrop, dim=-1, largest=False)[1]
        flat.scatter_(1, idx, 0.0)
    a = flat.reshape(B, R, N)
    eye = torch.eye(N, dtype=a.dtype)[:R]
    a = (a + ROLLOUT_IDENTITY * eye) / (1.0 + ROLLOUT_IDENTITY)
    return a / a.sum(dim=-1, keepdim=True)


def deit_rollout(probs: Sequence[Tensor]) -> Tensor:
    """attn_rollout (deit:99-124), head_fusion='mean': R = a_{L-1} ... a_0. Returns (B,N,N)."""
    B, _, N, _ = probs[0].shape
    R = torch.eye(N).unsqueeze(0).repeat(B, 1, 1)
    for p in probs:
        R = rollout_discard_normalize(p.mean(dim=1)) @ R
    return R


def topk_sorted(scores: Tensor, k: int) -> Tensor:
    """topk(k) followed by an ascending sort of the indices (deit:229-230, protopformer.py:157-158,273-274)."""
    return scores.topk(k, dim=-1)[1].sort(dim=-1)[0]


def deit_features(sd: SD, img: Tensor, heads: int, depth: int, reserve_layer: int, reserve_k: int,
                  droppath: Optional[List[Tuple[Optional[Tensor], Optional[Tensor]]]] = None,
                  pre: str = "features.", return_probs: bool = False, force_idx: Optional[Tensor] = None):
    """forward_feature_patch_embed_all + forward_feature_mask_train_direct (deit:172-181, 209-240).

    Returns x (B,1+Np,D) after the final norm, cls_token_attn (B,Np) and the reserved indices (B,k)."""
    B = img.shape[0]
    tok = patch_embed(sd, img, pre)
    x = torch.cat([sd[pre + "cls_token"].expand(B, -1, -1), tok], dim=1) + sd[pre + "pos_embed"]
    N = x.shape[1]
    policy = torch.ones(B, N)
    probs: List[Tensor] = []
    cls_attn = idx = None
    for i in range(depth):
        if i == reserve_layer:
            R = deit_rollout([p.detach() for p in probs[:i]])
            cls_attn = R[:, 0, 1:]
            # force_idx (tests only): follow a given token reservation, e.g. the one a bf16 run selected
            idx = topk_sorted(cls_attn, reserve_k) if force_idx is None else force_idx
            policy = torch.zeros(B, N)
            policy[:, 0] = 1.0
            policy.scatter_(1, idx + 1, 1.0)
        k1, k2 = (None, None) if droppath is None else droppath[i]
        x, p = deit_block(sd, f"{pre}blocks.{i}.", x, heads, policy, k1, k2)
        probs.append(p)
    x = layer_norm(x, sd[pre + "norm.weight"], sd[pre + "norm.bias"])
    if return_probs:
        return x, cls_attn, idx, probs
    return x, cls_attn, idx


# --------------------------------------------------------------------------------------
# CaiT backbone (tools/cait_models_attn.py)
# --------------------------------------------------------------------------------------
def cait_talking_heads_attention(sd: SD, pre: str, x: Tensor, heads: int) -> Tuple[Tensor, Tensor]:
    """TalkingHeadAttn.forward (cait:115-132). Returns (out, post-proj_w attention (B,H,N,N))."""
    B, N, C = x.shape
    hd = C // heads
    qkv = linear(x, sd[pre + "qkv.weight"], sd[pre + "qkv.bias"]).reshape(B, N, 3, heads, hd)
    q = qkv[:, :, 0].transpose(1, 2) * (hd ** -0.5)
    k = qkv[:, :, 1].transpose(1, 2)
    v = qkv[:, :, 2].transpose(1, 2)
    s = q @ k.transpose(-1, -2)                                       # (B,H,N,N)
    s = torch.einsum("bhnm,gh->bgnm", s, sd[pre + "proj_l.weight"]) + sd[pre + "proj_l.bias"].reshape(1, -1, 1, 1)
    p = s.softmax(dim=-1)
    p = torch.einsum("bhnm,gh->bgnm", p, sd[pre + "proj_w.weight"]) + sd[pre + "proj_w.bias"].reshape(1, -1, 1, 1)
    o = (p @ v).transpose(1, 2).reshape(B, N, C)
    return linear(o, sd[pre + "proj.weight"], sd[pre + "proj.bias"]), p


def cait_class_attention(sd: SD, pre: str, u: Tensor, heads: int, policy: Tensor) -> Tuple[Tensor, Tensor]:
    """ClassAttn.forward (cait:71-90): query = token 0 only. Returns (cls_out (B,1,C), attn (B,H,1,N))."""
    B, N, C = u.shape
    hd = C // heads
    q = linear(u[:, 0], sd[pre + "q.weight"], sd[pre + "q.bias"]).reshape(B, 1, heads, hd).transpose(1, 2) * (hd ** -0.5)
    k = linear(u, sd[pre + "k.weight"], sd[pre + "k.bias"]).reshape(B, N, heads, hd).transpose(1, 2)
    v = linear(u, sd[pre + "v.weight"], sd[pre + "v.bias"]).reshape(B, N, heads, hd).transpose(1, 2)
    s = q @ k.transpose(-1, -2)
    p = policy_softmax(s, policy, self_keep=False)
    o = (p @ v).transpose(1, 2).reshape(B, 1, C)
    return linear(o, sd[pre + "proj.weight"], sd[pre + "proj.bias"]), p


def cait_sa_block(sd: SD, pre: str, x: Tensor, heads: int,
                  keep1: Optional[Tensor] = None, keep2: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
    """LayerScaleBlock.forward (cait:153-158)."""
    a, p = cait_talking_heads_attention(sd, pre + "attn.", layer_norm(x, sd[pre + "norm1.weight"], sd[pre + "norm1.bias"]), heads)
    a = sd[pre + "gamma_1"] * a
    if keep1 is not None:
        a = a * keep1.reshape(-1, 1, 1)
    x = x + a
    m = sd[pre + "gamma_2"] * mlp(sd, pre + "mlp.", layer_norm(x, sd[pre + "norm2.weight"], sd[pre + "norm2.bias"]))
    if keep2 is not None:
        m = m * keep2.reshape(-1, 1, 1)
    return x + m, p


def cait_ca_block(sd: SD, pre: str, x: Tensor, cls: Tensor, heads: int, policy: Tensor) -> Tuple[Tensor, Tensor]:
    """LayerScaleBlockClassAttn.forward (cait:179-185); drop_path is 0 for these blocks (cait:217)."""
    u = torch.cat([cls, x], dim=1)
    a, p = cait_class_attention(sd, pre + "attn.", layer_norm(u, sd[pre + "norm1.weight"], sd[pre + "norm1.bias"]), heads, policy)
    cls = cls + sd[pre + "gamma_1"] * a
    cls = cls + sd[pre + "gamma_2"] * mlp(sd, pre + "mlp.", layer_norm(cls, sd[pre + "norm2.weight"], sd[pre + "norm2.bias"]))
    return cls, p


def cait_rollout(sa_probs: Sequence[Tensor], ca_probs: Sequence[Tensor]) -> Tensor:
    """attn_rollout_cait with head_fusion='mean' (cait:223-261). Returns cls_result (B,1,Np)."""
    B, _, N, _ = sa_probs[0].shape
    R = torch.eye(N).unsqueeze(0).repeat(B, 1, 1)
    for p in sa_probs:
        R = rollout_discard_normalize(p.mean(dim=1)) @ R
    rows = torch.cat([rollout_discard_normalize(p.mean(dim=1)) for p in ca_probs], dim=1)   # (B,n_ca,1+Np)
    row = rows.mean(dim=1, keepdim=True)[:, :, 1:]
    return row @ R


def cait_features(sd: SD, img: Tensor, heads: int, depth: int, reserve_layer: int, reserve_k: int,
                  droppath: Optional[List[Tuple[Optional[Tensor], Optional[Tensor]]]] = None,
                  depth_token_only: int = 2, pre: str = "features.", return_probs: bool = False,
                  force_idx: Optional[Tensor] = None):
    """forward_feature_patch_embed_all + forward_feature_mask_train_direct (cait:303-345)."""
    B = img.shape[0]
    x = patch_embed(sd, img, pre) + sd[pre + "pos_embed"]
    cls = sd[pre + "cls_token"].expand(B, -1, -1)
    Np = x.shape[1]
    sa_probs: List[Tensor] = []
    for i in range(depth):
        k1, k2 = (None, None) if droppath is None else droppath[i]
        x, p = cait_sa_block(sd, f"{pre}blocks.{i}.", x, heads, k1, k2)
        sa_probs.append(p)
    policy = torch.ones(B, 1 + Np)
    ca_probs: List[Tensor] = []
    cls_attn = idx = None
    for i in range(depth_token_only):
        if i == reserve_layer:
            # the reference's slice [depth:] of all_attn holds the CA rows produced so far (cait:328,249)
            res = cait_rollout([p.detach() for p in sa_probs], [p.detach() for p in ca_probs])
            cls_attn = res[:, 0]
            idx = topk_sorted(cls_attn, reserve_k) if force_idx is None else force_idx
            policy = torch.zeros(B, 1 + Np)
            policy[:, 0] = 1.0
            policy.scatter_(1, idx + 1, 1.0)
        cls, p = cait_ca_block(sd, f"{pre}blocks_token_only.{i}.", x, cls, heads, policy)
        ca_probs.append(p)
    x = layer_norm(torch.cat([cls, x], dim=1), sd[pre + "norm.weight"], sd[pre + "norm.bias"])
    if return_probs:
        return x, cls_attn, idx, sa_probs, ca_probs
    return x, cls_attn, idx


# --------------------------------------------------------------------------------------
# Prototype layer (protopformer.py)
# --------------------------------------------------------------------------------------
def addon_sigmoid(sd: SD, tokens: Tensor) -> Tensor:
    """add_on_layers on (B,T,D) tokens (protopformer.py:171-172).  'regular' (protopformer.py:111-114): Conv2d 1x1 + Sigmoid.
    'bottleneck' (protopformer.py:90-107, the signature default): pairs of 1x1 convolutions, the width halved pair by pair down to the
    prototype dimension, ReLU after every convolution except the last, which is followed by the Sigmoid.  The structure is read off the
    state dict: Sequential indices 0, 2, 4, ... hold the convolutions (the activations between them carry no parameters)."""
    idxs = sorted({int(k.split(".")[1]) for k in sd if k.startswith("add_on_layers.") and k.endswith(".weight")})
    x = tokens
    for n, i in enumerate(idxs):
        w = sd[f"add_on_layers.{i}.weight"]
        x = linear(x, w.reshape(w.shape[0], -1), sd[f"add_on_layers.{i}.bias"])
        x = torch.sigmoid(x) if n == len(idxs) - 1 else torch.relu(x)
    return x


def l2_distances(tokens: Tensor, protos: Tensor) -> Tensor:
    """_l2_convolution_single (protopformer.py:201-218): relu(|x|^2 - 2 x.p + |p|^2).

    tokens (B,T,Dp), protos (P,Dp[,1,1]) -> (B,P,T)."""
    pm = protos.reshape(protos.shape[0], -1)
    x2 = (tokens ** 2).sum(-1)                       # (B,T)   the ones-conv of the reference
    p2 = (pm ** 2).sum(-1)                           # (P,)
    xp = torch.einsum("btd,pd->bpt", tokens, pm)
    return F.relu(x2.unsqueeze(1) + (-2.0 * xp + p2.reshape(1, -1, 1)))


def log_similarity(d: Tensor, eps: float = PROTO_EPS) -> Tensor:
    """distance_2_similarity, 'log' (protopformer.py:228-230)."""
    return torch.log((d + 1.0) / (d + eps))


def proto_activations(tokens: Tensor, protos: Tensor, activation: str = "log") -> Tuple[Tensor, Tensor, Tensor]:
    """get_activations (protopformer.py:236-247): returns (max over tokens (B,P), distances (B,P,T), act (B,P,T))."""
    d = l2_distances(tokens, protos)
    act = log_similarity(d) if activation == "log" else -d
    return act.max(dim=-1)[0], d, act


def ppnet_forward(sd: SD, img: Tensor, cfg: dict, train: bool = True,
                  droppath: Optional[list] = None, force_idx: Optional[Tensor] = None, force_argmax: Optional[Tensor] = None) -> dict:
    """PPNet.forward (protopformer.py:290-335) for either backbone.

    cfg keys: arch ('deit'|'cait'), heads, depth, reserve_layer, reserve_k, global_coe.
    force_argmax (tests only, (B, P) int64): route the max-pool of the local branch through the given token per (sample, prototype)
    -- e.g. the arg-max a bf16 run took -- instead of this run's own arg-max (max_pool2d is discontinuous at near-ties)."""
    feats = deit_features if cfg["arch"] == "deit" else cait_features
    x, cls_attn, idx = feats(sd, img, cfg["heads"], cfg["depth"], cfg["reserve_layer"], cfg["reserve_k"], droppath,
                             force_idx=force_idx)
    B, _, D = x.shape
    cls_tok = x[:, :1]
    img_tok = torch.gather(x[:, 1:], 1, idx[:, :, None].expand(-1, -1, D))        # protopformer.py:156-162
    cls_f = addon_sigmoid(sd, cls_tok)
    img_f = addon_sigmoid(sd, img_tok)
    g_act, _, _ = proto_activations(cls_f, sd["prototype_vectors_global"])
    l_act, dist, act = proto_activations(img_f, sd["prototype_vectors"])
    if force_argmax is not None:
        l_act = act.gather(-1, force_argmax.unsqueeze(-1)).squeeze(-1)
    lg = g_act @ sd["last_layer_global.weight"].t()
    ll = l_act @ sd["last_layer.weight"].t()
    g = cfg["global_coe"]
    logits = g * lg + (1.0 - g) * ll
    side = int(round(math.sqrt(act.shape[-1])))
    return {
        "logits": logits, "logits_global": lg, "logits_local": ll,
        "cls_token_attn": cls_attn.detach(), "reserve_idx": idx,
        "distances": dist.reshape(B, -1, side, side),
        "total_proto_act": act.reshape(B, -1, side, side),
        "tokens": img_f, "cls_tokens": cls_f, "x": x,
    }


# --------------------------------------------------------------------------------------
# PPC loss (protopformer.py:249-288)
# --------------------------------------------------------------------------------------
def weighted_grid_moments(weights: Tensor, side: int) -> Tuple[Tensor, Tensor]:
    """batch_cov (protopformer.py:249-257) on the side x side integer grid; weights (R, side*side).

    Returns mean (R,2) and covariance (R,2,2) with the reference's normalisation
    (weights rescaled to sum N; mean = average of points*w; cov / (N-1))."""
    N = side * side
    ii = torch.arange(side, dtype=torch.float32)
    pts = torch.stack([ii.repeat_interleave(side), ii.repeat(side)], dim=-1)     # (N,2): (row, col) of flat index
    w = weights / weights.sum(dim=-1, keepdim=True) * N
    mean = (pts.unsqueeze(0) * w.unsqueeze(-1)).mean(dim=1)                      # (R,2)
    diff = pts.unsqueeze(0) - mean.unsqueeze(1)                                  # (R,N,2)
    cov = torch.einsum("rn,rni,rnj->rij", w, diff, diff) / (N - 1)
    return mean, cov


def ppc_loss(total_proto_act: Tensor, cls_attn_rollout: Tensor, original_fea_len: int, label: Tensor,
             protos_per_class: int, cov_thresh: float, mean_thresh: float, force_idx: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
    """get_PPC_loss (protopformer.py:259-288)."""
    B = total_proto_act.shape[0]
    side = int(original_fea_len ** 0.5)
    act = total_proto_act.flatten(2)                                   # (B,P,k)
    k = act.shape[-1]
    cols = label.reshape(B, 1) * protos_per_class + torch.arange(protos_per_class).reshape(1, -1)
    own = torch.gather(act, 1, cols[:, :, None].expand(-1, -1, k))     # (B,ppc,k)
    idx = topk_sorted(cls_attn_rollout, k) if force_idx is None else force_idx
    canvas = torch.zeros(B, protos_per_class, original_fea_len)
    canvas = canvas.scatter(2, idx[:, None, :].expand(-1, protos_per_class, -1), own)
    mean, cov = weighted_grid_moments(canvas.reshape(B * protos_per_class, -1), side)
    cov_loss = F.relu((cov[:, 0, 0] + cov[:, 1, 1]) / 2 - cov_thresh).mean()
    mu = mean.reshape(B, protos_per_class, 2)
    # torch.cdist restated explicitly (euclidean, p=2); like cdist's backward, a zero distance
    # (the diagonal, or coincident means) contributes zero gradient instead of sqrt'(0) = inf.
    diff = mu.unsqueeze(2) - mu.unsqueeze(1)
    d2 = (diff ** 2).sum(-1)
    pos = d2 > 0
    dist = torch.where(pos, torch.sqrt(torch.where(pos, d2, torch.ones_like(d2))), torch.zeros_like(d2))
    off = 1.0 - torch.eye(protos_per_class)
    mean_loss = F.relu((mean_thresh - dist) * off).mean()
    return cov_loss, mean_loss


def train_loss(out: dict, label: Tensor, cfg: dict, with_ppc: bool = True) -> Tuple[Tensor, dict]:
    """Loss of one train step (tools/engine_proto.py:51-64): CE + cov_coe*PPC_sigma + mean_coe*PPC_mu."""
    ce = F.cross_entropy(out["logits"], label)
    parts = {"ce": ce}
    loss = ce
    if with_ppc:
        cov, mean = ppc_loss(out["total_proto_act"], out["cls_token_attn"], out["cls_token_attn"].shape[-1], label,
                             cfg["protos_per_class"], cfg.get("ppc_cov_thresh", 1.0), cfg.get("ppc_mean_thresh", 2.0),
                             force_idx=out["reserve_idx"])
        parts.update(ppc_cov=cov, ppc_mean=mean)
        loss = loss + cfg.get("ppc_cov_coe", 0.1) * cov + cfg.get("ppc_mean_coe", 0.5) * mean
    return loss, parts


# --------------------------------------------------------------------------------------
# helpers used by the CPU baseline and the parity tests
# --------------------------------------------------------------------------------------
def droppath_scales(B: int, rates: Sequence[float], gen: torch.Generator) -> list:
    """Per-layer (keep1, keep2) per-sample scales of timm DropPath: floor(keep + U)/keep."""
    out = []
    for r in rates:
        if r <= 0.0:
            out.append((None, None))
            continue
        keep = 1.0 - r
        k1 = torch.floor(keep + torch.rand(B, generator=gen)) / keep
        k2 = torch.floor(keep + torch.rand(B, generator=gen)) / keep
        out.append((k1, k2))
    return out


def adamw_groups(sd: SD, lrs: Optional[dict] = None, default_wd: float = 0.05) -> List[dict]:
    """The reference's optimizer groups (tools/create_optimizer.py:27-39, main.py:364-366):
    features lr 1e-4 wd 1e-3; add_on lr 3e-3 wd 1e-3; prototypes(+global) lr 3e-3 wd = --weight_decay."""
    lrs = lrs or {"features": 1e-4, "add_on_layers": 3e-3, "prototype_vectors": 3e-3}
    feats = [v for k, v in sd.items() if k.startswith("features.")]
    addon = [v for k, v in sd.items() if k.startswith("add_on_layers.")]
    return [
        {"params": feats, "lr": lrs["features"], "weight_decay": 1e-3},
        {"params": addon, "lr": lrs["add_on_layers"], "weight_decay": 1e-3},
        {"params": [sd["prototype_vectors"]], "lr": lrs["prototype_vectors"], "weight_decay": default_wd},
        {"params": [sd["prototype_vectors_global"]], "lr": lrs["prototype_vectors"], "weight_decay": default_wd},
    ]


FROZEN_KEYS = ("ones", "last_layer.weight", "last_layer_global.weight")


def init_state_dict(cfg: dict, seed: int = 1028) -> SD:
    """Seeded random-init state dict with the reference's key names/shapes and initialisers
    (SURVEY.md 8(b),(d); protopformer.py:115-131,367-392; timm 0.5.4 ViT/CaiT init: trunc-normal
    std .02 linears, zero biases, LN 1/0)."""
    g = torch.Generator().manual_seed(seed)
    D, depth, heads, mlp_ratio = cfg["dim"], cfg["depth"], cfg["heads"], cfg.get("mlp_ratio", 4)
    p, img = cfg.get("patch", 16), cfg.get("img", 224)
    Np = (img // p) ** 2
    P, Dp, C = cfg["num_prototypes"], cfg["proto_dim"], cfg["num_classes"]
    gpc = cfg["global_per_class"]

    def tn(*shape):
        t = torch.empty(*shape)
        torch.nn.init.trunc_normal_(t, std=0.02, generator=g)
        return t

    sd: SD = {}
    f = "features."
    sd[f + "cls_token"] = tn(1, 1, D)
    sd[f + "pos_embed"] = tn(1, Np + (1 if cfg["arch"] == "deit" else 0), D)
    fan_in = 3 * p * p
    bound = 1.0 / math.sqrt(fan_in)
    sd[f + "patch_embed.proj.weight"] = (torch.rand(D, 3, p, p, generator=g) * 2 - 1) * bound
    sd[f + "patch_embed.proj.bias"] = (torch.rand(D, generator=g) * 2 - 1) * bound

    def block(pre, attn_kind):
        sd[pre + "norm1.weight"] = torch.ones(D); sd[pre + "norm1.bias"] = torch.zeros(D)
        sd[pre + "norm2.weight"] = torch.ones(D); sd[pre + "norm2.bias"] = torch.zeros(D)
        if attn_kind == "ca":
            for n in ("q", "k", "v"):
                sd[pre + f"attn.{n}.weight"] = tn(D, D); sd[pre + f"attn.{n}.bias"] = torch.zeros(D)
        else:
            sd[pre + "attn.qkv.weight"] = tn(3 * D, D); sd[pre + "attn.qkv.bias"] = torch.zeros(3 * D)
        sd[pre + "attn.proj.weight"] = tn(D, D); sd[pre + "attn.proj.bias"] = torch.zeros(D)
        if attn_kind == "th":
            sd[pre + "attn.proj_l.weight"] = tn(heads, heads); sd[pre + "attn.proj_l.bias"] = torch.zeros(heads)
            sd[pre + "attn.proj_w.weight"] = tn(heads, heads); sd[pre + "attn.proj_w.bias"] = torch.zeros(heads)
        if attn_kind in ("th", "ca"):
            sd[pre + "gamma_1"] = cfg.get("init_scale", 1e-5) * torch.ones(D)
            sd[pre + "gamma_2"] = cfg.get("init_scale", 1e-5) * torch.ones(D)
        Hd = int(D * mlp_ratio)
        sd[pre + "mlp.fc1.weight"] = tn(Hd, D); sd[pre + "mlp.fc1.bias"] = torch.zeros(Hd)
        sd[pre + "mlp.fc2.weight"] = tn(D, Hd); sd[pre + "mlp.fc2.bias"] = torch.zeros(D)

    for i in range(depth):
        block(f"{f}blocks.{i}.", "sa" if cfg["arch"] == "deit" else "th")
    if cfg["arch"] == "cait":
        for i in range(cfg.get("depth_token_only", 2)):
            block(f"{f}blocks_token_only.{i}.", "ca")
    sd[f + "norm.weight"] = torch.ones(D); sd[f + "norm.bias"] = torch.zeros(D)

    sd["add_on_layers.0.weight"] = torch.randn(Dp, D, 1, 1, generator=g) * math.sqrt(2.0 / Dp)   # kaiming fan_out
    sd["add_on_layers.0.bias"] = torch.zeros(Dp)
    sd["prototype_vectors"] = torch.rand(P, Dp, 1, 1, generator=g)
    sd["prototype_vectors_global"] = torch.rand(C * gpc, Dp, 1, 1, generator=g)
    sd["ones"] = torch.ones(P, Dp, 1, 1)

    def head(n_proto):
        per = n_proto // C
        ident = torch.zeros(n_proto, C)
        ident[torch.arange(n_proto), torch.arange(n_proto) // per] = 1.0
        return (ident - 0.5 * (1 - ident)).t().contiguous()
    sd["last_layer.weight"] = head(P)
    sd["last_layer_global.weight"] = head(C * gpc)
    return sd


ARCH_CFGS = {
    "deit_tiny_patch16_224": dict(arch="deit", dim=192, depth=12, heads=3),
    "deit_small_patch16_224": dict(arch="deit", dim=384, depth=12, heads=6),
    "deit_base_patch16_224": dict(arch="deit", dim=768, depth=12, heads=12),
    "cait_xxs24_224": dict(arch="cait", dim=192, depth=24, heads=4, init_scale=1e-5),
}


def make_cfg(base_architecture: str, num_prototypes: int, proto_dim: int, num_classes: int, reserve_layer: int,
             reserve_k: int, global_per_class: int = 10, global_coe: float = 0.5, **kw) -> dict:
    cfg = dict(ARCH_CFGS[base_architecture])
    cfg.update(num_prototypes=num_prototypes, proto_dim=proto_dim, num_classes=num_classes,
               reserve_layer=reserve_layer, reserve_k=reserve_k, global_per_class=global_per_class,
               global_coe=global_coe, protos_per_class=num_prototypes // num_classes)
    cfg.update(kw)
    return cfg


def train_step(sd: SD, opt: torch.optim.Optimizer, img: Tensor, label: Tensor, cfg: dict,
               droppath: Optional[list] = None, ema: Optional[SD] = None, ema_decay: float = 0.99996) -> float:
    """One step of tools/engine_proto.py:41-81 on CPU fp32 (forward, CE+PPC, backward, AdamW, EMA)."""
    out = ppnet_forward(sd, img, cfg, train=True, droppath=droppath)
    loss, _ = train_loss(out, label, cfg, with_ppc=True)
    opt.zero_grad()
    loss.backward()
    opt.step()
    if ema is not None:
        with torch.no_grad():
            for k, v in sd.items():
                ema[k].mul_(ema_decay).add_(v.detach(), alpha=1.0 - ema_decay)
    return float(loss.detach())
