"""DeiT backbone of ProtoPFormer on the HIP kernels (host-side orchestration only).

Mirrors the reference's ``MyVisionTransformer`` (tools/deit_models_attn.py:84-240): same parameter names
(``patch_embed.proj``, ``cls_token``, ``pos_embed``, ``blocks.{i}.{norm1,attn.qkv,attn.proj,norm2,mlp.fc1,mlp.fc2}``,
``norm``) so reference checkpoints load, same public methods.  nn.Linear / nn.LayerNorm / nn.Conv2d objects are used
purely as parameter containers; their forward() is never called -- every FLOP of the path runs in
protopformer_amd/csrc kernels through the C ABI.
"""
import math
from functools import partial

import torch
import torch.nn as nn

from . import ops

LN_EPS = 1e-6


class _PatchEmbed(nn.Module):
    def __init__(self, img_size, patch_size, in_chans, embed_dim):
        super().__init__()
        self.img_size, self.patch_size = img_size, patch_size
        self.num_patches = (img_size // patch_size) ** 2
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)


class _Attention(nn.Module):
    def __init__(self, dim, num_heads):
        super().__init__()
        self.num_heads = num_heads
        self.qkv = nn.Linear(dim, dim * 3, bias=True)
        self.proj = nn.Linear(dim, dim)


class _Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)


class Block(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio, drop_path):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=LN_EPS)
        self.attn = _Attention(dim, num_heads)
        self.drop_path_rate = float(drop_path)
        self.norm2 = nn.LayerNorm(dim, eps=LN_EPS)
        self.mlp = _Mlp(dim, int(dim * mlp_ratio))


def _init_vit(m):
    if isinstance(m, nn.Linear):
        nn.init.trunc_normal_(m.weight, std=.02)
        if m.bias is not None:
            nn.init.zeros_(m.bias)
    elif isinstance(m, nn.LayerNorm):
        nn.init.ones_(m.weight)
        nn.init.zeros_(m.bias)


class MyVisionTransformer(nn.Module):
    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=192, depth=12, num_heads=3, mlp_ratio=4.,
                 drop_path_rate=0.1, **_unused):
        super().__init__()
        self.embed_dim, self.depth, self.num_heads = embed_dim, depth, num_heads
        self.patch_embed = _PatchEmbed(img_size, patch_size, in_chans, embed_dim)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, self.patch_embed.num_patches + 1, embed_dim))
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, depth)]      # deit:89 stochastic depth decay rule
        self.blocks = nn.ModuleList([Block(embed_dim, num_heads, mlp_ratio, dpr[i]) for i in range(depth)])
        self.norm = nn.LayerNorm(embed_dim, eps=LN_EPS)
        nn.init.trunc_normal_(self.pos_embed, std=.02)
        nn.init.trunc_normal_(self.cls_token, std=.02)
        self.apply(_init_vit)

    def droppath_rates(self):
        """Two DropPath slots per block (attention branch, MLP branch), same rate (deit:71,79-80)."""
        return [r for blk in self.blocks for r in (blk.drop_path_rate, blk.drop_path_rate)]

    def __repr__(self):                      # PPNet inspects str(features).upper() (protopformer.py:78-84)
        return "MyVisionTransformer(hip)" + super().__repr__()[len("MyVisionTransformer"):]

    # ---- the reference's two entry points (deit:172-181, 209-240), inference-only compatibility wrappers
    @torch.no_grad()
    def forward_feature_patch_embed_all(self, x):
        from .backbone import deit_embed
        store = self._store()
        xe = deit_embed(self, store, x)
        return xe[:, :1], xe[:, 1:]

    @torch.no_grad()
    def forward_feature_mask_train_direct(self, cls_embed, x_embed, token_attn=None, reserve_layer_nums=[]):
        from .backbone import deit_blocks_fwd
        store = self._store()
        x = torch.cat([cls_embed, x_embed], dim=1).contiguous()
        (layer, k), = reserve_layer_nums
        # all 1+Np tokens come back, as in the reference (callers gather x[:, 1:] with indices up to Np-1, protopformer.py:156-162)
        x_out, cls_attn, idx, _ = deit_blocks_fwd(self, store, x, layer, k, dp=None, save=False, compact=False)
        B, N, D = x_out.shape
        y, _, _ = ops.layernorm_fwd(x_out.reshape(B * N, D), self.norm.weight, self.norm.bias, LN_EPS)
        return y.float().reshape(B, N, D), (cls_attn, None)

    def _store(self):
        root = getattr(self, "_ppf_root", None)
        if root is None:
            raise RuntimeError("features module must be owned by a protopformer_amd.PPNet (flat parameter store)")
        return root().flat_store()
