"""Stand-ins that let /root/reference/protopformer.py be imported in the build container.

TEST INFRASTRUCTURE ONLY (fixture generation).  The reference needs ``timm==0.5.4`` (README.md:59),
``turtle``/tkinter (tools/deit_models_attn.py:1) and CUDA (``.cuda()`` calls); none is available here.
Everything below was written from the timm 0.5.4 *public API* (constructor signatures, attribute
and state-dict names, documented forward semantics) -- NOT verified against timm's source, which is
not present.  Parity is therefore pinned to the code under /root/reference and to torch.nn
primitives for these pieces; see oracle/ppf_oracle.py header ("parity unpinned at the timm boundary").
Never imported by the product or by GPU-side tests (``/root/reference`` does not exist there).
"""
import sys
import types
from functools import partial

import torch
import torch.nn as nn

_REGISTRY = {}


def trunc_normal_(t, mean=0., std=1., a=-2., b=2.):
    return nn.init.trunc_normal_(t, mean=mean, std=std, a=a, b=b)


class PatchEmbed(nn.Module):
    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768, norm_layer=None, flatten=True):
        super().__init__()
        self.img_size = (img_size, img_size)
        self.patch_size = (patch_size, patch_size)
        self.grid_size = (img_size // patch_size, img_size // patch_size)
        self.num_patches = self.grid_size[0] * self.grid_size[1]
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)
        self.norm = nn.Identity()

    def forward(self, x):
        return self.norm(self.proj(x).flatten(2).transpose(1, 2))


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = act_layer()
        self.drop1 = nn.Dropout(drop)
        self.fc2 = nn.Linear(hidden_features, out_features)
        self.drop2 = nn.Dropout(drop)

    def forward(self, x):
        return self.drop2(self.fc2(self.drop1(self.act(self.fc1(x)))))


class DropPath(nn.Module):
    def __init__(self, drop_prob=0.):
        super().__init__()
        self.drop_prob = drop_prob

    def forward(self, x):
        if self.drop_prob == 0. or not self.training:
            return x
        keep = 1 - self.drop_prob
        shape = (x.shape[0],) + (1,) * (x.ndim - 1)
        mask = torch.floor(keep + torch.rand(shape, dtype=x.dtype, device=x.device))
        return x.div(keep) * mask


def _init_linear_ln(m):
    if isinstance(m, nn.Linear):
        trunc_normal_(m.weight, std=.02)
        if m.bias is not None:
            nn.init.zeros_(m.bias)
    elif isinstance(m, nn.LayerNorm):
        nn.init.zeros_(m.bias)
        nn.init.ones_(m.weight)


class VisionTransformer(nn.Module):
    """Parameter set of timm 0.5.4 VisionTransformer (blocks are replaced by the subclass)."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, num_classes=1000, embed_dim=768, depth=12,
                 num_heads=12, mlp_ratio=4., qkv_bias=True, representation_size=None, distilled=False,
                 drop_rate=0., attn_drop_rate=0., drop_path_rate=0., embed_layer=PatchEmbed, norm_layer=None,
                 act_layer=None, weight_init=''):
        super().__init__()
        self.num_classes = num_classes
        self.num_features = self.embed_dim = embed_dim
        self.num_tokens = 1
        norm_layer = norm_layer or partial(nn.LayerNorm, eps=1e-6)
        self.patch_embed = embed_layer(img_size=img_size, patch_size=patch_size, in_chans=in_chans, embed_dim=embed_dim)
        num_patches = self.patch_embed.num_patches
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.dist_token = None
        self.pos_embed = nn.Parameter(torch.zeros(1, num_patches + 1, embed_dim))
        self.pos_drop = nn.Dropout(p=drop_rate)
        self.blocks = nn.Sequential()
        self.norm = norm_layer(embed_dim)
        self.pre_logits = nn.Identity()
        self.head = nn.Linear(embed_dim, num_classes) if num_classes > 0 else nn.Identity()

    def init_weights(self, mode=''):
        trunc_normal_(self.pos_embed, std=.02)
        trunc_normal_(self.cls_token, std=.02)
        self.apply(_init_linear_ln)


class Cait(nn.Module):
    """Parameter set of timm 0.5.4 Cait (blocks / blocks_token_only are replaced by the subclass)."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, num_classes=1000, embed_dim=768, depth=12,
                 num_heads=12, mlp_ratio=4., qkv_bias=True, drop_rate=0., attn_drop_rate=0., drop_path_rate=0.,
                 norm_layer=partial(nn.LayerNorm, eps=1e-6), global_pool=None, init_scale=1e-4, **kwargs):
        super().__init__()
        self.num_classes = num_classes
        self.num_features = self.embed_dim = embed_dim
        self.patch_embed = PatchEmbed(img_size=img_size, patch_size=patch_size, in_chans=in_chans, embed_dim=embed_dim)
        num_patches = self.patch_embed.num_patches
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, num_patches, embed_dim))
        self.pos_drop = nn.Dropout(p=drop_rate)
        self.blocks = nn.ModuleList()
        self.blocks_token_only = nn.ModuleList()
        self.norm = norm_layer(embed_dim)
        self.head = nn.Linear(embed_dim, num_classes) if num_classes > 0 else nn.Identity()
        trunc_normal_(self.pos_embed, std=.02)
        trunc_normal_(self.cls_token, std=.02)
        self.apply(self._init_weights)

    def _init_weights(self, m):
        _init_linear_ln(m)


def register_model(fn):
    _REGISTRY[fn.__name__] = fn
    return fn


def create_model(model_name, pretrained=False, **kwargs):
    return _REGISTRY[model_name](pretrained=pretrained, **kwargs)


def build_model_with_cfg(model_cls, variant, pretrained, default_cfg=None, pretrained_filter_fn=None, **kwargs):
    assert not pretrained, "no network in the build container"
    model = model_cls(**kwargs)
    model.default_cfg = default_cfg
    return model


def overlay_external_default_cfg(default_cfg, kwargs):
    return default_cfg


def _cfg(url='', **kwargs):
    return dict(url=url, **kwargs)


def install(reference_root='/root/reference'):
    """Register the stand-in modules and make ``import protopformer`` (the reference) work on CPU."""
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    mod('turtle', forward=lambda *a, **k: None)
    timm = mod('timm')
    timm.models = mod('timm.models', create_model=create_model)
    mod('timm.models.vision_transformer', VisionTransformer=VisionTransformer, _cfg=_cfg)
    mod('timm.models.cait', Cait=Cait)
    mod('timm.models.registry', register_model=register_model)
    mod('timm.models.layers', trunc_normal_=trunc_normal_, PatchEmbed=PatchEmbed, Mlp=Mlp, DropPath=DropPath)
    mod('timm.models.helpers', build_model_with_cfg=build_model_with_cfg,
        overlay_external_default_cfg=overlay_external_default_cfg)
    mod('timm.data', IMAGENET_DEFAULT_MEAN=(0.485, 0.456, 0.406), IMAGENET_DEFAULT_STD=(0.229, 0.224, 0.225))
    torch.Tensor.cuda = lambda self, *a, **k: self          # the reference hard-codes .cuda()
    if reference_root not in sys.path:
        sys.path.insert(0, reference_root)
    import protopformer                                      # noqa: E402  (the reference module)
    return protopformer


def freeze_droppath(model):
    """Take the train branch deterministically: model.train() but every DropPath in eval mode."""
    model.train()
    for m in model.modules():
        if isinstance(m, DropPath):
            m.eval()
    return model
