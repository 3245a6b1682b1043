"""fp32 verification mode of the backbone (forward and backward, DeiT and CaiT): the same call sequence as backbone.py / cait.py with every operand and
intermediate in fp32 (csrc/precise.hip + ppf_sgemm), the masked full-length blocks of the reference after the token reservation
(no compaction), fp32 master weights.  Enabled by ``PPNet.precise = True`` or ``PPF_PRECISE=1``; with gradients enabled the forward
records what the fp32 backward needs (``PreciseTokensFn``).

Purpose: the bf16 product path carries ~4e-3 of operand rounding per GEMM, so end to end it can only be gated at a few 1e-3; this
mode holds the *whole* forward / loss to the north-star 1e-3 rel against the reference-generated fixtures
(tests/test_gpu_precise.py), which separates "bf16 rounding" from "kernel / orchestration bug" (rollout, reservation, prototype
layer, PPC and CE kernels are shared with the product path).  The backward does the same for the gradients: every weight
gradient of the reference fixture is held at 1e-3 where the bf16 step can only be gated on its direction.  Never on the measured path.
Reference lines: deit:172-240, cait:303-345, protopformer.py:141-173."""
import torch

from . import ops
from .backbone import LN_EPS, _dp, addon_convs


def _mlp(blk, x1, N, s2, colscale=None):
    n2 = ops.layernorm_fwd_f32(x1, blk.norm2.weight, blk.norm2.bias, LN_EPS)
    h = ops.linear_f32(n2, blk.mlp.fc1.weight, blk.mlp.fc1.bias, kind=1)
    return ops.linear_f32(h, blk.mlp.fc2.weight, blk.mlp.fc2.bias, kind=3, res=x1, rowscale=s2, rows_per_group=N, colscale=colscale)


def deit_tokens(ppnet, img, dp, saved=None):
    feats = ppnet.features
    (layer, k), = ppnet.reserve_layer_nums
    pe = feats.patch_embed
    cols = ops.im2col_patch_f32(img.contiguous().float(), pe.patch_size)
    tok = ops.linear_f32(cols, pe.proj.weight, pe.proj.bias)
    x = ops.assemble_tokens(tok, feats.cls_token, feats.pos_embed, img.shape[0], pe.num_patches, feats.embed_dim, 1)
    B, N, D = x.shape
    H = feats.num_heads
    NP = (N + 3) // 4 * 4
    hm = torch.empty((max(layer, 1), B, N, NP), dtype=torch.float32, device=x.device)
    x = x.reshape(B * N, D)
    policy = cls_attn = idx = None
    layers = []
    for i, blk in enumerate(feats.blocks):
        if i == layer:
            cls_attn, idx, policy = ops.rollout(hm, layer, B, N, k, lead=1)
        n1 = ops.layernorm_fwd_f32(x, blk.norm1.weight, blk.norm1.bias, LN_EPS)
        qkv = ops.linear_f32(n1, blk.attn.qkv.weight, blk.attn.qkv.bias)
        ao = ops.attn_fwd_f32(qkv, B, H, N, D, policy=policy, self_keep=True, headmean=hm[i] if i < layer else None)
        x1 = ops.linear_f32(ao, blk.attn.proj.weight, blk.attn.proj.bias, kind=3, res=x, rowscale=_dp(dp, 2 * i), rows_per_group=N)
        layers.append((x, qkv, ao, x1, policy))
        x = _mlp(blk, x1, N, _dp(dp, 2 * i + 1))
    rows = ops.reserved_rows_map(idx, N)
    nf = ops.layernorm_fwd_f32(x, feats.norm.weight, feats.norm.bias, LN_EPS, row_map=rows)          # deit:238 on the reserved rows
    f, acts = _addon_fwd(ppnet, nf)                                                                 # protopformer.py:162-172
    if saved is not None:
        saved.update(cols=cols, layers=layers, x_last=x, rows=rows, nf=nf, f=f, acts=acts, shape=(B, N, D), dp=dp)
    return f.reshape(B, 1 + k, f.shape[-1]), cls_attn, idx


def _addon_fwd(ppnet, nf):
    """add_on_layers on the normalised reserved rows: Linear (+ ReLU ...) + Sigmoid; returns (f, inputs of the convolutions after the first)."""
    convs = addon_convs(ppnet)
    h, acts = nf, []
    for j, c in enumerate(convs):
        if j:
            acts.append(h)
        h = ops.linear_f32(h, c.weight, c.bias, kind=2 if j == len(convs) - 1 else 4)
    return h, acts


def _gv(store, p):
    return store.grad_view(p) if p is not None and p.requires_grad else None


def _linear_bwd(store, dy, x, lin, want_dx=True):
    """gradients of y = x W^T + b: W.grad += dy^T x, b.grad += colsum(dy) (into the flat store's views); returns dy W."""
    gw = _gv(store, lin.weight)
    if gw is not None:
        ops.linear_wgrad_f32(dy, x, gw, _gv(store, lin.bias))
    return ops.linear_dgrad_f32(dy, lin.weight) if want_dx else None


def _head_bwd(ppnet, store, saved, df, rows_total, D):
    """add-on conv + sigmoid + final norm on the reserved rows: gradient w.r.t. the last token matrix (rows outside the reservation zero)."""
    feats = ppnet.features
    convs = addon_convs(ppnet)
    d = ops.ew_bwd_f32(1, df, saved["f"])                                                           # sigmoid'
    for j in reversed(range(1, len(convs))):                                                        # bottleneck tail (protopformer.py:90-107)
        x_in = saved["acts"][j - 1]
        d = ops.ew_bwd_f32(5, _linear_bwd(store, d, x_in, convs[j]), x_in)                          # ... through the ReLU that produced x_in
    dnf = _linear_bwd(store, d, saved["nf"], convs[0])
    dx = ops.zeros((rows_total, D), torch.float32, df.device)
    ops.layernorm_bwd_f32(dnf, saved["x_last"], feats.norm.weight, _gv(store, feats.norm.weight), _gv(store, feats.norm.bias), dx,
                          row_map=saved["rows"], eps=LN_EPS)
    return dx


def _mlp_bwd(store, blk, x1, d, N, rowscale, colscale=None):
    """x_out = x1 + rowscale * colscale * fc2(gelu(fc1(norm2(x1)))): d (gradient w.r.t. x_out) becomes the gradient w.r.t. x1, in place."""
    n2 = ops.layernorm_fwd_f32(x1, blk.norm2.weight, blk.norm2.bias, LN_EPS)
    pre = ops.linear_f32(n2, blk.mlp.fc1.weight, blk.mlp.fc1.bias)
    h = ops.linear_f32(n2, blk.mlp.fc1.weight, blk.mlp.fc1.bias, kind=1)
    if colscale is not None:
        dyb = _layerscale_bwd(store, d, ops.linear_f32(h, blk.mlp.fc2.weight, blk.mlp.fc2.bias), colscale, rowscale, N)
    else:
        dyb = ops.ew_bwd_f32(2, d, rowscale=rowscale, rows_per_group=N)
    dh = _linear_bwd(store, dyb, h, blk.mlp.fc2)
    dpre = ops.ew_bwd_f32(0, dh, pre)                                                               # gelu'
    dn2 = _linear_bwd(store, dpre, n2, blk.mlp.fc1)
    ops.layernorm_bwd_f32(dn2, x1, blk.norm2.weight, _gv(store, blk.norm2.weight), _gv(store, blk.norm2.bias), d, dres_in=d, eps=LN_EPS)
    return d


def deit_backward_f32(ppnet, saved, df):
    """fp32 backward of deit_tokens (deit:172-240 + the add-on head): every product through ppf_sgemm, the elementwise / LayerNorm /
    attention pieces through csrc/precise.hip; gradients ACCUMULATE into the flat store's views (p.grad), as autograd's would."""
    store = ppnet.flat_store()
    store.attach_all_grads()
    feats = ppnet.features
    B, N, D = saved["shape"]
    H, dp = feats.num_heads, saved["dp"]
    dx = _head_bwd(ppnet, store, saved, df, B * N, D)
    for i in reversed(range(len(feats.blocks))):
        blk = feats.blocks[i]
        x_in, qkv, ao, x1, policy = saved["layers"][i]
        _mlp_bwd(store, blk, x1, dx, N, _dp(dp, 2 * i + 1))
        # x1 = x + s1 * proj(attention(qkv(norm1(x))))
        dyb = ops.ew_bwd_f32(2, dx, rowscale=_dp(dp, 2 * i), rows_per_group=N)
        dao = _linear_bwd(store, dyb, ao, blk.attn.proj)
        dqkv = ops.attn_bwd_f32(qkv, dao, B, H, N, D, policy=policy, self_keep=True)
        n1 = ops.layernorm_fwd_f32(x_in, blk.norm1.weight, blk.norm1.bias, LN_EPS)
        dn1 = _linear_bwd(store, dqkv, n1, blk.attn.qkv)
        ops.layernorm_bwd_f32(dn1, x_in, blk.norm1.weight, _gv(store, blk.norm1.weight), _gv(store, blk.norm1.bias), dx, dres_in=dx, eps=LN_EPS)
    pe = feats.patch_embed
    Np = pe.num_patches
    ops.assemble_tokens_bwd(dx, _gv(store, feats.pos_embed), _gv(store, feats.cls_token), B, Np, D, 1)       # dpos / dcls (fp32, fixed order)
    dtok = dx.reshape(B, N, D)[:, 1:].reshape(B * Np, D).contiguous()
    _linear_bwd(store, dtok, saved["cols"], pe.proj, want_dx=False)


class PreciseTokensFn(torch.autograd.Function):
    """image -> (f, cls_token_attn, reserve idx) in fp32 with an fp32 backward; parameters enter only to hook autograd."""

    @staticmethod
    def forward(ctx, img, ppnet, dp, *params):
        saved = {} if any(ctx.needs_input_grad) else None
        ctx.deit = _is_deit(ppnet)
        f, cls_attn, idx = (deit_tokens if ctx.deit else cait_tokens)(ppnet, img, dp, saved)
        ctx.saved, ctx.ppnet = saved, ppnet
        ctx.mark_non_differentiable(cls_attn, idx)
        return f, cls_attn, idx

    @staticmethod
    def backward(ctx, df, _dcls, _didx):
        (deit_backward_f32 if ctx.deit else cait_backward_f32)(ctx.ppnet, ctx.saved, df.contiguous().reshape(-1, df.shape[-1]).float())
        ctx.saved = None
        return (None,) * len(ctx.needs_input_grad)


def cait_tokens(ppnet, img, dp, saved=None):
    feats = ppnet.features
    (layer, k), = ppnet.reserve_layer_nums
    pe = feats.patch_embed
    cols = ops.im2col_patch_f32(img.contiguous().float(), pe.patch_size)
    tok = ops.linear_f32(cols, pe.proj.weight, pe.proj.bias)
    x = ops.assemble_tokens(tok, feats.cls_token, feats.pos_embed, img.shape[0], pe.num_patches, feats.embed_dim, 0)
    B, N, D = x.shape
    H = feats.num_heads
    NP = (N + 3) // 4 * 4
    depth = len(feats.blocks)
    hm = torch.empty((depth, B, N, NP), dtype=torch.float32, device=x.device)
    x = x.reshape(B * N, D)
    layers, tlayers = [], []
    for i, blk in enumerate(feats.blocks):
        a = blk.attn
        n1 = ops.layernorm_fwd_f32(x, blk.norm1.weight, blk.norm1.bias, LN_EPS)
        qkv = ops.linear_f32(n1, a.qkv.weight, a.qkv.bias)
        ao = ops.th_attn_fwd_f32(qkv, a.proj_l.weight, a.proj_l.bias, a.proj_w.weight, a.proj_w.bias, B, H, N, D, hm[i])
        x1 = ops.linear_f32(ao, a.proj.weight, a.proj.bias, kind=3, res=x, rowscale=_dp(dp, 2 * i), rows_per_group=N, colscale=blk.gamma_1)
        layers.append((x, qkv, ao, x1))
        x = _mlp(blk, x1, N, _dp(dp, 2 * i + 1), colscale=blk.gamma_2)
    N1 = N + 1
    cls = feats.cls_token.detach().reshape(1, D).expand(B, D).contiguous()
    xt = x.reshape(B, N, D)
    policy = cls_attn = idx = None
    rowmeans = []
    for j, blk in enumerate(feats.blocks_token_only):
        a = blk.attn
        if j == layer:
            cls_attn, idx, policy = ops.rollout(hm, depth, B, N, k, lead=0, init_rows=torch.stack(rowmeans).contiguous())
        u = torch.cat([cls.reshape(B, 1, D), xt], dim=1).reshape(B * N1, D)
        n = ops.layernorm_fwd_f32(u, blk.norm1.weight, blk.norm1.bias, LN_EPS)
        kk = ops.linear_f32(n, a.k.weight, a.k.bias)
        vv = ops.linear_f32(n, a.v.weight, a.v.bias)
        qq = ops.linear_f32(n.reshape(B, N1 * D)[:, :D], a.q.weight, a.q.bias)           # cls rows only (cait:73)
        out, rowmean = ops.class_attn_fwd_f32(qq, kk, vv, policy, B, H, N1, D)
        rowmeans.append(rowmean)
        cls1 = ops.linear_f32(out, a.proj.weight, a.proj.bias, kind=3, res=cls, colscale=blk.gamma_1)
        tlayers.append((u, qq, kk, vv, policy, out, cls1))
        cls = _mlp(blk, cls1, 1, None, colscale=blk.gamma_2)
    u_out = torch.cat([cls.reshape(B, 1, D), xt], dim=1).reshape(B * N1, D)
    rows = ops.reserved_rows_map(idx, N1)
    nf = ops.layernorm_fwd_f32(u_out, feats.norm.weight, feats.norm.bias, LN_EPS, row_map=rows)     # cait:343 on the reserved rows
    f, acts = _addon_fwd(ppnet, nf)
    if saved is not None:
        saved.update(cols=cols, layers=layers, tlayers=tlayers, x_last=u_out, rows=rows, nf=nf, f=f, acts=acts, shape=(B, N, D), dp=dp)
    return f.reshape(B, 1 + k, f.shape[-1]), cls_attn, idx


def _layerscale_bwd(store, d, y, gamma, rowscale, N):
    """branch gradient of x_out = res + rowscale * gamma * y: gamma.grad += colsum(d * rowscale * y); returns d * rowscale * gamma."""
    g = _gv(store, gamma)
    if g is not None:
        ops.colsum_f32(ops.ew_bwd_f32(3, ops.ew_bwd_f32(2, d, rowscale=rowscale, rows_per_group=N), y), g)
    return ops.ew_bwd_f32(4, d, gamma, rowscale=rowscale, rows_per_group=N)


def cait_backward_f32(ppnet, saved, df):
    """fp32 backward of cait_tokens (cait:303-345 + the add-on head): talking-heads blocks, then the class-attention blocks whose patch
    rows are the (constant) output of the last talking-heads block -- their gradients add up in dxt."""
    store = ppnet.flat_store()
    store.attach_all_grads()
    feats = ppnet.features
    B, N, D = saved["shape"]
    N1, H, dp = N + 1, feats.num_heads, saved["dp"]
    du = _head_bwd(ppnet, store, saved, df, B * N1, D).reshape(B, N1, D)
    dcls = du[:, 0].contiguous()
    dxt = du[:, 1:].contiguous()
    for j in reversed(range(len(feats.blocks_token_only))):
        blk = feats.blocks_token_only[j]
        a = blk.attn
        u, qq, kk, vv, policy, out, cls1 = saved["tlayers"][j]
        dcls1 = _mlp_bwd(store, blk, cls1, dcls, 1, None, blk.gamma_2)
        y = ops.linear_f32(out, a.proj.weight, a.proj.bias)
        dout = _linear_bwd(store, _layerscale_bwd(store, dcls1, y, blk.gamma_1, None, 1), out, a.proj)
        dq, dk, dv = ops.class_attn_bwd_f32(qq, kk, vv, policy, dout, B, H, N1, D)
        n = ops.layernorm_fwd_f32(u, blk.norm1.weight, blk.norm1.bias, LN_EPS)
        dn = (_linear_bwd(store, dk, n, a.k) + _linear_bwd(store, dv, n, a.v)).reshape(B, N1, D)
        dn[:, 0] += _linear_bwd(store, dq, n.reshape(B, N1, D)[:, 0].contiguous(), a.q)
        du = torch.empty((B * N1, D), dtype=torch.float32, device=df.device)
        ops.layernorm_bwd_f32(dn.reshape(B * N1, D), u, blk.norm1.weight, _gv(store, blk.norm1.weight), _gv(store, blk.norm1.bias), du, eps=LN_EPS)
        du = du.reshape(B, N1, D)
        dcls = dcls1 + du[:, 0]
        dxt += du[:, 1:]
    g_cls = _gv(store, feats.cls_token)
    if g_cls is not None:
        ops.colsum_f32(dcls.contiguous(), g_cls.reshape(D))                                        # cls_token.expand(B, ...) (cait:320)
    dx = dxt.reshape(B * N, D)
    for i in reversed(range(len(feats.blocks))):
        blk = feats.blocks[i]
        a = blk.attn
        x_in, qkv, ao, x1 = saved["layers"][i]
        _mlp_bwd(store, blk, x1, dx, N, _dp(dp, 2 * i + 1), blk.gamma_2)
        y = ops.linear_f32(ao, a.proj.weight, a.proj.bias)
        dao = _linear_bwd(store, _layerscale_bwd(store, dx, y, blk.gamma_1, _dp(dp, 2 * i), N), ao, a.proj)
        mix = (a.proj_l.weight, a.proj_l.bias, a.proj_w.weight, a.proj_w.bias)
        dqkv = ops.th_attn_bwd_f32(qkv, dao, mix, [store.grad_view(p) for p in mix], B, H, N, D)
        n1 = ops.layernorm_fwd_f32(x_in, blk.norm1.weight, blk.norm1.bias, LN_EPS)
        dn1 = _linear_bwd(store, dqkv, n1, a.qkv)
        ops.layernorm_bwd_f32(dn1, x_in, blk.norm1.weight, _gv(store, blk.norm1.weight), _gv(store, blk.norm1.bias), dx, dres_in=dx, eps=LN_EPS)
    ops.assemble_tokens_bwd(dx, _gv(store, feats.pos_embed), None, B, N, D, 0)
    _linear_bwd(store, dx, saved["cols"], feats.patch_embed.proj, want_dx=False)


def _is_deit(ppnet):
    from .deit import MyVisionTransformer
    return isinstance(ppnet.features, MyVisionTransformer)


def tokens(ppnet, img, dp):
    if torch.is_grad_enabled() and any(p.requires_grad for p in ppnet.parameters()):
        return PreciseTokensFn.apply(img, ppnet, dp, *ppnet._hook_params())
    return (deit_tokens if _is_deit(ppnet) else cait_tokens)(ppnet, img, dp)
