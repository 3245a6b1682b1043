"""CaiT backbone of ProtoPFormer on the HIP kernels (host-side orchestration only).

Mirrors the reference's ``MyCait`` (tools/cait_models_attn.py:188-345): 24 LayerScale blocks with talking-heads attention on the
patch tokens, then 2 class-attention blocks that update only the cls token, attention rollout + token reservation before class
block ``reserve_layer`` (cait:326-339).  Parameter names equal the reference's (``blocks.{i}.{gamma_1,gamma_2,norm1,attn.{qkv,proj,
proj_l,proj_w},norm2,mlp.{fc1,fc2}}``, ``blocks_token_only.{i}.{gamma_1,gamma_2,norm1,attn.{q,k,v,proj},norm2,mlp}``, ``pos_embed``
``(1,Np,D)``, ``cls_token``, ``norm``).  nn modules are parameter containers only; all arithmetic is in csrc/cait.hip + the GEMMs.
"""
import functools
import os

import torch
import torch.nn as nn

from . import _lib, ops
from .backbone import LN_EPS, _dp, _wgrad, addon_bwd, addon_convs, head_tokens_fwd, wgrad_lane
from .deit import _Mlp, _PatchEmbed, _init_vit
from .ops import EPI_BF16, EPI_DGELU, EPI_F32, EPI_GELU, EPI_RESID


class _TalkingHeadAttn(nn.Module):
    def __init__(self, dim, num_heads):
        super().__init__()
        self.num_heads = num_heads
        self.qkv = nn.Linear(dim, dim * 3, bias=True)
        self.proj = nn.Linear(dim, dim)
        self.proj_l = nn.Linear(num_heads, num_heads)
        self.proj_w = nn.Linear(num_heads, num_heads)


class _ClassAttn(nn.Module):
    def __init__(self, dim, num_heads):
        super().__init__()
        self.num_heads = num_heads
        self.q = nn.Linear(dim, dim, bias=True)
        self.k = nn.Linear(dim, dim, bias=True)
        self.v = nn.Linear(dim, dim, bias=True)
        self.proj = nn.Linear(dim, dim)


class LayerScaleBlock(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio, drop_path, init_values, attn_cls):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=LN_EPS)
        self.attn = attn_cls(dim, num_heads)
        self.drop_path_rate = float(drop_path)
        self.norm2 = nn.LayerNorm(dim, eps=LN_EPS)
        self.mlp = _Mlp(dim, int(dim * mlp_ratio))
        self.gamma_1 = nn.Parameter(init_values * torch.ones(dim))
        self.gamma_2 = nn.Parameter(init_values * torch.ones(dim))


class MyCait(nn.Module):
    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=192, depth=24, num_heads=4, mlp_ratio=4., drop_path_rate=0.1,
                 init_scale=1e-5, depth_token_only=2, **_unused):
        super().__init__()
        self.embed_dim, self.depth, self.num_heads = embed_dim, depth, num_heads
        self.layer_nums = [depth, depth_token_only]
        self.patch_embed = _PatchEmbed(img_size, patch_size, in_chans, embed_dim)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, self.patch_embed.num_patches, embed_dim))
        self.blocks = nn.ModuleList([LayerScaleBlock(embed_dim, num_heads, mlp_ratio, drop_path_rate, init_scale, _TalkingHeadAttn)
                                     for _ in range(depth)])                                         # constant rate (cait:206)
        self.blocks_token_only = nn.ModuleList([LayerScaleBlock(embed_dim, num_heads, 4.0, 0.0, init_scale, _ClassAttn)
                                                for _ in range(depth_token_only)])
        self.norm = nn.LayerNorm(embed_dim, eps=LN_EPS)
        nn.init.trunc_normal_(self.pos_embed, std=.02)
        nn.init.trunc_normal_(self.cls_token, std=.02)
        self.apply(_init_vit)

    def droppath_rates(self):
        return [r for blk in self.blocks for r in (blk.drop_path_rate, blk.drop_path_rate)]

    def __repr__(self):
        return "MyCait(hip)" + super().__repr__()[len("MyCait"):]

    def _store(self):
        root = getattr(self, "_ppf_root", None)
        if root is None:
            raise RuntimeError("features module must be owned by a protopformer_amd.PPNet (flat parameter store)")
        return root().flat_store()

    @torch.no_grad()
    def forward_feature_patch_embed_all(self, x):
        """cait:303-312 (inference-only compatibility wrapper): returns (cls_tokens [B,1,D], x [B,Np,D])."""
        xe = cait_embed(self, self._store(), x)
        return self.cls_token.expand(x.shape[0], -1, -1), xe

    @torch.no_grad()
    def forward_feature_mask_train_direct(self, cls_embed, x_embed, token_attn=None, reserve_layer_nums=[]):
        """cait:314-345 (inference-only compatibility wrapper)."""
        (layer, k), = reserve_layer_nums
        u, cls_attn, idx, _ = cait_blocks_fwd(self, self._store(), x_embed.contiguous(), layer, k, dp=None, save=False)
        B, N1, D = u.shape
        y, _, _ = ops.layernorm_fwd(u.reshape(B * N1, D), self.norm.weight, self.norm.bias, LN_EPS)
        return y.float().reshape(B, N1, D), (cls_attn, None)


# ------------------------------------------------------------------------------------------------ forward
def cait_embed(feats, store, img, saved=None):
    pe = feats.patch_embed
    B, D, Np = img.shape[0], feats.embed_dim, pe.num_patches
    cols = ops.im2col_patch(img.contiguous().float(), pe.patch_size)
    tok = ops.gemm(cols, store.w16(pe.proj.weight).reshape(D, -1), epi=EPI_F32, bias=pe.proj.bias)
    x = ops.assemble_tokens(tok, feats.cls_token, feats.pos_embed, B, Np, D, 0)          # cls gets no position (cait:307-309)
    if saved is not None:
        saved["cols"] = cols
    return x


def _th_attention_fwd(blk, qkv, B, H, N, D, hm_slot):
    """Talking-heads attention (cait:115-130) from packed qkv; returns (out bf16 [B*N,D], P fp32, A bf16)."""
    hd = D // H
    if ops.th_fused_ok(H, N, D):
        # one launch up to and including A.V, nothing of size N x N in fp32: only the softmax statistics (and the bf16 A the dV product reads)
        # are kept for backward
        a16, rowmax, zinv, ao = ops.th_fwd(qkv, blk.attn.proj_l.weight, blk.attn.proj_l.bias, blk.attn.proj_w.weight, blk.attn.proj_w.bias, hm_slot, B, H, N, D,
                                           with_out=ops.th_pv_fused())
        sp = (rowmax, zinv)
        if ao is not None:
            return ao, sp, a16
    else:
        sp = ops.th_scores(qkv, blk.attn.proj_l.weight, blk.attn.proj_l.bias, B, H, N, D)
        a16 = ops.th_softmax_mix(sp, blk.attn.proj_w.weight, blk.attn.proj_w.bias, hm_slot)      # sp now holds P
    NPK = a16.shape[-1]
    ao = torch.empty((B * N, D), dtype=torch.bfloat16, device=qkv.device)
    # O_h[q][d] = sum_key A_h[q][key] V_h[key][d]
    ops.gemm_batched(a16, ops._Off(qkv, 2 * D), ao, N, hd, N, NPK, 3 * D, D, False, True, False, 1.0, B, H,
                     (H * N * NPK, N * NPK), (N * 3 * D, hd), (N * D, hd), kpad=1)
    return ao, sp, a16


def cait_blocks_fwd(feats, store, x, reserve_layer, reserve_k, dp, save):
    """24 talking-heads blocks + class-attention blocks with rollout/reservation (cait:314-345).
    x fp32 [B,Np,D] -> (u_out [B,1+Np,D] fp32 = cat(cls, x), cls_token_attn [B,Np], idx int32 [B,k], saved)."""
    B, N, D = x.shape
    H = feats.num_heads
    M = B * N
    NP = (N + 3) // 4 * 4
    depth = len(feats.blocks)
    hm = torch.empty((depth, B, N, NP), dtype=torch.float32, device=x.device)
    thr = torch.empty((depth, B), dtype=torch.int32, device=x.device)          # rollout discard thresholds per (layer, sample)
    lane = wgrad_lane(store)
    layers = []
    x = x.reshape(M, D)
    # The projection / fc2 GEMM, the LayerScale residual and the LayerNorm that follows run as one full-row kernel (csrc/rowgemm.hip) where
    # the shape is covered: `pre` carries the next block's norm1 output out of the previous block's fc2 launch.
    rpt = ops.rowgemm_tile_rows(M, N)
    rowk = (bool(feats.blocks) and ops.rowgemm_ok(D, D, rpt)
            and ops.rowgemm_ok(D, feats.blocks[0].mlp.fc1.out_features, rpt))
    pre = None
    rb = 3                                                                     # backbone._ROLL_BATCH
    for i, blk in enumerate(feats.blocks):
        n1, mean1, rstd1 = pre if pre is not None else ops.layernorm_fwd(x, blk.norm1.weight, blk.norm1.bias, LN_EPS)
        qkv = ops.gemm(n1, store.w16(blk.attn.qkv.weight), epi=EPI_BF16, bias=blk.attn.qkv.bias)
        ao, prob, a16 = _th_attention_fwd(blk, qkv, B, H, N, D, hm[i])
        # the rollout's order statistic, off the critical path (three layers per main-stream event record, as backbone.forward_blocks)
        side = lambda i=i: ops.rollout_threshold(hm[i], thr[i], N)
        lane.submit(side, (hm, thr), defer=(i % rb != rb - 1) and i != len(feats.blocks) - 1)
        s1, s2 = _dp(dp, 2 * i), _dp(dp, 2 * i + 1)
        raw1 = torch.empty((M, D), dtype=torch.bfloat16, device=x.device) if save else None
        if rowk:
            x1, n2, mean2, rstd2 = ops.rowgemm_resid_ln(ao, store.w16(blk.attn.proj.weight), x, rpt, bias=blk.attn.proj.bias, rowscale=s1, rows_per_group=N,
                                                        ln_w=blk.norm2.weight, ln_b=blk.norm2.bias, eps=LN_EPS, colscale=blk.gamma_1, aux_out=raw1)
        else:
            x1 = ops.gemm(ao, store.w16(blk.attn.proj.weight), epi=EPI_RESID, bias=blk.attn.proj.bias, res=x, rowscale=s1, rows_per_group=N,
                          colscale=blk.gamma_1, aux_out=raw1)
            n2, mean2, rstd2 = ops.layernorm_fwd(x1, blk.norm2.weight, blk.norm2.bias, LN_EPS)
        h = torch.empty((M, blk.mlp.fc1.out_features), dtype=torch.uint8, device=x.device)     # gelu'(pre-activation), 8-bit codes
        g = ops.gemm(n2, store.w16(blk.mlp.fc1.weight), epi=EPI_GELU, bias=blk.mlp.fc1.bias, aux_out=h)
        raw2 = torch.empty((M, D), dtype=torch.bfloat16, device=x.device) if save else None
        nxt = feats.blocks[i + 1] if i + 1 < depth else None
        if rowk:
            x2, nn1, nm1, nr1 = ops.rowgemm_resid_ln(g, store.w16(blk.mlp.fc2.weight), x1, rpt, bias=blk.mlp.fc2.bias, rowscale=s2, rows_per_group=N,
                                                     ln_w=nxt.norm1.weight if nxt is not None else None, ln_b=nxt.norm1.bias if nxt is not None else None,
                                                     eps=LN_EPS, colscale=blk.gamma_2, aux_out=raw2)
            pre = (nn1, nm1, nr1) if nxt is not None else None
        else:
            x2 = ops.gemm(g, store.w16(blk.mlp.fc2.weight), epi=EPI_RESID, bias=blk.mlp.fc2.bias, res=x1, rowscale=s2, rows_per_group=N,
                          colscale=blk.gamma_2, aux_out=raw2)
        if save:
            layers.append(dict(x=x, n1=n1, mean1=mean1, rstd1=rstd1, qkv=qkv, prob=prob, a16=a16, ao=ao, x1=x1, n2=n2, mean2=mean2,
                               rstd2=rstd2, h=h, g=g, raw1=raw1, raw2=raw2, s1=s1, s2=s2))
        x = x2
    # ---- class-attention stage: only the cls token changes
    N1 = N + 1
    # (every sub-matrix copy of this stage goes through the library -- ops.gather_rows / copy_2d / cat_rows -- not through ATen: a
    # recorded step, engine.ReplayedTrainStep, replays library calls only)
    cls = ops.gather_rows(feats.cls_token.detach().reshape(1, D), _zero_rows(B, x.device))        # cls_token broadcast to [B, D]
    xt = x.reshape(B, N, D)
    policy = cls_attn = idx = None
    ca_layers = []
    rowmeans = torch.empty((len(feats.blocks_token_only), B, N1), dtype=torch.float32, device=x.device)
    for j, blk in enumerate(feats.blocks_token_only):
        if j == reserve_layer:
            init_rows = rowmeans[:j]                                              # class-attention rows produced so far (cait:249-251)
            lane.join()
            cls_attn, idx, policy = ops.rollout(hm, depth, B, N, reserve_k, lead=0, init_rows=init_rows, thr=thr)
        u = ops.cat_rows(cls, xt).reshape(B * N1, D)
        n, mean1, rstd1 = ops.layernorm_fwd(u, blk.norm1.weight, blk.norm1.bias, LN_EPS)
        kk = ops.gemm(n, store.w16(blk.attn.k.weight), epi=EPI_BF16, bias=blk.attn.k.bias)
        vv = ops.gemm(n, store.w16(blk.attn.v.weight), epi=EPI_BF16, bias=blk.attn.v.bias)
        ncls = n.reshape(B, N1 * D)[:, :D]                                       # cls rows of n: row stride N1*D
        qq = torch.empty((B, D), dtype=torch.bfloat16, device=x.device)
        ops._lib.call("ppf_gemm_bf16", n, store.w16(blk.attn.q.weight), qq, B, D, D, N1 * D, D, D, 0, 0, EPI_BF16, blk.attn.q.bias, None, 0,
                      None, 1, None, None, None, 0, None, 1.0, None, 0)
        out, attn, zinv, _ = ops.class_attn_fwd(qq, kk, vv, policy, B, H, N1, D, rowmean=rowmeans[j])
        raw1 = torch.empty((B, D), dtype=torch.bfloat16, device=x.device) if save else None
        cls1 = ops.gemm(out, store.w16(blk.attn.proj.weight), epi=EPI_RESID, bias=blk.attn.proj.bias, res=cls, colscale=blk.gamma_1, aux_out=raw1)
        n2, mean2, rstd2 = ops.layernorm_fwd(cls1, blk.norm2.weight, blk.norm2.bias, LN_EPS)
        h = torch.empty((B, blk.mlp.fc1.out_features), dtype=torch.uint8, device=x.device)
        g = ops.gemm(n2, store.w16(blk.mlp.fc1.weight), epi=EPI_GELU, bias=blk.mlp.fc1.bias, aux_out=h)
        raw2 = torch.empty((B, D), dtype=torch.bfloat16, device=x.device) if save else None
        cls2 = ops.gemm(g, store.w16(blk.mlp.fc2.weight), epi=EPI_RESID, bias=blk.mlp.fc2.bias, res=cls1, colscale=blk.gamma_2, aux_out=raw2)
        if save:
            ca_layers.append(dict(u=u, n=n, mean1=mean1, rstd1=rstd1, q=qq, k=kk, v=vv, attn=attn, zinv=zinv, out=out, cls=cls, cls1=cls1,
                                  n2=n2, mean2=mean2, rstd2=rstd2, h=h, g=g, raw1=raw1, raw2=raw2, policy=policy))
        cls = cls2
    lane.join()
    u_out = ops.cat_rows(cls, xt)
    return u_out, cls_attn, idx, dict(sa=layers, ca=ca_layers)


_ZERO_ROWS = {}


def _zero_rows(B, device):
    """int32 [B] of zeros: the row map that broadcasts one source row to B rows (ops.gather_rows); cached per batch size."""
    key = (B, str(device))
    t = _ZERO_ROWS.get(key)
    if t is None:
        t = _ZERO_ROWS[key] = torch.zeros(B, dtype=torch.int32, device=device)
    return t


# ------------------------------------------------------------------------------------------------ backward
def _th_attention_bwd(store, blk, L, dao, B, H, N, D):
    """Backward of the talking-heads attention: returns dqkv bf16 [B*N, 3D]; accumulates proj_l / proj_w grads."""
    hd = D // H
    qkv, prob, a16 = L["qkv"], L["prob"], L["a16"]
    NPK = a16.shape[-1]
    dev = qkv.device
    scale = hd ** -0.5
    dqkv = torch.empty_like(qkv)
    gv = store.grad_view
    one_launch = isinstance(prob, tuple) and ops.th_grads_ok(H, N, D)      # dQ, dK, dV by ops.th_grads behind th_bwd
    if not one_launch:
        # dV_h[key][d] = sum_q A_h[q][key] dO_h[q][d]
        ops.gemm_batched(a16, dao, ops._Off(dqkv, 2 * D), N, hd, N, NPK, D, 3 * D, True, True, False, 1.0, B, H,
                         (H * N * NPK, N * NPK), (N * D, hd), (N * 3 * D, hd), kpad=1)
    if isinstance(prob, tuple):
        # fused: dA, the two head mixes and the softmax backward in registers; the parameter-gradient sums leave the main stream
        rowmax, zinv = prob
        ds16, partial = ops.th_bwd(qkv, dao, blk.attn.proj_l.weight, blk.attn.proj_l.bias, blk.attn.proj_w.weight, rowmax, zinv, B, H, N, D)
        wgrad_lane(store).submit(lambda: ops.th_param_reduce(partial, B, H, N, gv(blk.attn.proj_w.weight), gv(blk.attn.proj_w.bias), gv(blk.attn.proj_l.bias),
                                                             gv(blk.attn.proj_l.weight)), (partial,),
                                 defer=True)    # launched with the qkv weight gradient: one event record
    else:
        NP = prob.shape[-1]
        # dA_h[q][key] = sum_d dO_h[q][d] V_h[key][d]
        da = torch.empty((B, H, N, NP), dtype=torch.float32, device=dev)
        ops.gemm_batched(dao, ops._Off(qkv, 2 * D), da, N, N, hd, D, 3 * D, NP, False, False, True, 1.0, B, H,
                         (N * D, hd), (N * 3 * D, hd), (H * N * NP, N * NP))
        ds16 = ops.th_softmax_bwd(prob, da, blk.attn.proj_w.weight, blk.attn.proj_l.weight, gv(blk.attn.proj_w.weight), gv(blk.attn.proj_w.bias),
                                  gv(blk.attn.proj_l.bias))
        ops.th_dwl(qkv, da, gv(blk.attn.proj_l.weight), B, H, N, D)                # da now holds dS'
    if one_launch:
        return ops.th_grads(qkv, dao, ds16, a16, dqkv, B, H, N, D)
    # dQ_h = scale * dS_h K_h ;  dK_h = scale * dS_h^T Q_h
    ops.gemm_batched(ds16, ops._Off(qkv, D), dqkv, N, hd, N, NPK, 3 * D, 3 * D, False, True, False, scale, B, H,
                     (H * N * NPK, N * NPK), (N * 3 * D, hd), (N * 3 * D, hd), kpad=1)
    ops.gemm_batched(ds16, qkv, ops._Off(dqkv, D), N, hd, N, NPK, 3 * D, 3 * D, True, True, False, scale, B, H,
                     (H * N * NPK, N * NPK), (N * 3 * D, hd), (N * 3 * D, hd), kpad=1)
    return dqkv


def _mlp_bwd(store, blk, L, dyb, fc2_bias=False, want_dn=True):
    """Shared MLP backward: consumes dyb = bf16 gradient of the fc2 output; returns dn2 bf16 (want_dn) or dh, the gradient at fc1's
    output (the caller then fuses fc1's input gradient with the LayerNorm backward).  fc2_bias: dyb's producer did not accumulate
    fc2.bias' gradient, the weight-gradient GEMM that reads dyb anyway sums its columns."""
    _wgrad(store, dyb, L["g"], blk.mlp.fc2.weight, blk.mlp.fc2.bias if fc2_bias else None, defer=True)
    w2t = store.w16t(blk.mlp.fc2.weight)
    if w2t is not None:          # contraction-contiguous operands: the direct-to-LDS kernels (csrc/gemm_bf16.hip gemm224g / gemm128g)
        dh = ops.gemm(dyb, w2t, epi=EPI_DGELU, aux_in=L["h"])
    else:
        dh = ops.gemm(dyb, store.w16(blk.mlp.fc2.weight), trans_b=True, epi=EPI_DGELU, aux_in=L["h"])
    _wgrad(store, dh, L["n2"], blk.mlp.fc1.weight, blk.mlp.fc1.bias)
    if not want_dn:
        return dh
    return ops.gemm(dh, store.w16(blk.mlp.fc1.weight), trans_b=True, epi=EPI_BF16)


def cait_backward(ppnet, store, saved, df):
    feats = ppnet.features
    sa, ca = saved["layers"]["sa"], saved["layers"]["ca"]
    head = saved["head"]
    u_last = saved["x_last"]                               # [B, N1, D]
    B, N1, D = u_last.shape
    N, H = N1 - 1, feats.num_heads
    M = B * N
    dev = u_last.device
    conv = addon_convs(ppnet)[0]
    gv = store.grad_view
    lane = wgrad_lane(store)
    lnb = functools.partial(ops.layernorm_bwd, lane=lane)      # column-sum reductions (parameter grads) go to the side stream
    dz = addon_bwd(ppnet, store, head, saved["f"].reshape(-1, saved["f"].shape[-1]), df)
    _wgrad(store, dz, head["nf"], conv.weight)
    dnf = ops.gemm(dz, store.w16(conv.weight).reshape(conv.out_channels, D), trans_b=True, epi=EPI_BF16)
    du = ops.zeros((B * N1, D), torch.float32, dev)
    lnb(dnf, u_last.reshape(B * N1, D), feats.norm.weight, head["meanf"], head["rstdf"], gv(feats.norm.weight),
                      gv(feats.norm.bias), dx_out=du, row_map=head["row_map"])
    du3 = du.reshape(B, N1, D)
    du2 = du.reshape(B, N1 * D)
    dcls = ops.copy_2d(torch.empty((B, D), dtype=torch.float32, device=dev), du2[:, :D])
    gs = getattr(ppnet, "_grad_sync", None)            # data-parallel: the final norm + add-on + prototype gradients are complete (round 6: this
    if gs is not None:                                 # chunk was launched LAST, 45 us of exchange behind the last backward kernel)
        lane.flush()
        _lib.run_live(lambda: gs.chunk_ready(gs.tail_chunk, also=lane.streams))
    # ---- class-attention blocks (reverse)
    for j in range(len(ca) - 1, -1, -1):
        L, blk = ca[j], feats.blocks_token_only[j]
        dyb = lane.track(torch.empty((B, D), dtype=torch.bfloat16, device=dev))
        lnb(None, None, None, None, None, None, None, dres_in=dcls, cast_out=dyb, colscale=blk.gamma_2,
                          dbias_next=gv(blk.mlp.fc2.bias), branch=L["raw2"], dcolscale=gv(blk.gamma_2))
        dn2 = _mlp_bwd(store, blk, L, dyb)
        lane.before_overwrite(dyb)
        lnb(dn2, L["cls1"], blk.norm2.weight, L["mean2"], L["rstd2"], gv(blk.norm2.weight), gv(blk.norm2.bias), dres_in=dcls,
                          dx_out=dcls, cast_out=dyb, colscale=blk.gamma_1, dbias_next=gv(blk.attn.proj.bias), branch=L["raw1"],
                          dcolscale=gv(blk.gamma_1))
        _wgrad(store, dyb, L["out"], blk.attn.proj.weight)
        dout = ops.gemm(dyb, store.w16(blk.attn.proj.weight), trans_b=True, epi=EPI_BF16)
        dq, dk, dv = ops.class_attn_bwd(L["q"], L["k"], L["v"], L["attn"], L["zinv"], dout, B, H, N1, D)
        # projections q (cls rows only), k, v
        ncls = L["n"].reshape(B, N1 * D)[:, :D]
        ops._lib.call("ppf_gemm_bf16", dq, L["n"], gv(blk.attn.q.weight), D, D, B, D, N1 * D, D, 1, 1, ops.EPI_ATOMIC, None, None, 0, None, 1,
                      None, None, None, 0, gv(blk.attn.q.bias), 1.0, None, 0)
        _wgrad(store, dk, L["n"], blk.attn.k.weight, blk.attn.k.bias)
        _wgrad(store, dv, L["n"], blk.attn.v.weight, blk.attn.v.bias)
        dnk = ops.gemm(dk, store.w16(blk.attn.k.weight), trans_b=True, epi=EPI_F32)
        dnv = ops.gemm(dv, store.w16(blk.attn.v.weight), trans_b=True, epi=EPI_F32)
        dnq = ops.gemm(dq, store.w16(blk.attn.q.weight), trans_b=True, epi=EPI_F32)
        dn16 = ops.merge3_cast(dnk, dnv, dnq, N1)
        ops.copy_2d(du2[:, :D], dcls)                      # gradient reaching this block's cls input through the residual path
        lnb(dn16, L["u"], blk.norm1.weight, L["mean1"], L["rstd1"], gv(blk.norm1.weight), gv(blk.norm1.bias), dres_in=du, dx_out=du)
        dcls = ops.copy_2d(torch.empty((B, D), dtype=torch.float32, device=dev), du2[:, :D])
    # cls_token parameter: sum over the batch of the cls gradient (column sums via the scale/cast pass)
    scratch = torch.empty((B, D), dtype=torch.bfloat16, device=dev)
    lnb(None, None, None, None, None, None, None, dres_in=dcls, cast_out=scratch, dbias_next=gv(feats.cls_token).reshape(D))
    # ---- talking-heads blocks (reverse)
    dx = ops.copy_2d(torch.empty((B, N * D), dtype=torch.float32, device=dev), du2[:, D:]).reshape(M, D)
    dyb = lane.track(torch.empty((M, D), dtype=torch.bfloat16, device=dev))
    last = feats.blocks[-1]
    lnb(None, None, None, None, None, None, None, dres_in=dx, cast_out=dyb, rowscale=sa[-1]["s2"], rows_per_group=N,
                      colscale=last.gamma_2, dbias_next=gv(last.mlp.fc2.bias), branch=sa[-1]["raw2"], dcolscale=gv(last.gamma_2))
    # Input gradient of fc1 / qkv + LayerNorm backward + LayerScale terms of the branch below as one full-row kernel (csrc/rowgemm.hip,
    # RG_LNBWD_LS) where the shape is covered.  Half-sample tiles as in the forward pass (measured on this backbone, same box: 8 885 vs
    # 8 300 img/s with whole samples -- the opposite of deit_tiny, whose side stream carries relatively more weight-gradient work).
    rptb = ops.rowgemm_tile_rows(M, N)
    hid = feats.blocks[0].mlp.fc1.out_features if len(feats.blocks) else 0
    rowb = (len(sa) > 0 and store.w16t(feats.blocks[0].mlp.fc1.weight) is not None
            and ops.rowgemm_ok(D, hid, rptb) and ops.rowgemm_ok(D, 3 * D, rptb) and ops.rowgemm_ok(D, D, rptb))
    bias_done = True                       # the producer of the current dyb has already accumulated the bias gradient of the Linear above it
    # The bf16 branch gradient alternates between two buffers (as backbone.deit_backward): the kernel that produces the next one does not
    # wait for the side stream's weight-gradient GEMM that still reads the current one (measured: a 44 us stall per block otherwise).
    dyb_alt = None
    nring = 4 if D > 256 else 64       # buffers in rotation, as backbone.deit_backward (narrow: no reuse)
    ring = []

    def next_dyb(cur, alt):
        if nring >= 64:                                    # never reused: nothing to order against the side stream, no marks
            return torch.empty_like(cur), cur
        if not any(b.data_ptr() == cur.data_ptr() for b in ring):
            ring.append(cur)
        if len(ring) < nring:
            nxt = lane.track(torch.empty_like(cur))
            ring.append(nxt)
        else:
            i = next(j for j, b in enumerate(ring) if b.data_ptr() == cur.data_ptr())
            nxt = ring[(i + 1) % len(ring)]
        lane.before_overwrite(nxt)
        return nxt, cur

    for i in range(len(sa) - 1, -1, -1):
        L, blk = sa[i], feats.blocks[i]
        if rowb:
            dh = _mlp_bwd(store, blk, L, dyb, fc2_bias=not bias_done, want_dn=False)
            dyb, dyb_alt = next_dyb(dyb, dyb_alt)
            ops.rowgemm_lnbwd(dh, store.w16t(blk.mlp.fc1.weight), L["x1"], L["mean2"], L["rstd2"], blk.norm2.weight, gv(blk.norm2.weight), gv(blk.norm2.bias),
                              rptb, dres_in=dx, dx_out=dx, cast_out=dyb, rowscale=L["s1"], rows_per_group=N, lane=lane, defer_reduce=True,
                              colscale=blk.gamma_1, branch=L["raw1"], dcolscale=gv(blk.gamma_1))
            _wgrad(store, dyb, L["ao"], blk.attn.proj.weight, blk.attn.proj.bias, defer=True)
            dao = ops.rowgemm_bf16(dyb, store.w16t(blk.attn.proj.weight), rptb)
        else:
            dn2 = _mlp_bwd(store, blk, L, dyb, fc2_bias=not bias_done)
            dyb, dyb_alt = next_dyb(dyb, dyb_alt)
            lnb(dn2, L["x1"], blk.norm2.weight, L["mean2"], L["rstd2"], gv(blk.norm2.weight), gv(blk.norm2.bias), dres_in=dx, dx_out=dx,
                              cast_out=dyb, rowscale=L["s1"], rows_per_group=N, colscale=blk.gamma_1, dbias_next=gv(blk.attn.proj.bias),
                              branch=L["raw1"], dcolscale=gv(blk.gamma_1))
            _wgrad(store, dyb, L["ao"], blk.attn.proj.weight, defer=True)
            dao = ops.gemm(dyb, store.w16(blk.attn.proj.weight), trans_b=True, epi=EPI_BF16)
        dqkv = _th_attention_bwd(store, blk, L, dao, B, H, N, D)
        _wgrad(store, dqkv, L["n1"], blk.attn.qkv.weight, blk.attn.qkv.bias)
        dn1 = None if rowb else ops.gemm(dqkv, store.w16(blk.attn.qkv.weight), trans_b=True, epi=EPI_BF16)
        if i > 0:
            prev, Lp = feats.blocks[i - 1], sa[i - 1]
            dyb, dyb_alt = next_dyb(dyb, dyb_alt)
            if rowb:
                ops.rowgemm_lnbwd(dqkv, store.w16t(blk.attn.qkv.weight), L["x"], L["mean1"], L["rstd1"], blk.norm1.weight, gv(blk.norm1.weight), gv(blk.norm1.bias),
                                  rptb, dres_in=dx, dx_out=dx, cast_out=dyb, rowscale=Lp["s2"], rows_per_group=N, lane=lane, defer_reduce=True,
                                  colscale=prev.gamma_2, branch=Lp["raw2"], dcolscale=gv(prev.gamma_2))
            else:
                lnb(dn1, L["x"], blk.norm1.weight, L["mean1"], L["rstd1"], gv(blk.norm1.weight), gv(blk.norm1.bias), dres_in=dx,
                                  dx_out=dx, cast_out=dyb, rowscale=Lp["s2"], rows_per_group=N, colscale=prev.gamma_2,
                                  dbias_next=gv(prev.mlp.fc2.bias), branch=Lp["raw2"], dcolscale=gv(prev.gamma_2))
            bias_done = not rowb
        else:
            if rowb:
                ops.rowgemm_lnbwd(dqkv, store.w16t(blk.attn.qkv.weight), L["x"], L["mean1"], L["rstd1"], blk.norm1.weight, gv(blk.norm1.weight), gv(blk.norm1.bias),
                                  rptb, dres_in=dx, dx_out=dx, lane=lane, defer_reduce=True)
            else:
                lnb(dn1, L["x"], blk.norm1.weight, L["mean1"], L["rstd1"], gv(blk.norm1.weight), gv(blk.norm1.bias), dres_in=dx, dx_out=dx)
        if gs is not None and i in gs.block_chunk:
            lane.flush()
            _lib.run_live(lambda c=gs.block_chunk[i]: gs.chunk_ready(c, also=lane.streams))
    pe = feats.patch_embed
    dtok = ops.assemble_tokens_bwd(dx, gv(feats.pos_embed).reshape(N, D), None, B, N, D, 0)
    _wgrad(store, dtok, saved["cols"], pe.proj.weight, pe.proj.bias)
    if gs is not None:
        lane.flush()
        _lib.run_live(lambda: gs.chunk_ready(gs.head_chunk, also=lane.streams))
    lane.join()


def cait_t16_params(feats):
    """Weights whose input-gradient products read W^T contraction-contiguous (csrc/rowgemm.hip): fc1, qkv, proj of the talking-heads blocks."""
    return [w for blk in feats.blocks for w in (blk.mlp.fc1.weight, blk.attn.qkv.weight, blk.attn.proj.weight, blk.mlp.fc2.weight)]


CAIT_FNS = dict(embed=cait_embed, blocks=cait_blocks_fwd, backward=cait_backward, t16_params=cait_t16_params)
