"""Forward / backward orchestration of the backbone on the HIP kernels (host-side launch sequencing only).

The whole "image -> reserved, add-on-projected tokens" stage is ONE autograd node (TokensFn) whose backward is the
hand-written kernel sequence below; weight gradients are accumulated by the kernels straight into the flat gradient
buffer (param.grad views), so the node returns no parameter gradients to autograd.
Reference call path: protopformer.py:141-173 (conv_features) -> deit:172-181, 209-240.
"""
import functools
import os

import torch
import torch.nn as nn

from . import _lib, ops
from .ops import EPI_ATOMIC, EPI_BF16, EPI_DGELU, EPI_F32, EPI_GELU, EPI_RESID, EPI_SIGMOID_F32

LN_EPS = 1e-6
_KEEP_CACHE = {}
# The full-row GEMMs with fused LayerNorm (csrc/rowgemm.hip) serve the forward pass wherever the shape allows; in the BACKWARD pass they own whole
# CUs while the weight-gradient GEMMs of the side stream want to share them -- a win where the step is launch-bound (deit_tiny batch 128:
# +7.5 %), neutral in throughput at deit_small batch 256 where they stretch the weight-gradient kernels from 100 to 137 us
# (profiles/r3_rowgemm.txt section 7); so: small problems only.
_ROW_BWD_MAX_ELEMS = 12_000_000          # rows x width of the residual stream up to which the backward pass uses them


def _row_bwd(M, D):
    return M * D <= _ROW_BWD_MAX_ELEMS


def droppath_scales(rates, B, device, training):
    """Per-layer, per-sample DropPath factors floor(keep + U)/keep (timm DropPath); None where the rate is 0.
    Returns (scales [n_active, B] tensor or None, index map layer-slot -> row or -1).
    The draws come from the library's Philox kernel (ops.droppath_draw) keyed by torch.initial_seed(); the step counter lives in
    device memory and is advanced by the launch itself, so a captured HIP graph of the step replays fresh draws."""
    slots = [r if training else 0.0 for r in rates]
    active = [i for i, r in enumerate(slots) if r > 0.0]
    if not active:
        return None, [-1] * len(slots)
    key = (tuple(slots), str(device))
    st = _KEEP_CACHE.get(key)
    if st is None:              # the H2D copy of a fresh host tensor is a blocking call: build the constants once per schedule
        st = _KEEP_CACHE[key] = dict(keep=torch.tensor([1.0 - slots[i] for i in active], device=device, dtype=torch.float32),
                                     state=torch.zeros(1, dtype=torch.int64, device=device), seed=torch.initial_seed())
    scales = ops.droppath_draw(torch.empty((len(active), B), dtype=torch.float32, device=device), st["keep"], st["seed"], st["state"])
    index = [-1] * len(slots)
    for row, i in enumerate(active):
        index[i] = row
    return scales, index


def _dp(dp, slot):
    if dp is None:
        return None
    scales, index = dp
    return None if scales is None or index[slot] < 0 else scales[index[slot]]


# ------------------------------------------------------------------------------------------------ DeiT forward
def deit_embed(feats, store, img, saved=None):
    """PatchEmbed (im2col + GEMM) + cls token + position embedding (deit:172-181). Returns x fp32 [B, 1+Np, D]."""
    pe = feats.patch_embed
    B = img.shape[0]
    D, Np = feats.embed_dim, pe.num_patches
    cols = ops.im2col_patch(img.contiguous().float(), pe.patch_size)
    w16 = store.w16(pe.proj.weight).reshape(D, -1)
    tok = ops.gemm(cols, w16, epi=EPI_F32, bias=pe.proj.bias)
    x = ops.assemble_tokens(tok, feats.cls_token, feats.pos_embed, B, Np, D, 1)
    if saved is not None:
        saved["cols"] = cols
    return x


def deit_blocks_fwd(feats, store, x, reserve_layer, reserve_k, dp, save, compact=None):
    """The 12 blocks with attention rollout + token reservation at `reserve_layer` (deit:209-236).
    x fp32 [B,N,D] -> (x_out, cls_token_attn [B,N-1], idx int32 [B,k], per-layer saved activations).

    From `reserve_layer` on, the reference keeps all N tokens and masks the dropped ones out of every softmax (deit:29-43,
    233-236); their block outputs are never read again (the head gathers the reserved rows, protopformer.py:156-162) and as keys
    they weigh eps/N ~ 5e-9.  Those blocks therefore run on the reserved rows only ([cls, 1+idx...], N' = 1+k, no policy, the
    softmax's eps/N term kept at the original N): same results to ~1e-6, 58 % less work in the last block.  x_out is then
    [B, 1+k, D].  compact=False (or PPF_COMPACT_RESERVED=0) runs the masked full-length blocks and returns all N tokens, as the
    reference's forward_feature_mask_train_direct does."""
    B, N, D = x.shape
    H = feats.num_heads
    NP = (N + 3) // 4 * 4
    hm = torch.empty((max(reserve_layer, 1), B, N, NP), dtype=torch.float32, device=x.device)
    thr = torch.empty((max(reserve_layer, 1), B), dtype=torch.int32, device=x.device)       # rollout discard thresholds per (layer, sample)
    # the rollout's per-layer order statistic runs on the side stream right behind the layer's head-mean map.  (Round 4 also built a
    # column-compressed record per layer that shrinks the main stream's wait for the chain from 168 to 33 us; its heavier per-layer kernel cost
    # more than that -- deit_small 16 842 -> 16 580 img/s, profiles/r4_queue_gaps.txt -- and it was removed in round 6.)
    side_thr = reserve_layer > 0
    if compact is None:
        compact = os.environ.get("PPF_COMPACT_RESERVED", "1") != "0"
    policy = None
    cls_attn = idx = rows = None
    layers = []
    Nc, eps_n = N, 0                              # tokens per sample in the current block, N of the softmax eps term (0 = Nc)
    x = x.reshape(B * N, D)
    lane = wgrad_lane(store)
    rolled = None                                 # rollout outputs when the chain itself ran on the side stream
    roll_side = reserve_layer > 0
    pre = None                                    # (n1, mean1, rstd1) of the coming block when the previous block's fc2 GEMM produced them
    nblk = len(feats.blocks)
    for i, blk in enumerate(feats.blocks):
        if i == reserve_layer:
            lane.join()
            if rolled is not None:
                cls_attn, idx, policy = rolled
            else:
                cls_attn, idx, policy = ops.rollout(hm, reserve_layer, B, N, reserve_k, lead=1, thr=thr if side_thr else None)
            if compact:
                rows = ops.reserved_rows_map(idx, N)
                x = ops.gather_rows(x, rows)
                Nc, eps_n, policy = 1 + reserve_k, N, None
                pre = None
        M = B * Nc
        hid = blk.mlp.fc1.out_features
        # full-row GEMMs (csrc/rowgemm.hip): the residual products also emit the LayerNorm that follows them
        rpt = ops.rowgemm_tile_rows(B * Nc, Nc)
        fused = ops.rowgemm_ok(D, D, rpt) and ops.rowgemm_ok(D, hid, rpt)
        n1, mean1, rstd1 = pre if pre is not None else ops.layernorm_fwd(x, blk.norm1.weight, blk.norm1.bias, LN_EPS)
        pre = None
        qkv = ops.gemm(n1, store.w16(blk.attn.qkv.weight), epi=EPI_BF16, bias=blk.attn.qkv.bias)
        # the head-mean map of a layer in front of the reservation (the rollout's input) comes out of the attention launch itself where the
        # one-launch kernel covers the shape; otherwise it is recomputed from the saved statistics on the side stream
        hm_fused = i < reserve_layer and ops.attn_fwd_hm_ok(H, Nc, D)
        ao, rowmax, zinv = ops.attn_fwd(qkv, B, H, Nc, D, policy=policy, self_keep=True, eps_n=eps_n, headmean=hm[i] if hm_fused else None)
        if i < reserve_layer:
            def side(qkv=qkv, rowmax=rowmax, zinv=zinv, i=i, hm_fused=hm_fused):
                if not hm_fused:
                    ops.attn_headmean(qkv, rowmax, zinv, B, H, N, D, policy=policy, self_keep=True, out=hm[i])
                if side_thr:
                    ops.rollout_threshold(hm[i], thr[i], N)        # the rollout's order statistic of this layer, off the critical path
            # (the side launches of _ROLL_BATCH consecutive layers go out under one main-stream event record)
            # ... except in front of the reservation: the chain needs the last layers' thresholds first, and the main stream waits for it
            lane.submit(side, (qkv, rowmax, zinv, hm, thr), defer=(i % _ROLL_BATCH != _ROLL_BATCH - 1) and i < reserve_layer - _ROLL_TAIL)
            if roll_side and i == reserve_layer - 1:
                # the rollout chain (176 us, one workgroup per sample) depends only on the head-mean maps and thresholds the lane has
                # produced: it runs there, right behind the last map, under the rest of this block instead of in front of the next one
                rolled = ops.rollout_outputs(B, N, reserve_k, 1, x.device)
                lane.submit(lambda: ops.rollout(hm, reserve_layer, B, N, reserve_k, lead=1, thr=thr if side_thr else None, out=rolled),
                            (hm, thr) + rolled)
        s1, s2 = _dp(dp, 2 * i), _dp(dp, 2 * i + 1)
        if fused:
            x1, n2, mean2, rstd2 = ops.rowgemm_resid_ln(ao, store.w16(blk.attn.proj.weight), x, rpt, bias=blk.attn.proj.bias, rowscale=s1, rows_per_group=Nc,
                                                        ln_w=blk.norm2.weight, ln_b=blk.norm2.bias, eps=LN_EPS)
        else:
            x1 = ops.gemm(ao, store.w16(blk.attn.proj.weight), epi=EPI_RESID, bias=blk.attn.proj.bias, res=x, rowscale=s1, rows_per_group=Nc)
            n2, mean2, rstd2 = ops.layernorm_fwd(x1, blk.norm2.weight, blk.norm2.bias, LN_EPS)
        h = torch.empty((M, hid), dtype=torch.uint8, device=x.device)            # gelu'(pre-activation), 8-bit codes (csrc/gemm_common.h)
        g = ops.gemm(n2, store.w16(blk.mlp.fc1.weight), epi=EPI_GELU, bias=blk.mlp.fc1.bias, aux_out=h)
        if fused:
            # the next block's norm1 rides on this block's fc2 product, unless the token gather of the reservation comes in between
            nxt = feats.blocks[i + 1] if (i + 1 < nblk and not (compact and i + 1 == reserve_layer)) else None
            x2, nn1, nm1, nr1 = ops.rowgemm_resid_ln(g, store.w16(blk.mlp.fc2.weight), x1, rpt, bias=blk.mlp.fc2.bias, rowscale=s2, rows_per_group=Nc,
                                                     ln_w=nxt.norm1.weight if nxt is not None else None, ln_b=nxt.norm1.bias if nxt is not None else None,
                                                     eps=LN_EPS)
            pre = (nn1, nm1, nr1) if nxt is not None else None
        else:
            x2 = ops.gemm(g, store.w16(blk.mlp.fc2.weight), epi=EPI_RESID, bias=blk.mlp.fc2.bias, res=x1, rowscale=s2, rows_per_group=Nc)
        if save:
            layers.append(dict(x=x, n1=n1, mean1=mean1, rstd1=rstd1, qkv=qkv, ao=ao, rowmax=rowmax, zinv=zinv, x1=x1, n2=n2,
                               mean2=mean2, rstd2=rstd2, h=h, g=g, policy=policy, s1=s1, s2=s2, N=Nc, eps_n=eps_n,
                               rows=rows if (compact and i == reserve_layer) else None))
        x = x2
    lane.join()
    return x.reshape(B, Nc, D), cls_attn, idx, layers


def gather_rows_map(idx, N):
    """Source-row map of the reserved tokens: per sample [cls, 1+idx...] as flat rows of the [B*N] token matrix
    (integer index plumbing of protopformer.py:156-162)."""
    return ops.reserved_rows_map(idx, N)


def head_tokens_fwd(ppnet, store, x, idx):
    """Final norm on the reserved rows only + add-on 1x1 conv + sigmoid (deit:238; protopformer.py:162-172).
    x is either the full token matrix [B, N, D] (rows gathered here) or already the reserved rows [B, 1+k, D].
    Returns f fp32 [B, 1+k, Dp] (token 0 = cls) and what backward needs."""
    feats = ppnet.features
    B, N, D = x.shape
    k = idx.shape[1]
    row_map = gather_rows_map(idx, N) if N != 1 + k else None
    nf, meanf, rstdf = ops.layernorm_fwd(x.reshape(B * N, D), feats.norm.weight, feats.norm.bias, LN_EPS, row_map=row_map)
    f, chain = addon_fwd(ppnet, store, nf)
    return f.reshape(B, 1 + k, f.shape[-1]), dict(row_map=row_map, nf=nf, meanf=meanf, rstdf=rstdf, chain=chain)


def addon_convs(ppnet):
    """The 1x1 convolutions of add_on_layers in order: one ('regular', protopformer.py:111-114) or the bottleneck chain
    (protopformer.py:90-107) -- ReLU between them, Sigmoid after the last."""
    return [m for m in ppnet.add_on_layers if isinstance(m, nn.Conv2d)]


def addon_fwd(ppnet, store, nf):
    """nf bf16 [rows, D] (final-norm output of the reserved rows) -> (f fp32 [rows, Dp], what addon_bwd needs).
    'regular': one MFMA GEMM with the sigmoid in its epilogue.  'bottleneck': the first convolution on the same GEMM (fp32 out), the narrow
    tail (widths D/2 ... Dp on 1 + k rows per sample: < 1 % of the step's flops) in fp32 through ppf_sgemm."""
    convs = addon_convs(ppnet)
    c0 = convs[0]
    w0 = store.w16(c0.weight).reshape(c0.out_channels, -1)
    if len(convs) == 1:
        return ops.gemm(nf, w0, epi=EPI_SIGMOID_F32, bias=c0.bias), None
    h = ops.relu_f32_(ops.gemm(nf, w0, epi=EPI_F32, bias=c0.bias))
    acts = []
    for j, c in enumerate(convs[1:]):
        acts.append(h)
        h = ops.linear_f32(h, c.weight, c.bias, kind=2 if j == len(convs) - 2 else 4)
    return h, acts


def addon_bwd(ppnet, store, head, f, df):
    """-> dz bf16 [rows, C0]: gradient w.r.t. the FIRST convolution's pre-activation (its bias gradient and every parameter gradient of the
    tail are accumulated here); the callers do the first convolution's weight / input gradients on the bf16 GEMMs as before."""
    convs = addon_convs(ppnet)
    gv = store.grad_view
    if len(convs) == 1:
        return ops.sigmoid_bwd(df, f, gv(convs[0].bias))
    acts = head["chain"]
    d = ops.ew_bwd_f32(1, df.contiguous(), f)                                    # sigmoid'
    for j in reversed(range(1, len(convs))):
        c, x_in = convs[j], acts[j - 1]
        ops.linear_wgrad_f32(d, x_in, gv(c.weight), gv(c.bias))
        d = ops.ew_bwd_f32(5, ops.linear_dgrad_f32(d, c.weight), x_in)          # through the ReLU that produced x_in
    ops.colsum_f32(d, gv(convs[0].bias))
    return ops.cast_bf16(d)


# ------------------------------------------------------------------------------------------------ DeiT backward
class WgradLane:
    """Second HIP stream for the weight-gradient GEMMs.  dW = dy^T x is off the critical path of backward (only the
    optimizer consumes it), and the dgrad / LayerNorm / attention kernels of the chain rarely fill all 256 CUs, so the
    wgrads run concurrently with them.  Hazards: a wgrad reads tensors produced on the main stream (the lane waits for the
    main stream before each launch), temporaries may be freed while the lane still reads them (they are held until the next
    join), and a buffer the main stream writes again while the lane may still read it must be registered with track() and
    guarded with before_overwrite().
    Ordering goes through the library's pooled events (ppf_stream_wait_stream / _mark / _wait_mark).  Every event record is a
    packet in its queue (measured ~3 us of main-stream time each, profiles/r2_lane_events.txt), so they are kept few:
    submit(defer=True) launches together with the NEXT submit under one main-stream event (the LayerNorm column-sum reduction and
    the weight gradient that follows it), and only launches that read a tracked buffer are followed by a side-stream mark.
    The functions submitted here must not allocate torch memory (they run with torch's current stream unchanged and only the
    library's launch stream redirected).  PPF_WGRAD_STREAM=0 runs everything on the main stream."""

    def __init__(self, device):
        self.enabled = os.environ.get("PPF_WGRAD_STREAM", "1") != "0"
        # ONE side stream at the default priority (two or three streams, and a high- / low-priority one, measured within noise in rounds 2-4)
        self.streams = [torch.cuda.Stream(device=device)] if self.enabled else []
        self.raws = [st.cuda_stream for st in self.streams]
        self.last_read = {}         # data_ptr of a tracked buffer -> ticket of the last side-stream launch that reads it
        self.tracked = set()
        self.held = []              # tensors the lane reads, kept alive (so their memory is not reused) until the next join()
        self.pending = []           # deferred launches (fn, reads)

    def track(self, t):
        """t is a buffer the main stream will write again during this backward pass (see before_overwrite)."""
        self.tracked.add(t.data_ptr())
        return t

    def submit(self, fn, reads, tag=None, defer=False):
        if not self.enabled:
            fn()
            return
        self.pending.append((fn, reads, tag))
        self.held.extend(reads)
        if not defer:
            self.flush()

    def flush(self, tag=None):
        if not self.pending:
            return
        raw = self.raws[0]
        try:
            _lib.call("ppf_stream_wait_stream", raw, _lib.stream_ptr())      # ONE main-stream event record for everything pending
            for fn, reads, tg in self.pending:
                _lib.push_stream(raw)
                try:
                    fn()
                    ptrs = [t.data_ptr() for t in reads if t.data_ptr() in self.tracked]
                    if ptrs:
                        ticket = _lib.stream_mark(raw)
                        for q in ptrs:
                            self.last_read[q] = ticket
                finally:
                    _lib.pop_stream()
        finally:
            self.pending.clear()

    def before_overwrite(self, t):
        """The main stream is about to write t: wait for the side stream's last launch that reads it."""
        if t.data_ptr() not in self.tracked:
            raise RuntimeError("WgradLane.before_overwrite: buffer was not registered with track()")
        q = t.data_ptr()
        if any(r.data_ptr() == q for _, reads, _t in self.pending for r in reads):
            self.flush()
        ticket = self.last_read.pop(q, None)
        if ticket is not None:
            _lib.call("ppf_stream_wait_mark", _lib.stream_ptr(), ticket)

    def join(self):
        if self.enabled:
            self.flush()
            for raw in self.raws:
                _lib.call("ppf_stream_wait_stream", _lib.stream_ptr(), raw)
        self.last_read.clear()
        self.tracked.clear()
        self.held.clear()


def wgrad_lane(store):
    lane = getattr(store, "_wgrad_lane", None)
    if lane is None:
        lane = store._wgrad_lane = WgradLane(store.device)
    return lane


# Every non-deferred submit costs the MAIN stream one event-record packet (the side stream waits for it), and a packet boundary is
# 4-9 us of main-queue time (profiles/r4_queue_gaps.txt).  The first weight gradient of each branch (fc2, proj) is therefore parked and
# launched together with the second one (fc1, qkv) under a single record: two records per block instead of four -- for the narrow
# models (embed dim <= 256: deit_tiny +1.8 % same-box, cait_xxs24 +-0), whose kernels are short against the packet boundaries; at
# D = 384 the later start of the parked GEMM costs more than the records (-0.5 %).
_ROLL_TAIL = 1        # layers in front of the reservation whose side launches are never parked
_ROLL_BATCH = 3       # +0.9 % deit_tiny, +0.6 % deit_small same-box against 1


def _wgrad(store, dy16, x16, weight, bias=None, defer=False):
    """dW[N,K] += dy^T x (deterministic split-K into the flat grad), optional fused bias grad (column sums of dy); queued on
    the weight-gradient lane (defer: launched with the next submit, see above)."""
    gw = store.grad_view(weight)
    gb = store.grad_view(bias) if bias is not None else None
    wgrad_lane(store).submit(lambda: ops.gemm(dy16, x16, trans_a=True, trans_b=True, epi=EPI_ATOMIC, out=gw.reshape(weight.shape[0], -1),
                                              colsum=gb), (dy16, x16), defer=defer and min(weight.shape[0], weight.shape[1]) <= 256)


# Round 5: the input gradients of fc1 and qkv (N = 384 outputs, K = 1536 / 1152) through the PLAIN full-row GEMM (rowgemm_bf16: double-buffered
# LDS-DMA stages, 57 / 46 us stand-alone) instead of the 224 x 128 tiles (single-buffered, ~100 us), the LayerNorm backward staying its own
# launch: +1.0 % same-box at deit_small (fc1 +0.4, qkv +0.6, proj -0.2, all three +0.8; profiles/r5_dgrad_row.txt).  Bit mask fc1 (1) / qkv (2) / proj (4).
_DGRAD_ROW = 3


def _dgrad(dy16, store, weight, wt, rows=None, which=0):
    """dx = dy W as bf16.  With the transposed weight shadow (FlatStore.register_transposed) both operands are contraction-contiguous and
    the product can take the 224 x 128 direct-to-LDS kernel (csrc/gemm_bf16.hip gemm224g_kernel: one round of the chip instead of 1.54 for
    the N = 384 outputs); otherwise the [K][N] weight is read transposed by the generic kernel."""
    if wt is not None and (_DGRAD_ROW & which) and rows is not None and ops.rowgemm_ok(wt.shape[0], wt.shape[1], rows):
        return ops.rowgemm_bf16(dy16, wt, rows)
    if wt is not None:
        return ops.gemm(dy16, wt, epi=EPI_BF16)
    return ops.gemm(dy16, store.w16(weight), trans_b=True, epi=EPI_BF16)


def deit_backward(ppnet, store, saved, df):
    """Backward of image -> f. df: fp32 [B*(1+k), Dp] gradient w.r.t. the sigmoid outputs."""
    feats = ppnet.features
    layers, head = saved["layers"], saved["head"]
    x_last = saved["x_last"]                      # [B, N, D] fp32 (pre-final-norm)
    B, N, D = x_last.shape
    M = B * N
    dev = x_last.device
    conv = addon_convs(ppnet)[0]
    # add-on: sigmoid' (and the fp32 tail of a bottleneck head) then the two GEMMs of its first convolution
    lane = wgrad_lane(store)
    lnb = functools.partial(ops.layernorm_bwd, lane=lane, defer_reduce=True)      # column-sum reductions (parameter grads): side stream
    dz = addon_bwd(ppnet, store, head, saved["f"].reshape(-1, saved["f"].shape[-1]), df)
    _wgrad(store, dz, head["nf"], conv.weight)
    dnf = ops.gemm(dz, store.w16(conv.weight).reshape(conv.out_channels, D), trans_b=True, epi=EPI_BF16)
    # final norm backward scatters into the (zero) residual-stream gradient; also emits the bf16 gradient of the last fc2
    # (x_last is already the reserved rows when the last blocks ran compacted: then nothing is scattered here)
    alloc = ops.zeros if head["row_map"] is not None else (lambda shape, dtype, device: torch.empty(shape, dtype=dtype, device=device))
    dx = alloc((M, D), torch.float32, dev)
    dyb = lane.track(alloc((M, D), torch.bfloat16, dev))
    last = feats.blocks[-1]
    lnb(dnf, x_last.reshape(M, D), feats.norm.weight, head["meanf"], head["rstdf"], store.grad_view(feats.norm.weight),
                      store.grad_view(feats.norm.bias), dx_out=dx, row_map=head["row_map"], cast_out=dyb, rowscale=layers[-1]["s2"],
                      rows_per_group=N, dbias_next=store.grad_view(last.mlp.fc2.bias))
    gs = getattr(ppnet, "_grad_sync", None)           # data-parallel: all-reduce chunks as their layers complete
    if gs is not None:
        lane.flush()
        _lib.run_live(lambda: gs.chunk_ready(gs.tail_chunk, also=lane.streams))
    # The bf16 branch gradient alternates between two buffers: the LayerNorm backward that produces the next one does not have to
    # wait for the side stream's weight-gradient GEMM that still reads the current one (the main stream would otherwise be tied to
    # the progress of the side stream twice per block).  Round 4: four buffers in
    # rotation instead of two (the side stream lags up to a block behind at the start of backward: +0.6 % deit_small, +0.5 % deit_tiny
    # same-box; six = four).  Parking the prototype-gradient kernel until a few blocks' weight gradients have run was measured
    # 1 % SLOWER: at the start of backward it overlaps the serial head section, later it lands in the saturated part.
    dyb_alt = None

    # buffers the branch gradient rotates through (2 = ping-pong; 4: +0.6 % same-box at D = 384).  The narrow models never reuse one
    # (64 > two per block): their side stream lags whole blocks behind and every reuse is a main-stream wait (cait_xxs24 +4 % same-box,
    # deit_tiny +0.2 %; 24-48 buffers of B*N*D bf16 = 0.2-0.5 GB of the 288); at D = 384 no reuse measured -0.7 % (larger footprint).
    nring = 4 if D > 256 else 64
    ring = {}

    def next_dyb(cur, alt):
        if nring >= 64:                                   # never reused: nothing to order against the side stream, no marks
            return torch.empty_like(cur), cur
        bufs = ring.setdefault(tuple(cur.shape), [cur])
        if not any(b.data_ptr() == cur.data_ptr() for b in bufs):
            bufs.append(cur)
        if len(bufs) < nring:
            nxt = lane.track(torch.empty_like(cur))
            bufs.append(nxt)
        else:
            i = next(j for j, b in enumerate(bufs) if b.data_ptr() == cur.data_ptr())
            nxt = bufs[(i + 1) % len(bufs)]
        lane.before_overwrite(nxt)
        return nxt, cur

    # Bias gradients of proj / fc2 are column sums of the bf16 branch gradient `dyb`: the LayerNorm-backward KERNEL adds them while it
    # writes dyb (dbias_next); the fused GEMM + LayerNorm-backward (csrc/rowgemm.hip) does not, there the weight-gradient GEMM that
    # reads dyb anyway sums its columns (COLSUM).  bias_done: the producer of the current dyb has already accumulated the bias gradient.
    bias_done = True
    for i in range(len(layers) - 1, -1, -1):
        L, blk = layers[i], feats.blocks[i]
        Nl = L["N"]                                       # tokens per sample in this block (1+k once compacted)
        hid = blk.mlp.fc1.out_features
        w1t, wqt, wpt = store.w16t(blk.mlp.fc1.weight), store.w16t(blk.attn.qkv.weight), store.w16t(blk.attn.proj.weight)
        rpt = ops.rowgemm_tile_rows(B * Nl, Nl, backward=True)
        fused = (_row_bwd(B * Nl, D) and w1t is not None and wqt is not None and wpt is not None and ops.rowgemm_ok(D, hid, rpt) and ops.rowgemm_ok(D, 3 * D, rpt)
                 and ops.rowgemm_ok(D, D, rpt))
        # MLP branch: x2 = x1 + s2 * (gelu(n2 W1^T + b1) W2^T + b2)
        _wgrad(store, dyb, L["g"], blk.mlp.fc2.weight, None if bias_done else blk.mlp.fc2.bias, defer=True)
        w2t = store.w16t(blk.mlp.fc2.weight)
        if w2t is not None:      # contraction-contiguous operands: the direct-to-LDS kernel (gemm128g, four workgroups per CU) takes K = 384
            dh = ops.gemm(dyb, w2t, epi=EPI_DGELU, aux_in=L["h"])
        else:
            dh = ops.gemm(dyb, store.w16(blk.mlp.fc2.weight), trans_b=True, epi=EPI_DGELU, aux_in=L["h"])
        _wgrad(store, dh, L["n2"], blk.mlp.fc1.weight, blk.mlp.fc1.bias)
        dyb, dyb_alt = next_dyb(dyb, dyb_alt)
        if fused:
            ops.rowgemm_lnbwd(dh, w1t, L["x1"], L["mean2"], L["rstd2"], blk.norm2.weight, store.grad_view(blk.norm2.weight), store.grad_view(blk.norm2.bias),
                              rpt, dres_in=dx, dx_out=dx, cast_out=dyb, rowscale=L["s1"], rows_per_group=Nl, lane=lane, defer_reduce=True)
        else:
            dn2 = _dgrad(dh, store, blk.mlp.fc1.weight, w1t, Nl, 1)
            lnb(dn2, L["x1"], blk.norm2.weight, L["mean2"], L["rstd2"], store.grad_view(blk.norm2.weight),
                              store.grad_view(blk.norm2.bias), dres_in=dx, dx_out=dx, cast_out=dyb, rowscale=L["s1"], rows_per_group=Nl,
                              dbias_next=store.grad_view(blk.attn.proj.bias))
        bias_done = not fused
        # attention branch: x1 = x + s1 * (attn(n1) Wp^T + bp)
        _wgrad(store, dyb, L["ao"], blk.attn.proj.weight, None if bias_done else blk.attn.proj.bias, defer=True)
        if fused:
            dao = ops.rowgemm_bf16(dyb, wpt, rpt)
        else:
            dao = _dgrad(dyb, store, blk.attn.proj.weight, wpt, Nl, 4)
        dqkv = ops.attn_bwd(L["qkv"], L["ao"], dao, L["rowmax"], L["zinv"], B, feats.num_heads, Nl, D, policy=L["policy"], self_keep=True, eps_n=L["eps_n"])
        _wgrad(store, dqkv, L["n1"], blk.attn.qkv.weight, blk.attn.qkv.bias)
        dn1 = None if fused else _dgrad(dqkv, store, blk.attn.qkv.weight, wqt, Nl, 2)
        if i > 0:
            prev = feats.blocks[i - 1]
            dyb, dyb_alt = next_dyb(dyb, dyb_alt)
            if fused:
                ops.rowgemm_lnbwd(dqkv, wqt, L["x"], L["mean1"], L["rstd1"], blk.norm1.weight, store.grad_view(blk.norm1.weight), store.grad_view(blk.norm1.bias),
                                  rpt, dres_in=dx, dx_out=dx, cast_out=dyb, rowscale=layers[i - 1]["s2"], rows_per_group=Nl, lane=lane, defer_reduce=True)
            else:
                lnb(dn1, L["x"], blk.norm1.weight, L["mean1"], L["rstd1"], store.grad_view(blk.norm1.weight),
                                  store.grad_view(blk.norm1.bias), dres_in=dx, dx_out=dx, cast_out=dyb, rowscale=layers[i - 1]["s2"],
                                  rows_per_group=Nl, dbias_next=store.grad_view(prev.mlp.fc2.bias))
            bias_done = not fused
            if L["rows"] is not None:
                # this block ran on the reserved rows: hand its input gradient back to the full token matrix (zeros elsewhere)
                Mf = B * layers[i - 1]["N"]
                dx = ops.scatter_rows(dx, L["rows"], Mf)
                dyb = lane.track(ops.scatter_rows(dyb, L["rows"], Mf))
                dyb_alt = None
        else:
            if fused:
                ops.rowgemm_lnbwd(dqkv, wqt, L["x"], L["mean1"], L["rstd1"], blk.norm1.weight, store.grad_view(blk.norm1.weight), store.grad_view(blk.norm1.bias),
                                  rpt, dres_in=dx, dx_out=dx, lane=lane, defer_reduce=True)
            else:
                lnb(dn1, L["x"], blk.norm1.weight, L["mean1"], L["rstd1"], store.grad_view(blk.norm1.weight),
                                  store.grad_view(blk.norm1.bias), dres_in=dx, dx_out=dx)
        if gs is not None and i in gs.block_chunk:
            lane.flush()
            _lib.run_live(lambda c=gs.block_chunk[i]: gs.chunk_ready(c, also=lane.streams))
    # token assembly + patch embedding
    pe = feats.patch_embed
    Np = pe.num_patches
    if layers and layers[0]["rows"] is not None:          # reservation in front of block 0: the embedding sees all tokens
        dx = ops.scatter_rows(dx, layers[0]["rows"], B * (Np + 1))
    dtok = ops.assemble_tokens_bwd(dx, store.grad_view(feats.pos_embed).reshape(Np + 1, D), store.grad_view(feats.cls_token).reshape(D), B, Np, D, 1)
    _wgrad(store, dtok, saved["cols"], pe.proj.weight, pe.proj.bias)
    if gs is not None:
        lane.flush()
        _lib.run_live(lambda: gs.chunk_ready(gs.head_chunk, also=lane.streams))
    lane.join()


class TokensFn(torch.autograd.Function):
    """image -> (f [B,1+k,Dp], cls_token_attn [B,Np], reserve idx [B,k]); parameters enter only to hook autograd."""

    @staticmethod
    def forward(ctx, img, ppnet, dp, *params):
        store = ppnet.flat_store()
        store.refresh_bf16()
        feats = ppnet.features
        need_bwd = any(ctx.needs_input_grad)
        if need_bwd and "t16_params" in ppnet._arch_fns:
            if store._t16 is None:
                store.register_transposed(ppnet._arch_fns["t16_params"](feats))
            # W^T shadow of the input-gradient products: first needed in backward, so for the narrow models the transposition runs on the
            # side stream under the first blocks of the forward pass (launched with the lane's first flush; every path joins the lane before
            # backward): deit_tiny +0.8 % same-box; at D = 384 it is 47 us of a saturated chip wherever it runs (16 625 vs 16 597).
            if store._t16 is not None and not store._t16["fresh"] and feats.embed_dim <= 256:
                wgrad_lane(store).submit(store.refresh_t16, (store.bf16, store._t16["buf"]), defer=True)
            else:
                store.refresh_t16()
        saved = {} if need_bwd else None
        (layer, k), = ppnet.reserve_layer_nums
        fwd = ppnet._arch_fns
        x = fwd["embed"](feats, store, img, saved)
        x_out, cls_attn, idx, layers = fwd["blocks"](feats, store, x, layer, k, dp, save=need_bwd)
        f, head = head_tokens_fwd(ppnet, store, x_out, idx)
        if need_bwd:
            saved.update(layers=layers, head=head, x_last=x_out, f=f)
            ctx.saved = saved
            ctx.ppnet = ppnet
        ctx.mark_non_differentiable(cls_attn, idx)
        return f, cls_attn, idx

    @staticmethod
    def backward(ctx, df, _dcls, _didx):
        ppnet, saved = ctx.ppnet, ctx.saved
        store = ppnet.flat_store()
        store.attach_all_grads()
        ppnet._arch_fns["backward"](ppnet, store, saved, df.contiguous().reshape(-1, df.shape[-1]))
        ctx.saved = None
        return (None, None, None) + (None,) * (len(ctx.needs_input_grad) - 3)


def deit_t16_params(feats):
    """Weights whose input-gradient products read W^T contraction-contiguous (csrc/rowgemm.hip, gemm224g / gemm128g): fc1, qkv, proj, fc2."""
    return [w for blk in feats.blocks for w in (blk.mlp.fc1.weight, blk.attn.qkv.weight, blk.attn.proj.weight, blk.mlp.fc2.weight)]


DEIT_FNS = dict(embed=deit_embed, blocks=deit_blocks_fwd, backward=deit_backward, t16_params=deit_t16_params)
