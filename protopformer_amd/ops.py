"""Thin host-side wrappers over the C-ABI kernels (no arithmetic here: allocation, shapes, launch)."""
import os

import torch

from . import _lib

EPI_BF16, EPI_F32, EPI_GELU, EPI_SIGMOID_F32, EPI_RESID, EPI_DGELU, EPI_ATOMIC = range(7)


# the kernel bench.py's roofline probe (ppf_gemm_probe, HIP events inside the C entry point) reports on
DOMINANT_NAME = ("weight-gradient bf16 MFMA GEMM, split over the contraction, ordered partial tiles: wgrad8_kernel<4,2> / <2,4> (256x128 / 128x256 tiles) "
                 "and gemm_kernel<TA=1,TB=1,EPI_PARTIAL,COLSUM> (128x128)")


_WORKSPACE = {}


def _workspace(device, nbytes):
    """Reusable split-K scratch, grown on demand: one per (device, stream) -- the weight-gradient lane (backbone.WgradLane)
    runs its GEMMs on a second stream, concurrently with main-stream kernels that also use a workspace."""
    key = (device, _lib.stream_ptr())
    buf = _WORKSPACE.get(key)
    if buf is None or buf.numel() < nbytes:
        if buf is not None:
            _RETIRED_WS.append(buf)           # a side stream may still be using it (see _gemm_workspace)
        buf = torch.empty(max(nbytes, 64 << 20), dtype=torch.uint8, device=device)
        _WORKSPACE[key] = buf
    return buf


_GEMM_WS = {}
_RETIRED_WS = []


def _gemm_workspace(device, nbytes):
    """Split-K scratch of the weight-gradient GEMMs, one per (device, stream), used by nothing else: every weight gradient writes its
    partial tiles into the same 28 MB and the ordered reduce reads them straight back (they stay in the L2 / memory-side cache;
    giving every GEMM its own range and summing later was measured slower: profiles/r4_reduce_batch.txt)."""
    key = (device, _lib.stream_ptr())
    buf = _GEMM_WS.get(key)
    if buf is None or buf.numel() < nbytes:
        # a buffer that is outgrown stays allocated: the lane may still be using it and torch's allocator knows nothing of that stream
        if buf is not None:
            _RETIRED_WS.append(buf)
        buf = torch.empty(max(nbytes, 64 << 20), dtype=torch.uint8, device=device)
        _GEMM_WS[key] = buf
    return buf


def _chk(t, dtype=None):
    assert t.is_cuda and t.is_contiguous(), "kernel operands must be contiguous CUDA tensors"
    if dtype is not None:
        assert t.dtype == dtype, f"expected {dtype}, got {t.dtype}"
    return t


def gemm(a, b, *, trans_a=False, trans_b=False, epi=EPI_BF16, out=None, bias=None, res=None, rowscale=None,
         rows_per_group=1, colscale=None, aux_in=None, aux_out=None, colsum=None, alpha=1.0):
    """C[M,N] (+)= epi(sum_kc A(m,kc) B(n,kc)); a/b are 2-D bf16. Storage: a is [M,K] ([K,M] if trans_a),
    b is [N,K] ([K,N] if trans_b)."""
    _chk(a, torch.bfloat16), _chk(b, torch.bfloat16)
    M, K = (a.shape[1], a.shape[0]) if trans_a else a.shape
    N = b.shape[1] if trans_b else b.shape[0]
    Kb = b.shape[0] if trans_b else b.shape[1]
    assert K == Kb, f"contraction mismatch {K} vs {Kb}"
    if out is None:
        odt = torch.float32 if epi in (EPI_F32, EPI_SIGMOID_F32, EPI_RESID, EPI_ATOMIC) else torch.bfloat16
        out = torch.empty((M, N), dtype=odt, device=a.device)
        if epi == EPI_ATOMIC:
            zero_(out)
    ldaux = 0
    for t in (aux_in, aux_out):
        if t is not None:
            # EPI_GELU / EPI_DGELU: gelu' as 8-bit codes, ldaux counts BYTES (ABI >= 6); EPI_RESID: the bf16 unscaled branch
            _chk(t, torch.uint8 if epi in (EPI_GELU, EPI_DGELU) else torch.bfloat16)
            ldaux = t.shape[-1]
    ws = None
    if epi == EPI_ATOMIC:
        ws = _gemm_workspace(a.device, _lib.lib().ppf_gemm_workspace_bytes(M, N, K))
    _lib.call("ppf_gemm_bf16", a, b, out, M, N, K, a.shape[1], b.shape[1], out.shape[-1], int(trans_a), int(trans_b), epi,
              bias, res, res.shape[-1] if res is not None else 0, rowscale, rows_per_group, colscale, aux_in, aux_out, ldaux,
              colsum, float(alpha), ws, ws.numel() if ws is not None else 0)
    return out


# ------------------------------------------------------------------------------------------------ full-row GEMM + fused LayerNorm
def rowgemm_ok(D, K, rows_per_tile):
    """True when csrc/rowgemm.hip takes a product with output width D, contraction K and tiles of rows_per_tile rows."""
    return bool(_lib.lib().ppf_rowgemm_supported(int(D), int(K), int(rows_per_tile)))


_CU_COUNT = {}


def rowgemm_tile_rows(M, rows_per_sample, device=None, backward=False):
    """Tile height for the full-row GEMMs: one sample per workgroup, or less when whole samples would leave a third of the CUs without a
    workgroup (batch 128 on 256 CUs: 128 tiles of 197 rows).  Tiles need not end on sample boundaries: every epilogue is row-wise and the
    DropPath scale is indexed by the global row (rows_per_group).
    Forward: half a sample (255 tiles of 99 rows).  backward=True (callers whose side stream is busy with weight-gradient GEMMs while
    these one-per-CU workgroups run): tiles for ~62 % of the CUs, the rest stays free for the side stream -- measured on deit_tiny batch
    128, img/s: whole samples 24.3k, 144-176 rows 24.9-25.0k, half samples 23.6k."""
    dev = torch.cuda.current_device() if device is None else device
    cus = _CU_COUNT.get(dev)
    if cus is None:
        cus = _CU_COUNT[dev] = torch.cuda.get_device_properties(dev).multi_processor_count
    tiles = (M + rows_per_sample - 1) // rows_per_sample
    if 3 * tiles > 2 * cus or rows_per_sample <= 16:
        return rows_per_sample
    half = (rows_per_sample + 1) // 2
    if backward:
        target = max(1, int(0.62 * cus))
        return max(half, min(rows_per_sample, (M + target - 1) // target))
    return half


def rowgemm_bf16(a, b, rows_per_tile, bias=None):
    """bf16 [M, D] = a [M, K] @ b [D, K]^T (+ bias)."""
    _chk(a, torch.bfloat16), _chk(b, torch.bfloat16)
    M, K = a.shape
    D = b.shape[0]
    out = torch.empty((M, D), dtype=torch.bfloat16, device=a.device)
    _lib.call("ppf_rowgemm_bf16", a, b, M, D, K, K, b.shape[1], rows_per_tile, bias, out)
    return out


def rowgemm_resid_ln(a, b, res, rows_per_tile, bias=None, rowscale=None, rows_per_group=1, ln_w=None, ln_b=None, eps=1e-6, colscale=None, aux_out=None):
    """x_out = res + rowscale * colscale * (a @ b^T + bias) (fp32) and, with ln_w / ln_b, n = bf16(LN(x_out)), mean, rstd of the LayerNorm
    that follows; aux_out (bf16 [M, D], optional) receives the unscaled branch a @ b^T + bias.
    Returns (x_out, n, mean, rstd) (the last three None without a LayerNorm)."""
    _chk(a, torch.bfloat16), _chk(b, torch.bfloat16), _chk(res, torch.float32)
    M, K = a.shape
    D = b.shape[0]
    xout = torch.empty((M, D), dtype=torch.float32, device=a.device)
    n = mean = rstd = None
    if ln_w is not None:
        n = torch.empty((M, D), dtype=torch.bfloat16, device=a.device)
        mean = torch.empty(M, dtype=torch.float32, device=a.device)
        rstd = torch.empty(M, dtype=torch.float32, device=a.device)
    _lib.call("ppf_rowgemm_resid_ln", a, b, M, D, K, K, b.shape[1], rows_per_tile, bias, res, xout, rowscale, rows_per_group, colscale, aux_out,
              ln_w, ln_b, n, mean, rstd, float(eps))
    return xout, n, mean, rstd


def rowgemm_lnbwd(a, b, x, mean, rstd, w, dw, db, rows_per_tile, dres_in=None, dx_out=None, cast_out=None, rowscale=None, rows_per_group=1,
                  lane=None, defer_reduce=False, colscale=None, branch=None, dcolscale=None):
    """dn = a @ b^T is the gradient w.r.t. the output of LN(x); dx_out = dres_in + LN'(dn) (fp32), cast_out = bf16(rowscale * dx_out);
    dw / db (the LayerNorm's parameter gradients) += the column sums, reduced in a fixed order by a second small kernel (on `lane`,
    the side stream, when given).  colscale / branch / dcolscale: CaiT's LayerScale on the branch below (cast_out additionally times
    colscale, dcolscale += sum_m rowscale * dx_out * branch)."""
    _chk(a, torch.bfloat16), _chk(b, torch.bfloat16), _chk(x, torch.float32)
    M, K = a.shape
    D = b.shape[0]
    tiles = (M + rows_per_tile - 1) // rows_per_tile
    nparts = 3 if colscale is not None else 2
    part = torch.empty(tiles * nparts * D, dtype=torch.float32, device=a.device)
    if dx_out is None:
        dx_out = torch.empty((M, D), dtype=torch.float32, device=a.device)
    _lib.call("ppf_rowgemm_lnbwd", a, b, M, D, K, K, b.shape[1], rows_per_tile, x, mean, rstd, w, dres_in, dx_out, cast_out, rowscale, rows_per_group,
              colscale, branch, part, part.numel() * 4)
    red = lambda: _lib.call("ppf_rowgemm_colsum", part, tiles, D, nparts, dw, db, dcolscale)
    if lane is not None:
        lane.submit(red, (part,), defer=defer_reduce)
    else:
        red()
    return dx_out


def transpose_bf16_batched(src, dst, desc, n, total_tiles):
    _lib.call("ppf_transpose_bf16_batched", src, dst, desc, n, total_tiles)


def layernorm_fwd(x, w, b, eps=1e-6, row_map=None):
    """x fp32 [R, D] -> (y bf16 [rows, D], mean, rstd); rows = len(row_map) gathers source rows."""
    _chk(x, torch.float32)
    D = x.shape[-1]
    rows = row_map.numel() if row_map is not None else x.numel() // D
    y = torch.empty((rows, D), dtype=torch.bfloat16, device=x.device)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    _lib.call("ppf_layernorm_fwd", x, row_map, w, b, y, mean, rstd, rows, D, float(eps))
    return y, mean, rstd


def layernorm_bwd(dy, x, w, mean, rstd, dw, db, *, dres_in=None, dx_out=None, row_map=None, cast_out=None, rowscale=None,
                  rows_per_group=1, colscale=None, dbias_next=None, branch=None, dcolscale=None, lane=None, defer_reduce=False):
    """dx_out[src] = dres_in[src] + LN'(dy); dw/db accumulate (+=). dy=None: pure scale/cast/colsum pass.
    Large calls write per-workgroup column partials and add them up in a second, deterministic kernel; with `lane`
    (backbone.WgradLane) that reduction -- parameter gradients only -- runs on the side stream; defer_reduce launches it together with
    the lane's next submit (one main-stream ordering event for both: +0.5 % on DeiT, where a weight gradient follows immediately;
    measured -6 % on CaiT, whose scale / cast passes are followed by main-stream work first)."""
    D = x.shape[-1] if x is not None else dres_in.shape[-1]
    if dy is not None:
        rows = dy.numel() // D
    else:
        rows = dres_in.numel() // D
    sums = (dw if dy is not None else None, db if dy is not None else None, dbias_next if cast_out is not None else None,
            dcolscale if (cast_out is not None and branch is not None) else None)
    part = None
    # fixed-order two-pass column sums at every size (bit-identical from run to run)
    if any(t is not None for t in sums):
        dev = (dy if dy is not None else dres_in).device
        part = torch.empty(_lib.lib().ppf_layernorm_bwd_blocks(rows) * 4 * D, dtype=torch.float32, device=dev)
    _lib.call("ppf_layernorm_bwd", dy, x, row_map, w, mean, rstd, dres_in, dx_out, dw, db, cast_out, rowscale, rows_per_group,
              colscale, dbias_next, branch, dcolscale, rows, D, part, part.numel() * 4 if part is not None else 0)
    if part is not None:
        red = lambda: _lib.call("ppf_layernorm_bwd_reduce", part, rows, D, *sums)
        if lane is not None:
            lane.submit(red, (part,), defer=defer_reduce, tag="LNRED")
        else:
            red()


def cast_bf16(src, dst=None):
    _chk(src, torch.float32)
    n = src.numel()
    assert n % 8 == 0
    if dst is None:
        dst = torch.empty(src.shape, dtype=torch.bfloat16, device=src.device)
    _lib.call("ppf_cast_f32_bf16", src, dst, n)
    return dst


def im2col_patch(img, patch):
    _chk(img, torch.float32)
    B, C, H, W = img.shape
    cols = torch.empty((B * (H // patch) * (W // patch), C * patch * patch), dtype=torch.bfloat16, device=img.device)
    _lib.call("ppf_im2col_patch", img, cols, B, C, H, W, patch)
    return cols


def assemble_tokens(tok, cls, pos, B, Np, D, lead):
    x = torch.empty((B, Np + lead, D), dtype=torch.float32, device=tok.device)
    _lib.call("ppf_assemble_tokens", tok, cls, pos, x, B, Np, D, lead)
    return x


def assemble_tokens_bwd(dx, dpos, dcls, B, Np, D, lead):
    dtok = torch.empty((B * Np, D), dtype=torch.bfloat16, device=dx.device)
    _lib.call("ppf_assemble_tokens_bwd", dx, dtok, dpos, dcls, B, Np, D, lead)
    return dtok


def attn_fwd_hm_ok(H, N, D):
    """The one-launch forward + head-mean kernel (csrc/attention.hip attn_fwd16_kernel) covers this (heads, tokens, width)."""
    return bool(_lib.lib().ppf_attn_fwd_hm_supported(H, N, D))


def attn_fwd(qkv, B, H, N, D, policy=None, self_keep=True, eps_n=0, headmean=None):
    """qkv bf16 [B*N, 3D] -> (out bf16 [B*N, D], rowmax, zinv [B,H,N]).  headmean (fp32 [B, N, NP], only where attn_fwd_hm_ok): the
    head-mean probability map of deit:104, written by the same launch."""
    _chk(qkv, torch.bfloat16)
    out = torch.empty((B * N, D), dtype=torch.bfloat16, device=qkv.device)
    rowmax = torch.empty((B, H, N), dtype=torch.float32, device=qkv.device)
    zinv = torch.empty((B, H, N), dtype=torch.float32, device=qkv.device)
    if headmean is not None:
        NP = (N + 3) // 4 * 4
        _lib.call("ppf_attn_fwd_hm", qkv, out, policy, rowmax, zinv, headmean, NP, B, H, N, D, int(self_keep), int(eps_n))
    else:
        _lib.call("ppf_attn_fwd", qkv, out, policy, rowmax, zinv, B, H, N, D, int(self_keep), int(eps_n))
    return out, rowmax, zinv


def attn_headmean(qkv, rowmax, zinv, B, H, N, D, policy=None, self_keep=True, out=None, eps_n=0):
    NP = (N + 3) // 4 * 4
    if out is None:
        out = torch.empty((B, N, NP), dtype=torch.float32, device=qkv.device)
    _lib.call("ppf_attn_headmean", qkv, policy, rowmax, zinv, out, NP, B, H, N, D, int(self_keep), int(eps_n))
    return out


def attn_bwd(qkv, out, dout, rowmax, zinv, B, H, N, D, policy=None, self_keep=True, eps_n=0):
    dqkv = torch.empty_like(qkv)
    delta = torch.empty((B, H, N), dtype=torch.float32, device=qkv.device)
    _lib.call("ppf_attn_bwd", qkv, out, dout, dqkv, policy, rowmax, zinv, delta, B, H, N, D, int(self_keep), int(eps_n))
    return dqkv


def rollout_threshold(hm_layer, thr_out, N, discard_ratio=0.9):
    """thr_out [B] int32 <- per-sample key of the int(N*N*ratio)-th smallest entry of one layer's head-mean map [B, N, NP]."""
    B, _, NP = hm_layer.shape
    _lib.call("ppf_rollout_threshold", hm_layer, B, N, NP, int(N * N * discard_ratio), thr_out)


def rollout_outputs(B, N, k, lead, device):
    """(cls_attn [B,N-lead], idx int32 [B,k], policy [B,N-lead+1]) buffers for rollout(out=...)."""
    Nk = N - lead
    return (torch.empty((B, Nk), dtype=torch.float32, device=device), torch.empty((B, k), dtype=torch.int32, device=device),
            torch.empty((B, Nk + 1), dtype=torch.float32, device=device))


def rollout(hm, L, B, N, k, lead=1, init_rows=None, discard_ratio=0.9, identity=0.2, thr=None, out=None):
    """hm: [L,B,N,NP] fp32 head-mean attention. Returns (cls_attn [B,N-lead], idx int32 [B,k] ascending, policy [B,N-lead+1]);
    out = rollout_outputs(...) to write into existing buffers (no allocation: usable on the side-stream lane)."""
    _chk(hm, torch.float32)
    NP = hm.shape[-1]
    cls_attn, idx, policy = out if out is not None else rollout_outputs(B, N, k, lead, hm.device)
    n_init = init_rows.shape[0] if init_rows is not None else 0
    # discard counts in double precision, exactly like the reference's int(numel * ratio)
    kdrop, kdrop_init = int(N * N * discard_ratio), int((N + 1) * discard_ratio)
    _lib.call("ppf_rollout", hm, B * N * NP, L, B, N, NP, init_rows, n_init, lead, kdrop, kdrop_init, float(identity), k, thr, cls_attn, idx, policy)
    return cls_attn, idx, policy


def proto_fwd(tokens, t0, T, protos, act_kind=0, eps=1e-4, want_dist=True, want_act=True):
    """tokens fp32 [B, Ttot, Dp]; uses tokens[:, t0:t0+T]. protos fp32 [P, Dp]."""
    _chk(tokens, torch.float32), _chk(protos, torch.float32)
    B, Ttot, Dp = tokens.shape
    P = protos.shape[0]
    dev = tokens.device
    act_max = torch.empty((B, P), dtype=torch.float32, device=dev)
    argmax = torch.empty((B, P), dtype=torch.int32, device=dev) if T > 1 else None
    dist = torch.empty((B, P, T), dtype=torch.float32, device=dev) if want_dist else None
    act = torch.empty((B, P, T), dtype=torch.float32, device=dev) if want_act else None
    ws = _workspace(dev, _lib.lib().ppf_proto_fwd_workspace(P, Dp)) if T > 1 else None      # pre-split prototype planes (per-stream scratch)
    _lib.call("ppf_proto_fwd", tokens, Ttot * Dp, t0, T, protos, B, P, Dp, act_kind, float(eps), act_max, argmax, dist, act,
              ws, ws.numel() if ws is not None else 0)
    return act_max, argmax, dist, act


def proto_bwd(tokens, t0, T, protos, dist, g_full, g_max, argmax, dtok, dprotos, act_kind=0, eps=1e-4, rows=None, from_act=False):
    """Backward of proto_fwd.  g_full [B,P,T] is the dense gradient of the activation maps; rows = (g_rows [B,ppc,T], label int64 [B], ppc)
    gives the same gradient in the block form of the PPC loss instead (ppf_proto_bwd_rows: nothing of shape (B,P,T) is touched).
    from_act: `dist` is the activation map act_full of the forward (the derivative is taken from it; T > 1 only)."""
    B, Ttot, Dp = tokens.shape
    P = protos.shape[0]
    if T == 1 and from_act:
        raise ValueError("proto_bwd: from_act applies to the pooled branch (T > 1)")
    if T == 1 and (g_full is None) != (g_max is None):
        # one token per sample: every (sample, prototype) pair carries a gradient -> two dense fp32 products instead of the gather
        ws = _workspace(tokens.device, _lib.lib().ppf_proto_bwd_single_workspace(B, P, Dp))
        _lib.call("ppf_proto_bwd_single", tokens, Ttot * Dp, t0, protos, B, P, Dp, act_kind, float(eps), dist, g_max if g_max is not None else g_full,
                  dtok, Ttot * Dp, dprotos, ws, ws.numel())
        return
    need = _lib.lib().ppf_proto_bwd_workspace(B, T, P, Dp, int(dtok is not None), int(dprotos is not None))
    if dtok is not None:
        ws = torch.empty(need, dtype=torch.uint8, device=tokens.device)
        _lib.call("ppf_memset_zero", ws, _lib.lib().ppf_proto_bwd_workspace(B, T, P, Dp, 1, 0))     # the bitmap; the scratch behind it is write-first
    else:
        ws = _workspace(tokens.device, need) if need else None    # prototype gradients only: per-stream scratch, any content
    if rows is not None:
        if g_full is not None:
            raise ValueError("proto_bwd: pass the activation-map gradient either dense (g_full) or in block form (rows), not both")
        g_rows, label, ppc = rows
        _lib.call("ppf_proto_bwd_rows", tokens, Ttot * Dp, t0, T, protos, B, P, Dp, act_kind, float(eps), dist, int(from_act), g_rows, label, int(ppc), g_max,
                  argmax, dtok, Ttot * Dp, dprotos, ws, need)
        return
    _lib.call("ppf_proto_bwd", tokens, Ttot * Dp, t0, T, protos, B, P, Dp, act_kind, float(eps), dist, int(from_act), g_full, g_max, argmax, dtok,
              Ttot * Dp, dprotos, ws, need)


def ppc_loss(act, idx, label, ppc, side, cov_thresh, mean_thresh):
    """act [B,P,T] fp32, idx [B,T] int32, label [B] int64 -> (loss[2] = (cov, mean), gcov, gmean [B,ppc,T])."""
    B, P, T = act.shape
    dev = act.device
    partial = torch.empty((B, 2), dtype=torch.float32, device=dev)
    gcov = torch.empty((B, ppc, T), dtype=torch.float32, device=dev)
    gmean = torch.empty((B, ppc, T), dtype=torch.float32, device=dev)
    loss = torch.empty(2, dtype=torch.float32, device=dev)
    _lib.call("ppf_ppc_loss", act, idx, label, B, P, T, ppc, side, float(cov_thresh), float(mean_thresh), partial, gcov, gmean, loss)
    return loss, gcov, gmean


def ppc_loss_bwd_rows(gcov, gmean, up_cov, up_mean):
    """up_cov * gcov + up_mean * gmean as [B, ppc, T]: the PPC gradient in block form (row k of sample b belongs to prototype label[b]*ppc + k)."""
    B, ppc, T = gcov.shape
    rows = torch.empty_like(gcov)
    _lib.call("ppf_ppc_loss_bwd", gcov, gmean, up_cov, up_mean, _zero_labels(gcov.device, B), rows, B, ppc, T, ppc)
    return rows


_ZERO_LABELS = {}


def _zero_labels(device, B):
    t = _ZERO_LABELS.get(device)
    if t is None or t.numel() < B:
        t = zeros((max(B, 1024),), torch.int64, device)
        _ZERO_LABELS[device] = t
    return t


def ppc_loss_bwd(gcov, gmean, up_cov, up_mean, label, P):
    B, ppc, T = gcov.shape
    g_full = zeros((B, P, T), torch.float32, gcov.device)
    _lib.call("ppf_ppc_loss_bwd", gcov, gmean, up_cov, up_mean, label, g_full, B, P, T, ppc)
    return g_full


def cross_entropy(logits, label):
    B, C = logits.shape
    per = torch.empty(B, dtype=torch.float32, device=logits.device)
    dlogits = torch.empty_like(logits)
    loss = torch.empty(1, dtype=torch.float32, device=logits.device)
    _lib.call("ppf_cross_entropy", logits, label, per, dlogits, loss, B, C)
    return loss, dlogits


def sgemm(a, b, out, M, N, K, sam, sak, sbn, sbk, alpha=1.0, beta=0.0):
    ws = _workspace(out.device, 16 * M * N * 4)
    _lib.call("ppf_sgemm", a, b, out, M, N, K, sam, sak, sbn, sbk, out.shape[-1], float(alpha), float(beta), ws, ws.numel() // 4)
    return out


def sgemm_pair(a0, b0, out0, N0, K0, sa0, sb0, alpha0, a1, b1, out1, N1, K1, sa1, sb1, alpha1, M, total=None, c0=0.0, c1=0.0):
    """out_i = alpha_i A_i B_i^T (i = 0, 1) and total = c0 A0 B0^T + c1 A1 B1^T in one launch + one reduction; sa_i = (row stride, k stride)
    of A_i, sb_i = (row stride, k stride) of B_i (B_i indexed [n, k])."""
    nfl = _lib.lib().ppf_sgemm_pair_workspace(M, N0, K0, N1, K1)
    ws = _workspace(a0.device, nfl * 4)
    _lib.call("ppf_sgemm_pair", a0, b0, out0, N0, K0, sa0[0], sa0[1], sb0[0], sb0[1], out0.shape[-1] if out0 is not None else 0, float(alpha0),
              a1, b1, out1, N1, K1, sa1[0], sa1[1], sb1[0], sb1[1], out1.shape[-1] if out1 is not None else 0, float(alpha1),
              total, total.shape[-1] if total is not None else 0, float(c0), float(c1), M, ws, ws.numel() // 4)


def axpbypcz(x, y, z, a, b, c):
    out = torch.empty_like(x)
    _lib.call("ppf_axpbypcz", x, y, z, out, float(a), float(b), float(c), x.numel())
    return out


_CONST = {}


def const_scalar(device, value):
    """A cached one-element fp32 device tensor (never written): seed / fixed coefficients of the loss backward without a fill launch."""
    key = (device, float(value))
    t = _CONST.get(key)
    if t is None:
        t = _CONST[key] = torch.full((), float(value), dtype=torch.float32, device=device)
    return t


def is_const_one(t):
    c = _CONST.get((t.device, 1.0))
    return c is not None and t.data_ptr() == c.data_ptr()


def axpby(x, y, a, b, out=None):
    if out is None:
        out = torch.empty_like(x)
    _lib.call("ppf_axpby", x, y, out, float(a), float(b), x.numel())
    return out


def topk_sorted(scores, k):
    """Ascending indices (int32) of the k largest entries of each row of scores [B, n<=256]."""
    _chk(scores, torch.float32)
    B, n = scores.shape
    idx = torch.empty((B, k), dtype=torch.int32, device=scores.device)
    _lib.call("ppf_topk_sorted", scores, B, n, k, idx)
    return idx


def sigmoid_bwd(df, f, dbias):
    rows, cols = f.shape
    dz = torch.empty((rows, cols), dtype=torch.bfloat16, device=f.device)
    part = torch.empty(_lib.lib().ppf_sigmoid_bwd_blocks(rows) * cols, dtype=torch.float32, device=f.device) if dbias is not None else None
    _lib.call("ppf_sigmoid_bwd", df, f, dz, dbias, rows, cols, part, part.numel() * 4 if part is not None else 0)
    return dz


# ------------------------------------------------------------------------------------------------ plumbing kernels
def zeros(shape, dtype, device):
    """torch.zeros on the library's own memset node (keeps a captured step free of ATen fill kernels)."""
    t = torch.empty(shape, dtype=dtype, device=device)
    _lib.call("ppf_memset_zero", t, t.numel() * t.element_size())
    return t


def zero_(t):
    _lib.call("ppf_memset_zero", t, t.numel() * t.element_size())
    return t


def copy_2d(dst, src):
    """dst[...] = src[...] for 2-D views (rows, cols) whose rows are contiguous: strided sub-matrix copy on the library's launch path
    (recordable; replaces torch.cat / slice.contiguous() / slice.copy_ in the CaiT class-attention stage)."""
    assert dst.dim() == 2 and src.dim() == 2 and dst.shape == src.shape and dst.stride(1) == 1 and src.stride(1) == 1 and dst.dtype == src.dtype
    es = dst.element_size()
    _lib.call("ppf_copy_2d", dst, dst.stride(0) * es, src, src.stride(0) * es, dst.shape[1] * es, dst.shape[0])
    return dst


def cat_rows(first, rest):
    """torch.cat([first [B,1,D], rest [B,N,D]], dim=1) -> [B, 1+N, D] with two strided copies."""
    B, N, D = rest.shape
    out = torch.empty((B, N + 1, D), dtype=rest.dtype, device=rest.device)
    o2 = out.reshape(B, (N + 1) * D)
    copy_2d(o2[:, :D], first.reshape(B, D))
    copy_2d(o2[:, D:], rest.reshape(B, N * D))
    return out


def reserved_rows_map(idx, N):
    """Flat source rows [B*(1+k)] int32 of [cls, 1+idx...] in the [B*N] token matrix."""
    B, k = idx.shape
    rows = torch.empty(B * (k + 1), dtype=torch.int32, device=idx.device)
    _lib.call("ppf_reserved_rows_map", idx, rows, B, k, N)
    return rows


def gather_rows(src, rows):
    """dst[r] = src[rows[r]] for a 2-D contiguous src."""
    _chk(src)
    dst = torch.empty((rows.numel(), src.shape[1]), dtype=src.dtype, device=src.device)
    _lib.call("ppf_gather_rows", src, rows, dst, rows.numel(), src.shape[1] * src.element_size())
    return dst


def scatter_rows(src, rows, nrows_dst):
    """zeros(nrows_dst, D) with dst[rows[r]] = src[r]."""
    _chk(src)
    dst = torch.empty((nrows_dst, src.shape[1]), dtype=src.dtype, device=src.device)
    _lib.call("ppf_scatter_rows", src, rows, dst, src.shape[0], nrows_dst, src.shape[1] * src.element_size())
    return dst


def droppath_draw(out, keep, seed, state):
    """out [nslot, B] <- floor(keep + U)/keep; advances the device-resident step counter `state` (int64[1])."""
    nslot, B = out.shape
    _lib.call("ppf_droppath_scales", out, keep, nslot, B, int(seed) & 0xFFFFFFFFFFFFFFFF, state)
    return out


def scale_by_scalar(x, scalar):
    out = torch.empty_like(x)
    _lib.call("ppf_scale_by_scalar", x, scalar, out, x.numel())
    return out


# ------------------------------------------------------------------------------------------------ CaiT
def gemm_batched(a, b, c, M, N, K, lda, ldb, ldc, trans_a, trans_b, out_f32, alpha, bo, bi, sa, sb, sc, kpad=0):
    """Raw batched GEMM on pointers with element offsets: sa/sb/sc = (outer stride, inner stride)."""
    _lib.call("ppf_gemm_bf16_batched", a, b, c, M, N, K, lda, ldb, ldc, int(trans_a), int(trans_b), int(out_f32), float(alpha), bo, bi,
              sa[0], sa[1], sb[0], sb[1], sc[0], sc[1], int(kpad))


class _Off:
    """A device pointer at an element offset into a tensor (for sub-matrix views handed to the C ABI)."""

    def __init__(self, t, off):
        self.t, self.off = t, off

    def data_ptr(self):
        return self.t.data_ptr() + self.off * self.t.element_size()


def th_scores(qkv, wl, bl, B, H, N, D):
    NP = (N + 3) // 4 * 4
    sp = torch.empty((B, H, N, NP), dtype=torch.float32, device=qkv.device)
    _lib.call("ppf_th_scores", qkv, wl, bl, sp, B, H, N, D, NP)
    return sp


def th_softmax_mix(sp, ww, bw, hm_out):
    B, H, N, NP = sp.shape
    NPK = (N + 7) // 8 * 8
    a16 = torch.empty((B, H, N, NPK), dtype=torch.bfloat16, device=sp.device)
    _lib.call("ppf_th_softmax_mix", sp, a16, hm_out, ww, bw, B, H, N, NP, NPK)
    return a16


def th_softmax_bwd(prob, da, ww, wl, dww, dbw, dbl):
    B, H, N, NP = prob.shape
    NPK = (N + 7) // 8 * 8
    ds16 = torch.empty((B, H, N, NPK), dtype=torch.bfloat16, device=prob.device)
    _lib.call("ppf_th_softmax_bwd", prob, da, ds16, ww, wl, dww, dbw, dbl, B, H, N, NP, NPK)
    return ds16


def th_dwl(qkv, ds_prime, dwl, B, H, N, D):
    _lib.call("ppf_th_dwl", qkv, ds_prime, dwl, B, H, N, D, ds_prime.shape[-1])


def _th_mode():
    """PPF_TH_FUSED: "1" (default) the fused talking-heads kernels incl. A.V forward and one launch for dQ / dK / dV backward; "pv0" the fused
    kernels up to A / dS with the per-head products as batched GEMMs; "0" the materialising kernels (what shapes outside the fused kernels'
    cover get).  The two non-default values exist so that tests/test_gpu_cait.py exercises those fallback kernels at a shape the fused ones cover."""
    return os.environ.get("PPF_TH_FUSED", "1")


def th_fused_ok(H, N, D):
    """The fused talking-heads kernels (csrc/cait.hip th_fwd / th_bwd) cover this (heads, tokens, width)."""
    return _th_mode() != "0" and bool(_lib.lib().ppf_th_fused_supported(H, N, D))


def th_pv_fused():
    return _th_mode() == "1"


def th_fwd(qkv, wl, bl, ww, bw, hm_out, B, H, N, D, with_out=True):
    """cait:119-128 in one launch: returns (a16 bf16 [B,H,N,NPK] = proj_w(softmax(proj_l(scale q k^T))), rowmax, zinv [B,H,N], out);
    hm_out [B,N,NP] receives the head mean of the mixed probabilities (rollout input); out = a16 @ v as bf16 [B*N, D] (None when
    with_out is False: the caller multiplies)."""
    NP, NPK = (N + 3) // 4 * 4, (N + 7) // 8 * 8
    a16 = torch.empty((B, H, N, NPK), dtype=torch.bfloat16, device=qkv.device)
    rowmax = torch.empty((B, H, N), dtype=torch.float32, device=qkv.device)
    zinv = torch.empty_like(rowmax)
    out = torch.empty((B * N, D), dtype=torch.bfloat16, device=qkv.device) if with_out else None
    _lib.call("ppf_th_fwd", qkv, wl, bl, ww, bw, a16, hm_out, rowmax, zinv, out, B, H, N, D, NP, NPK)
    return a16, rowmax, zinv, out


def th_bwd(qkv, dout, wl, bl, ww, rowmax, zinv, B, H, N, D):
    """Backward of th_fwd up to dS: returns (ds16 bf16 [B,H,N,NPK], partial) -- partial holds the per-workgroup parameter-gradient
    sums for th_param_reduce."""
    NPK = (N + 7) // 8 * 8
    ds16 = torch.empty((B, H, N, NPK), dtype=torch.bfloat16, device=qkv.device)
    partial = torch.empty((int(_lib.lib().ppf_th_bwd_partial_floats(B, H, N)),), dtype=torch.float32, device=qkv.device)
    _lib.call("ppf_th_bwd", qkv, dout, wl, bl, ww, rowmax, zinv, ds16, partial, B, H, N, D, NPK)
    return ds16, partial


def th_grads_ok(H, N, D):
    return _th_mode() == "1" and bool(_lib.lib().ppf_th_grads_supported(H, N, D))


def th_grads(qkv, dout, ds16, a16, dqkv, B, H, N, D):
    """dqkv bf16 [B*N, 3D] <- dQ | dK | dV of the talking-heads attention from dS (th_bwd), A (th_fwd), q / k (packed qkv) and dO: one launch."""
    _lib.call("ppf_th_grads", qkv, dout, ds16, a16, dqkv, B, H, N, D, a16.shape[-1])
    return dqkv


def th_param_reduce(partial, B, H, N, dww, dbw, dbl, dwl):
    _lib.call("ppf_th_param_reduce", partial, B, H, N, dww, dbw, dbl, dwl)


def class_attn_fwd(q, k, v, policy, B, H, N1, D, rowmean=None):
    dev = q.device
    attn = torch.empty((B, H, N1), dtype=torch.float32, device=dev)
    zinv = torch.empty((B, H), dtype=torch.float32, device=dev)
    if rowmean is None:
        rowmean = torch.empty((B, N1), dtype=torch.float32, device=dev)
    out = torch.empty((B, D), dtype=torch.bfloat16, device=dev)
    _lib.call("ppf_class_attn_fwd", q, k, v, policy, attn, zinv, rowmean, out, B, H, N1, D)
    return out, attn, zinv, rowmean


def class_attn_bwd(q, k, v, attn, zinv, dout, B, H, N1, D):
    dq = torch.empty_like(q)
    dk = torch.empty_like(k)
    dv = torch.empty_like(v)
    _lib.call("ppf_class_attn_bwd", q, k, v, attn, zinv, dout, dq, dk, dv, B, H, N1, D)
    return dq, dk, dv


def merge3_cast(a, b, cq, N1):
    rows, D = a.shape
    out = torch.empty((rows, D), dtype=torch.bfloat16, device=a.device)
    _lib.call("ppf_merge3_cast", a, b, cq, out, rows, D, N1)
    return out


# ------------------------------------------------------------------------------------------------ fp32 verification path
def linear_f32(x, w, bias=None, kind=0, res=None, rowscale=None, rows_per_group=1, colscale=None):
    """fp32 y = epilogue(x W^T + b) through ppf_sgemm (fp32 FMA) + ppf_epilogue_f32.  x [M,K] (row stride may exceed K), w [N,K]."""
    M, K = x.shape
    N = w.shape[0]
    w2 = w.reshape(N, -1)
    assert w2.shape[1] == K and w2.is_contiguous() and x.stride(1) == 1
    out = torch.empty((M, N), dtype=torch.float32, device=x.device)
    ws = _workspace(x.device, 16 * M * N * 4) if M * N <= (1 << 22) else None
    _lib.call("ppf_sgemm", x, w2, out, M, N, K, x.stride(0), 1, K, 1, N, 1.0, 0.0, ws, ws.numel() // 4 if ws is not None else 0)
    _lib.call("ppf_epilogue_f32", out, bias, kind, res, rowscale, rows_per_group, colscale, M, N)
    return out


def layernorm_fwd_f32(x, w, b, eps=1e-6, row_map=None):
    D = x.shape[-1]
    rows = row_map.numel() if row_map is not None else x.numel() // D
    y = torch.empty((rows, D), dtype=torch.float32, device=x.device)
    _lib.call("ppf_layernorm_fwd_f32", x, row_map, w, b, y, rows, D, float(eps))
    return y


def im2col_patch_f32(img, patch):
    B, C, H, W = img.shape
    cols = torch.empty((B * (H // patch) * (W // patch), C * patch * patch), dtype=torch.float32, device=img.device)
    _lib.call("ppf_im2col_patch_f32", img, cols, B, C, H, W, patch)
    return cols


def attn_fwd_f32(qkv, B, H, N, D, policy=None, self_keep=True, headmean=None, eps_n=0):
    out = torch.empty((B * N, D), dtype=torch.float32, device=qkv.device)
    NP = headmean.shape[-1] if headmean is not None else 0
    _lib.call("ppf_attn_fwd_f32", qkv, out, policy, headmean, NP, B, H, N, D, int(self_keep), int(eps_n))
    return out


def th_attn_fwd_f32(qkv, wl, bl, ww, bw, B, H, N, D, headmean):
    out = torch.empty((B * N, D), dtype=torch.float32, device=qkv.device)
    _lib.call("ppf_th_attn_fwd_f32", qkv, wl, bl, ww, bw, out, headmean, headmean.shape[-1], B, H, N, D)
    return out


def class_attn_fwd_f32(q, k, v, policy, B, H, N1, D):
    out = torch.empty((B, D), dtype=torch.float32, device=q.device)
    attn_mean = torch.empty((B, N1), dtype=torch.float32, device=q.device)
    _lib.call("ppf_class_attn_fwd_f32", q, k, v, policy, attn_mean, out, B, H, N1, D)
    return out, attn_mean


# ------------------------------------------------------------------------------------------------ fp32 verification mode: backward pieces
def layernorm_bwd_f32(dy, x, w, dw, db, dx_out, dres_in=None, row_map=None, eps=1e-6):
    """dx_out[xrow] = (dres_in[xrow] or 0) + LN'(dy); dw / db += (fp32 atomics); rows of dy <-> rows row_map[r] of x (None: identity)."""
    rows, D = dy.shape
    _lib.call("ppf_layernorm_bwd_f32", dy, x, row_map, w, dres_in, dx_out, dw, db, rows, D, float(eps))
    return dx_out


def ew_bwd_f32(kind, a, b=None, rowscale=None, rows_per_group=1):
    """kind 0: a * gelu'(b) | 1: a * b * (1 - b) | 2: a * rowscale[row // rows_per_group] (rowscale None: copy) | 3: a * b |
    4: a * rowscale[..] * b[col] (b a LayerScale vector) | 5: a where b > 0 else 0 (ReLU', b = the ReLU's output)."""
    M, N = a.shape
    out = torch.empty_like(a)
    _lib.call("ppf_ew_bwd_f32", int(kind), a, b, out, rowscale, rows_per_group, M, N)
    return out


def colsum_f32(x, out):
    """out[n] += sum_m x[m, n]."""
    M, N = x.shape
    _lib.call("ppf_colsum_f32", x, out, M, N)


def attn_bwd_f32(qkv, dout, B, H, N, D, policy=None, self_keep=True, eps_n=0):
    dqkv = torch.empty_like(qkv)
    scratch = torch.empty(B * H * 2 * N * N, dtype=torch.float32, device=qkv.device)
    _lib.call("ppf_attn_bwd_f32", qkv, dout, policy, dqkv, scratch, B, H, N, D, int(self_keep), int(eps_n))
    return dqkv


def linear_dgrad_f32(dy, w):
    """dx [M, K] = dy [M, N] @ w [N, K] through ppf_sgemm (fp32 FMA)."""
    M, N = dy.shape
    w2 = w.reshape(N, -1)
    K = w2.shape[1]
    out = torch.empty((M, K), dtype=torch.float32, device=dy.device)
    return sgemm(dy, w2, out, M, K, N, N, 1, 1, K)


def linear_wgrad_f32(dy, x, gw, gb=None):
    """gw [N, K] += dy [M, N]^T @ x [M, K] (and gb [N] += column sums of dy) through ppf_sgemm, accumulating into the flat gradient views."""
    M, N = dy.shape
    K = x.shape[1]
    sgemm(dy, x, gw.reshape(N, K), N, K, M, 1, N, 1, K, alpha=1.0, beta=1.0)
    if gb is not None:
        colsum_f32(dy, gb)


def th_attn_bwd_f32(qkv, dout, mix, gmix, B, H, N, D):
    """mix = (wl, bl, ww, bw) of the talking-heads mixers, gmix their gradient views (accumulated)."""
    dqkv = zeros(qkv.shape, torch.float32, qkv.device)
    _lib.call("ppf_th_attn_bwd_f32", qkv, dout, mix[0], mix[1], mix[2], mix[3], dqkv, gmix[0], gmix[1], gmix[2], gmix[3], B, H, N, D)
    return dqkv


def class_attn_bwd_f32(q, k, v, policy, dout, B, H, N1, D):
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    _lib.call("ppf_class_attn_bwd_f32", q, k, v, policy, dout, dq, dk, dv, B, H, N1, D)
    return dq, dk, dv


def relu_f32_(x):
    """in place max(x, 0) on an fp32 [M, N] matrix."""
    M, N = x.shape
    _lib.call("ppf_epilogue_f32", x, None, 4, None, None, 1, None, M, N)
    return x

