"""Thin host-side wrappers over the C-ABI kernels (no arithmetic here: allocation, shapes, launch)."""
import torch

from . import _lib

EPI_BF16, EPI_F32, EPI_GELU, EPI_SIGMOID_F32, EPI_RESID, EPI_DGELU, EPI_ATOMIC = range(7)


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
        out = (torch.zeros if epi == EPI_ATOMIC else torch.empty)((M, N), dtype=odt, device=a.device)
    ldaux = 0
    for t in (aux_in, aux_out):
        if t is not None:
            ldaux = t.shape[-1]
    _lib.call("ppf_gemm_bf16", a, b, out, M, N, K, a.shape[1], b.shape[1], out.shape[-1], int(trans_a), int(trans_b), epi,
              bias, res, res.shape[-1] if res is not None else 0, rowscale, rows_per_group, colscale, aux_in, aux_out, ldaux,
              colsum, float(alpha))
    return out
