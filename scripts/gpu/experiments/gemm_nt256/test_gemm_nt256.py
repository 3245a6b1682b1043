"""Parity of the retired 256x256 persistent GEMM through its stand-alone entry (bash build.sh first; MI355X)."""
import ctypes
import os
import sys

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(HERE))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import assert_close          # noqa: E402


def _lib():
    lib = ctypes.CDLL(os.path.join(HERE, "libgemm_nt256.so"))
    lib.ppf_gemm_nt256.restype = ctypes.c_int
    lib.ppf_gemm_nt256.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 4 + [ctypes.c_void_p] * 4
    return lib


@pytest.mark.parametrize("M,N,K,epi", [(24576, 512, 1024, 0), (24600, 520, 768, 4), (24576, 768, 1536, 1)])
def test_gemm_nt256(M, N, K, epi):
    g = torch.Generator().manual_seed(1)
    a = (0.5 * torch.randn(M, K, generator=g)).bfloat16().cuda(); b = (0.05 * torch.randn(N, K, generator=g)).bfloat16().cuda()
    bias = (0.1 * torch.randn(N, generator=g)).cuda(); res = torch.randn(M, N, generator=g).cuda() if epi == 4 else None
    out = torch.empty((M, N), dtype=torch.bfloat16 if epi == 0 else torch.float32, device="cuda")
    run = lambda: _lib().ppf_gemm_nt256(a.data_ptr(), b.data_ptr(), out.data_ptr(), M, N, K, epi, bias.data_ptr(), res.data_ptr() if res is not None else None,
                                        None, torch.cuda.current_stream().cuda_stream)
    assert run() == 0
    ref = a.float() @ b.float().t() + bias + (res if res is not None else 0)
    assert_close(out.float(), ref, rtol=8e-3 if epi == 0 else 1e-3, atol=2e-3, what=f"nt256 epi {epi}")
    first = out.clone()
    for _ in range(5):
        assert run() == 0 and torch.equal(out, first)
