"""ctypes wrapper of the retired fused MLP forward (scripts/gpu/experiments/mlpfwd/mlpfwd.hip -> libmlpfwd.so, see build.sh).
This was protopformer_amd.ops.mlp_fwd until round 6; it is NOT part of libppf_hip.so or of the train step any more
(290-305 us per deit_small layer against 214 us for the two launches it replaced: profiles/r5_mlp_fused.txt)."""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, "libmlpfwd.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} not found: run scripts/gpu/experiments/mlpfwd/build.sh")
        _lib = ctypes.CDLL(path)
        _lib.ppf_mlp_fwd_supported.restype = ctypes.c_int
        _lib.ppf_mlp_fwd_supported.argtypes = [ctypes.c_int] * 3
        P, I, F = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
        _lib.ppf_mlp_fwd.restype = ctypes.c_int
        _lib.ppf_mlp_fwd.argtypes = [P] * 5 + [I] * 4 + [P, P] + [P, P] + [P, I] + [P] + [P] + [P] * 5 + [F] + [P]
    return _lib


def _p(t):
    return None if t is None else t.data_ptr()


def mlp_fwd_supported(D, hid, rows_per_tile):
    return bool(lib().ppf_mlp_fwd_supported(int(D), int(hid), int(rows_per_tile)))


def mlp_fwd(a, w1, b1, w2, b2, res, rows_per_tile, rowscale=None, rows_per_group=1, ln_w=None, ln_b=None, eps=1e-6, colscale=None, aux_out=None):
    """timm Mlp + residual + the following LayerNorm in one launch: returns (x_out, n, mean, rstd, h, dgelu)."""
    M, D = a.shape
    hid = w1.shape[0]
    dev = a.device
    h = torch.empty((M, hid), dtype=torch.bfloat16, device=dev)
    dg = torch.empty((M, hid), dtype=torch.uint8, device=dev)
    xout = torch.empty((M, D), dtype=torch.float32, device=dev)
    n = mean = rstd = None
    if ln_w is not None:
        n = torch.empty((M, D), dtype=torch.bfloat16, device=dev)
        mean = torch.empty(M, dtype=torch.float32, device=dev)
        rstd = torch.empty(M, dtype=torch.float32, device=dev)
    rc = lib().ppf_mlp_fwd(_p(a), _p(w1), _p(b1), _p(w2), _p(b2), M, D, hid, rows_per_tile, _p(h), _p(dg), _p(res), _p(xout), _p(rowscale), rows_per_group,
                           _p(colscale), _p(aux_out), _p(ln_w), _p(ln_b), _p(n), _p(mean), _p(rstd), float(eps), torch.cuda.current_stream().cuda_stream)
    if rc != 0:
        raise RuntimeError(f"ppf_mlp_fwd failed (rc={rc})")
    return xout, n, mean, rstd, h, dg
