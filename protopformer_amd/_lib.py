"""ctypes binding of the C-ABI HIP library (include/ppf_hip.h).

There is NO CPU or eager fallback: if the shared library is missing or a call fails, this raises."""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libppf_hip.so")

# signature spec per entry point: p = device/host pointer, i = int32, l = int64, f = float, s = hipStream_t
SIGS = {
    "ppf_gemm_bf16": "pppiiiiiiiiippipipppipf" "pz" "s",
    "ppf_device_info": "pppi",
    "ppf_gemm_probe": "i",
    "ppf_gemm_probe_read": "pppp",
    "ppf_layernorm_fwd": "ppppppp" "iif" "s",
    "ppf_layernorm_bwd": "pppppp" "pppp" "pp" "i" "pppp" "ii" "pz" "s",
    "ppf_layernorm_bwd_reduce": "p" "ii" "pppp" "s",
    "ppf_cast_f32_bf16": "ppls",
    "ppf_im2col_patch": "ppiiiiis",
    "ppf_assemble_tokens": "ppppiiiis",
    "ppf_assemble_tokens_bwd": "ppppiiiis",
    "ppf_adamw_step": "pppppp" "li" "ppp" "fff" "i" "ff" "s",
    "ppf_attn_fwd": "ppppp" "iiiii" "i" "s",
    "ppf_attn_headmean": "ppppp" "i" "iiiii" "i" "s",
    "ppf_attn_bwd": "pppppppp" "iiiii" "i" "s",
    "ppf_rollout": "pl" "iiii" "p" "iiii" "f" "i" "ppp" "s",
    "ppf_proto_fwd": "pliip" "iiii" "f" "pppp" "s",
    "ppf_proto_bwd": "pliip" "iiii" "f" "ppppp" "l" "p" "pz" "s",
    "ppf_ppc_loss": "ppp" "iiiii" "ff" "pppp" "s",
    "ppf_ppc_loss_bwd": "pppppp" "iiii" "s",
    "ppf_cross_entropy": "ppppp" "ii" "s",
    "ppf_sgemm": "ppp" "iii" "llll" "i" "ff" "pl" "s",
    "ppf_axpby": "ppp" "ff" "l" "s",
    "ppf_topk_sorted": "piiips",
    "ppf_gemm_bf16_batched": "ppp" "iiiiii" "iii" "f" "ii" "llllll" "i" "s",
    "ppf_th_scores": "pppp" "iiiii" "s",
    "ppf_th_dwl": "ppp" "iiiii" "s",
    "ppf_th_softmax_mix": "ppppp" "iiiii" "s",
    "ppf_th_softmax_bwd": "pppppppp" "iiiii" "s",
    "ppf_class_attn_fwd": "pppppppp" "iiii" "s",
    "ppf_class_attn_bwd": "ppppppppp" "iiii" "s",
    "ppf_merge3_cast": "pppp" "iii" "s",
    "ppf_sigmoid_bwd": "ppppii" "pz" "s",
    "ppf_im2col_patch_f32": "pp" "iiiii" "s",
    "ppf_layernorm_fwd_f32": "ppppp" "iif" "s",
    "ppf_epilogue_f32": "pp" "i" "pp" "i" "p" "ii" "s",
    "ppf_attn_fwd_f32": "pppp" "i" "iiii" "ii" "s",
    "ppf_th_attn_fwd_f32": "ppppp" "pp" "i" "iiii" "s",
    "ppf_class_attn_fwd_f32": "pppppp" "iiii" "s",
    "ppf_adamw_step_dev": "pppppp" "li" "pp" "ffff" "s",
    "ppf_hyper_set": "ppi" "s",
    "ppf_clip_grad_scale": "pl" "ff" "ppp" "s",
    "ppf_droppath_scales": "pp" "ii" "Lp" "s",
    "ppf_reserved_rows_map": "pp" "iii" "s",
    "ppf_gather_rows": "ppp" "ii" "s",
    "ppf_scatter_rows": "ppp" "iii" "s",
    "ppf_memset_zero": "pz" "s",
    "ppf_image_finish_u8": "pp" "iii" "ppp" "Lp" "s",
    "ppf_scale_by_scalar": "ppp" "l" "s",
}

_CT = {"p": ctypes.c_void_p, "i": ctypes.c_int, "l": ctypes.c_int64, "L": ctypes.c_uint64, "f": ctypes.c_float, "s": ctypes.c_void_p, "z": ctypes.c_size_t}
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} not found: build it with `python -m protopformer_amd.build` "
                               "(there is no fallback path)")
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.ppf_last_error.restype = ctypes.c_char_p
        _lib.ppf_abi_version.restype = ctypes.c_int
        _lib.ppf_gemm_workspace_bytes.restype = ctypes.c_size_t
        _lib.ppf_gemm_workspace_bytes.argtypes = [ctypes.c_int] * 3
        _lib.ppf_layernorm_bwd_blocks.restype = ctypes.c_int
        _lib.ppf_layernorm_bwd_blocks.argtypes = [ctypes.c_int]
        _lib.ppf_sigmoid_bwd_blocks.restype = ctypes.c_int
        _lib.ppf_sigmoid_bwd_blocks.argtypes = [ctypes.c_int]
        _lib.ppf_clip_grad_blocks.restype = ctypes.c_int
        _lib.ppf_clip_grad_blocks.argtypes = []
        for name, spec in SIGS.items():
            fn = getattr(_lib, name)
            fn.restype = ctypes.c_int
            fn.argtypes = [_CT[c] for c in spec]
    return _lib


def _ptr(x):
    if x is None:
        return None
    if isinstance(x, torch.Tensor) or hasattr(x, "data_ptr"):
        return x.data_ptr()
    return x


def stream_ptr():
    return torch.cuda.current_stream().cuda_stream


def call(name, *args):
    """Invoke an entry point on the current torch stream; the stream argument is appended automatically
    when the signature ends in 's'."""
    L = lib()
    spec = SIGS[name]
    if spec.endswith("s") and len(args) == len(spec) - 1:
        args = args + (stream_ptr(),)
    if len(args) != len(spec):
        raise TypeError(f"{name}: expected {len(spec)} arguments, got {len(args)}")
    conv = [(_ptr(a) if c in "ps" else a) for a, c in zip(args, spec)]
    rc = getattr(L, name)(*conv)
    if rc != 0:
        raise RuntimeError(f"{name} failed (rc={rc}): {L.ppf_last_error().decode()}")
