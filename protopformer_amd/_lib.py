"""ctypes binding of the C-ABI HIP library (include/ppf_hip.h).

There is NO CPU or eager fallback: if the shared library is missing or a call fails, this raises."""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PPF_LIB_PATH") or os.path.join(_HERE, "lib", "libppf_hip.so")      # PPF_LIB_PATH: an alternative build (same-box A/B of two builds)

# signature spec per entry point: p = device/host pointer, i = int32, l = int64, L = uint64, f = float, z = size_t, s = hipStream_t
SIGS = {
    "ppf_gemm_bf16": "pppiiiiiiiiippipipppipf" "pz" "s",
    "ppf_device_info": "pppi",
    "ppf_rowgemm_bf16": "pp" "iiiiii" "pp" "s",
    "ppf_rowgemm_resid_ln": "pp" "iiiiii" "p" "pp" "pi" "pp" "pp" "ppp" "f" "s",
    "ppf_rowgemm_lnbwd": "pp" "iiiiii" "pppp" "ppp" "pi" "pp" "pz" "s",
    "ppf_rowgemm_colsum": "p" "iii" "ppp" "s",
    "ppf_layernorm_bwd_f32": "pppppppp" "ii" "f" "s",
    "ppf_ew_bwd_f32": "i" "ppp" "p" "i" "ii" "s",
    "ppf_colsum_f32": "pp" "ii" "s",
    "ppf_attn_bwd_f32": "ppppp" "iiiiii" "s",
    "ppf_th_attn_bwd_f32": "ppppppp" "pppp" "iiii" "s",
    "ppf_class_attn_bwd_f32": "pppppppp" "iiii" "s",
    "ppf_transpose_bf16_batched": "ppp" "ii" "s",
    "ppf_gemm_probe": "i",
    "ppf_gemm_test_force_g224": "i",
    "ppf_gemm_probe_read": "pppp",
    "ppf_path_probe": "i",
    "ppf_path_probe_read": "ipppp",
    "ppf_layernorm_fwd": "ppppppp" "iif" "s",
    "ppf_layernorm_bwd": "pppppp" "pppp" "pp" "i" "pppp" "ii" "pz" "s",
    "ppf_layernorm_bwd_reduce": "p" "ii" "pppp" "s",
    "ppf_cast_f32_bf16": "ppls",
    "ppf_cast_bf16_f32": "ppls",
    "ppf_im2col_patch": "ppiiiiis",
    "ppf_assemble_tokens": "ppppiiiis",
    "ppf_assemble_tokens_bwd": "ppppiiiis",
    "ppf_adamw_step": "pppppp" "li" "ppp" "fff" "i" "ff" "s",
    "ppf_attn_fwd": "ppppp" "iiiii" "i" "s",
    "ppf_attn_fwd_hm": "pppppp" "i" "iiiii" "i" "s",
    "ppf_attn_headmean": "ppppp" "i" "iiiii" "i" "s",
    "ppf_attn_bwd": "pppppppp" "iiiii" "i" "s",
    "ppf_rollout": "pl" "iiii" "p" "iiii" "f" "i" "pppp" "s",
    "ppf_rollout_threshold": "p" "iiii" "p" "s",
    "ppf_proto_fwd": "pliip" "iiii" "f" "pppp" "pz" "s",
    "ppf_proto_bwd": "pliip" "iiii" "f" "pi" "pppp" "l" "p" "pz" "s",
    "ppf_proto_bwd_rows": "pliip" "iiii" "f" "pi" "pp" "i" "ppp" "l" "p" "pz" "s",
    "ppf_proto_bwd_single": "plip" "iiii" "f" "ppp" "l" "p" "pz" "s",
    "ppf_ppc_loss": "ppp" "iiiii" "ff" "pppp" "s",
    "ppf_ppc_loss_bwd": "pppppp" "iiii" "s",
    "ppf_cross_entropy": "ppppp" "ii" "s",
    "ppf_sgemm": "ppp" "iii" "llll" "i" "ff" "pl" "s",
    "ppf_sgemm_pair": "ppp" "ii" "llll" "i" "f" "ppp" "ii" "llll" "i" "f" "p" "i" "ff" "i" "pl" "s",
    "ppf_axpby": "ppp" "ff" "l" "s",
    "ppf_axpbypcz": "pppp" "fff" "l" "s",
    "ppf_topk_sorted": "piiips",
    "ppf_gemm_bf16_batched": "ppp" "iiiiii" "iii" "f" "ii" "llllll" "i" "s",
    "ppf_th_scores": "pppp" "iiiii" "s",
    "ppf_th_dwl": "ppp" "iiiii" "s",
    "ppf_th_softmax_mix": "ppppp" "iiiii" "s",
    "ppf_th_softmax_bwd": "pppppppp" "iiiii" "s",
    "ppf_th_fwd": "pppppppppp" "iiiiii" "s",
    "ppf_th_bwd": "ppppppppp" "iiiii" "s",
    "ppf_th_param_reduce": "p" "iii" "pppp" "s",
    "ppf_th_grads": "ppppp" "iiiii" "s",
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
    "ppf_adamw_step_guarded": "pppppp" "li" "pp" "ffff" "pp" "s",
    "ppf_hyper_set": "ppi" "s",
    "ppf_clip_grad_scale": "pl" "ff" "ppp" "s",
    "ppf_droppath_scales": "pp" "ii" "Lp" "s",
    "ppf_reserved_rows_map": "pp" "iii" "s",
    "ppf_gather_rows": "ppp" "ii" "s",
    "ppf_scatter_rows": "ppp" "iii" "s",
    "ppf_memset_zero": "pz" "s",
    "ppf_copy_2d": "pl" "pl" "ll" "s",
    "ppf_image_finish_u8": "pp" "iii" "ppp" "Lp" "s",
    "ppf_scale_by_scalar": "ppp" "l" "s",
    "ppf_stream_wait_stream": "pp",
    "ppf_stream_arm": "pi",
    "ppf_stream_wait_mark": "pl",
}

EXPECTED_ABI = 10              # == PPF_ABI_VERSION of include/ppf_hip.h this table was written against (tests/test_abi_cpu.py)

_CT = {"p": ctypes.c_void_p, "i": ctypes.c_int, "l": ctypes.c_int64, "L": ctypes.c_uint64, "f": ctypes.c_float, "s": ctypes.c_void_p, "z": ctypes.c_size_t}
_lib = None
_FAST = {}                     # name -> (bound function, pointer-argument positions, has trailing stream, arity)
_stream_override = []          # raw hipStream_t stack: set by the weight-gradient lane while it enqueues on its own stream


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} not found: build it with `python -m protopformer_amd.build` "
                               "(there is no fallback path)")
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.ppf_last_error.restype = ctypes.c_char_p
        _lib.ppf_abi_version.restype = ctypes.c_int
        got = _lib.ppf_abi_version()
        if got != EXPECTED_ABI:
            _lib = None
            raise RuntimeError(f"{LIB_PATH} was built for ABI version {got}, this binding expects {EXPECTED_ABI}: rebuild it with "
                               "`python -m protopformer_amd.build --force` (a stale library would receive shifted arguments)")
        _lib.ppf_gemm_workspace_bytes.restype = ctypes.c_size_t
        _lib.ppf_gemm_workspace_bytes.argtypes = [ctypes.c_int] * 3
        _lib.ppf_sgemm_pair_workspace.restype = ctypes.c_int64
        _lib.ppf_sgemm_pair_workspace.argtypes = [ctypes.c_int] * 5
        _lib.ppf_proto_bwd_single_workspace.restype = ctypes.c_size_t
        _lib.ppf_proto_bwd_single_workspace.argtypes = [ctypes.c_int] * 3
        _lib.ppf_proto_bwd_workspace.restype = ctypes.c_size_t
        _lib.ppf_proto_fwd_workspace.restype = ctypes.c_size_t
        _lib.ppf_proto_fwd_workspace.argtypes = [ctypes.c_int] * 2
        _lib.ppf_proto_bwd_workspace.argtypes = [ctypes.c_int] * 6
        _lib.ppf_layernorm_bwd_blocks.restype = ctypes.c_int
        _lib.ppf_layernorm_bwd_blocks.argtypes = [ctypes.c_int]
        _lib.ppf_sigmoid_bwd_blocks.restype = ctypes.c_int
        _lib.ppf_sigmoid_bwd_blocks.argtypes = [ctypes.c_int]
        _lib.ppf_clip_grad_blocks.restype = ctypes.c_int
        _lib.ppf_clip_grad_blocks.argtypes = []
        _lib.ppf_rowgemm_supported.restype = ctypes.c_int
        _lib.ppf_rowgemm_supported.argtypes = [ctypes.c_int] * 3
        _lib.ppf_attn_fwd_hm_supported.restype = ctypes.c_int
        _lib.ppf_attn_fwd_hm_supported.argtypes = [ctypes.c_int] * 3
        _lib.ppf_th_fused_supported.restype = ctypes.c_int
        _lib.ppf_th_fused_supported.argtypes = [ctypes.c_int] * 3
        _lib.ppf_th_grads_supported.restype = ctypes.c_int
        _lib.ppf_th_grads_supported.argtypes = [ctypes.c_int] * 3
        _lib.ppf_th_bwd_partial_floats.restype = ctypes.c_size_t
        _lib.ppf_th_bwd_partial_floats.argtypes = [ctypes.c_int] * 3
        _lib.ppf_stream_mark.restype = ctypes.c_int64
        _lib.ppf_stream_mark.argtypes = [ctypes.c_void_p]
        for name, spec in SIGS.items():
            fn = getattr(_lib, name)
            fn.restype = ctypes.c_int
            fn.argtypes = [_CT[c] for c in spec]
            _FAST[name] = (fn, tuple(i for i, c in enumerate(spec) if c == "p"), spec.endswith("s"), len(spec))
    return _lib


def _ptr(x):
    if x is None:
        return None
    if isinstance(x, torch.Tensor) or hasattr(x, "data_ptr"):
        return x.data_ptr()
    return x


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)     # C entry: ~0.2 us (torch.cuda.current_stream() costs ~8 us)
_dev_index = None


def stream_ptr():
    """Raw hipStream_t the next launch goes to: the weight-gradient lane's while it is enqueueing, else torch's current stream
    on this process's device (one process drives one GPU; the device index is read once)."""
    global _dev_index
    if _stream_override:
        return _stream_override[-1]
    if _raw_stream is None:
        return torch.cuda.current_stream().cuda_stream
    if _dev_index is None:
        _dev_index = torch.cuda.current_device()
    return _raw_stream(_dev_index)


def push_stream(raw):
    _stream_override.append(raw)


def pop_stream():
    _stream_override.pop()


def call(name, *args):
    """Invoke an entry point on the current stream (stream_ptr()); the stream argument is appended automatically
    when the signature ends in 's'."""
    if _lib is None:
        lib()
    fn, ptr_pos, has_stream, arity = _FAST[name]
    if has_stream and len(args) == arity - 1:
        args = args + (stream_ptr(),)
    if len(args) != arity:
        raise TypeError(f"{name}: expected {arity} arguments, got {len(args)}")
    a = list(args)
    for i in ptr_pos:
        x = a[i]
        if x is not None and not isinstance(x, int):
            a[i] = x.data_ptr()
    rc = fn(*a)
    if rc != 0:
        raise RuntimeError(f"{name} failed (rc={rc}): {_lib.ppf_last_error().decode()}")
    if _rec is not None and not _rec.suspended:
        _rec.add_call(name, fn, a, args)


# ------------------------------------------------------------------------------------------------ recorded command lists
# One train step is ~300-800 launches; enqueueing them from Python costs ~20 us each (allocation, shape logic, autograd), which makes
# the small configurations host-bound (HIP graphs do not help on ROCm 7.2: hipGraphLaunch re-enqueues every node from the host at
# the same cost, profiles/r2_graph_timeline.txt).  A Recorder keeps, for ONE eagerly executed step, every library call with its
# converted arguments (raw pointers, sizes, streams) and keeps the tensors alive, so their addresses stay valid; replay() then walks
# the list: one pre-bound ctypes call (~2-3 us) per launch, no Python orchestration, no allocation, no autograd.  What cannot be
# frozen runs as a LIVE Python callable in its place in the list (optimizer scalars, the RCCL collectives of the gradient exchange).
class Recorder:
    CALL, MARK, WAIT, LIVE = 0, 1, 2, 3

    def __init__(self):
        self.cmds, self.keep, self.tickets, self.nslots, self.suspended = [], [], {}, 0, 0
        self.main_stream = stream_ptr() if torch.cuda.is_available() else None      # torch's current stream while the step was recorded

    def add_call(self, name, fn, a, args):
        if name == "ppf_stream_wait_mark":
            slot = self.tickets.get(a[1])
            if slot is not None:              # (a ticket taken before the recording started guards work that is long complete)
                self.cmds.append((Recorder.WAIT, fn, a[0], slot))
            return
        self.cmds.append((Recorder.CALL, fn, tuple(a), name))
        self.keep.append(args)                # the tensors behind the recorded pointers stay allocated


_rec = None


def start_recording():
    global _rec
    if _rec is not None:
        raise RuntimeError("a recording is already in progress")
    _rec = Recorder()
    return _rec


def stop_recording():
    global _rec
    r, _rec = _rec, None
    return r


def stream_mark(raw):
    """Ticket for 'everything enqueued on `raw` so far' (ppf_stream_mark); recorded symbolically: a replay takes a fresh ticket."""
    t = lib().ppf_stream_mark(raw)
    if t < 0:
        raise RuntimeError(_lib.ppf_last_error().decode())
    if _rec is not None and not _rec.suspended:
        _rec.tickets[t] = _rec.nslots
        _rec.cmds.append((Recorder.MARK, raw, _rec.nslots))
        _rec.nslots += 1
    return t


def run_live(fn):
    """Run fn() now; inside a recording it also becomes a LIVE entry of the command list (executed again by every replay, at this
    position): host-side state that changes from step to step (optimizer scalars) and calls outside the library (collectives)."""
    if _rec is None or _rec.suspended:
        return fn()
    _rec.suspended += 1
    try:
        out = fn()
    finally:
        _rec.suspended -= 1
    _rec.cmds.append((Recorder.LIVE, fn))
    return out


def _arm_plan(cmds):
    """{index of a recorded library call: its launch stream} for the calls that are directly followed by ppf_stream_wait_stream(dst, that
    stream): their kernels are launched with a completion event attached (ppf_stream_arm), so the wait does not have to put an event-record
    packet into the producer's queue.  Same-box A/B (profiles/r6_armed_events.txt): deit_small +0.3 %, deit_tiny +0.5 %, cait_xxs24 +0.4 %; main-queue idle
    between kernels 0.87 -> 0.67 ms per deit_small step."""
    plan = {}
    for i, c in enumerate(cmds[:-1]):
        n = cmds[i + 1]
        if c[0] != 0 or n[0] != 0 or n[3] != "ppf_stream_wait_stream" or c[3].startswith("ppf_stream_"):
            continue
        if not _FAST[c[3]][2]:
            continue
        stream = c[2][-1]
        if stream is not None and stream == n[2][1]:
            plan[i] = stream
    return plan


def replay(rec):
    """Enqueue a recorded step again: same kernels, same arguments, same streams and cross-stream dependencies.  The list holds the raw
    streams of the recording; the caller's input copies and the live collectives go to torch's CURRENT stream, so that must be the one the
    step was recorded under."""
    if rec.main_stream is not None and not _stream_override and stream_ptr() != rec.main_stream:
        raise RuntimeError("replay(): the current torch stream differs from the stream the step was recorded on (input copies and "
                           "collectives would not be ordered with the recorded launches); replay under the recording's stream")
    slots = [0] * rec.nslots
    mark = _lib.ppf_stream_mark
    arm_at = getattr(rec, "arm_at", None)
    if arm_at is None:
        arm_at = rec.arm_at = _arm_plan(rec.cmds)
    arm = _FAST["ppf_stream_arm"][0]
    for i, c in enumerate(rec.cmds):
        k = c[0]
        if k == 0:
            if i in arm_at:
                arm(arm_at[i], 1)           # this call's launches carry their completion event: the wait that follows records nothing
            rc = c[1](*c[2])
            if rc != 0:
                raise RuntimeError(f"{c[3]} failed in replay (rc={rc}): {_lib.ppf_last_error().decode()}")
        elif k == 1:
            t = mark(c[1])
            if t < 0:
                raise RuntimeError(f"ppf_stream_mark failed in replay: {_lib.ppf_last_error().decode()}")
            slots[c[2]] = t
        elif k == 2:
            rc = c[1](c[2], slots[c[3]])
            if rc != 0:
                raise RuntimeError(f"ppf_stream_wait_mark failed in replay (rc={rc}): {_lib.ppf_last_error().decode()}")
        else:
            c[1]()
