"""ctypes binding of libcpc2_hip.so (the C ABI declared in include/cpc2_hip.h).

No fallback: if the library is missing the import of cpc2_amd fails loudly with build
instructions.  Every call goes through `check()` which raises with cpc_last_error().
"""
import ctypes
import os
import time

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CPC2_HIP_LIB") or os.path.join(_HERE, "libcpc2_hip.so")     # (override: probe builds under tools/)

c_float_p = ctypes.c_void_p      # raw device / host addresses (tensor.data_ptr())
c_ptr = ctypes.c_void_p
c_int = ctypes.c_int
c_long = ctypes.c_long
c_size_t = ctypes.c_size_t
c_float = ctypes.c_float

# name -> (restype, argtypes); must list EVERY symbol declared in include/cpc2_hip.h
SIGNATURES = {
    "cpc_version": (c_int, []),
    "cpc_last_error": (ctypes.c_char_p, []),
    "cpc_async_error_check": (c_int, [c_ptr]),
    "cpc_prof_enable": (c_int, [c_int]),
    "cpc_gemm_set_mode": (c_int, [c_int]),
    "cpc_prof_read": (c_int, [ctypes.c_char_p, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(c_long)]),
    "cpc_gemm_nt": (c_int, [c_ptr, c_long, c_ptr, c_long, c_ptr, c_long, c_ptr, c_int, c_int, c_int, c_ptr]),
    "cpc_gemm_tn_scratch_bytes": (c_size_t, [c_int, c_int, c_long]),
    "cpc_gemm_tn": (c_int, [c_ptr, c_long, c_ptr, c_long, c_ptr, c_long, c_int, c_int, c_long, c_ptr, c_size_t, c_ptr]),
    "cpc_split_planes": (c_int, [c_ptr, c_long, c_long, c_int, c_ptr, c_long, c_int, c_long, c_ptr]),
    "cpc_gemm_nt_planes": (c_int, [c_ptr, c_long, c_int, c_int, c_long, c_int, c_long, c_ptr, c_long, c_ptr, c_long, c_ptr, c_long, c_int, c_int, c_ptr]),
    "cpc_gemm_tn_planes_scratch_bytes": (c_size_t, [c_int, c_int, c_long]),
    "cpc_gemm_tn_planes": (c_int, [c_ptr, c_long, c_int, c_long, c_int, c_int, c_ptr, c_long, c_int, c_long, c_int, c_int, c_ptr, c_long,
                                   c_int, c_int, c_long, c_ptr, c_size_t, c_ptr]),
    "cpc_channelnorm_forward": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, c_float, c_ptr]),
    "cpc_channelnorm_backward": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, c_float, c_ptr]),
    "cpc_encoder_frames": (c_int, [c_int]),
    "cpc_encoder_saved_bytes": (c_size_t, [c_int, c_int, c_int]),
    "cpc_encoder_scratch_bytes": (c_size_t, [c_int, c_int, c_int]),
    "cpc_encoder_forward": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, c_float, c_ptr]),
    "cpc_encoder_backward": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, c_float, c_ptr]),
    "cpc_encoder_backward_deferred": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, c_float, c_ptr]),
    "cpc_encoder_forward2": (c_int, [c_ptr, c_ptr, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, c_float, c_ptr]),
    "cpc_encoder_backward2": (c_int, [c_ptr, c_ptr, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, c_float, c_int, c_ptr]),
    "cpc_coop_launches": (c_long, []),
    "cpc_coop_comm_buffers": (c_int, []),
    "cpc_coop_set_policy": (c_int, [c_int]),
    "cpc_recurrent_backward_calls": (c_long, []),
    "cpc_encoder_saved_layout": (c_int, [c_int, c_int, c_int, c_int, c_ptr]),
    "cpc_gru_saved_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "cpc_gru_scratch_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "cpc_gru_forward": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_int, c_ptr]),
    "cpc_gru_backward": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_int, c_ptr]),
    "cpc_gru_backward_deferred": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_int, c_ptr]),
    "cpc_side_tail_join": (c_int, [c_ptr]),
    "cpc_side_tail_wait": (c_int, [c_ptr]),
    "cpc_lstm_saved_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "cpc_lstm_scratch_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "cpc_lstm_forward": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_int, c_ptr]),
    "cpc_lstm_backward": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_int, c_ptr]),
    "cpc_lstm_backward_deferred": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_int, c_ptr]),
    "cpc_rnn_saved_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "cpc_rnn_scratch_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "cpc_rnn_forward": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_int, c_ptr]),
    "cpc_rnn_backward": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_int, c_ptr]),
    "cpc_transformer_param_count": (c_int, []),
    "cpc_transformer_saved_bytes": (c_size_t, [c_int] * 7),
    "cpc_transformer_scratch_bytes": (c_size_t, [c_int] * 7),
    "cpc_transformer_forward": (c_int, [c_ptr] * 5 + [c_int] * 7 + [c_float, ctypes.c_ulonglong, c_ptr]),
    "cpc_transformer_backward": (c_int, [c_ptr] * 7 + [c_int] * 7 + [c_float, ctypes.c_ulonglong, c_ptr]),
    "cpc_transformer_backward_deferred": (c_int, [c_ptr] * 7 + [c_int] * 7 + [c_float, ctypes.c_ulonglong, c_ptr]),
    "cpc_mt_create": (c_ptr, [ctypes.c_uint32]),
    "cpc_mt_destroy": (None, [c_ptr]),
    "cpc_mt_seed": (c_int, [c_ptr, ctypes.c_uint32]),
    "cpc_mt_get_state": (c_int, [c_ptr, c_ptr, ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
    "cpc_mt_set_state": (c_int, [c_ptr, c_ptr, c_int, c_int]),
    "cpc_negidx_sample_host": (c_int, [c_ptr, c_int, c_int, c_int, c_int, c_int, c_ptr, c_ptr, c_ptr]),
    "cpc_negidx_sample_host_async": (c_int, [c_ptr, c_int, c_int, c_int, c_int, c_int, c_ptr]),
    "cpc_negidx_wait": (c_int, [c_ptr]),
    "cpc_mt_draw_host": (c_int, [c_ptr, c_ptr, c_size_t]),
    "cpc_mt_draw_host_async": (c_int, [c_ptr, c_ptr, c_size_t]),
    "cpc_mt_draw_device_async": (c_int, [c_ptr, c_ptr, c_ptr, c_size_t, c_int, c_ptr]),
    "cpc_mt_draw_expand_device_async": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_int, c_ptr]),
    "cpc_mt_redraw_expand_device_async": (c_int, [c_ptr, c_ptr, c_int, c_int, c_size_t, c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_int, c_ptr]),
    "cpc_negidx_wait_on": (c_int, [c_ptr, c_ptr]),
    "cpc_negidx_stream": (c_int, [c_ptr, c_ptr]),
    "cpc_stream_create_apart": (c_int, [c_ptr, c_int, c_ptr]),
    "cpc_streams_overlap": (c_int, [c_ptr, c_ptr]),
    "cpc_stream_apart_failures": (c_long, []),
    "cpc_stream_spin": (c_int, [c_ptr, c_long]),
    "cpc_side_stream": (c_int, [c_ptr, c_ptr]),
    "cpc_negidx_expand": (c_int, [c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_ptr]),
    "cpc_infonce_saved_bytes": (c_size_t, [c_int] * 6),
    "cpc_infonce_scratch_bytes": (c_size_t, [c_int] * 6),
    "cpc_infonce_logits_offset": (c_size_t, [c_int] * 6),
    "cpc_infonce_perm_offset": (c_size_t, [c_int] * 6),
    "cpc_infonce_forward": (c_int, [c_ptr] * 9 + [c_int] * 6 + [c_ptr]),
    "cpc_infonce_backward": (c_int, [c_ptr] * 11 + [c_int] * 6 + [c_ptr]),
    "cpc_infonce_backward_deferred": (c_int, [c_ptr] * 11 + [c_int] * 6 + [c_ptr]),
    "cpc_infonce_join": (c_int, [c_ptr]),
    "cpc_infonce_forward_cw": (c_int, [c_ptr] * 9 + [c_int] * 6 + [c_ptr]),
    "cpc_infonce_backward_cw": (c_int, [c_ptr] * 11 + [c_int] * 7 + [c_ptr]),
    "cpc_infonce_forward_pred": (c_int, [c_ptr] * 8 + [c_int] * 5 + [c_ptr]),
    "cpc_infonce_backward_pred": (c_int, [c_ptr] * 9 + [c_int] * 5 + [c_ptr]),
    "cpc_flac_info": (c_int, [ctypes.c_char_p, ctypes.POINTER(c_int), ctypes.POINTER(c_int), ctypes.POINTER(c_int),
                              ctypes.POINTER(c_long)]),
    "cpc_flac_decode_f32": (c_int, [ctypes.c_char_p, c_ptr, c_long, ctypes.POINTER(c_int)]),
    "cpc_window_gather": (c_int, [c_ptr, c_long, c_ptr, c_ptr, c_int, c_int, c_ptr]),
    "cpc_adam_step": (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_long, c_int, c_float, c_float, c_float, c_float, c_float, c_ptr]),
}

_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: the HIP extension is not built and cpc2_amd has no fallback. "
                "Run `python -c 'import __graft_entry__ as g; g.build()'` (or `python cpc2_amd/build.py`).")
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


# Seconds the HOST thread has spent blocked on something other than enqueueing work, by cause (bench.py's `host` record reads and
# resets it): the sampler's worker thread (cpc_negidx_wait), the sampler's buffer-release events, a collective's work.wait().
HOST_WAITS = {}


class host_wait:
    """`with host_wait("cause"):` around a call that may block the host."""
    __slots__ = ("cause", "t0")

    def __init__(self, cause):
        self.cause = cause

    def __enter__(self):
        self.t0 = time.perf_counter()

    def __exit__(self, *exc):
        HOST_WAITS[self.cause] = HOST_WAITS.get(self.cause, 0.0) + time.perf_counter() - self.t0
        return False


def check(status, what=""):
    if status != 0:
        msg = load().cpc_last_error().decode(errors="replace")
        kind = ValueError if status == -1 else RuntimeError
        raise kind(f"libcpc2_hip {what} failed ({status}): {msg}")


def stream_ptr(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("cpc2_amd runs only on a GPU (HIP) device: got a tensor on "
                               f"'{t.device}'. There is no CPU fallback.")


def f32c(t):
    """contiguous fp32 view/copy of t (no copy when already so)."""
    if t.dtype != torch.float32:
        raise TypeError(f"cpc2_amd kernels are fp32 only (got {t.dtype})")
    return t if t.is_contiguous() else t.contiguous()


def ptr(t):
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def ptr_array(tensors):
    return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() if t is not None else None for t in tensors])


# grow-only scratch arena per device; every kernel is enqueued on the current stream in program
# order, so one buffer serves all calls
_scratch = {}


def scratch(nbytes, device, tag=None):
    # one buffer per (device, stream): ops enqueued on different streams may run at the same time.  `tag`: a buffer of its
    # own for an op whose kernels outlive the call on a stream of the library's (the deferred criterion backward)
    key = (device.type, device.index, torch.cuda.current_stream(device).cuda_stream if device.type == "cuda" else 0, tag)
    buf = _scratch.get(key)
    if buf is None or buf.numel() < nbytes:
        if buf is not None:
            del _scratch[key]
        buf = torch.empty(int(nbytes) + 4096, dtype=torch.uint8, device=device)
        _scratch[key] = buf
    return buf


def grad_buffers(params):
    """Output buffers for parameter gradients.  A parameter re-homed by FlatAdam(direct_grads=True) carries
    `_cpc_flat` = (flat_grad, offset): its gradient is then written straight into the flat gradient buffer (a
    FRESH view each time, so autograd adopts it as .grad instead of adding it) -- but only while .grad is None;
    otherwise (accumulation across backward passes) a private buffer is returned and autograd adds it."""
    out = []
    for p in params:
        if p is None:                      # an absent optional parameter (e.g. Krelpos with abspos=True)
            out.append(None)
            continue
        home = getattr(p, "_cpc_flat", None)
        if home is not None and p.grad is None:
            flat, off = home
            out.append(flat[off:off + p.numel()].view(p.shape))
        else:
            out.append(torch.empty_like(p))
    return out
