"""Deferred parameter-gradient work (cpc2_hip.h: cpc_gru_backward_deferred, cpc_encoder_backward_deferred,
cpc_transformer_backward_deferred + cpc_side_tail_join): the bookkeeping the autograd Functions of model.py / transformers.py share."""
import os

import torch

from . import _lib
from ._lib import check, stream_ptr

# --------------------------------------------------------------------------- deferred parameter-gradient work
# cpc_gru_backward_deferred / cpc_encoder_backward_deferred leave work that only finishes PARAMETER gradients on a stream of the
# library's, under the kernels the backward pass enqueues next.  Whoever reads those gradients sits behind join_tail(): the end of
# the backward pass (autograd callback), DataParallelContext's all-reduces, FlatAdam.step.
_tail = {}              # device index -> [tensors the side stream still uses, one tuple per deferred backward]


def join_tail(device):
    """Make the current stream of `device` wait for the parameter-gradient work deferred backward calls left on the library's side
    stream (no-op when none is pending)."""
    device = torch.device(device)
    if device.type != "cuda":
        return
    idx = device.index if device.index is not None else torch.cuda.current_device()
    if _tail.pop(idx, None) is not None:
        check(_lib.load().cpc_side_tail_join(stream_ptr(device)), "side_tail_join")


def _keep_for_tail(device, tensors):
    idx = device.index if device.index is not None else torch.cuda.current_device()
    _tail.setdefault(idx, []).append(tensors)
    # one callback per deferred call (join_tail is idempotent): an entry left behind by a backward pass that raised must not keep
    # the next pass from queueing its own
    torch.autograd.Variable._execution_engine.queue_callback(lambda: join_tail(device))


def _tail_tag(name, device):
    """Scratch tag of a deferred backward call: distinct for every call that is pending on the device at the same time (the side
    stream still reads the earlier calls' buffers until the join empties the list)."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    return (name, len(_tail.get(idx, ())))


def _all_in_place(params, grads):
    """Every gradient buffer is the parameter's own piece of FlatAdam's flat gradient buffer (grad_buffers): autograd then adopts
    it as .grad without reading it.  A private buffer would be ADDED to .grad the moment backward() returns."""
    return all(getattr(p, "_cpc_flat", None) is not None and g.data_ptr() == p._cpc_flat[0].data_ptr() + 4 * p._cpc_flat[1]
               for p, g in zip(params, grads))


def _no_hooks(params):
    if os.environ.get("CPC_NO_GRAD_TAIL"):                        # A/B switch
        return False
    return not any(getattr(p, "_backward_hooks", None) or getattr(p, "_post_accumulate_grad_hooks", None) for p in params)


class _TailScope:
    """`with module.deferred_weight_gradients():` around the FORWARD call -- the caller's promise that nothing reads the module's
    parameter gradients before the backward pass has ended: no wrapper whose reducer copies a gradient the moment autograd has
    accumulated it (DistributedDataParallel / DataParallel), no tensor hook on them.  cpcStep opens it for the bare model."""

    def __init__(self, module):
        self.module = module

    def __enter__(self):
        self.prev = self.module._defer_tail
        self.module._defer_tail = True
        return self

    def __exit__(self, *exc):
        self.module._defer_tail = self.prev
        return False




# --------------------------------------------------------------------------- gradients with a fixed home
# Buffers for gradients whose layout is fixed by the caller (criterion.py: the zero halves of train.py:102-103's slices; cpcStep's
# split of the encoder output into context windows and target windows): allocated and zeroed once per (shape, device, slot), the
# producers write their part in place and the consumer hands the whole buffer on.  A buffer is reused only after the backward pass
# that read it has been enqueued on the same stream; the slot carries every number the "stays zero" / "who writes what" invariant
# depends on (window counts), so two users with the same full shape but different splits never share one.
_grad_cache = {}


def _cached_grad_buffer(shape, device, slot):
    key = (tuple(shape), str(device), slot)
    buf = _grad_cache.get(key)
    if buf is None:
        if len(_grad_cache) > 8:
            _grad_cache.clear()
        buf = _grad_cache[key] = torch.zeros(shape, dtype=torch.float32, device=device)
    return buf


def grad_home(t):
    """(full_shape, slot, start) a tensor was marked with by split_windows (criterion.py), if it can be honoured: the gradient with
    respect to `t` should be written into windows start.. of that cached buffer -- the consumer then takes it without a copy."""
    home = getattr(t, "_cpc_grad_home", None)
    if home is None or not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
        return None
    # ONE consumer per home: a second function that takes the same tagged tensor would write the same memory, and autograd would
    # then add the buffer to itself (twice the last gradient, no error).  The first forward that honours the home claims it.
    if getattr(t, "_cpc_grad_home_claimed", False):
        return None
    full_shape, _slot, start = home
    if tuple(t.shape[1:]) != tuple(full_shape[1:]) or start + t.shape[0] > full_shape[0]:
        return None
    t._cpc_grad_home_claimed = True
    return home


def grad_home_view(home, like):
    """The piece of its home buffer a gradient shaped like `like` is written into (None: no home, allocate)."""
    if home is None:
        return torch.empty_like(like)
    full_shape, slot, start = home
    return _cached_grad_buffer(full_shape, like.device, slot)[start:start + like.shape[0]]
