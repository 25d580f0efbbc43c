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
    first = idx not in _tail
    _tail.setdefault(idx, []).append(tensors)
    if first:
        torch.autograd.Variable._execution_engine.queue_callback(lambda: join_tail(device))


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


