"""Drop-in counterpart of the reference's light transformer used as autoregressive network.

Mirrors /root/reference/cpc/transformers.py: ScaledDotProductAttention (:10-70), MultiHeadAttention
(:73-104), FFNetwork (:107-116), TransformerLayer (:119-134), buildTransformerAR (:176-187) -- same
class names, constructor signatures, parameter/buffer names (so state dicts are interchangeable:
`multihead.{Wo,Wk,Wq,Wv}.weight`, `multihead.Att.{Krelpos,z,mask}`, `ln_multihead.*`,
`ffnetwork.lin{1,2}.*`, `last_linear.*`, `ln_ffnetwork.*`) and the same construction order (identical
default initialisation under a given torch seed).  The sub-modules are parameter containers: the whole
layer runs as ONE fused call into libcpc2_hip.so (cpc_transformer_forward / _backward).

abspos=True (StaticPositionEmbedding :161-173 in front, no relative-position bias) is supported; sequence lengths
that are not a multiple of sizeSeq are zero-padded to the next multiple like the reference does (transformers.py:41-48
pads Q, K and V, which have no bias: the same as padding the layer's input) and the outputs of the padding dropped.
"""
import math

import torch
import torch.nn as nn

from . import _lib
from ._lib import check, f32c, grad_buffers, ptr, ptr_array, require_gpu, scratch, stream_ptr
from ._tail import _TailScope, _all_in_place, _keep_for_tail, _no_hooks, _tail_tag, grad_home, grad_home_view


class ScaledDotProductAttention(nn.Module):
    def __init__(self, sizeSeq, dk, dropout, relpos=False):
        super(ScaledDotProductAttention, self).__init__()
        self.drop = nn.Dropout(dropout)
        self.softmax = nn.Softmax(dim=2)
        self.relpos = relpos
        self.sizeSeq = sizeSeq
        if relpos:
            self.Krelpos = nn.Parameter(torch.Tensor(dk, sizeSeq))
            self.initmat_(self.Krelpos)
            self.register_buffer('z', torch.zeros(1, sizeSeq, 1))
        mask = torch.tril(torch.ones(sizeSeq, sizeSeq), diagonal=0)
        mask = 1 - mask
        mask[mask == 1] = -float('inf')
        self.register_buffer('mask', mask.unsqueeze(0))

    def initmat_(self, mat, dim=0):
        stdv = 1. / math.sqrt(mat.size(dim))
        mat.data.uniform_(-stdv, stdv)


class MultiHeadAttention(nn.Module):
    def __init__(self, sizeSeq, dropout, dmodel, nheads, abspos):
        super(MultiHeadAttention, self).__init__()
        self.Wo = nn.Linear(dmodel, dmodel, bias=False)
        self.Wk = nn.Linear(dmodel, dmodel, bias=False)
        self.Wq = nn.Linear(dmodel, dmodel, bias=False)
        self.Wv = nn.Linear(dmodel, dmodel, bias=False)
        self.nheads = nheads
        self.dk = dmodel // nheads
        self.Att = ScaledDotProductAttention(sizeSeq, self.dk, dropout, not abspos)


class FFNetwork(nn.Module):
    def __init__(self, din, dout, dff, dropout):
        super(FFNetwork, self).__init__()
        self.lin1 = nn.Linear(din, dff, bias=True)
        self.lin2 = nn.Linear(dff, dout, bias=True)
        self.relu = nn.ReLU()
        self.drop = nn.Dropout(dropout)


class _TransformerFn(torch.autograd.Function):
    """`layers` stacked TransformerLayers in one call; params = 15 tensors per layer in the C-ABI order.
    n_classifiers > 1: the last layer is a MultiClassifierTransformerHead and the output is [n, s, k, d_out]."""

    @staticmethod
    def forward(ctx, x, size_seq, n_layers, n_classifiers, dropout_p, seed, defer_tail, *params):
        require_gpu(x, *[p for p in params if p is not None])
        lib = _lib.load()
        ctx.dx_home = grad_home(x)             # (cpcStep's split_windows)
        x = f32c(x)
        ctx.param_refs = params
        ctx.defer_tail = bool(defer_tail) and n_classifiers == 1
        params = tuple(f32c(p) if p is not None else None for p in params)
        n, s, d_model = x.shape
        per = lib.cpc_transformer_param_count()
        d_out = params[(n_layers - 1) * per + 11].shape[0]            # last layer's last_linear.weight [d_out, d]
        nsaved = lib.cpc_transformer_saved_bytes(n, s, d_model, d_out, size_seq, n_layers, n_classifiers)
        nscratch = lib.cpc_transformer_scratch_bytes(n, s, d_model, d_out, size_seq, n_layers, n_classifiers)
        if nsaved == 0:
            check(-1, "transformer shape query")
        shape = (n, s, d_out) if n_classifiers == 1 else (n, s, n_classifiers, d_out)
        out = torch.empty(shape, dtype=torch.float32, device=x.device)
        saved = torch.empty(nsaved, dtype=torch.uint8, device=x.device)
        sc = scratch(nscratch, x.device)
        check(lib.cpc_transformer_forward(ptr(x), ptr_array(params), ptr(out), ptr(saved), ptr(sc), n, s, d_model, d_out,
                                          size_seq, n_layers, n_classifiers, dropout_p, seed, stream_ptr(x.device)),
              "transformer_forward")
        ctx.save_for_backward(x, saved, *params)
        ctx.cfg = (n, s, d_model, d_out, size_seq, n_layers, n_classifiers, dropout_p, seed)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        x, saved, *params = ctx.saved_tensors
        n, s, d_model, d_out, size_seq, n_layers, n_classifiers, dropout_p, seed = ctx.cfg
        dout = f32c(dout)
        dx = grad_home_view(ctx.dx_home, x) if ctx.needs_input_grad[0] else None
        grads = grad_buffers(ctx.param_refs)
        nscratch = lib.cpc_transformer_scratch_bytes(n, s, d_model, d_out, size_seq, n_layers, n_classifiers)
        per = lib.cpc_transformer_param_count()
        l0 = [(p, g) for p, g in zip(ctx.param_refs[:per], grads[:per]) if p is not None]
        # the deferred form (cpc2_hip.h): inside the caller's scope (TransformerLayer.deferred_weight_gradients) and with every
        # gradient of layer 0 written in place into the flat gradient buffer (model.py, _GruFn.backward)
        if ctx.defer_tail and _all_in_place([p for p, _g in l0], [g for _p, g in l0]):
            # a buffer of its own, and one PER PENDING CALL: the side stream reads it after this call has returned, and a second
            # TransformerLayer of the same context network (nLevelsGRU >= 2) runs its backward before the join
            sc = scratch(nscratch, x.device, tag=_tail_tag("tr_tail", x.device))
            check(lib.cpc_transformer_backward_deferred(ptr(x), ptr_array(params), ptr(dout), ptr(saved), ptr(sc), ptr(dx),
                                                        ptr_array(grads), n, s, d_model, d_out, size_seq, n_layers, n_classifiers,
                                                        dropout_p, seed, stream_ptr(x.device)), "transformer_backward_deferred")
            _keep_for_tail(x.device, (x, saved, params, dout, sc))
        else:
            sc = scratch(nscratch, x.device)
            check(lib.cpc_transformer_backward(ptr(x), ptr_array(params), ptr(dout), ptr(saved), ptr(sc), ptr(dx), ptr_array(grads),
                                               n, s, d_model, d_out, size_seq, n_layers, n_classifiers, dropout_p, seed,
                                               stream_ptr(x.device)),
                  "transformer_backward")
        return (dx, None, None, None, None, None, None) + tuple(grads)


class TransformerLayer(nn.Module):
    def __init__(self, sizeSeq=32, dmodel=512, dout=512, dff=2048, dropout=0.1, nheads=8, abspos=False):
        super(TransformerLayer, self).__init__()
        if nheads != 8 or dff != 2048:
            raise NotImplementedError("the MI355X transformer kernels are built for nheads=8, dff=2048 (the reference's values)")
        self.multihead = MultiHeadAttention(sizeSeq, dropout, dmodel, nheads, abspos)
        self.ln_multihead = nn.LayerNorm(dmodel)
        self.ffnetwork = FFNetwork(dmodel, dmodel, dff, dropout)
        self.last_linear = nn.Linear(dmodel, dout)
        self.ln_ffnetwork = nn.LayerNorm(dout)
        self.sizeSeq = sizeSeq
        self.dropout_p = float(dropout)
        self._calls = 0
        self._defer_tail = False       # set by deferred_weight_gradients() for the duration of the caller's scope

    def deferred_weight_gradients(self):
        """Context manager around the FORWARD call (model.py, _TailScope): the backward of a forward pass made inside may leave this
        layer's parameter-gradient work on the library's side stream until the end of the backward pass."""
        return _TailScope(self)

    def _param_list(self):
        m = self.multihead
        return [m.Wq.weight, m.Wk.weight, m.Wv.weight, m.Wo.weight, getattr(m.Att, "Krelpos", None),
                self.ln_multihead.weight, self.ln_multihead.bias,
                self.ffnetwork.lin1.weight, self.ffnetwork.lin1.bias,
                self.ffnetwork.lin2.weight, self.ffnetwork.lin2.bias,
                self.last_linear.weight, self.last_linear.bias,
                self.ln_ffnetwork.weight, self.ln_ffnetwork.bias]

    def forward(self, x):
        p = self.dropout_p if self.training else 0.0
        # a fresh dropout stream per call, derived from torch's CPU generator (so torch.manual_seed governs it)
        seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if p > 0.0 else 0
        x, s = _pad_to_blocks(x, self.sizeSeq)
        params = self._param_list()
        defer = self._defer_tail and _no_hooks([q for q in params if q is not None])
        return _TransformerFn.apply(x, self.sizeSeq, 1, 1, p, seed, defer, *params)[:, :s]


def _pad_to_blocks(x, size_seq):
    """transformers.py:38-50: inputs are attended in blocks of sizeSeq frames; a ragged tail is zero-padded."""
    s = x.size(1)
    r = s % size_seq
    if r > 0:
        x = torch.nn.functional.pad(x, (0, 0, 0, size_seq - r))
    return x, s


class StaticPositionEmbedding(nn.Module):
    """transformers.py:161-173 (abspos=True): fixed sinusoidal position table added to the input (a torch add:
    this variant is not on the measured path)."""

    def __init__(self, seqlen, dmodel):
        super(StaticPositionEmbedding, self).__init__()
        pos = torch.arange(0., seqlen).unsqueeze(1).repeat(1, dmodel)
        dim = torch.arange(0., dmodel).unsqueeze(0).repeat(seqlen, 1)
        div = torch.exp(- math.log(10000) * (2 * (dim // 2) / dmodel))
        pos *= div
        pos[:, 0::2] = torch.sin(pos[:, 0::2])
        pos[:, 1::2] = torch.cos(pos[:, 1::2])
        self.register_buffer('pe', pos.unsqueeze(0))

    def forward(self, x):
        return x + self.pe[:, :x.size(1), :]


class MultiClassifierTransformerHead(nn.Module):
    """transformers.py:137-158: one attention block whose feed-forward net emits `nclassifiers` residual branches;
    forward(x [B, S, dmodel]) -> [B, S, nclassifiers, dout].  One fused library call, like TransformerLayer."""

    def __init__(self, nclassifiers, sizeSeq=32, dmodel=512, dout=512, dff=2048, dropout=0.1, nheads=8, abspos=False):
        super(MultiClassifierTransformerHead, self).__init__()
        if nheads != 8 or dff != 2048:
            raise NotImplementedError("the MI355X transformer kernels are built for nheads=8, dff=2048 (the reference's values)")
        self.multihead = MultiHeadAttention(sizeSeq, dropout, dmodel, nheads, abspos)
        self.ln_multihead = nn.LayerNorm(dmodel)
        self.ffnetwork = FFNetwork(dmodel, dmodel * nclassifiers, dff, dropout)
        self.last_linear = nn.Linear(dmodel, dout)
        self.ln_ffnetwork = nn.LayerNorm(dout)
        self.nclassifiers = nclassifiers
        self.dout = dmodel
        self.sizeSeq = sizeSeq
        self.dropout_p = float(dropout)

    _param_list = TransformerLayer._param_list

    def forward(self, x):
        p = self.dropout_p if self.training else 0.0
        seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if p > 0.0 else 0
        x, s = _pad_to_blocks(x, self.sizeSeq)
        return _TransformerFn.apply(x, self.sizeSeq, 1, self.nclassifiers, p, seed, False, *self._param_list())[:, :s]


def buildTransformerAR(dimEncoded, dimAR, nLayers, sizeSeq, abspos):
    """transformers.py:176-187."""
    layerSequence = [StaticPositionEmbedding(sizeSeq, dimAR)] if abspos else []
    layerSequence += [TransformerLayer(sizeSeq=sizeSeq, dmodel=dimAR, dout=dimEncoded, abspos=abspos)
                      for _ in range(nLayers)]
    return nn.Sequential(*layerSequence)


def buildMultHeadTransformerAR(dimEncoded, dimAR, nLayers, sizeSeq, abspos, nHeads):
    """transformers.py:190-212: nLayers - 1 TransformerLayers, then the multi-classifier head."""
    layerSequence = [StaticPositionEmbedding(sizeSeq, dimAR)] if abspos else []
    layerSequence += [TransformerLayer(sizeSeq=sizeSeq, dmodel=dimAR, dout=dimEncoded, abspos=abspos)
                     for _ in range(nLayers - 1)]
    layerSequence += [MultiClassifierTransformerHead(nHeads, dmodel=dimAR, dout=dimEncoded, sizeSeq=sizeSeq, abspos=abspos)]
    return nn.Sequential(*layerSequence)
