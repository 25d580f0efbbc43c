"""Drop-in counterparts of the reference's cpc/model.py classes on the hot path.

Same class names, constructor and forward signatures, attributes and state-dict keys as
/root/reference/cpc/model.py (ChannelNorm :27-60, CPCEncoder :63-108, CPCAR :158-207,
CPCModel :279-390), so `cpc/train.py` can construct and drive them unchanged and reference
checkpoints load with load_state_dict.  The arithmetic runs in libcpc2_hip.so.

Parameter containers are the same torch modules the reference instantiates (nn.Conv1d,
nn.GRU) so that default initialisation under a given torch seed is identical; their own
forward() is never called.
"""
import os

import torch
import torch.nn as nn

from . import _lib
from ._lib import check, f32c, grad_buffers, ptr, ptr_array, require_gpu, scratch, stream_ptr
from ._tail import _TailScope, _all_in_place, _keep_for_tail, _no_hooks, _tail, _tail_tag, grad_home, grad_home_view, join_tail  # noqa: F401


# --------------------------------------------------------------------------- ChannelNorm
class _ChannelNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        require_gpu(x)
        lib = _lib.load()
        x = f32c(x)
        n, c, l = x.shape
        y = torch.empty_like(x)
        rstd = torch.empty(n * l, dtype=torch.float32, device=x.device)
        w = f32c(weight) if weight is not None else None
        b = f32c(bias) if bias is not None else None
        check(lib.cpc_channelnorm_forward(ptr(x), ptr(w), ptr(b), ptr(y), ptr(rstd), n, c, l, eps,
                                          stream_ptr(x.device)), "channelnorm_forward")
        ctx.save_for_backward(x, w, rstd)
        ctx.eps = eps
        ctx.affine = weight is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, w, rstd = ctx.saved_tensors
        dy = f32c(dy)
        n, c, l = x.shape
        dx = torch.empty_like(x)
        dw = torch.empty(1, c, 1, dtype=torch.float32, device=x.device) if ctx.affine else None
        db = torch.empty(1, c, 1, dtype=torch.float32, device=x.device) if ctx.affine else None
        check(lib.cpc_channelnorm_backward(ptr(x), ptr(w), ptr(dy), ptr(rstd), ptr(dx), ptr(dw), ptr(db),
                                           n, c, l, ctx.eps, stream_ptr(x.device)), "channelnorm_backward")
        return dx, dw, db, None


class ChannelNorm(nn.Module):
    """model.py:27-60 -- normalisation over the channel axis of [N, C, L], unbiased variance."""

    def __init__(self, numFeatures, epsilon=1e-05, affine=True):
        super(ChannelNorm, self).__init__()
        if affine:
            self.weight = nn.parameter.Parameter(torch.Tensor(1, numFeatures, 1))
            self.bias = nn.parameter.Parameter(torch.Tensor(1, numFeatures, 1))
        else:
            self.weight = None
            self.bias = None
        self.epsilon = epsilon
        self.p = 0
        self.affine = affine
        self.reset_parameters()

    def reset_parameters(self):
        if self.affine:
            torch.nn.init.ones_(self.weight)
            torch.nn.init.zeros_(self.bias)

    def forward(self, x):
        return _ChannelNormFn.apply(x, self.weight, self.bias, float(self.epsilon))


# --------------------------------------------------------------------------- CPCEncoder
_ENC_GEOMETRY = ((10, 5, 3), (8, 4, 2), (4, 2, 1), (4, 2, 1), (4, 2, 1))


class _EncoderFn(torch.autograd.Function):
    """relu(norm_i(conv_i(.))) x5 in one call; returns the channel-LAST output [N, T, H].  x2: None, or a second batch of
    windows that follows x (cpc_encoder_forward2: train.py:99's cat([past, future]) without the copy)."""

    @staticmethod
    def forward(ctx, x, x2, eps, defer_tail, *params):
        require_gpu(x, x2, *params)
        lib = _lib.load()
        x = f32c(x)
        x2 = f32c(x2) if x2 is not None else None
        ctx.param_refs = params
        ctx.defer_tail = bool(defer_tail)
        params = tuple(f32c(p) for p in params)
        n_first, cin, length = x.shape
        if cin != 1 or (x2 is not None and tuple(x2.shape[1:]) != (1, length)):
            raise ValueError(f"CPCEncoder expects [N, 1, L] waveforms (got {tuple(x.shape)}" + (f" and {tuple(x2.shape)})" if x2 is not None else ")"))
        n = n_first + (x2.shape[0] if x2 is not None else 0)
        hidden = params[0].shape[0]
        frames = lib.cpc_encoder_frames(length)
        nsaved = lib.cpc_encoder_saved_bytes(n, length, hidden)
        nscratch = lib.cpc_encoder_scratch_bytes(n, length, hidden)
        if nsaved == 0:
            check(-1, "encoder shape query")
        z = torch.empty(n, frames, hidden, dtype=torch.float32, device=x.device)
        saved = torch.empty(nsaved, dtype=torch.uint8, device=x.device)
        sc = scratch(nscratch, x.device)
        if x2 is None:
            check(lib.cpc_encoder_forward(ptr(x), ptr_array(params), ptr(z), ptr(saved), ptr(sc), n, length, hidden,
                                          eps, stream_ptr(x.device)), "encoder_forward")
        else:
            check(lib.cpc_encoder_forward2(ptr(x), ptr(x2), n_first, ptr_array(params), ptr(z), ptr(saved), ptr(sc), n, length, hidden,
                                           eps, stream_ptr(x.device)), "encoder_forward2")
        ctx.save_for_backward(x, x2, saved, *params)
        ctx.eps = eps
        ctx.dims = (n, length, hidden, n_first)
        return z

    @staticmethod
    def backward(ctx, dz):
        lib = _lib.load()
        x, x2, saved, *params = ctx.saved_tensors
        n, length, hidden, n_first = ctx.dims
        dz = f32c(dz)
        grads = grad_buffers(ctx.param_refs)
        nscratch = lib.cpc_encoder_scratch_bytes(n, length, hidden)
        # the deferred form (cpc2_hip.h): inside the caller's scope (CPCEncoder.deferred_weight_gradients) and with every gradient
        # of conv1-4 written in place into the flat gradient buffer
        defer = ctx.defer_tail and _all_in_place(ctx.param_refs[4:], grads[4:])
        # (deferred: a scratch buffer of its own -- the side stream outlives this call)
        sc = scratch(nscratch, x.device, tag=_tail_tag("enc_tail", x.device)) if defer else scratch(nscratch, x.device)
        if x2 is not None:
            check(lib.cpc_encoder_backward2(ptr(x), ptr(x2), n_first, ptr_array(params), ptr(dz), ptr(saved), ptr(sc), ptr_array(grads),
                                            n, length, hidden, ctx.eps, int(defer), stream_ptr(x.device)), "encoder_backward2")
        elif defer:
            check(lib.cpc_encoder_backward_deferred(ptr(x), ptr_array(params), ptr(dz), ptr(saved), ptr(sc), ptr_array(grads),
                                                    n, length, hidden, ctx.eps, stream_ptr(x.device)), "encoder_backward_deferred")
        else:
            check(lib.cpc_encoder_backward(ptr(x), ptr_array(params), ptr(dz), ptr(saved), ptr(sc), ptr_array(grads),
                                           n, length, hidden, ctx.eps, stream_ptr(x.device)), "encoder_backward")
        if defer:
            _keep_for_tail(x.device, (x, x2, saved, params, dz, sc))  # (not `grads`: see _GruFn.backward)
        return (None, None, None, None) + tuple(grads)


class CPCEncoder(nn.Module):
    """model.py:63-108.  Only normMode="layerNorm" (ChannelNorm, the default of the reference's
    config) runs on the fused HIP path; the other modes are not on the hot path."""

    def __init__(self, sizeHidden=512, normMode="layerNorm"):
        super(CPCEncoder, self).__init__()
        validModes = ["batchNorm", "instanceNorm", "ID", "layerNorm"]
        if normMode not in validModes:
            raise ValueError(f"Norm mode must be in {validModes}")
        if normMode != "layerNorm":
            raise NotImplementedError(
                f"normMode={normMode!r}: only 'layerNorm' (ChannelNorm) has an MI355X kernel path")
        self.dimEncoded = sizeHidden
        cin = 1
        for i, (k, s, p) in enumerate(_ENC_GEOMETRY):
            setattr(self, f"conv{i}", nn.Conv1d(cin, sizeHidden, k, stride=s, padding=p))
            setattr(self, f"batchNorm{i}", ChannelNorm(sizeHidden))
            cin = sizeHidden
        self.DOWNSAMPLING = 160
        self._defer_tail = False       # set by deferred_weight_gradients() for the duration of the caller's scope

    def getDimOutput(self):
        return self.conv4.out_channels

    def deferred_weight_gradients(self):
        """Context manager around the FORWARD call (see _TailScope): the backward of a forward pass made inside may leave the small
        passes that finish conv1-4's parameter gradients on the library's side stream (cpc_encoder_backward_deferred)."""
        return _TailScope(self)

    def _param_list(self):
        out = []
        for i in range(5):
            conv, norm = getattr(self, f"conv{i}"), getattr(self, f"batchNorm{i}")
            out += [conv.weight, conv.bias, norm.weight, norm.bias]
        return out

    def forward_channel_last(self, x, x_rest=None):
        """[N, 1, L] -> [N, T, H] (what CPCModel consumes); eps taken from batchNorm0.  x_rest: a second batch whose windows
        follow x's (the output is that of forward_channel_last(cat([x, x_rest])) without the concatenation)."""
        params = self._param_list()
        return _EncoderFn.apply(x, x_rest, float(self.batchNorm0.epsilon), self._defer_tail and _no_hooks(params[4:]), *params)

    def forward(self, x):
        # reference layout [N, H, T]; a permuted view of the channel-last buffer
        return self.forward_channel_last(x).permute(0, 2, 1)


# --------------------------------------------------------------------------- CPCAR (GRU / LSTM)
class _GruFn(torch.autograd.Function):
    """kind = "gru" or "rnn" (tanh): the two single-state recurrences share one calling convention."""

    @staticmethod
    def forward(ctx, x, h0, n_layers, want_hidden, kind, defer_tail, *params):
        require_gpu(x, *params)
        lib = _lib.load()
        ctx.dx_home = grad_home(x)             # (cpcStep's split_windows: dx has a fixed place in the encoder output's gradient)
        x = f32c(x)
        ctx.param_refs = params
        ctx.kind = kind
        ctx.defer_tail = bool(defer_tail) and kind == "gru"
        params = tuple(f32c(p) for p in params)
        n, t, dim_in = x.shape
        hidden = params[1].shape[1]
        nsaved = getattr(lib, f"cpc_{kind}_saved_bytes")(n, t, dim_in, hidden, n_layers)
        nscratch = getattr(lib, f"cpc_{kind}_scratch_bytes")(n, t, dim_in, hidden, n_layers)
        if nsaved == 0:
            check(-1, f"{kind} shape query")
        out = torch.empty(n, t, hidden, dtype=torch.float32, device=x.device)
        h_last = torch.empty(n_layers, n, hidden, dtype=torch.float32, device=x.device) if want_hidden else None
        h0c = f32c(h0) if h0 is not None else None
        saved = torch.empty(nsaved, dtype=torch.uint8, device=x.device)
        sc = scratch(nscratch, x.device)
        check(getattr(lib, f"cpc_{kind}_forward")(ptr(x), ptr_array(params), ptr(h0c), ptr(out), ptr(h_last), ptr(saved),
                                                  ptr(sc), n, t, dim_in, hidden, n_layers, stream_ptr(x.device)),
              f"{kind}_forward")
        ctx.save_for_backward(x, saved, *params)
        ctx.dims = (n, t, dim_in, hidden, n_layers)
        if want_hidden:
            ctx.mark_non_differentiable(h_last)
            return out, h_last
        return out, None

    @staticmethod
    def backward(ctx, dout, _dh):
        lib = _lib.load()
        x, saved, *params = ctx.saved_tensors
        n, t, dim_in, hidden, n_layers = ctx.dims
        dout = f32c(dout)
        need_dx = ctx.needs_input_grad[0]
        dx = grad_home_view(ctx.dx_home, x) if need_dx else None
        grads = grad_buffers(ctx.param_refs)
        kind = ctx.kind
        # The deferred form (cpc2_hip.h, cpc_gru_backward_deferred): every layer's weight gradients finish on a stream of the library's
        # while the encoder's backward runs.  Only inside the caller's scope (CPCAR.deferred_weight_gradients: nothing reads these
        # gradients before the backward pass has ended) and only when every one of them is written IN PLACE into the flat gradient
        # buffer -- a private buffer would be added to .grad by autograd the moment this function returns.
        defer = ctx.defer_tail and _all_in_place(ctx.param_refs, grads)
        nscratch = getattr(lib, f"cpc_{kind}_scratch_bytes")(n, t, dim_in, hidden, n_layers)
        if defer:
            sc = scratch(nscratch, x.device, tag=_tail_tag("gru_tail", x.device))      # a buffer of its own: the side stream outlives this call
            check(lib.cpc_gru_backward_deferred(ptr(x), ptr_array(params), ptr(dout), ptr(saved), ptr(sc), ptr(dx), ptr_array(grads),
                                                n, t, dim_in, hidden, n_layers, stream_ptr(x.device)), "gru_backward_deferred")
            # alive until the join.  NOT `grads`: autograd adopts a returned gradient as .grad only while nobody else holds it -- with a
            # second reference it CLONES it on the spot (the flat buffer's not yet written bytes) and the clone becomes .grad
            _keep_for_tail(x.device, (x, saved, params, dout, sc))
        else:
            sc = scratch(nscratch, x.device)
            check(getattr(lib, f"cpc_{kind}_backward")(ptr(x), ptr_array(params), ptr(dout), ptr(saved), ptr(sc), ptr(dx),
                                                       ptr_array(grads), n, t, dim_in, hidden, n_layers,
                                                       stream_ptr(x.device)), f"{kind}_backward")
        return (dx, None, None, None, None, None) + tuple(grads)


class _LstmFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, h0, c0, n_layers, want_hidden, defer_tail, *params):
        require_gpu(x, *params)
        lib = _lib.load()
        ctx.dx_home = grad_home(x)
        x = f32c(x)
        ctx.param_refs = params
        ctx.defer_tail = bool(defer_tail)
        params = tuple(f32c(p) for p in params)
        n, t, dim_in = x.shape
        hidden = params[1].shape[1]
        nsaved = lib.cpc_lstm_saved_bytes(n, t, dim_in, hidden, n_layers)
        nscratch = lib.cpc_lstm_scratch_bytes(n, t, dim_in, hidden, n_layers)
        if nsaved == 0:
            check(-1, "lstm shape query")
        out = torch.empty(n, t, hidden, dtype=torch.float32, device=x.device)
        h_last = torch.empty(n_layers, n, hidden, dtype=torch.float32, device=x.device) if want_hidden else None
        c_last = torch.empty_like(h_last) if want_hidden else None
        h0c = f32c(h0) if h0 is not None else None
        c0c = f32c(c0) if c0 is not None else None
        saved = torch.empty(nsaved, dtype=torch.uint8, device=x.device)
        sc = scratch(nscratch, x.device)
        check(lib.cpc_lstm_forward(ptr(x), ptr_array(params), ptr(h0c), ptr(c0c), ptr(out), ptr(h_last), ptr(c_last),
                                   ptr(saved), ptr(sc), n, t, dim_in, hidden, n_layers, stream_ptr(x.device)),
              "lstm_forward")
        ctx.save_for_backward(x, saved, *params)
        ctx.dims = (n, t, dim_in, hidden, n_layers)
        if want_hidden:
            ctx.mark_non_differentiable(h_last, c_last)
            return out, h_last, c_last
        return out, None, None

    @staticmethod
    def backward(ctx, dout, _dh, _dc):
        lib = _lib.load()
        x, saved, *params = ctx.saved_tensors
        n, t, dim_in, hidden, n_layers = ctx.dims
        dout = f32c(dout)
        need_dx = ctx.needs_input_grad[0]
        dx = grad_home_view(ctx.dx_home, x) if need_dx else None
        grads = grad_buffers(ctx.param_refs)
        nscratch = lib.cpc_lstm_scratch_bytes(n, t, dim_in, hidden, n_layers)
        if ctx.defer_tail and _all_in_place(ctx.param_refs, grads):                # (the deferred form: see _GruFn.backward)
            sc = scratch(nscratch, x.device, tag=_tail_tag("lstm_tail", x.device))
            check(lib.cpc_lstm_backward_deferred(ptr(x), ptr_array(params), ptr(dout), ptr(saved), ptr(sc), ptr(dx), ptr_array(grads),
                                                 n, t, dim_in, hidden, n_layers, stream_ptr(x.device)), "lstm_backward_deferred")
            _keep_for_tail(x.device, (x, saved, params, dout, sc))
        else:
            sc = scratch(nscratch, x.device)
            check(lib.cpc_lstm_backward(ptr(x), ptr_array(params), ptr(dout), ptr(saved), ptr(sc), ptr(dx),
                                        ptr_array(grads), n, t, dim_in, hidden, n_layers, stream_ptr(x.device)),
                  "lstm_backward")
        return (dx, None, None, None, None, None) + tuple(grads)


class CPCAR(nn.Module):
    """model.py:158-207.  mode="GRU", "LSTM" (this fork's default arMode) and "RNN" all run on the HIP path."""

    def __init__(self, dimEncoded, dimOutput, keepHidden, nLevelsGRU, mode="GRU", reverse=False):
        super(CPCAR, self).__init__()
        self.RESIDUAL_STD = 0.1
        rnn = {"LSTM": nn.LSTM, "RNN": nn.RNN}.get(mode, nn.GRU)        # anything else is a GRU (model.py:177-179)
        self.baseNet = rnn(dimEncoded, dimOutput, num_layers=nLevelsGRU, batch_first=True)
        self.hidden = None
        self.keepHidden = keepHidden
        self.reverse = reverse
        self._defer_tail = False       # set by deferred_weight_gradients() for the duration of the caller's scope

    def getDimOutput(self):
        return self.baseNet.hidden_size

    def deferred_weight_gradients(self):
        """Context manager around the FORWARD call: the backward of a forward pass made inside may leave layer 0's weight gradients
        on the library's side stream until the end of the backward pass (cpc_gru_backward_deferred).  The caller promises that
        nothing reads those gradients earlier -- no wrapper whose reducer copies a gradient the moment autograd has accumulated it
        (DistributedDataParallel / DataParallel around the model), no tensor hook on them; cpcStep opens it for the bare model."""
        return _TailScope(self)

    def _may_defer(self):
        return _no_hooks(self._param_list())

    def _param_list(self):
        out = []
        for layer in range(self.baseNet.num_layers):
            out += [getattr(self.baseNet, f"{n}_l{layer}") for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
        return out

    def forward(self, x):
        if self.reverse:
            x = torch.flip(x, [1])
        layers, keep = self.baseNet.num_layers, bool(self.keepHidden)
        if isinstance(self.baseNet, nn.LSTM):
            h0, c0 = self.hidden if self.hidden is not None else (None, None)
            x, h, c = _LstmFn.apply(x, h0, c0, layers, keep, self._defer_tail and self._may_defer(), *self._param_list())
            if self.keepHidden:
                self.hidden = (h.detach(), c.detach())
        else:
            kind = "rnn" if isinstance(self.baseNet, nn.RNN) else "gru"
            x, h = _GruFn.apply(x, self.hidden, layers, keep, kind, self._defer_tail and self._may_defer(), *self._param_list())
            if self.keepHidden:
                self.hidden = h.detach()
        # a sequence's order is preserved by each module (model.py:203-206)
        if self.reverse:
            x = torch.flip(x, [1])
        return x


class NoAr(nn.Module):
    """model.py:210-216 (arMode='no_ar'): the context IS the encoder output."""

    def __init__(self, *args):
        super(NoAr, self).__init__()

    def forward(self, x):
        return x


def _gru_layer_params(gru, layer, suffix=""):
    return [getattr(gru, f"{n}_l{layer}{suffix}") for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]


class BiDIRARTangled(nn.Module):
    """model.py:219-241 (cpc_mode='bert'): one bidirectional multi-layer GRU (keys `ARNet.*`, the backward direction's
    with the `_reverse` suffix).  Every (layer, direction) is one single-layer run of the GRU kernels; a layer's input is
    the concatenation of both directions of the layer below, as in torch.nn.GRU."""

    def __init__(self, dimEncoded, dimOutput, nLevelsGRU):
        super(BiDIRARTangled, self).__init__()
        assert dimOutput % 2 == 0
        self.ARNet = nn.GRU(dimEncoded, dimOutput // 2, num_layers=nLevelsGRU, batch_first=True, bidirectional=True)

    def getDimOutput(self):
        return self.ARNet.hidden_size * 2

    def forward(self, x):
        for layer in range(self.ARNet.num_layers):
            xf = _GruFn.apply(x, None, 1, False, "gru", False, *_gru_layer_params(self.ARNet, layer))[0]
            xb = _GruFn.apply(torch.flip(x, [1]), None, 1, False, "gru", False, *_gru_layer_params(self.ARNet, layer, "_reverse"))[0]
            x = torch.cat([xf, torch.flip(xb, [1])], dim=2)
        return x


class BiDIRAR(nn.Module):
    """model.py:244-272: two independent GRUs, one over the sequence and one over its mirror image."""

    def __init__(self, dimEncoded, dimOutput, nLevelsGRU):
        super(BiDIRAR, self).__init__()
        assert dimOutput % 2 == 0
        self.netForward = nn.GRU(dimEncoded, dimOutput // 2, num_layers=nLevelsGRU, batch_first=True)
        self.netBackward = nn.GRU(dimEncoded, dimOutput // 2, num_layers=nLevelsGRU, batch_first=True)

    def getDimOutput(self):
        return self.netForward.hidden_size * 2

    def _run(self, gru, x):
        params = [p for layer in range(gru.num_layers) for p in _gru_layer_params(gru, layer)]
        return _GruFn.apply(x, None, gru.num_layers, False, "gru", False, *params)[0]

    def forward(self, x):
        xf = self._run(self.netForward, x)
        xb = self._run(self.netBackward, torch.flip(x, [1]))
        return torch.cat([xf, torch.flip(xb, [1])], dim=2)


class LSTMPredictor(nn.LSTM):
    """nn.LSTM(dimOutputAR, dimOutputEncoder, batch_first=True) as a predictor (criterion.py:119-123): same parameters and
    state-dict keys, forward on the HIP kernels; returns (output, None) -- the criterion only takes element 0."""

    def forward(self, x, hx=None):
        if hx is not None or not self.batch_first or self.bidirectional or self.proj_size:
            raise NotImplementedError("LSTMPredictor: batch_first, unidirectional, zero initial state only")
        params = [getattr(self, f"{n}_l{layer}") for layer in range(self.num_layers)
                  for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
        return _LstmFn.apply(x, None, None, self.num_layers, False, False, *params)[0], None


class RNNPredictor(nn.RNN):
    """nn.RNN(dimOutputAR, dimOutputEncoder) as a predictor (criterion.py:115-118).  Like the reference's it is NOT
    batch_first: fed c [b, W, H], the recurrence runs along b and W is the batch."""

    def forward(self, x, hx=None):
        if hx is not None or self.batch_first or self.bidirectional or self.nonlinearity != "tanh":
            raise NotImplementedError("RNNPredictor: time-major, unidirectional tanh RNN with zero initial state only")
        params = [getattr(self, f"{n}_l{layer}") for layer in range(self.num_layers)
                  for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
        out = _GruFn.apply(x.transpose(0, 1).contiguous(), None, self.num_layers, False, "rnn", False, *params)[0]
        return out.transpose(0, 1).contiguous(), None


# --------------------------------------------------------------------------- CPCModel
def span_mask(batch, frames, mask_prob, mask_length, min_masks=0):
    """Boolean [batch, frames] mask of model.py:300-365 (the simplified wav2vec 2.0 span sampler): the same draws from
    numpy's global generator in the same order, so a given np.random.seed masks the same frames as the reference.
    Every sequence gets the same number of spans (int(mask_prob * 100 * frames / mask_length + U[0,1))), span starts
    are drawn without replacement, overlapping spans merge, and the rows are then thinned to the shortest one."""
    import numpy as np
    n_spans = max(min_masks, int(mask_prob * 100 * frames / float(mask_length) + np.random.rand()))
    rows = []
    for _ in range(batch):
        lengths = np.full(n_spans, mask_length)
        if lengths.sum() == 0:
            lengths[0] = min(mask_length, frames - 1)
        shortest = min(lengths)
        if frames - shortest <= n_spans:
            shortest = frames - n_spans - 1
        starts = np.random.choice(frames - shortest, n_spans, replace=False)
        covered = np.asarray([st + off for st, ln in zip(starts, lengths) for off in range(ln)])
        rows.append(np.unique(covered[covered < frames]))
    keep = min(len(r) for r in rows)
    mask = np.zeros((batch, frames), dtype=bool)
    for i, r in enumerate(rows):
        if len(r) > keep:
            r = np.random.choice(r, keep, replace=False)
        mask[i, r] = True
    return mask


class CPCModel(nn.Module):
    """model.py:279-390.  mask_prob > 0: spans of encoder frames are overwritten by a learned embedding before the
    context network (host-side span sampling + one torch index_put; like the reference, the returned encodedData is the
    masked tensor too, :373-385)."""

    def __init__(self, encoder, AR, mask_prob=0.0, mask_length=10):
        super(CPCModel, self).__init__()
        self.gEncoder = encoder
        self.gAR = AR
        self.mask_prob = mask_prob
        self.mask_length = mask_length
        if mask_prob > 0.0:
            self.mask_emb = nn.Parameter(torch.FloatTensor(encoder.dimEncoded).uniform_())

    def compute_mask_indices(self, shape, mask_prob, mask_length, min_masks=0):
        return span_mask(shape[0], shape[1], mask_prob, mask_length, min_masks)

    def getMask(self, features):
        batchSize, seqSize, _ = features.shape
        mask = self.compute_mask_indices((batchSize, seqSize), self.mask_prob, self.mask_length, min_masks=2)
        if mask.mean() > 0.6:
            import warnings
            warnings.warn("We detected that %.2f of all encoded frames have been masked. This might be too much." % mask.mean())
        features[torch.from_numpy(mask).to(features.device)] = self.mask_emb
        return features

    def forward(self, batchData, label):
        # (an encoder with forward hooks -- feature taps, profilers -- is CALLED, so that they fire; its channel-first output is
        #  then permuted as the reference does, model.py:382)
        hooked = bool(self.gEncoder._forward_hooks or self.gEncoder._forward_pre_hooks)
        if hasattr(self.gEncoder, "forward_channel_last") and not hooked:
            encodedData = self.gEncoder.forward_channel_last(batchData)      # [N, T, H], contiguous
        else:
            encodedData = self.gEncoder(batchData).permute(0, 2, 1)
        if self.mask_prob > 0.0:
            encodedData = self.getMask(encodedData)
        # (before the context network: autograd then runs this node's backward AFTER the context network's, and a criterion
        #  backward that was deferred beside it is joined there -- criterion.py, grad_join)
        from .criterion import grad_join
        encodedOut = grad_join(encodedData)
        cFeature = self.gAR(encodedData)
        return cFeature, encodedOut, label
