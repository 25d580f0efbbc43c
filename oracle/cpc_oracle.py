"""CPU restatement of the CPC training hot path (oracle, test infrastructure only).

Plain functional torch-CPU code over explicit parameter dicts (state-dict key
names of the reference).  Works in float32 (the reference's arithmetic) or
float64 (a tighter checker for the fp32 HIP kernels).  Each function cites the
reference lines it follows; parity with the reference is pinned by
tests/test_oracle_golden.py on fixtures made by tools/make_golden.py.

Never imported by the product package (cpc2_amd/).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from .mt19937 import MT19937, negative_indices

# (kernel, stride, padding) of the five encoder convolutions -- model.py:85-94
ENCODER_GEOMETRY = ((10, 5, 3), (8, 4, 2), (4, 2, 1), (4, 2, 1), (4, 2, 1))
DOWNSAMPLING = 160  # model.py:96


# --------------------------------------------------------------------------- #
# Encoder
# --------------------------------------------------------------------------- #
def channel_norm(x, weight, bias, eps=1e-5):
    """model.py:52-60.  x [N, C, L]; statistics over the CHANNEL axis, variance
    unbiased (torch.var default, divide by C-1); affine params shaped [1, C, 1]."""
    mean = x.mean(dim=1, keepdim=True)
    centred = x - mean
    var = (centred * centred).sum(dim=1, keepdim=True) / (x.shape[1] - 1)
    y = centred * torch.rsqrt(var + eps)
    if weight is not None:
        y = y * weight + bias
    return y


def encoder_forward(x, p, prefix="", return_all=False, masks=None, pre_out=None):
    """model.py:102-108.  x [N, 1, L] -> [N, H, L/160]; relu(norm_i(conv_i(.))).

    masks (test infrastructure): five 0/1 tensors shaped like the layer outputs; layer i then computes norm_i(.) * masks[i]
    instead of relu(norm_i(.)) -- the same function wherever the mask is the sign of the pre-activation, with the ReLU
    DECISIONS taken from somewhere else (an fp32 evaluation decides a pre-activation within rounding of zero the other way;
    the gradients below that element then differ by whole terms, which says nothing about the arithmetic under test).
    pre_out: a list that receives the five pre-activations (detached)."""
    outs = []
    for i, (_k, s, pad) in enumerate(ENCODER_GEOMETRY):
        x = F.conv1d(x, p[f"{prefix}conv{i}.weight"], p[f"{prefix}conv{i}.bias"], stride=s, padding=pad)
        x = channel_norm(x, p[f"{prefix}batchNorm{i}.weight"], p[f"{prefix}batchNorm{i}.bias"])
        if pre_out is not None:
            pre_out.append(x.detach())
        x = torch.relu(x) if masks is None else x * masks[i].to(x.dtype)
        outs.append(x)
    return outs if return_all else x


# --------------------------------------------------------------------------- #
# Autoregressive network: GRU (model.py:178-207; torch.nn.GRU, gate order r,z,n)
# --------------------------------------------------------------------------- #
def gru_forward(x, p, n_layers, prefix="", h0=None, reverse=False):
    """x [N, T, Hin] batch_first.  Returns (out [N, T, H], h_last [n_layers, N, H]).

        r = sigmoid(W_ir x + b_ir + W_hr h + b_hr)
        z = sigmoid(W_iz x + b_iz + W_hz h + b_hz)
        n = tanh  (W_in x + b_in + r * (W_hn h + b_hn))
        h' = (1 - z) * n + z * h
    reverse=True flips time before and after (model.py:190-191, 205-206)."""
    if reverse:
        x = torch.flip(x, [1])
    n, t_len, _ = x.shape
    h_last = []
    inp = x
    for layer in range(n_layers):
        w_ih, w_hh = p[f"{prefix}weight_ih_l{layer}"], p[f"{prefix}weight_hh_l{layer}"]
        b_ih, b_hh = p[f"{prefix}bias_ih_l{layer}"], p[f"{prefix}bias_hh_l{layer}"]
        hid = w_hh.shape[1]
        h = torch.zeros(n, hid, dtype=x.dtype) if h0 is None else h0[layer]
        gi_all = inp @ w_ih.t() + b_ih
        steps = []
        for t in range(t_len):
            gi = gi_all[:, t]
            gh = h @ w_hh.t() + b_hh
            r = torch.sigmoid(gi[:, :hid] + gh[:, :hid])
            z = torch.sigmoid(gi[:, hid:2 * hid] + gh[:, hid:2 * hid])
            cand = torch.tanh(gi[:, 2 * hid:] + r * gh[:, 2 * hid:])
            h = (1 - z) * cand + z * h
            steps.append(h)
        inp = torch.stack(steps, dim=1)
        h_last.append(h)
    out = inp
    if reverse:
        out = torch.flip(out, [1])
    return out, torch.stack(h_last, dim=0)


def lstm_forward(x, p, n_layers, prefix="", h0=None, c0=None, reverse=False):
    """CPCAR(mode="LSTM"), model.py:171-173 -> torch.nn.LSTM (batch_first, gate order i, f, g, o).
    x [N, T, Hin].  Returns (out [N, T, H], h_last [n_layers, N, H], c_last [n_layers, N, H]).

        i = sigmoid(W_ii x + b_ii + W_hi h + b_hi)      f = sigmoid(W_if x + b_if + W_hf h + b_hf)
        g = tanh   (W_ig x + b_ig + W_hg h + b_hg)      o = sigmoid(W_io x + b_io + W_ho h + b_ho)
        c' = f * c + i * g                               h' = o * tanh(c')"""
    if reverse:
        x = torch.flip(x, [1])
    n, t_len, _ = x.shape
    h_last, c_last = [], []
    inp = x
    for layer in range(n_layers):
        w_ih, w_hh = p[f"{prefix}weight_ih_l{layer}"], p[f"{prefix}weight_hh_l{layer}"]
        b_ih, b_hh = p[f"{prefix}bias_ih_l{layer}"], p[f"{prefix}bias_hh_l{layer}"]
        hid = w_hh.shape[1]
        h = torch.zeros(n, hid, dtype=x.dtype) if h0 is None else h0[layer]
        c = torch.zeros(n, hid, dtype=x.dtype) if c0 is None else c0[layer]
        gi_all = inp @ w_ih.t() + b_ih
        steps = []
        for t in range(t_len):
            pre = gi_all[:, t] + h @ w_hh.t() + b_hh
            i = torch.sigmoid(pre[:, :hid])
            f = torch.sigmoid(pre[:, hid:2 * hid])
            g = torch.tanh(pre[:, 2 * hid:3 * hid])
            o = torch.sigmoid(pre[:, 3 * hid:])
            c = f * c + i * g
            h = o * torch.tanh(c)
            steps.append(h)
        inp = torch.stack(steps, dim=1)
        h_last.append(h)
        c_last.append(c)
    out = inp
    if reverse:
        out = torch.flip(out, [1])
    return out, torch.stack(h_last, dim=0), torch.stack(c_last, dim=0)


def rnn_forward(x, p, n_layers, prefix="", h0=None, reverse=False):
    """CPCAR(mode="RNN"), model.py:174-176 -> torch.nn.RNN (tanh):  h' = tanh(W_ih x + b_ih + W_hh h + b_hh)."""
    if reverse:
        x = torch.flip(x, [1])
    n, t_len, _ = x.shape
    h_last = []
    inp = x
    for layer in range(n_layers):
        w_ih, w_hh = p[f"{prefix}weight_ih_l{layer}"], p[f"{prefix}weight_hh_l{layer}"]
        b_ih, b_hh = p[f"{prefix}bias_ih_l{layer}"], p[f"{prefix}bias_hh_l{layer}"]
        hid = w_hh.shape[1]
        h = torch.zeros(n, hid, dtype=x.dtype) if h0 is None else h0[layer]
        gi_all = inp @ w_ih.t() + b_ih
        steps = []
        for t in range(t_len):
            h = torch.tanh(gi_all[:, t] + h @ w_hh.t() + b_hh)
            steps.append(h)
        inp = torch.stack(steps, dim=1)
        h_last.append(h)
    out = inp
    if reverse:
        out = torch.flip(out, [1])
    return out, torch.stack(h_last, dim=0)


# --------------------------------------------------------------------------- #
# Autoregressive network: light causal transformer (transformers.py:10-134)
# --------------------------------------------------------------------------- #
def layer_norm(x, w, b, eps=1e-5):
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * w + b


def relpos_bias(q, krelpos):
    """Closed form of the 'skew' trick of transformers.py:61-66.
    q [B, S, dk], krelpos [dk, S] -> R [B, S, S] with R[i, j] = q_i . Krelpos[:, S-1-(i-j)]
    for j <= i.  Entries with j > i are masked to -inf by the caller, so their value is
    irrelevant; here they follow the same view arithmetic as the reference."""
    b, s, _ = q.shape
    qp = q @ krelpos                                       # [B, S, S]
    qp = torch.cat([torch.zeros(b, s, 1, dtype=q.dtype), qp], dim=2)   # [B, S, S+1]
    return qp.reshape(b, s + 1, s)[:, 1:, :]


def transformer_layer_forward(x, p, prefix, n_heads=8, size_seq=None, n_classifiers=1):
    """One TransformerLayer (transformers.py:119-134) in eval mode (dropout off).
    x [N, S, D] with S == sizeSeq (the training window).
    n_classifiers > 1: MultiClassifierTransformerHead (transformers.py:137-158): lin2 emits n_classifiers
    residual branches, output [N, S, n_classifiers, Dout]."""
    n, s, d = x.shape
    dk = d // n_heads
    if size_seq is not None and size_seq != s:
        # transformers.py:38-50: blocks of size_seq frames, a ragged tail zero-padded (Q, K, V have no bias, so padding
        # the layer input is the same thing); the rows of the padding are dropped (:69)
        pad = (-s) % size_seq
        xp = torch.cat([x, torch.zeros(n, pad, d, dtype=x.dtype)], dim=1) if pad else x
        blocks = xp.reshape(n * ((s + pad) // size_seq), size_seq, d)
        out = transformer_layer_forward(blocks, p, prefix, n_heads=n_heads, size_seq=size_seq, n_classifiers=n_classifiers)
        return out.reshape((n, s + pad) + tuple(out.shape[2:]))[:, :s]

    def split(v):   # trans_ (transformers.py:89-91)
        return v.view(n, s, n_heads, dk).transpose(1, 2).reshape(n * n_heads, s, dk)

    q = split(x @ p[f"{prefix}multihead.Wq.weight"].t())
    k = split(x @ p[f"{prefix}multihead.Wk.weight"].t())
    v = split(x @ p[f"{prefix}multihead.Wv.weight"].t())
    scores = q @ k.transpose(1, 2)
    key = f"{prefix}multihead.Att.Krelpos"
    if key in p:
        scores = scores + relpos_bias(q, p[key])
    mask = torch.triu(torch.full((s, s), float("-inf"), dtype=x.dtype), diagonal=1)
    att = torch.softmax(scores / math.sqrt(dk) + mask, dim=2)
    y = (att @ v).view(n, n_heads, s, dk).transpose(1, 2).reshape(n, s, d)
    y = y @ p[f"{prefix}multihead.Wo.weight"].t()
    y = layer_norm(x + y, p[f"{prefix}ln_multihead.weight"], p[f"{prefix}ln_multihead.bias"])
    ff = torch.relu(y @ p[f"{prefix}ffnetwork.lin1.weight"].t() + p[f"{prefix}ffnetwork.lin1.bias"])
    ff = ff @ p[f"{prefix}ffnetwork.lin2.weight"].t() + p[f"{prefix}ffnetwork.lin2.bias"]
    if n_classifiers > 1:            # transformers.py:153-158
        ff = ff.view(n, s, n_classifiers, d)
        y = y.view(n, s, 1, d)
    out = (y + ff) @ p[f"{prefix}last_linear.weight"].t() + p[f"{prefix}last_linear.bias"]
    return layer_norm(out, p[f"{prefix}ln_ffnetwork.weight"], p[f"{prefix}ln_ffnetwork.bias"])


# --------------------------------------------------------------------------- #
# CPCModel.forward (model.py:381-390)
# --------------------------------------------------------------------------- #
def model_forward(x, p, n_layers_gru=1, ar="GRU", reverse=False):
    """Returns (cFeature [N,T,H], encodedData [N,T,H])."""
    z = encoder_forward(x, p, "gEncoder.").permute(0, 2, 1)
    if ar == "GRU":
        c, _ = gru_forward(z, p, n_layers_gru, "gAR.baseNet.", reverse=reverse)
    elif ar == "LSTM":
        c = lstm_forward(z, p, n_layers_gru, "gAR.baseNet.", reverse=reverse)[0]
    elif ar == "RNN":
        c = rnn_forward(z, p, n_layers_gru, "gAR.baseNet.", reverse=reverse)[0]
    elif ar == "transformer":
        c = z
        for layer in range(n_layers_gru):
            c = transformer_layer_forward(c, p, f"gAR.{layer}.")
    else:
        raise ValueError(ar)
    return c, z


# --------------------------------------------------------------------------- #
# CPCUnsupersivedCriterion (criterion.py:193-363), linear predictors (:144-150)
# --------------------------------------------------------------------------- #
def quality_weights(signal_quality, growth_rate, inflection_point_x, window):
    """criterion.py:230, 334-338: per-window weight repeated over the W time steps."""
    q = signal_quality.mean(dim=1)
    w = 0.00001 + 1 / (1 + torch.exp(-growth_rate * (q - inflection_point_x)))
    return w.unsqueeze(1).repeat(1, window).reshape(-1)


def transformer_predictors(p, k_steps, prefix="wPrediction.predictors."):
    """rnnMode='transformer' (criterion.py:136-143): K one-layer transformers applied to c[:, :W]."""
    return [(lambda c, i=i: transformer_layer_forward(c, p, f"{prefix}{i}.0.")) for i in range(k_steps)]


def recurrent_predictors(p, k_steps, mode, prefix="wPrediction.predictors."):
    """rnnMode='LSTM' (criterion.py:119-123: nn.LSTM, batch_first) or 'RNN' (:115-118: nn.RNN WITHOUT batch_first, so
    the recurrence runs along the batch axis of c [b, W, H] and W plays the batch)."""
    if mode == "LSTM":
        return [(lambda c, i=i: lstm_forward(c, p, 1, f"{prefix}{i}.")[0]) for i in range(k_steps)]
    return [(lambda c, i=i: rnn_forward(c.transpose(0, 1), p, 1, f"{prefix}{i}.")[0].transpose(0, 1)) for i in range(k_steps)]


def static_position_embedding(seqlen, dmodel, dtype=torch.float32):
    """transformers.py:161-173."""
    pos = torch.arange(0., seqlen, dtype=torch.float64).unsqueeze(1).repeat(1, dmodel)
    dim = torch.arange(0., dmodel, dtype=torch.float64).unsqueeze(0).repeat(seqlen, 1)
    pos = pos * torch.exp(-math.log(10000) * (2 * torch.div(dim, 2, rounding_mode="floor") / dmodel))
    pos[:, 0::2] = torch.sin(pos[:, 0::2])
    pos[:, 1::2] = torch.cos(pos[:, 1::2])
    return pos.unsqueeze(0).to(dtype)


def multihead_predictors(p, k_steps, prefix="wPrediction.predictor."):
    """--multihead_rnn (criterion.py:44-94): one MultiClassifierTransformerHead, prediction k = head(c)[:, :, k].
    (Evaluated once per k here: the oracle favours brevity.)"""
    return [(lambda c, i=i: transformer_layer_forward(c, p, f"{prefix}0.", n_classifiers=k_steps)[:, :, i])
            for i in range(k_steps)]


def criterion_forward(c, z, predictors, ext_idx, n_neg, mode=None, n_skipped=0, weights=None):
    """c [b,T,Har], z [b,T,Henc], predictors = list of K matrices [Henc, Har] (nn.Linear
    weights, no bias) or of K callables (predictor modules), ext_idx int64 [n_neg*W*b] from negative_indices().

    Returns (losses [1, K-n_skipped], acc [1, K-n_skipped]).

    For step k (1-based): candidates = [z[:, k:k+W] (positive), z_flat[ext_idx] (negatives)],
    logits = <W_k c_t, cand> / Henc  (the reference takes .mean over the feature axis,
    criterion.py:171), loss_k = mean_i(w_i * CE(logits_i, 0)), acc_k = #(argmax == 0)/(W b).
    The same negatives serve every k (criterion.py:267-284)."""
    if mode == "reverse":            # criterion.py:292-294
        z = torch.flip(z, [1])
        c = torch.flip(c, [1])
    b, t_len, h_enc = z.shape
    k_steps = len(predictors)
    w_len = t_len - k_steps
    c = c[:, :w_len]
    idx = torch.as_tensor(np.asarray(ext_idx), dtype=torch.long)
    neg = z.reshape(-1, h_enc)[idx].view(b, n_neg, w_len, h_enc)
    if weights is None:
        weights = torch.ones(b * w_len, dtype=z.dtype)
    losses, accs = [], []
    for k in range(1, k_steps + 1):
        wk = predictors[k - 1]
        pred = (wk(c) if callable(wk) else c @ wk.t()).unsqueeze(1)  # [b,1,W,H]; callable: a predictor module
        pos = z[:, k:k + w_len].unsqueeze(1)                        # [b,1,W,H]
        cand = torch.cat([pos, neg], dim=1)                         # [b,1+n,W,H]
        logits = (pred * cand).mean(dim=3)                          # [b,1+n,W]
        logits = logits.permute(0, 2, 1).reshape(b * w_len, 1 + n_neg)
        ce = torch.logsumexp(logits, dim=1) - logits[:, 0]
        losses.append((weights * ce).mean().view(1, 1))
        accs.append((logits.argmax(dim=1) == 0).sum().to(z.dtype).view(1, 1))
    losses, accs = losses[n_skipped:], accs[n_skipped:]
    return torch.cat(losses, dim=1), torch.cat(accs, dim=1) / (w_len * b)


def criterion_forward_sparse(c, z, predictors, ext_idx, n_neg, mode=None, n_skipped=0, weights=None, dlosses=None,
                             windows_per_chunk=8):
    """The same criterion (criterion.py:237-286 sampleClean, :144-171 linear predictors + mean-dot, :329-363 CE / acc) with its
    gradients written out by hand, for batch sizes at which the dense restatement above (K tensors [b, 1+n, W, H] under autograd)
    does not fit: the negatives are gathered once per chunk of windows, every step k reuses them (criterion.py:267-284: one
    draw serves all K), dz collects the contributions of the negatives by index_add_.  Computes in the dtype of z (use float64).

    dlosses [K - n_skipped]: gradient of the returned losses (default ones = `losses.sum().backward()`, train.py:108).
    Returns dict(losses [1, K-s], acc [1, K-s], dc like c, dz like z, dW list of K [Henc, Har])."""
    if mode == "reverse":            # criterion.py:292-294
        z, c = torch.flip(z, [1]), torch.flip(c, [1])
    c, z = c.detach(), z.detach()
    b, t_len, h_enc = z.shape
    k_steps = len(predictors)
    w_len = t_len - k_steps
    dt = z.dtype
    idx = torch.as_tensor(np.asarray(ext_idx), dtype=torch.long).view(b, n_neg, w_len)   # criterion.py:263-265
    wts = (torch.ones(b * w_len, dtype=dt) if weights is None else weights.to(dt)).view(b, w_len)
    g_loss = torch.zeros(k_steps, dtype=dt)
    g_loss[n_skipped:] = 1.0 if dlosses is None else torch.as_tensor(dlosses, dtype=dt).reshape(-1)
    preds_w = [p.detach().to(dt) for p in predictors]
    zflat = z.reshape(-1, h_enc)
    loss = torch.zeros(k_steps, dtype=dt)
    hits = torch.zeros(k_steps, dtype=dt)
    dc = torch.zeros_like(c)
    dz = torch.zeros_like(z)
    dzflat = dz.view(-1, h_enc)
    dws = [torch.zeros_like(p) for p in preds_w]
    scale = 1.0 / (b * w_len)        # torch.mean over the b * W rows (criterion.py:349)
    rows = torch.arange(b).view(b, 1) * t_len + torch.arange(w_len).view(1, w_len)       # flat z row of frame t of window b
    for lo in range(0, b, windows_per_chunk):
        hi = min(b, lo + windows_per_chunk)
        neg = zflat[idx[lo:hi].reshape(-1)].view(hi - lo, n_neg, w_len, h_enc)           # criterion.py:266
        g_neg = torch.zeros_like(neg)
        cw = c[lo:hi, :w_len]
        for k in range(1, k_steps + 1):
            wk = preds_w[k - 1]
            pred = cw @ wk.t()                                                           # [cb, W, Henc]
            pos = z[lo:hi, k:k + w_len]
            s_pos = (pred * pos).sum(-1) / h_enc                                         # mean over features, criterion.py:171
            s_neg = torch.einsum("bwh,bnwh->bwn", pred, neg) / h_enc
            # a negative that IS the positive's row scores the same bit for bit in the reference (one product, one mean), and
            # max(1) then answers 0 (criterion.py:356); two summation orders here would decide that tie by rounding
            same = idx[lo:hi].permute(0, 2, 1) == (rows[lo:hi] + k).unsqueeze(-1)
            s_neg = torch.where(same, s_pos.unsqueeze(-1), s_neg)
            logits = torch.cat([s_pos.unsqueeze(-1), s_neg], dim=-1)                     # positive first = label 0
            lse = torch.logsumexp(logits, dim=-1)
            loss[k - 1] += (wts[lo:hi] * (lse - s_pos)).sum() * scale
            hits[k - 1] += (s_pos >= s_neg.max(dim=-1).values).sum()
            d_logits = torch.exp(logits - lse.unsqueeze(-1))
            d_logits[..., 0] -= 1.0
            d_logits *= (g_loss[k - 1] * scale * wts[lo:hi]).unsqueeze(-1) / h_enc
            d_pos, d_neg = d_logits[..., 0], d_logits[..., 1:]
            d_pred = d_pos.unsqueeze(-1) * pos + torch.einsum("bwn,bnwh->bwh", d_neg, neg)
            dz[lo:hi, k:k + w_len] += d_pos.unsqueeze(-1) * pred
            g_neg += d_neg.permute(0, 2, 1).unsqueeze(-1) * pred.unsqueeze(1)
            dc[lo:hi, :w_len] += d_pred @ wk
            dws[k - 1] += d_pred.reshape(-1, h_enc).t() @ cw.reshape(-1, cw.shape[-1])
        dzflat.index_add_(0, idx[lo:hi].reshape(-1), g_neg.view(-1, h_enc))
    if mode == "reverse":
        dc, dz = torch.flip(dc, [1]), torch.flip(dz, [1])
    return {"losses": loss[n_skipped:].view(1, -1), "acc": (hits[n_skipped:] * scale).view(1, -1), "dc": dc, "dz": dz, "dW": dws}


def candidates(z, ext_idx, n_neg, k_steps):
    """sampleClean (criterion.py:237-286) given the indices: K tensors [b, 1 + n_neg, W, Henc], positive first."""
    b, t_len, h_enc = z.shape
    w_len = t_len - k_steps
    idx = torch.as_tensor(np.asarray(ext_idx), dtype=torch.long)
    neg = z.reshape(-1, h_enc)[idx].view(b, n_neg, w_len, h_enc)
    return [torch.cat([z[:, k:k + w_len].unsqueeze(1), neg], dim=1) for k in range(1, k_steps + 1)]


def prediction_scores(c, z, predictors, ext_idx, n_neg, mode=None):
    """getPrediction (criterion.py:291-302): K score tensors [b, 1 + n_neg, W] = mean over channels of
    prediction * candidate.  ext_idx None: getCosineDistances (:304-327), the positive alone, [b, 1, W]."""
    if mode == "reverse":
        z = torch.flip(z, [1])
        c = torch.flip(c, [1])
    k_steps = len(predictors)
    w_len = z.shape[1] - k_steps
    c = c[:, :w_len]
    if ext_idx is None:
        cands = [z[:, k:k + w_len].unsqueeze(1) for k in range(1, k_steps + 1)]
    else:
        cands = candidates(z, ext_idx, n_neg, k_steps)
    out = []
    for wk, cand in zip(predictors, cands):
        pred = (wk(c) if callable(wk) else c @ wk.t()).unsqueeze(1)
        out.append((pred * cand).mean(dim=3))
    return out


# --------------------------------------------------------------------------- #
# Adam (torch.optim.Adam defaults as used at train.py:477-479: no weight decay, no amsgrad)
# --------------------------------------------------------------------------- #
class Adam:
    def __init__(self, params, lr=2e-4, betas=(0.9, 0.999), eps=1e-8):
        self.params = params            # dict name -> tensor (updated in place)
        self.lr, self.betas, self.eps = lr, betas, eps
        self.m = {k: torch.zeros_like(v) for k, v in params.items()}
        self.v = {k: torch.zeros_like(v) for k, v in params.items()}
        self.t = 0

    def step(self, grads):
        self.t += 1
        b1, b2 = self.betas
        c1, c2 = 1 - b1 ** self.t, 1 - b2 ** self.t
        for k, prm in self.params.items():
            g = grads[k]
            self.m[k].mul_(b1).add_(g, alpha=1 - b1)
            self.v[k].mul_(b2).addcmul_(g, g, value=1 - b2)
            denom = self.v[k].sqrt() / math.sqrt(c2) + self.eps
            prm.addcdiv_(self.m[k], denom, value=-self.lr / c1)


# --------------------------------------------------------------------------- #
# trainStep (train.py:87-113): one optimisation step
# --------------------------------------------------------------------------- #
def predictor_list(p, k_steps, prefix="wPrediction.predictors."):
    return [p[f"{prefix}{k}.weight"] for k in range(k_steps)]


def train_step_loss(past, future, model_p, crit_p, mt, k_steps, n_neg, n_layers_gru=1, ar="GRU"):
    """Forward of one training step exactly as train.py:95-108 wires it: the model runs on
    cat([past, future]) (2b windows); context comes from the PAST half, targets from the
    FUTURE half.  mt is an MT19937 whose stream supplies the negatives.
    Returns (total_loss, losses [1,K], acc [1,K])."""
    b = past.shape[0]
    c, z = model_forward(torch.cat([past, future], dim=0), model_p, n_layers_gru, ar)
    c, z = c[:b], z[b:]
    t_len = z.shape[1]
    _, _, ext = negative_indices(mt, b, t_len, t_len - k_steps, n_neg)
    losses, acc = criterion_forward(c, z, predictor_list(crit_p, k_steps), ext, n_neg)
    return losses.sum(), losses, acc


def train_steps(past, future, model_p, crit_p, seed, n_steps, k_steps, n_neg,
                n_layers_gru=1, ar="GRU", lr=2e-4):
    """n_steps Adam steps on a fixed batch (the loss-curve anchor G6).  Parameter order for
    the optimiser follows train.py:472: criterion parameters, then model parameters."""
    params = {}
    for name, v in list(crit_p.items()) + list(model_p.items()):
        params[name] = v.clone().requires_grad_(True)
    opt = Adam({k: v.data for k, v in params.items()}, lr=lr)
    mt = MT19937(seed)
    curve = []
    for _ in range(n_steps):
        mp = {k: params[k] for k in model_p}
        cp = {k: params[k] for k in crit_p}
        tot, losses, _acc = train_step_loss(past, future, mp, cp, mt, k_steps, n_neg, n_layers_gru, ar)
        grads = torch.autograd.grad(tot, list(params.values()), allow_unused=True)
        g = {k: (gi if gi is not None else torch.zeros_like(params[k])) for k, gi in zip(params, grads)}
        opt.step(g)
        curve.append(losses.detach().clone())
    return torch.cat(curve, dim=0), {k: v.detach() for k, v in params.items()}
