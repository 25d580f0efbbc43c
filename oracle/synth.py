"""Deterministic synthetic parameters / inputs (numpy RandomState, stable across torch
versions).  Shared by tools/make_golden.py (which feeds them to the reference), the
tests and bench.py, so that large fixtures only need to store seeds and outputs.

Test infrastructure only -- never imported by cpc2_amd/.
"""
import numpy as np
import torch

from .cpc_oracle import ENCODER_GEOMETRY


def _uniform(rs, shape, bound):
    return torch.from_numpy(rs.uniform(-bound, bound, size=shape).astype(np.float32))


def encoder_params(hidden, seed, prefix="gEncoder."):
    rs = np.random.RandomState(seed)
    p = {}
    cin = 1
    for i, (k, _s, _p) in enumerate(ENCODER_GEOMETRY):
        bound = 1.0 / np.sqrt(cin * k)
        p[f"{prefix}conv{i}.weight"] = _uniform(rs, (hidden, cin, k), bound)
        p[f"{prefix}conv{i}.bias"] = _uniform(rs, (hidden,), bound)
        p[f"{prefix}batchNorm{i}.weight"] = torch.from_numpy(
            (1.0 + 0.1 * rs.standard_normal((1, hidden, 1))).astype(np.float32))
        p[f"{prefix}batchNorm{i}.bias"] = torch.from_numpy(
            (0.1 * rs.standard_normal((1, hidden, 1))).astype(np.float32))
        cin = hidden
    return p


def gru_params(dim_in, hidden, n_layers, seed, prefix="gAR.baseNet.", gates=3):
    rs = np.random.RandomState(seed)
    bound = 1.0 / np.sqrt(hidden)
    p = {}
    for layer in range(n_layers):
        d = dim_in if layer == 0 else hidden
        p[f"{prefix}weight_ih_l{layer}"] = _uniform(rs, (gates * hidden, d), bound)
        p[f"{prefix}weight_hh_l{layer}"] = _uniform(rs, (gates * hidden, hidden), bound)
        p[f"{prefix}bias_ih_l{layer}"] = _uniform(rs, (gates * hidden,), bound)
        p[f"{prefix}bias_hh_l{layer}"] = _uniform(rs, (gates * hidden,), bound)
    return p


def lstm_params(dim_in, hidden, n_layers, seed, prefix="gAR.baseNet."):
    return gru_params(dim_in, hidden, n_layers, seed, prefix, gates=4)


def rnn_params(dim_in, hidden, n_layers, seed, prefix="gAR.baseNet."):
    return gru_params(dim_in, hidden, n_layers, seed, prefix, gates=1)


def predictor_params(k_steps, dim_ar, dim_enc, seed, prefix="wPrediction.predictors.", scale=1.0):
    rs = np.random.RandomState(seed)
    bound = scale / np.sqrt(dim_ar)
    return {f"{prefix}{k}.weight": _uniform(rs, (dim_enc, dim_ar), bound) for k in range(k_steps)}


def transformer_params(d_model, d_out, size_seq, seed, prefix="gAR.0.", n_heads=8, dff=2048, n_classifiers=1):
    rs = np.random.RandomState(seed)
    dk = d_model // n_heads
    b = 1.0 / np.sqrt(d_model)
    p = {}
    for w in ("Wo", "Wk", "Wq", "Wv"):
        p[f"{prefix}multihead.{w}.weight"] = _uniform(rs, (d_model, d_model), b)
    p[f"{prefix}multihead.Att.Krelpos"] = _uniform(rs, (dk, size_seq), 1.0 / np.sqrt(dk))
    p[f"{prefix}ln_multihead.weight"] = torch.from_numpy((1 + 0.1 * rs.standard_normal(d_model)).astype(np.float32))
    p[f"{prefix}ln_multihead.bias"] = torch.from_numpy((0.1 * rs.standard_normal(d_model)).astype(np.float32))
    p[f"{prefix}ffnetwork.lin1.weight"] = _uniform(rs, (dff, d_model), b)
    p[f"{prefix}ffnetwork.lin1.bias"] = _uniform(rs, (dff,), b)
    p[f"{prefix}ffnetwork.lin2.weight"] = _uniform(rs, (d_model * n_classifiers, dff), 1.0 / np.sqrt(dff))
    p[f"{prefix}ffnetwork.lin2.bias"] = _uniform(rs, (d_model * n_classifiers,), 1.0 / np.sqrt(dff))
    p[f"{prefix}last_linear.weight"] = _uniform(rs, (d_out, d_model), b)
    p[f"{prefix}last_linear.bias"] = _uniform(rs, (d_out,), b)
    p[f"{prefix}ln_ffnetwork.weight"] = torch.from_numpy((1 + 0.1 * rs.standard_normal(d_out)).astype(np.float32))
    p[f"{prefix}ln_ffnetwork.bias"] = torch.from_numpy((0.1 * rs.standard_normal(d_out)).astype(np.float32))
    return p


def audio_windows(n, length, seed, rms=0.05):
    """Speech-like-RMS gaussian windows [n, 1, length]."""
    rs = np.random.RandomState(seed)
    return torch.from_numpy((rms * rs.standard_normal((n, 1, length))).astype(np.float32))


def features(shape, seed, scale=1.0, relu=False):
    rs = np.random.RandomState(seed)
    a = scale * rs.standard_normal(shape)
    if relu:
        a = np.maximum(a, 0.0)
    return torch.from_numpy(a.astype(np.float32))
