"""Pins the CPU oracle (oracle/) to golden vectors produced by the reference
implementation (tools/make_golden.py).  CPU only."""
import ctypes
import hashlib
import os
import subprocess

import numpy as np
import pytest
import torch

from oracle import cpc_oracle as O
from oracle import synth
from oracle.mt19937 import MT19937, negative_indices

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def t(a):
    return torch.from_numpy(np.asarray(a))


def strip(p, prefix):
    return {k[len(prefix):]: v for k, v in p.items() if k.startswith(prefix)}


# ----------------------------------------------------------------------------- G1
@pytest.mark.parametrize("tag", ["tiny", "mid"])
def test_negative_indices_bit_exact(golden, tag):
    g = golden("g1_negidx.npz")
    seed, b, t_len, k, nn = (int(v) for v in g[f"{tag}_cfg"])
    bi, si, ext = negative_indices(MT19937(seed), b, t_len, t_len - k, nn)
    assert np.array_equal(bi, g[f"{tag}_batchIdx"])
    assert np.array_equal(si, g[f"{tag}_seqIdx"])
    assert np.array_equal(ext, g[f"{tag}_extIdx"])


def test_negative_indices_full_size_and_stream_continuity(golden):
    g = golden("g1_negidx.npz")
    seed, b, t_len, k, nn = (int(v) for v in g["full_cfg"])
    mt = MT19937(seed)
    _, _, ext = negative_indices(mt, b, t_len, t_len - k, nn)
    assert hashlib.sha256(ext.astype("<i8").tobytes()).hexdigest() == str(g["full_ext_sha256"])
    assert np.array_equal(ext[:64], g["full_ext_head"]) and np.array_equal(ext[-64:], g["full_ext_tail"])
    _, _, ext2 = negative_indices(mt, b, t_len, t_len - k, nn)     # second step, same stream
    assert hashlib.sha256(ext2.astype("<i8").tobytes()).hexdigest() == str(g["full_ext2_sha256"])


def test_mt_state_roundtrip_with_torch():
    torch.manual_seed(4321)
    torch.randint(0, 10, (1000,))
    st = torch.get_rng_state()
    mt = MT19937.from_torch_state(st.numpy().tobytes())
    mine = mt.randint(1, 128, 5000)
    theirs = torch.randint(1, 128, (5000,)).numpy()
    assert np.array_equal(mine, theirs)
    back = np.frombuffer(mt.to_torch_state(st.numpy().tobytes()), dtype=np.uint8)
    assert np.array_equal(back, torch.get_rng_state().numpy())


def test_c_oracle_matches_numpy_oracle(tmp_path):
    so = tmp_path / "liboracle_mt.so"
    subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", str(so),
                           os.path.join(ROOT, "oracle", "mt19937.c")])
    lib = ctypes.CDLL(str(so))
    lib.oracle_mt_sizeof.restype = ctypes.c_size_t
    st = ctypes.create_string_buffer(lib.oracle_mt_sizeof())
    lib.oracle_mt_seed(st, ctypes.c_uint32(1234))
    b, t_len, w, nn = 8, 128, 116, 128
    n = nn * w * b
    outs = [np.empty(n, dtype=np.int64) for _ in range(3)]
    lib.oracle_negative_indices(st, b, t_len, w, nn, *[o.ctypes.data_as(ctypes.c_void_p) for o in outs])
    ref = negative_indices(MT19937(1234), b, t_len, w, nn)
    for a, r in zip(outs, ref):
        assert np.array_equal(a, r)


# ----------------------------------------------------------------------------- G2/G3
def test_channel_norm(golden):
    g = golden("g3_channelnorm.npz")
    x = t(g["x"]).requires_grad_(True)
    w, b = t(g["w"]).requires_grad_(True), t(g["b"]).requires_grad_(True)
    y = O.channel_norm(x, w, b)
    (y * t(g["g"])).sum().backward()
    assert torch.allclose(y, t(g["y"]), atol=1e-6, rtol=1e-6)
    assert torch.allclose(x.grad, t(g["dx"]), atol=2e-6, rtol=1e-5)
    assert torch.allclose(w.grad, t(g["dw"]), atol=1e-5, rtol=1e-5)
    assert torch.allclose(b.grad, t(g["db"]), atol=1e-5, rtol=1e-5)


def test_channel_norm_variance_is_unbiased():
    x = torch.randn(2, 8, 5)
    y = O.channel_norm(x, None, None)
    # with the unbiased estimator the normalised rows have sum of squares C-1, not C
    assert torch.allclose((y * y).sum(1), torch.full((2, 5), 7.0), atol=1e-3)


def test_encoder(golden):
    g = golden("g2_encoder_h32.npz")
    hidden = int(g["hidden"])
    p = {k: v.clone().requires_grad_(True) for k, v in synth.encoder_params(hidden, int(g["param_seed"])).items()}
    x = synth.audio_windows(2, 20480, int(g["x_seed"]))
    acts = O.encoder_forward(x, p, "gEncoder.", return_all=True)
    for i, a in enumerate(acts):
        assert torch.allclose(a[:, :, :4], t(g[f"act{i}_head"]), atol=2e-6, rtol=1e-5)
        assert torch.allclose(a[:, :, -4:], t(g[f"act{i}_tail"]), atol=2e-6, rtol=1e-5)
        assert abs(float(a.detach().double().abs().sum()) - float(g[f"act{i}_abs"])) <= 1e-6 * float(g[f"act{i}_abs"])
    out = acts[-1]
    assert out.shape == (2, hidden, 128)
    assert torch.allclose(out, t(g["out"]), atol=2e-6, rtol=1e-5)
    (out * synth.features((2, hidden, 128), int(g["gout_seed"]))).sum().backward()
    for k, v in p.items():
        ref = t(g["grad." + k[len("gEncoder."):]])
        scale = float(ref.abs().max())
        assert torch.allclose(v.grad, ref, atol=2e-5 * scale, rtol=1e-4), k


# ----------------------------------------------------------------------------- G4
@pytest.mark.parametrize("tag", ["l1", "l2"])
def test_gru(golden, tag):
    g = golden("g4_gru.npz")
    hin, hid, layers, n, t_len = (int(v) for v in g[f"{tag}_cfg"])
    p = {k: v.clone().requires_grad_(True) for k, v in synth.gru_params(hin, hid, layers, 41).items()}
    x = synth.features((n, t_len, hin), 42, relu=True).requires_grad_(True)
    out, _ = O.gru_forward(x, p, layers, "gAR.baseNet.")
    assert torch.allclose(out, t(g[f"{tag}_out"]), atol=1e-6, rtol=1e-5)
    (out * synth.features((n, t_len, hid), 43)).sum().backward()
    assert torch.allclose(x.grad, t(g[f"{tag}_dx"]), atol=1e-5, rtol=1e-4)
    for k, v in p.items():
        assert torch.allclose(v.grad, t(g[f"{tag}_grad." + k[len("gAR."):]]), atol=1e-5, rtol=1e-4), k


def test_gru_reverse(golden):
    g = golden("g4_gru.npz")
    p = synth.gru_params(32, 32, 1, 41)
    x = synth.features((3, 20, 32), 42, relu=True)
    out, _ = O.gru_forward(x, p, 1, "gAR.baseNet.", reverse=True)
    assert torch.allclose(out, t(g["rev_out"]), atol=1e-6, rtol=1e-5)


# ----------------------------------------------------------------------------- G12
_RECURRENT = {"LSTM": (synth.lstm_params, O.lstm_forward), "RNN": (synth.rnn_params, O.rnn_forward)}


@pytest.mark.parametrize("mode", ["LSTM", "RNN"])
@pytest.mark.parametrize("tag", ["l1", "l2"])
def test_lstm_rnn(golden, mode, tag):
    g = golden("g12_lstm_rnn.npz")
    make, fwd = _RECURRENT[mode]
    hin, hid, layers, n, t_len = (int(v) for v in g[f"{mode}_{tag}_cfg"])
    p = {k: v.clone().requires_grad_(True) for k, v in make(hin, hid, layers, 51).items()}
    x = synth.features((n, t_len, hin), 52, relu=True).requires_grad_(True)
    out = fwd(x, p, layers, "gAR.baseNet.")[0]
    assert torch.allclose(out, t(g[f"{mode}_{tag}_out"]), atol=1e-6, rtol=1e-5)
    (out * synth.features((n, t_len, hid), 53)).sum().backward()
    assert torch.allclose(x.grad, t(g[f"{mode}_{tag}_dx"]), atol=1e-5, rtol=1e-4)
    for k, v in p.items():
        assert torch.allclose(v.grad, t(g[f"{mode}_{tag}_grad." + k[len("gAR."):]]), atol=1e-5, rtol=1e-4), k


@pytest.mark.parametrize("mode", ["LSTM", "RNN"])
def test_lstm_rnn_keep_hidden(golden, mode):
    g = golden("g12_lstm_rnn.npz")
    make, fwd = _RECURRENT[mode]
    p = make(32, 32, 2, 51)
    xa = synth.features((2, 9, 32), 54, relu=True)
    xb = synth.features((2, 7, 32), 55, relu=True)
    state = fwd(xa, p, 2, "gAR.baseNet.")[1:]
    if mode == "LSTM":
        out2, h, c = fwd(xb, p, 2, "gAR.baseNet.", h0=state[0], c0=state[1])
        assert torch.allclose(c, t(g["LSTM_keep_c"]), atol=1e-6, rtol=1e-5)
    else:
        out2, h = fwd(xb, p, 2, "gAR.baseNet.", h0=state[0])
    assert torch.allclose(out2, t(g[f"{mode}_keep_out2"]), atol=1e-6, rtol=1e-5)
    assert torch.allclose(h, t(g[f"{mode}_keep_h"]), atol=1e-6, rtol=1e-5)


# ----------------------------------------------------------------------------- G5
def _criterion_case(b, t_len, har, henc, k, nn, seed, pseed, mode=None, n_skipped=0, weights=None):
    p = {kk: v.clone().requires_grad_(True)
         for kk, v in synth.predictor_params(k, har, henc, seed=pseed, scale=4.0).items()}
    c = synth.features((b, t_len, har), pseed + 1).requires_grad_(True)
    z = synth.features((b, t_len, henc), pseed + 2, relu=True).requires_grad_(True)
    _, _, ext = negative_indices(MT19937(seed), b, t_len, t_len - k, nn)
    losses, acc = O.criterion_forward(c, z, O.predictor_list(p, k), ext, nn, mode=mode,
                                      n_skipped=n_skipped, weights=weights)
    losses.sum().backward()
    return losses, acc, c, z, p


@pytest.mark.parametrize("tag", ["plain", "skip", "reverse", "quality", "rect"])
def test_criterion_small(golden, tag):
    g = golden("g5_criterion_small.npz")
    kw = dict(b=4, t_len=32, har=32, henc=32, k=4, nn=16, seed=99, pseed=50)
    if tag == "skip":
        kw["n_skipped"] = 1
    if tag == "reverse":
        kw["mode"] = "reverse"
    if tag == "rect":
        kw["har"] = 24
    if tag == "quality":
        kw["weights"] = O.quality_weights(t(g["quality_signal"]), 2.0, 0.1, 32 - 4)
    losses, acc, c, z, p = _criterion_case(**kw)
    assert torch.equal(losses, t(g[f"{tag}_losses"])) or torch.allclose(losses, t(g[f"{tag}_losses"]), atol=0, rtol=2e-7)
    assert torch.allclose(acc, t(g[f"{tag}_acc"]), atol=1e-6)
    assert torch.allclose(c.grad, t(g[f"{tag}_dc"]), atol=1e-7, rtol=1e-4)
    assert torch.allclose(z.grad, t(g[f"{tag}_dz"]), atol=1e-7, rtol=1e-4)
    for i in range(4):
        grad = p[f"wPrediction.predictors.{i}.weight"].grad
        if grad is None:      # skipped step: no gradient reaches its predictor
            grad = torch.zeros_like(p[f"wPrediction.predictors.{i}.weight"])
        assert torch.allclose(grad, t(g[f"{tag}_dW{i}"]), atol=1e-7, rtol=1e-4)


def test_criterion_full_shapes(golden):
    g = golden("g5_criterion_full.npz")
    losses, acc, c, z, p = _criterion_case(b=8, t_len=128, har=256, henc=256, k=12, nn=128, seed=1234, pseed=60)
    assert torch.allclose(losses, t(g["losses"]), atol=0, rtol=1e-6)
    assert torch.allclose(acc, t(g["acc"]), atol=2e-3)          # argmax ties
    assert torch.allclose(c.grad[:, :3, :8], t(g["dc_head"]), atol=1e-8, rtol=1e-4)
    assert torch.allclose(z.grad[:, :3, :8], t(g["dz_head"]), atol=1e-8, rtol=1e-4)
    assert torch.allclose(z.grad[:, -3:, :8], t(g["dz_tail"]), atol=1e-8, rtol=1e-4)
    for name, ten in (("dc", c.grad), ("dz", z.grad)):
        assert abs(float(ten.double().abs().sum()) - float(g[f"{name}_abs"])) <= 1e-5 * float(g[f"{name}_abs"])
    for i in range(12):
        w = p[f"wPrediction.predictors.{i}.weight"].grad
        assert abs(float(w.double().abs().sum()) - float(g[f"dW{i}_abs"])) <= 1e-5 * float(g[f"dW{i}_abs"])
        assert torch.allclose(w[:4, :8], t(g[f"dW{i}_head"]), atol=1e-8, rtol=1e-4)


def _sparse_vs(ref_losses, ref_acc, ref_dc, ref_dz, ref_dw, got, tol):
    def close(a, b_, what):
        err = float((a.double() - b_.double()).abs().max() / (b_.double().abs().max() + 1e-300))
        assert err <= tol, f"{what}: {err:.2e}"
    close(got["losses"], ref_losses, "losses")
    close(got["acc"], ref_acc, "acc")
    close(got["dc"], ref_dc, "dc")
    close(got["dz"], ref_dz, "dz")
    for i, w in enumerate(ref_dw):
        if w is None or float(w.abs().max()) == 0.0:
            assert float(got["dW"][i].abs().max()) == 0.0
        else:
            close(got["dW"][i], w, f"dW{i}")


@pytest.mark.parametrize("tag", ["plain", "skip", "reverse", "quality", "rect", "many_negatives", "full_b8"])
def test_sparse_criterion_is_the_dense_one(golden, tag):
    """oracle.criterion_forward_sparse (hand-written gradients, negatives gathered per chunk of windows: the checker of the HIP
    criterion at b = 64) against the autograd restatement that the goldens above pin, in float64, chunk sizes that do not
    divide the batch."""
    kw = dict(b=4, t_len=32, har=32, henc=32, k=4, nn=16, seed=99, pseed=50)
    okw = {}
    if tag == "skip":
        okw["n_skipped"] = 1
    if tag == "reverse":
        okw["mode"] = "reverse"
    if tag == "rect":
        kw["har"] = 24
    if tag == "quality":
        okw["weights"] = O.quality_weights(t(golden("g5_criterion_small.npz")["quality_signal"]).double(), 2.0, 0.1, 28)
    if tag == "many_negatives":
        kw = dict(b=5, t_len=33, har=128, henc=128, k=7, nn=129, seed=7, pseed=3)
    if tag == "full_b8":
        kw = dict(b=8, t_len=128, har=256, henc=256, k=12, nn=128, seed=1234, pseed=60)
    b, t_len, k, nn = kw["b"], kw["t_len"], kw["k"], kw["nn"]
    p = {n: v.double().requires_grad_(True) for n, v in synth.predictor_params(k, kw["har"], kw["henc"], seed=kw["pseed"], scale=4.0).items()}
    c = synth.features((b, t_len, kw["har"]), kw["pseed"] + 1).double().requires_grad_(True)
    z = synth.features((b, t_len, kw["henc"]), kw["pseed"] + 2, relu=True).double().requires_grad_(True)
    _, _, ext = negative_indices(MT19937(kw["seed"]), b, t_len, t_len - k, nn)
    losses, acc = O.criterion_forward(c, z, O.predictor_list(p, k), ext, nn, **okw)
    dl = torch.linspace(0.5, 1.5, losses.shape[1], dtype=torch.float64)           # (a non-uniform gradient of the losses)
    (losses * dl).sum().backward()
    got = O.criterion_forward_sparse(c, z, O.predictor_list(p, k), ext, nn, dlosses=dl, windows_per_chunk=3, **okw)
    _sparse_vs(losses.detach(), acc, c.grad, z.grad, [p[f"wPrediction.predictors.{i}.weight"].grad for i in range(k)], got, 1e-12)


def test_sparse_criterion_at_full_batch_vs_reference_golden(golden):
    """g18: the reference's criterion at b = 64 (BASELINE configs[1]: T 128, H 256, K 12, 128 negatives), predictors at trained
    scale.  The reference ran in float32; the sparse oracle in float64 agrees with its losses to 1e-6 and with every stored
    gradient element (heads, tails, a strided sample) and digest to float32 rounding."""
    g = golden("g18_criterion_b64.npz")
    b, t_len, har, henc, k, nn, seed, pseed = (int(v) for v in g["cfg"])
    p = {n: v.double() for n, v in synth.predictor_params(k, har, henc, seed=pseed, scale=float(g["scale"])).items()}
    c = synth.features((b, t_len, har), pseed + 1).double()
    z = synth.features((b, t_len, henc), pseed + 2, relu=True).double()
    _, _, ext = negative_indices(MT19937(seed), b, t_len, t_len - k, nn)
    got = O.criterion_forward_sparse(c, z, O.predictor_list(p, k), ext, nn)
    assert torch.allclose(got["losses"], t(g["losses"]).double(), atol=0, rtol=2e-6)
    assert torch.allclose(got["acc"], t(g["acc"]).double(), atol=2.5 / (b * (t_len - k)))          # float32 argmax ties
    for name in ("dc", "dz"):
        ten = got[name]
        scale = float(ten.abs().max())
        for part, view in (("head", ten[:, :3, :8]), ("tail", ten[:, -14:, :8]), ("sample", ten[::9, ::5, ::37])):
            assert torch.allclose(view, t(g[f"{name}_{part}"]).double(), atol=2e-6 * scale, rtol=1e-4), (name, part)
        assert abs(float(ten.abs().sum()) - float(g[f"{name}_abs"])) <= 1e-5 * float(g[f"{name}_abs"])
        assert abs(float(ten.sum()) - float(g[f"{name}_sum"])) <= 1e-5 * float(g[f"{name}_abs"])
    assert float(got["dc"][:, t_len - k:].abs().max()) == 0.0
    for i in range(k):
        w = got["dW"][i]
        assert torch.allclose(w[::17, ::13], t(g[f"dW{i}_sample"]).double(), atol=2e-6 * float(w.abs().max()), rtol=1e-4)
        assert abs(float(w.abs().sum()) - float(g[f"dW{i}_abs"])) <= 1e-5 * float(g[f"dW{i}_abs"])


# ----------------------------------------------------------------------------- G6
def test_train_steps_loss_curve(golden):
    g = golden("g6_trainsteps.npz")
    hidden, b, k, nn, steps, seed = (int(v) for v in g["cfg"])
    mp = synth.encoder_params(hidden, 21)
    mp.update(synth.gru_params(hidden, hidden, 1, 22))
    cp = synth.predictor_params(k, hidden, hidden, 23)
    x = synth.audio_windows(b, 20480, 24)
    steps = 6     # the first 6 of the 20 reference steps keep this test short
    curve, final = O.train_steps(x, x, mp, cp, seed, steps, k, nn)
    ref = t(g["curve"])[:steps]
    assert torch.allclose(curve, ref, atol=0, rtol=2e-5), (curve - ref).abs().max()


# ----------------------------------------------------------------------------- G7
def test_transformer_layer(golden):
    g = golden("g7_transformer.npz")
    d_model, s, n = (int(v) for v in g["cfg"])
    p = {k: v.clone().requires_grad_(True) for k, v in synth.transformer_params(d_model, d_model, s, 71).items()}
    x = synth.features((n, s, d_model), 72, relu=True).requires_grad_(True)
    out = O.transformer_layer_forward(x, p, "gAR.0.")
    assert torch.allclose(out, t(g["out"]), atol=2e-6, rtol=1e-5)
    (out * synth.features((n, s, d_model), 73)).sum().backward()
    assert torch.allclose(x.grad, t(g["dx"]), atol=1e-5, rtol=1e-4)
    for k, v in p.items():
        ref = t(g["grad." + k[len("gAR."):]])
        assert torch.allclose(v.grad, ref, atol=2e-5 * float(ref.abs().max()) + 1e-7, rtol=1e-4), k


# ----------------------------------------------------------------------------- G8
def test_criterion_with_transformer_predictors(golden):
    g = golden("g8_criterion_transformer_pred.npz")
    b, t_len, h, k, nn, seed = (int(v) for v in g["cfg"])
    p = {}
    for i in range(k):
        p.update(synth.transformer_params(h, h, t_len - k, seed=80 + i, prefix=f"wPrediction.predictors.{i}.0."))
    p = {n: v.clone().requires_grad_(True) for n, v in p.items()}
    c = synth.features((b, t_len, h), 90).requires_grad_(True)
    z = synth.features((b, t_len, h), 91, relu=True).requires_grad_(True)
    _, _, ext = negative_indices(MT19937(seed), b, t_len, t_len - k, nn)
    losses, acc = O.criterion_forward(c, z, O.transformer_predictors(p, k), ext, nn)
    losses.sum().backward()
    assert torch.allclose(losses, t(g["losses"]), atol=0, rtol=1e-6)
    assert torch.allclose(c.grad, t(g["dc"]), atol=1e-8, rtol=1e-4)
    assert torch.allclose(z.grad, t(g["dz"]), atol=1e-8, rtol=1e-4)


# ----------------------------------------------------------------------------- G14
@pytest.mark.parametrize("mode,gates", [("LSTM", 4), ("RNN", 1)])
def test_criterion_with_recurrent_predictors(golden, mode, gates):
    """rnnMode='LSTM' / 'RNN' (criterion.py:115-123); the RNN is not batch_first and recurs along the batch axis."""
    g = golden("g14_criterion_recurrent_pred.npz")
    b, t_len, har, henc, k, nn, seed = (int(v) for v in g["cfg"])
    p = {}
    for i in range(k):
        p.update(synth.gru_params(har, henc, 1, seed=110 + i, prefix=f"wPrediction.predictors.{i}.", gates=gates))
    p = {n: v.clone().requires_grad_(True) for n, v in p.items()}
    c = synth.features((b, t_len, har), 120).requires_grad_(True)
    z = synth.features((b, t_len, henc), 121, relu=True).requires_grad_(True)
    _, _, ext = negative_indices(MT19937(seed), b, t_len, t_len - k, nn)
    losses, acc = O.criterion_forward(c, z, O.recurrent_predictors(p, k, mode), ext, nn)
    losses.sum().backward()
    assert torch.allclose(losses, t(g[f"{mode}_losses"]), atol=0, rtol=1e-6)
    assert torch.allclose(c.grad, t(g[f"{mode}_dc"]), atol=1e-8, rtol=1e-4)
    assert torch.allclose(z.grad, t(g[f"{mode}_dz"]), atol=1e-8, rtol=1e-4)
    for n, v in p.items():
        assert torch.allclose(v.grad, t(g[f"{mode}_grad." + n]), atol=1e-8, rtol=1e-4), n


# ----------------------------------------------------------------------------- G9
def test_criterion_with_multihead_predictor(golden):
    """--multihead_rnn (criterion.py:44-94): one transformer head with nPredicts residual branches."""
    g = golden("g9_criterion_multihead_pred.npz")
    b, t_len, h, k, nn, seed = (int(v) for v in g["cfg"])
    p = synth.transformer_params(h, h, t_len - k, seed=95, prefix="wPrediction.predictor.0.", n_classifiers=k)
    p = {n: v.clone().requires_grad_(True) for n, v in p.items()}
    c = synth.features((b, t_len, h), 96).requires_grad_(True)
    z = synth.features((b, t_len, h), 97, relu=True).requires_grad_(True)
    _, _, ext = negative_indices(MT19937(seed), b, t_len, t_len - k, nn)
    losses, acc = O.criterion_forward(c, z, O.multihead_predictors(p, k), ext, nn)
    losses.sum().backward()
    assert torch.allclose(losses, t(g["losses"]), atol=0, rtol=1e-6)
    assert torch.allclose(c.grad, t(g["dc"]), atol=1e-8, rtol=1e-4)
    assert torch.allclose(z.grad, t(g["dz"]), atol=1e-8, rtol=1e-4)
    for name, v in p.items():
        ref = t(g["grad." + name])
        assert torch.allclose(v.grad, ref, atol=2e-5 * float(ref.abs().max()) + 1e-9, rtol=1e-4), name


# ----------------------------------------------------------------------------- G10
@pytest.mark.parametrize("tag", ["abspos", "ragged", "abspos_ragged"])
def test_transformer_variants(golden, tag):
    """abspos=True (StaticPositionEmbedding, no relative positions) and lengths that are not a multiple of sizeSeq."""
    g = golden("g10_transformer_variants.npz")
    d_model, size_seq, n = (int(v) for v in g["cfg"])
    s_len = int(g[tag + "_len"])
    abspos = tag.startswith("abspos")
    p = synth.transformer_params(d_model, d_model, size_seq, 171)
    if abspos:
        p = {k: v for k, v in p.items() if not k.endswith("Krelpos")}
    p = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    x = synth.features((n, s_len, d_model), 172, relu=True).requires_grad_(True)
    xin = x + O.static_position_embedding(size_seq, d_model)[:, :s_len] if abspos else x
    out = O.transformer_layer_forward(xin, p, "gAR.0.", size_seq=size_seq)
    assert torch.allclose(out, t(g[tag + "_out"]), atol=2e-6, rtol=1e-5)
    (out * synth.features((n, s_len, d_model), 173)).sum().backward()
    assert torch.allclose(x.grad, t(g[tag + "_dx"]), atol=1e-5, rtol=1e-4)
    layer = "1." if abspos else "0."
    for k, v in p.items():
        ref = t(g[tag + "_grad." + layer + k[len("gAR.0."):]])
        assert torch.allclose(v.grad, ref, atol=2e-5 * float(ref.abs().max()) + 1e-7, rtol=1e-4), k


# ----------------------------------------------------------------------------- G16 criterion, inference-side API
@pytest.mark.parametrize("tag,mode", [("plain", None), ("reverse", "reverse")])
def test_prediction_scores_and_candidates(golden, tag, mode):
    """getPrediction / getCosineDistances / sampleClean (criterion.py:237-327) restated by the oracle."""
    g = golden("g16_criterion_inference.npz")
    b, t_len, har, henc, k, nn, seed, pseed = 4, 32, 32, 32, 4, 16, 99, 50
    p = synth.predictor_params(k, har, henc, seed=pseed, scale=4.0)
    c = synth.features((b, t_len, har), pseed + 1)
    z = synth.features((b, t_len, henc), pseed + 2, relu=True)
    mt = MT19937(seed)
    _, _, ext = negative_indices(mt, b, t_len, t_len - k, nn)
    scores = O.prediction_scores(c, z, O.predictor_list(p, k), ext, nn, mode=mode)
    assert torch.allclose(torch.stack(scores), t(g[f"{tag}_pred"]), atol=1e-7, rtol=2e-6)
    assert (g[f"{tag}_label"] == 0).all() and g[f"{tag}_label"].shape == (b * (t_len - k),)
    # the generator stands where the reference's stood after the call
    assert np.array_equal(mt.randint(0, 1000, 4), g[f"{tag}_next_draws"])
    cos = O.prediction_scores(c, z, O.predictor_list(p, k), None, nn, mode=mode)
    assert torch.allclose(torch.stack(cos), t(g[f"{tag}_cos"]), atol=1e-7, rtol=2e-6)
    # sampleClean is called on the tensors as given (no flip): candidates of the unflipped z
    cands = O.candidates(z, ext, nn, k)
    assert torch.equal(cands[0], t(g[f"{tag}_cand_first"])) and torch.equal(cands[-1], t(g[f"{tag}_cand_last"]))
