#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE implementation (CPU).

Runs only in the build container where /root/reference exists:
    PYTHONDONTWRITEBYTECODE=1 python tools/make_golden.py

The reference is imported unmodified.  Two modules it imports but never uses on this
path (torchaudio, progressbar) are absent from the image and are registered as empty
stubs; `torch.ones(..., device='cuda')` at criterion.py:340 is redirected to the CPU
(arithmetic untouched).  Only inputs and outputs are written -- no reference code.
"""
import hashlib
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
for _m in ("torchaudio", "progressbar"):
    sys.modules.setdefault(_m, types.ModuleType(_m))

import cpc.model as ref_model                      # noqa: E402
import cpc.criterion.criterion as ref_crit          # noqa: E402
import cpc.transformers as ref_tr                   # noqa: E402

from oracle import synth                            # noqa: E402

_real_ones = torch.ones


def _ones_cpu(*a, **kw):
    if kw.get("device") == "cuda":
        kw["device"] = "cpu"
    return _real_ones(*a, **kw)


torch.ones = _ones_cpu
OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)
torch.set_num_threads(8)


def npy(d):
    return {k: (v.detach().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in d.items()}


def save(name, **arrays):
    arrays["torch_version"] = np.array(torch.__version__)
    arrays = npy(arrays)
    np.savez_compressed(os.path.join(OUT, name), **arrays)
    print("wrote", name, sum(v.nbytes for v in arrays.values()), "bytes (raw)")


def strip(p, prefix):
    return {k[len(prefix):]: v for k, v in p.items() if k.startswith(prefix)}


# ------------------------------------------------------------------ G1 indices
def ref_indices(seed, b, t_len, k_steps, n_neg):
    """Runs the reference's sampleClean and recovers its index tensors by replaying
    torch.randint on the same generator state (the reference does not return them)."""
    w = t_len - k_steps
    crit = ref_crit.CPCUnsupersivedCriterion(k_steps, 8, 8, n_neg, rnnMode="linear", sizeInputSeq=t_len)
    # z rows carry their own flat index so the gathered negatives reveal extIdx exactly
    z = torch.arange(b * t_len, dtype=torch.float32).view(b, t_len, 1).repeat(1, 1, 8)
    torch.manual_seed(seed)
    outs, _ = crit.sampleClean(z, w)
    ext = outs[0][:, 1:, :, 0].reshape(-1).to(torch.int64)       # [b, n_neg, W] flat
    torch.manual_seed(seed)
    n = n_neg * w * b
    batch_idx = torch.randint(0, b, (n,))
    seq_idx = torch.randint(1, t_len, (n,))
    return batch_idx, seq_idx, ext


def g1():
    out = {}
    for tag, (seed, b, t_len, k, nn) in {"tiny": (1234, 2, 8, 4, 3), "mid": (7, 4, 32, 4, 16)}.items():
        bi, si, ext = ref_indices(seed, b, t_len, k, nn)
        out[f"{tag}_cfg"] = np.array([seed, b, t_len, k, nn])
        out[f"{tag}_batchIdx"], out[f"{tag}_seqIdx"], out[f"{tag}_extIdx"] = bi, si, ext
    seed, b, t_len, k, nn = 1234, 64, 128, 12, 128
    bi, si, ext = ref_indices(seed, b, t_len, k, nn)
    out["full_cfg"] = np.array([seed, b, t_len, k, nn])
    out["full_ext_sha256"] = np.array(hashlib.sha256(ext.numpy().astype("<i8").tobytes()).hexdigest())
    out["full_ext_head"], out["full_ext_tail"] = ext[:64], ext[-64:]
    # second call on the SAME generator (stream continuity across steps)
    crit = ref_crit.CPCUnsupersivedCriterion(k, 8, 8, nn, rnnMode="linear", sizeInputSeq=t_len)
    z = torch.arange(b * t_len, dtype=torch.float32).view(b, t_len, 1).repeat(1, 1, 8)
    torch.manual_seed(seed)          # seed AFTER construction: module init draws from the same generator
    crit.sampleClean(z, t_len - k)
    outs, _ = crit.sampleClean(z, t_len - k)
    ext2 = outs[0][:, 1:, :, 0].reshape(-1).to(torch.int64)
    out["full_ext2_sha256"] = np.array(hashlib.sha256(ext2.numpy().astype("<i8").tobytes()).hexdigest())
    out["full_ext2_head"] = ext2[:64]
    save("g1_negidx.npz", **out)


# ------------------------------------------------------------------ G2/G3 encoder, ChannelNorm
def g2():
    hidden = 32
    p = synth.encoder_params(hidden, seed=11)
    enc = ref_model.CPCEncoder(hidden, "layerNorm")
    enc.load_state_dict(strip(p, "gEncoder."))
    x = synth.audio_windows(2, 20480, seed=12)
    gout = synth.features((2, hidden, 128), seed=13)
    # per-layer activations through the reference's own modules
    acts, h = [], x
    for i in range(5):
        h = torch.relu(getattr(enc, f"batchNorm{i}")(getattr(enc, f"conv{i}")(h)))
        acts.append(h)
    out = enc(x)
    assert torch.equal(out, acts[-1])
    (out * gout).sum().backward()
    d = {"hidden": hidden, "param_seed": 11, "x_seed": 12, "gout_seed": 13, "out": out}
    for i, a in enumerate(acts):
        d[f"act{i}_sum"] = a.double().sum()
        d[f"act{i}_abs"] = a.double().abs().sum()
        d[f"act{i}_head"] = a[:, :, :4]
        d[f"act{i}_tail"] = a[:, :, -4:]
    for k, v in enc.named_parameters():
        d["grad." + k] = v.grad
    save("g2_encoder_h32.npz", **d)


def g3():
    rs = np.random.RandomState(3)
    x = torch.from_numpy(rs.standard_normal((3, 16, 50)).astype(np.float32)).requires_grad_(True)
    w = torch.from_numpy((1 + 0.2 * rs.standard_normal((1, 16, 1))).astype(np.float32))
    b = torch.from_numpy((0.2 * rs.standard_normal((1, 16, 1))).astype(np.float32))
    g = torch.from_numpy(rs.standard_normal((3, 16, 50)).astype(np.float32))
    cn = ref_model.ChannelNorm(16)
    cn.weight.data.copy_(w)
    cn.bias.data.copy_(b)
    y = cn(x)
    (y * g).sum().backward()
    save("g3_channelnorm.npz", x=x, w=w, b=b, g=g, y=y, dx=x.grad, dw=cn.weight.grad, db=cn.bias.grad)


# ------------------------------------------------------------------ G4 GRU
def g4():
    d = {}
    for tag, (hin, hid, layers, n, t_len) in {"l1": (32, 32, 1, 3, 20), "l2": (24, 32, 2, 2, 16)}.items():
        p = synth.gru_params(hin, hid, layers, seed=41)
        ar = ref_model.CPCAR(hin, hid, False, layers, mode="GRU", reverse=False)
        ar.load_state_dict(strip(p, "gAR."))
        x = synth.features((n, t_len, hin), seed=42, relu=True).requires_grad_(True)
        g = synth.features((n, t_len, hid), seed=43)
        out = ar(x)
        (out * g).sum().backward()
        d[f"{tag}_cfg"] = np.array([hin, hid, layers, n, t_len])
        d[f"{tag}_out"], d[f"{tag}_dx"] = out, x.grad
        for k, v in ar.named_parameters():
            d[f"{tag}_grad." + k] = v.grad
    # reverse mode (model.py:190-191,205-206)
    p = synth.gru_params(32, 32, 1, seed=41)
    ar = ref_model.CPCAR(32, 32, False, 1, mode="GRU", reverse=True)
    ar.load_state_dict(strip(p, "gAR."))
    x = synth.features((3, 20, 32), seed=42, relu=True)
    d["rev_out"] = ar(x)
    save("g4_gru.npz", **d)


# ------------------------------------------------------------------ G5 criterion
def run_criterion(b, t_len, har, henc, k, nn, seed, pseed, mode=None, n_skipped=0, quality=None,
                  growth=None, infl=None, scale=4.0):
    p = synth.predictor_params(k, har, henc, seed=pseed, scale=scale)
    crit = ref_crit.CPCUnsupersivedCriterion(k, har, henc, nn, mode=mode, rnnMode="linear",
                                             sizeInputSeq=t_len, n_skipped=n_skipped,
                                             growth_rate=growth, inflection_point_x=infl)
    crit.load_state_dict(p)
    c = synth.features((b, t_len, har), seed=pseed + 1).requires_grad_(True)
    z = synth.features((b, t_len, henc), seed=pseed + 2, relu=True).requires_grad_(True)
    torch.manual_seed(seed)
    losses, acc = crit(c, z, None, quality)
    losses.sum().backward()
    grads = {}
    for i in range(k):      # a skipped step receives no gradient at all (None) -> store zeros
        w = crit.wPrediction.predictors[i].weight
        grads[f"dW{i}"] = w.grad if w.grad is not None else torch.zeros_like(w)
    return losses, acc, c.grad, z.grad, grads


def g5():
    d = {}
    base = dict(b=4, t_len=32, har=32, henc=32, k=4, nn=16, seed=99, pseed=50)
    quality = synth.features((4, 12), seed=77)
    variants = {
        "plain": {},
        "skip": {"n_skipped": 1},
        "reverse": {"mode": "reverse"},
        "quality": {"quality": quality, "growth": 2.0, "infl": 0.1},
        "rect": {"har": 24},          # dimOutputEncoder > dimOutputAR
    }
    for tag, kw in variants.items():
        cfg = dict(base, **kw)
        losses, acc, dc, dz, gw = run_criterion(**cfg)
        d[f"{tag}_losses"], d[f"{tag}_acc"], d[f"{tag}_dc"], d[f"{tag}_dz"] = losses, acc, dc, dz
        for k, v in gw.items():
            d[f"{tag}_{k}"] = v
    d["quality_signal"] = quality
    save("g5_criterion_small.npz", **d)

    # full-size shapes of config C2 at b=8 (fits memory); store outputs + gradient digests
    losses, acc, dc, dz, gw = run_criterion(b=8, t_len=128, har=256, henc=256, k=12, nn=128, seed=1234, pseed=60)
    f = {"losses": losses, "acc": acc,
         "dc_sum": dc.double().sum(), "dc_abs": dc.double().abs().sum(), "dc_head": dc[:, :3, :8],
         "dz_sum": dz.double().sum(), "dz_abs": dz.double().abs().sum(), "dz_head": dz[:, :3, :8],
         "dz_tail": dz[:, -3:, :8]}
    for k, v in gw.items():
        f[f"{k}_abs"] = v.double().abs().sum()
        f[f"{k}_head"] = v[:4, :8]
    save("g5_criterion_full.npz", **f)


def g18():
    """The criterion at the FULL batch of BASELINE configs[1] (b = 64, T = 128, H = 256, K = 12, 128 negatives; criterion.py:237-286,
    329-363), predictors at trained scale: the launch geometry the HIP kernels run at in the benchmark.  The reference holds twelve
    [64, 129, 116, 256] candidate tensors for it (~12 GB + autograd); stored: outputs, digests and a strided sample of every gradient
    (tests/test_oracle_golden.py holds oracle.criterion_forward_sparse to them; the GPU test compares every element with that)."""
    cfg = dict(b=64, t_len=128, har=256, henc=256, k=12, nn=128, seed=4321, pseed=160, scale=2.0)
    losses, acc, dc, dz, gw = run_criterion(**cfg)
    f = {"cfg": np.array([cfg[n] for n in ("b", "t_len", "har", "henc", "k", "nn", "seed", "pseed")]), "scale": np.array(cfg["scale"]),
         "losses": losses, "acc": acc}
    for name, ten in (("dc", dc), ("dz", dz)):
        f[f"{name}_sum"], f[f"{name}_abs"] = ten.double().sum(), ten.double().abs().sum()
        f[f"{name}_head"], f[f"{name}_tail"] = ten[:, :3, :8], ten[:, -14:, :8]
        f[f"{name}_sample"] = ten[::9, ::5, ::37]
    for k, v in gw.items():
        f[f"{k}_abs"] = v.double().abs().sum()
        f[f"{k}_sample"] = v[::17, ::13]
    save("g18_criterion_b64.npz", **f)


# ------------------------------------------------------------------ G6 train steps (loss-curve anchor)
def g6():
    hidden, b, k, nn, steps, seed = 64, 4, 12, 128, 20, 1234
    mp = synth.encoder_params(hidden, seed=21)
    mp.update(synth.gru_params(hidden, hidden, 1, seed=22))
    cp = synth.predictor_params(k, hidden, hidden, seed=23)
    enc = ref_model.CPCEncoder(hidden, "layerNorm")
    ar = ref_model.CPCAR(hidden, hidden, False, 1, mode="GRU", reverse=False)
    model = ref_model.CPCModel(enc, ar)
    model.load_state_dict(mp)
    crit = ref_crit.CPCUnsupersivedCriterion(k, hidden, hidden, nn, rnnMode="linear", sizeInputSeq=128)
    crit.load_state_dict(cp)
    # train.py:472-479 -- criterion parameters first, Adam(lr=2e-4, betas=(0.9,0.999), eps=1e-8)
    opt = torch.optim.Adam(list(crit.parameters()) + list(model.parameters()), lr=2e-4,
                           betas=(0.9, 0.999), eps=1e-8)
    x = synth.audio_windows(b, 20480, seed=24)
    label = torch.zeros(b, dtype=torch.long)
    torch.manual_seed(seed)
    model.train()
    crit.train()
    curve, accs = [], []
    for _ in range(steps):
        # train.py:95-113
        past, future = x, x
        combined = torch.cat([past, future], dim=0)
        lab = torch.cat([label, label])
        c_feature, encoded, lab = model(combined, lab)
        c_feature, encoded = c_feature[:b], encoded[b:]
        all_losses, all_acc = crit(c_feature, encoded, lab[:b], None)
        all_losses.sum().backward()
        opt.step()
        opt.zero_grad()
        curve.append(all_losses.detach().clone())
        accs.append(all_acc.detach().clone())
    d = {"cfg": np.array([hidden, b, k, nn, steps, seed]), "curve": torch.cat(curve), "acc": torch.cat(accs)}
    for name, v in list(model.state_dict().items()) + list(crit.state_dict().items()):
        d["final_abs." + name] = v.double().abs().sum()
    d["final.gEncoder.conv0.weight"] = model.state_dict()["gEncoder.conv0.weight"]
    d["final.wPrediction.predictors.0.weight_head"] = crit.state_dict()["wPrediction.predictors.0.weight"][:4, :8]
    save("g6_trainsteps.npz", **d)

    # default-init digests under torch.manual_seed(0) (module construction order of train.py:442-462)
    torch.manual_seed(0)
    enc = ref_model.CPCEncoder(64, "layerNorm")
    ar = ref_model.CPCAR(64, 64, False, 1, mode="GRU", reverse=False)
    model = ref_model.CPCModel(enc, ar)
    crit = ref_crit.CPCUnsupersivedCriterion(12, 64, 64, 128, rnnMode="linear", sizeInputSeq=128)
    i = {}
    for name, v in list(model.state_dict().items()) + list(crit.state_dict().items()):
        i["abs." + name] = v.double().abs().sum()
        i["shape." + name] = np.array(v.shape)
    save("g6_init_seed0_h64.npz", **i)


# ------------------------------------------------------------------ G7 transformer AR
def g7():
    d_model, s, n = 32, 16, 2
    p = synth.transformer_params(d_model, d_model, s, seed=71, dff=2048)
    net = ref_tr.buildTransformerAR(d_model, d_model, 1, s, False)
    sd = strip(p, "gAR.")
    sd.update({k: v for k, v in net.state_dict().items() if k.endswith(".z") or k.endswith(".mask")})
    net.load_state_dict(sd)
    net.eval()
    x = synth.features((n, s, d_model), seed=72, relu=True).requires_grad_(True)
    g = synth.features((n, s, d_model), seed=73)
    out = net(x)
    (out * g).sum().backward()
    d = {"cfg": np.array([d_model, s, n]), "out": out, "dx": x.grad}
    for k, v in net.named_parameters():
        d["grad." + k] = v.grad
    save("g7_transformer.npz", **d)


# ------------------------------------------------------------------ G8 criterion with transformer predictors
def g8():
    """rnnMode='transformer' (the fork's default predictor, criterion.py:136-143), eval mode (dropout off)."""
    b, t_len, h, k, nn, seed = 3, 32, 32, 4, 8, 5
    crit = ref_crit.CPCUnsupersivedCriterion(k, h, h, nn, rnnMode="transformer", sizeInputSeq=t_len)
    sd = crit.state_dict()
    for i in range(k):
        p = synth.transformer_params(h, h, t_len - k, seed=80 + i, prefix=f"wPrediction.predictors.{i}.0.")
        sd.update(p)
    crit.load_state_dict(sd)
    crit.eval()
    c = synth.features((b, t_len, h), seed=90).requires_grad_(True)
    z = synth.features((b, t_len, h), seed=91, relu=True).requires_grad_(True)
    torch.manual_seed(seed)
    losses, acc = crit(c, z, None)
    losses.sum().backward()
    d = {"cfg": np.array([b, t_len, h, k, nn, seed]), "losses": losses, "acc": acc, "dc": c.grad, "dz": z.grad}
    for name, prm in crit.named_parameters():
        if name.startswith("wPrediction.predictors.0.") or name.endswith("Krelpos"):
            d["grad." + name] = prm.grad
    d["param_names"] = np.array(sorted(n for n, _ in crit.named_parameters()))
    save("g8_criterion_transformer_pred.npz", **d)


# ------------------------------------------------------------------ G9 criterion with the multi-head predictor
def g9():
    """--multihead_rnn with rnnMode='transformer' (criterion.py:44-94, transformers.py:137-158,190-212), eval mode."""
    b, t_len, h, k, nn, seed = 3, 32, 32, 4, 8, 6
    crit = ref_crit.CPCUnsupersivedCriterion(k, h, h, nn, rnnMode="transformer", sizeInputSeq=t_len, multihead_rnn=True)
    sd = crit.state_dict()
    sd.update(synth.transformer_params(h, h, t_len - k, seed=95, prefix="wPrediction.predictor.0.", n_classifiers=k))
    crit.load_state_dict(sd)
    crit.eval()
    c = synth.features((b, t_len, h), seed=96).requires_grad_(True)
    z = synth.features((b, t_len, h), seed=97, relu=True).requires_grad_(True)
    torch.manual_seed(seed)
    losses, acc = crit(c, z, None)
    losses.sum().backward()
    d = {"cfg": np.array([b, t_len, h, k, nn, seed]), "losses": losses, "acc": acc, "dc": c.grad, "dz": z.grad}
    for name, prm in crit.named_parameters():
        d["grad." + name] = prm.grad
    d["param_names"] = np.array(sorted(n for n, _ in crit.named_parameters()))
    d["param_shapes"] = np.array([str(tuple(v.shape)) for _, v in sorted(crit.state_dict().items())])
    save("g9_criterion_multihead_pred.npz", **d)


# ------------------------------------------------------------------ G10 transformer variants
def g10():
    """abspos=True (StaticPositionEmbedding, no Krelpos) and an input whose length is not a multiple of sizeSeq."""
    d_model, size_seq, n = 32, 16, 2
    d = {"cfg": np.array([d_model, size_seq, n])}
    for tag, abspos, s_len in (("abspos", True, 16), ("ragged", False, 41), ("abspos_ragged", True, 11)):
        p = synth.transformer_params(d_model, d_model, size_seq, seed=171, dff=2048)
        net = ref_tr.buildTransformerAR(d_model, d_model, 1, size_seq, abspos)
        layer = "1." if abspos else "0."
        sd = {layer + k[len("gAR.0."):]: v for k, v in p.items() if not (abspos and k.endswith("Krelpos"))}
        sd.update({k: v for k, v in net.state_dict().items() if k.endswith(".z") or k.endswith(".mask") or k.endswith(".pe")})
        net.load_state_dict(sd)
        net.eval()
        x = synth.features((n, s_len, d_model), seed=172, relu=True).requires_grad_(True)
        g = synth.features((n, s_len, d_model), seed=173)
        out = net(x)
        (out * g).sum().backward()
        d[tag + "_len"] = np.array(s_len)
        d[tag + "_out"] = out
        d[tag + "_dx"] = x.grad
        d[tag + "_keys"] = np.array(sorted(net.state_dict().keys()))
        for k, v in net.named_parameters():
            d[tag + "_grad." + k] = v.grad
    save("g10_transformer_variants.npz", **d)


# ------------------------------------------------------------------ G11 CPCModel with span masking
def g11():
    """mask_prob > 0 (model.py:300-390): spans from numpy's global generator, frames overwritten by mask_emb."""
    hidden, n = 32, 3
    p = synth.encoder_params(hidden, seed=11)
    p.update(synth.gru_params(hidden, hidden, 1, seed=41))
    d = {"cfg": np.array([hidden, n])}
    # the span sampler alone, several shapes, one generator stream
    np.random.seed(123)
    probe = ref_model.CPCModel(ref_model.CPCEncoder(hidden, "layerNorm"), ref_model.CPCAR(hidden, hidden, False, 1, mode="GRU"))
    for i, (bsz, frames, prob, length, mn) in enumerate([(4, 128, 0.01, 10, 2), (3, 50, 0.05, 3, 0), (2, 20, 0.02, 8, 2), (5, 128, 0.065, 10, 2)]):
        d[f"mask{i}_cfg"] = np.array([bsz, frames, prob, length, mn])
        d[f"mask{i}"] = probe.compute_mask_indices((bsz, frames), prob, length, min_masks=mn)
    d["rand_after"] = np.random.rand()
    # the model
    model = ref_model.CPCModel(ref_model.CPCEncoder(hidden, "layerNorm"), ref_model.CPCAR(hidden, hidden, False, 1, mode="GRU"),
                               mask_prob=0.02, mask_length=4)
    sd = dict(p)
    sd["mask_emb"] = torch.from_numpy(np.random.RandomState(77).uniform(size=hidden).astype(np.float32))
    model.load_state_dict(sd)
    x = synth.audio_windows(n, 20480, seed=12)
    np.random.seed(5)
    with torch.no_grad():      # the reference's own backward fails here (in-place write into the ReLU output it needs)
        c, z, _ = model(x, None)
    d.update({"c": c, "z": z, "mask_emb": sd["mask_emb"]})
    save("g11_model_span_mask.npz", **d)


# ------------------------------------------------------------------ G12 CPCAR LSTM / RNN
def g12():
    """CPCAR(mode="LSTM") -- this fork's default arMode -- and mode="RNN" (model.py:171-176), forward + backward,
    plus keepHidden carry-over across two calls (model.py:197-201)."""
    d = {}
    for mode, make in (("LSTM", synth.lstm_params), ("RNN", synth.rnn_params)):
        for tag, (hin, hid, layers, n, t_len) in {"l1": (32, 32, 1, 3, 20), "l2": (24, 32, 2, 2, 16)}.items():
            p = make(hin, hid, layers, seed=51)
            ar = ref_model.CPCAR(hin, hid, False, layers, mode=mode, reverse=False)
            ar.load_state_dict(strip(p, "gAR."))
            x = synth.features((n, t_len, hin), seed=52, relu=True).requires_grad_(True)
            g = synth.features((n, t_len, hid), seed=53)
            out = ar(x)
            (out * g).sum().backward()
            key = f"{mode}_{tag}"
            d[f"{key}_cfg"] = np.array([hin, hid, layers, n, t_len])
            d[f"{key}_out"], d[f"{key}_dx"] = out, x.grad
            for k, v in ar.named_parameters():
                d[f"{key}_grad." + k] = v.grad
        # keepHidden: the second call starts from the first call's final state
        p = make(32, 32, 2, seed=51)
        ar = ref_model.CPCAR(32, 32, True, 2, mode=mode, reverse=False)
        ar.load_state_dict(strip(p, "gAR."))
        xa = synth.features((2, 9, 32), seed=54, relu=True)
        xb = synth.features((2, 7, 32), seed=55, relu=True)
        ar(xa)
        d[f"{mode}_keep_out2"] = ar(xb)
        hid_state = ar.hidden
        if isinstance(hid_state, tuple):
            d[f"{mode}_keep_h"], d[f"{mode}_keep_c"] = hid_state
        else:
            d[f"{mode}_keep_h"] = hid_state
    save("g12_lstm_rnn.npz", **d)


# ------------------------------------------------------------------ G13 learning-rate schedules
def g13():
    """train.py:501-520 with the reference's own utils: learning rate after each of 14 epochs for the three scheduler
    set-ups (step only, ramp only, ramp + step through SchedulerCombiner) and a resumed run (5 epochs done)."""
    import cpc.utils.misc as ref_utils

    def make(step, ramp, done):
        w = torch.nn.Parameter(torch.zeros(3))
        opt = torch.optim.Adam([w], lr=2e-4)
        sched = None
        if step > 0:
            sched = torch.optim.lr_scheduler.StepLR(opt, step, gamma=0.5)
        if ramp is not None:
            r = torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda=lambda e: ref_utils.ramp_scheduling_function(ramp, e),
                                                  last_epoch=-1)
            sched = r if sched is None else ref_utils.SchedulerCombiner([r, sched], [0, ramp])
        for _ in range(done):
            sched.step()
        lrs = []
        for _ in range(14):
            lrs.append(opt.param_groups[0]["lr"])
            opt.step()
            sched.step()
        return np.array(lrs)

    save("g13_lr_schedules.npz", step3=make(3, None, 0), ramp4=make(-1, 4, 0), ramp4_step3=make(3, 4, 0),
         ramp4_step3_resumed5=make(3, 4, 5))


# ------------------------------------------------------------------ G14 recurrent predictors
def g14():
    """rnnMode='LSTM' (nn.LSTM, batch_first) and rnnMode='RNN' (nn.RNN -- NOT batch_first: the recurrence then runs
    over the batch axis of c, criterion.py:115-123,163) as predictors; forward + backward."""
    d = {}
    b, t_len, har, henc, k, nn, seed = 4, 32, 24, 32, 4, 8, 7
    for mode, gates in (("LSTM", 4), ("RNN", 1)):
        crit = ref_crit.CPCUnsupersivedCriterion(k, har, henc, nn, rnnMode=mode, sizeInputSeq=t_len)
        sd = {}
        for i in range(k):
            sd.update(synth.gru_params(har, henc, 1, seed=110 + i, prefix=f"wPrediction.predictors.{i}.", gates=gates))
        crit.load_state_dict(sd)
        c = synth.features((b, t_len, har), seed=120).requires_grad_(True)
        z = synth.features((b, t_len, henc), seed=121, relu=True).requires_grad_(True)
        torch.manual_seed(seed)
        losses, acc = crit(c, z, None)
        losses.sum().backward()
        d[f"{mode}_losses"], d[f"{mode}_acc"], d[f"{mode}_dc"], d[f"{mode}_dz"] = losses, acc, c.grad, z.grad
        for name, prm in crit.named_parameters():
            d[f"{mode}_grad." + name] = prm.grad
    d["cfg"] = np.array([b, t_len, har, henc, k, nn, seed])
    save("g14_criterion_recurrent_pred.npz", **d)


# ------------------------------------------------------------------ G15 bidirectional context networks
def g15():
    """BiDIRARTangled (cpc_mode='bert', model.py:219-241) and BiDIRAR (model.py:244-272), forward + backward, with the
    modules' default initialisation under a fixed torch seed (the state dict is stored)."""
    d = {}
    hin, hout, layers, n, t_len = 24, 32, 2, 3, 20
    for name, cls in (("tangled", ref_model.BiDIRARTangled), ("bidir", ref_model.BiDIRAR)):
        torch.manual_seed(31)
        net = cls(hin, hout, layers)
        x = synth.features((n, t_len, hin), seed=131, relu=True).requires_grad_(True)
        g = synth.features((n, t_len, hout), seed=132)
        out = net(x)
        (out * g).sum().backward()
        d[f"{name}_out"], d[f"{name}_dx"] = out, x.grad
        for k, v in net.state_dict().items():
            d[f"{name}_param." + k] = v
        for k, v in net.named_parameters():
            d[f"{name}_grad." + k] = v.grad
    d["cfg"] = np.array([hin, hout, layers, n, t_len])
    save("g15_bidirectional_ar.npz", **d)


# ------------------------------------------------------------------ G16 criterion, inference-side API
def g16():
    """getPrediction / getCosineDistances / sampleClean (criterion.py:237-327) on the g5 'plain' / 'reverse' setting."""
    b, t_len, har, henc, k, nn, seed, pseed = 4, 32, 32, 32, 4, 16, 99, 50
    d = {}
    for tag, mode in (("plain", None), ("reverse", "reverse")):
        crit = ref_crit.CPCUnsupersivedCriterion(k, har, henc, nn, mode=mode, rnnMode="linear", sizeInputSeq=t_len)
        crit.load_state_dict(synth.predictor_params(k, har, henc, seed=pseed, scale=4.0))
        c = synth.features((b, t_len, har), seed=pseed + 1)
        z = synth.features((b, t_len, henc), seed=pseed + 2, relu=True)
        with torch.no_grad():
            torch.manual_seed(seed)
            preds, label = crit.getPrediction(c, z, None)
            after_pred = torch.randint(0, 1000, (4,))            # where the generator stands after the call
            cos = crit.getCosineDistances(c, z)
            torch.manual_seed(seed)
            cands, label2 = crit.sampleClean(z, t_len - k)
        d[f"{tag}_pred"] = torch.stack(preds)                       # [K, b, 1 + nn, W]
        d[f"{tag}_label"] = label
        d[f"{tag}_next_draws"] = after_pred
        d[f"{tag}_cos"] = torch.stack(cos)                          # [K, b, 1, W]
        d[f"{tag}_cand_first"], d[f"{tag}_cand_last"] = cands[0], cands[-1]      # [b, 1 + nn, W, henc]
        d[f"{tag}_cand_label"] = label2
    save("g16_criterion_inference.npz", **d)


# ------------------------------------------------------------------ G17 a checkpoint WRITTEN by the reference
def g17():
    """A run directory as the reference leaves it (checkpoint_N.pt in its {"gEncoder", "cpcCriterion", "optimizer", "best"}
    layout, checkpoint_args.json, checkpoint_logs.json) + what its model computes on a fixed input."""
    import argparse
    import json
    from cpc.cpc_default_config import set_default_cpc_config
    parser = set_default_cpc_config(argparse.ArgumentParser())
    args = parser.parse_args([])
    args.hiddenEncoder, args.hiddenGar, args.arMode, args.nLevelsGRU = 32, 32, "GRU", 1
    args.negativeSamplingExt, args.nPredicts, args.rnnMode, args.sizeWindow = 16, 4, "linear", 20480
    torch.manual_seed(4242)
    enc = ref_model.CPCEncoder(args.hiddenEncoder, args.normMode)
    ar = ref_model.CPCAR(args.hiddenEncoder, args.hiddenGar, False, args.nLevelsGRU, mode=args.arMode)
    model = ref_model.CPCModel(enc, ar)
    crit = ref_crit.CPCUnsupersivedCriterion(args.nPredicts, args.hiddenGar, args.hiddenEncoder, args.negativeSamplingExt,
                                             rnnMode=args.rnnMode, sizeInputSeq=args.sizeWindow // 160)
    opt = torch.optim.Adam(list(crit.parameters()) + list(model.parameters()), lr=2e-4)
    x = synth.audio_windows(2, 20480, 5)
    with torch.no_grad():
        c, z, _ = model(x, None)
    out = os.path.join(OUT, "ref_checkpoint")
    os.makedirs(out, exist_ok=True)
    torch.save({"gEncoder": model.state_dict(), "cpcCriterion": crit.state_dict(), "optimizer": opt.state_dict(),
                "best": model.state_dict()}, os.path.join(out, "checkpoint_7.pt"))
    with open(os.path.join(out, "checkpoint_args.json"), "w") as f:
        json.dump(vars(args), f, indent=2)
    with open(os.path.join(out, "checkpoint_logs.json"), "w") as f:
        json.dump({"epoch": [7], "locLoss_train": [[4.8] * args.nPredicts]}, f)
    save("g17_ref_checkpoint_outputs.npz", x=x, c=c, z=z)


if __name__ == "__main__":
    which = sys.argv[1:] or ["g1", "g2", "g3", "g4", "g5", "g6", "g7", "g8", "g9", "g10", "g11", "g12", "g13", "g14", "g15", "g16", "g17", "g18"]
    for name in which:
        globals()[name]()
