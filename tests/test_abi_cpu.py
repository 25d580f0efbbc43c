"""CPU-only checks of the boundary: the C-ABI library loads and exports every symbol that
include/cpc2_hip.h declares, the host-side negative-index sampler is bit-exact against the
reference golden vectors, the module classes keep the reference's state-dict contract, and the
product refuses to run without a GPU (no CPU fallback)."""
import ctypes
import hashlib
import os
import re

import numpy as np
import pytest
import torch

import cpc2_amd
from cpc2_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "cpc2_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(cpc_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 25
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in cpc2_hip.h but not exported"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    assert _lib.load().cpc_version() >= 100


def test_shape_queries_and_error_reporting():
    lib = _lib.load()
    assert lib.cpc_encoder_frames(20480) == 128
    assert lib.cpc_encoder_frames(64000) == 400
    assert lib.cpc_encoder_saved_bytes(4, 20480, 256) > 0
    assert lib.cpc_encoder_saved_bytes(4, 20480, 100) == 0          # unsupported hidden size
    assert b"not supported" in lib.cpc_last_error()
    assert lib.cpc_infonce_saved_bytes(8, 128, 12, 256, 256, 128) > 0
    assert lib.cpc_infonce_saved_bytes(8, 128, 17, 256, 256, 128) == 0


@pytest.mark.parametrize("tag", ["tiny", "mid"])
def test_sampler_bit_exact_vs_reference(golden, tag):
    g = golden("g1_negidx.npz")
    seed, b, t_len, k, nn = (int(v) for v in g[f"{tag}_cfg"])
    s = cpc2_amd.criterion.NegativeSampler()
    s.seed(seed)
    ext, bi, si = s.sample_host(b, t_len, t_len - k, nn, want_parts=True)
    assert np.array_equal(ext.numpy().astype(np.int64), g[f"{tag}_extIdx"])
    assert np.array_equal(bi.numpy(), g[f"{tag}_batchIdx"])
    assert np.array_equal(si.numpy(), g[f"{tag}_seqIdx"])


def test_sampler_full_size_follows_torch_generator(golden):
    g = golden("g1_negidx.npz")
    seed, b, t_len, k, nn = (int(v) for v in g["full_cfg"])
    s = cpc2_amd.criterion.NegativeSampler()          # default: consumes torch's global CPU generator
    torch.manual_seed(seed)
    ext = s.sample_host(b, t_len, t_len - k, nn).numpy().astype("<i8")
    assert hashlib.sha256(ext.tobytes()).hexdigest() == str(g["full_ext_sha256"])
    ext2 = s.sample_host(b, t_len, t_len - k, nn).numpy().astype("<i8")     # next step, same stream
    assert hashlib.sha256(ext2.tobytes()).hexdigest() == str(g["full_ext2_sha256"])
    # the global generator was advanced by exactly 4 * n draws
    after = torch.randint(0, 1000, (8,))
    torch.manual_seed(seed)
    n = b * nn * (t_len - k)
    for _ in range(2):
        torch.randint(0, b, (n,))
        torch.randint(1, t_len, (n,))
    assert torch.equal(after, torch.randint(0, 1000, (8,)))


def test_state_dict_contract_and_default_init(golden):
    g = golden("g6_init_seed0_h64.npz")
    torch.manual_seed(0)
    enc = cpc2_amd.CPCEncoder(64, "layerNorm")
    ar = cpc2_amd.CPCAR(64, 64, False, 1, mode="GRU", reverse=False)
    model = cpc2_amd.CPCModel(enc, ar)
    crit = cpc2_amd.CPCUnsupersivedCriterion(12, 64, 64, 128, rnnMode="linear", sizeInputSeq=128)
    sd = dict(model.state_dict())
    sd.update(crit.state_dict())
    ref_names = {k[len("abs."):] for k in g.files if k.startswith("abs.")}
    assert set(sd) == ref_names
    for name, v in sd.items():
        assert tuple(v.shape) == tuple(int(x) for x in g["shape." + name]), name
        ref = float(g["abs." + name])
        assert abs(float(v.double().abs().sum()) - ref) <= 1e-6 * max(ref, 1.0), name
    assert enc.DOWNSAMPLING == 160 and enc.dimEncoded == 64 and enc.getDimOutput() == 64 and ar.getDimOutput() == 64


def test_transformer_state_dict_contract(golden):
    from cpc2_amd.transformers import buildTransformerAR
    g = golden("g7_transformer.npz")
    net = buildTransformerAR(32, 32, 1, 16, False)
    ref_params = {k[len("grad."):] for k in g.files if k.startswith("grad.")}
    assert {k for k, _ in net.named_parameters()} == ref_params
    for name, p in net.named_parameters():
        assert tuple(p.shape) == tuple(g["grad." + name].shape), name
    assert {"0.multihead.Att.z", "0.multihead.Att.mask"} <= set(net.state_dict())
    assert tuple(net.state_dict()["0.multihead.Att.mask"].shape) == (1, 16, 16)


def test_constructor_errors_mirror_reference():
    with pytest.raises(ValueError):
        cpc2_amd.CPCEncoder(64, "nope")
    with pytest.raises(ValueError):
        cpc2_amd.CPCUnsupersivedCriterion(12, 64, 64, 128, mode="sideways", rnnMode="linear")


def test_no_cpu_fallback():
    model = cpc2_amd.CPCModel(cpc2_amd.CPCEncoder(32), cpc2_amd.CPCAR(32, 32, False, 1))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        model(torch.zeros(2, 1, 20480), None)
    crit = cpc2_amd.CPCUnsupersivedCriterion(4, 32, 32, 16, rnnMode="linear", sizeInputSeq=32)
    with pytest.raises(RuntimeError):
        crit(torch.zeros(2, 32, 32), torch.zeros(2, 32, 32), None)


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "cpc2_amd")
    for dirpath, _d, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("no oracle", ""), f"{f} mentions the oracle"


# ----------------------------------------------------------------------------- factories (reference: cpc/unit_tests.py:279-348)
def _default_args(**kw):
    """The reference's architecture defaults (cpc_default_config.py:18-78) for the flags the factories read."""
    import types
    args = types.SimpleNamespace(hiddenEncoder=256, hiddenGar=256, nPredicts=12, negativeSamplingExt=128, sizeWindow=20480,
                                 samplingType="samespeaker", cpc_mode=None, encoder_type="cpc", normMode="layerNorm",
                                 arMode="GRU", nLevelsGRU=1, rnnMode="linear", dropout=False, abspos=False)
    for k, v in kw.items():
        setattr(args, k, v)
    return args


def test_build_cpc_encoder_like_reference_TestEncoderBuilder():
    from cpc2_amd.train import getEncoder
    enc = getEncoder(_default_args())
    assert isinstance(enc, cpc2_amd.CPCEncoder) and enc.dimEncoded == 256 and enc.DOWNSAMPLING == 160
    for other in ("mfcc", "lfb"):                      # not on the hot path: refused, never silently replaced
        with pytest.raises(NotImplementedError):
            getEncoder(_default_args(encoder_type=other))


def test_build_ar_like_reference_TestARBuilder():
    import torch
    from cpc2_amd.train import getAR
    ar = getAR(_default_args(arMode="GRU"))
    assert isinstance(ar, cpc2_amd.CPCAR) and isinstance(ar.baseNet, torch.nn.GRU) and ar.getDimOutput() == 256
    assert getAR(_default_args(arMode="GRU", samplingType="sequential")).keepHidden
    assert getAR(_default_args(arMode="GRU", cpc_mode="reverse")).reverse
    args = _default_args(arMode="transformer", hiddenGar=256)
    tr = getAR(args)
    assert isinstance(tr, torch.nn.Sequential) and len(tr) == 1 and tr[0].sizeSeq == 128 and args.hiddenGar == 256
    lstm = getAR(_default_args(arMode="LSTM", nLevelsGRU=2))                 # the fork's default arMode
    assert isinstance(lstm.baseNet, torch.nn.LSTM) and lstm.baseNet.num_layers == 2 and lstm.getDimOutput() == 256
    assert isinstance(getAR(_default_args(arMode="RNN")).baseNet, torch.nn.RNN)
    from cpc2_amd.model import BiDIRARTangled, NoAr
    bert = getAR(_default_args(cpc_mode="bert", nLevelsGRU=2))
    assert isinstance(bert, BiDIRARTangled) and bert.ARNet.bidirectional and bert.getDimOutput() == 256
    assert isinstance(getAR(_default_args(arMode="no_ar")), NoAr)


def test_build_criterion_variants():
    from cpc2_amd.train import getCriterion
    from cpc2_amd.criterion import MultiHeadPredictionNetwork, NoneCriterion, PredictionNetwork
    crit = getCriterion(_default_args(), 160)
    assert isinstance(crit.wPrediction, PredictionNetwork) and len(crit.wPrediction.predictors) == 12
    assert crit.wPrediction.predictors[0].weight.shape == (256, 256)
    tr = getCriterion(_default_args(rnnMode="transformer"), 160)
    assert tr.wPrediction.predictors[0][0].sizeSeq == 116
    mh = getCriterion(_default_args(rnnMode="transformer", multihead_rnn=True), 160)
    assert isinstance(mh.wPrediction, MultiHeadPredictionNetwork) and mh.wPrediction.predictor[0].nclassifiers == 12
    assert isinstance(getCriterion(_default_args(cpc_mode="none"), 160), NoneCriterion)
    with pytest.raises(ValueError):
        getCriterion(_default_args(cpc_mode="bert"), 160)


def test_span_mask_draws_like_the_reference(golden):
    """model.py:300-365: same spans for the same numpy seed, and the generator is left in the same state."""
    import numpy as np
    from cpc2_amd.model import span_mask
    g = golden("g11_model_span_mask.npz")
    np.random.seed(123)
    for i in range(4):
        bsz, frames, prob, length, mn = g[f"mask{i}_cfg"]
        m = span_mask(int(bsz), int(frames), float(prob), int(length), min_masks=int(mn))
        assert m.dtype == bool and np.array_equal(m, g[f"mask{i}"]), i
    assert np.random.rand() == float(g["rand_after"])


# ----------------------------------------------------------------------------- learning-rate schedules (train.py:501-520)
@pytest.mark.parametrize("tag,step,ramp,done", [("step3", 3, None, 0), ("ramp4", -1, 4, 0), ("ramp4_step3", 3, 4, 0),
                                               ("ramp4_step3_resumed5", 3, 4, 5)])
def test_lr_schedules_follow_the_reference(tag, step, ramp, done):
    import numpy as np
    import torch
    from cpc2_amd.train import buildScheduler
    ref = np.load(os.path.join(ROOT, "tests", "golden", "g13_lr_schedules.npz"))[tag]
    opt = torch.optim.Adam([torch.nn.Parameter(torch.zeros(3))], lr=2e-4)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")                 # the reference steps the scheduler before the optimiser when resuming
        sched = buildScheduler(opt, step, ramp, done)
        lrs = []
        for _ in range(len(ref)):
            lrs.append(opt.param_groups[0]["lr"])
            opt.step()
            sched.step()
    assert np.allclose(lrs, ref, rtol=1e-12, atol=0)


# ----------------------------------------------------------------------------- sampler: random shapes (criterion.py:247-266)
def test_sampler_matches_torch_randint_on_random_shapes():
    """Property test over ragged shapes: the native MT19937 sampler reproduces, bit for bit, the index arithmetic of
    sampleClean run on torch's own CPU generator -- and the oracle's restatement does too -- for any (b, T, K, Nneg),
    both index layouts, two consecutive steps on one stream."""
    import numpy as np
    import torch
    from hypothesis import given, settings, strategies as st
    from oracle.mt19937 import MT19937, negative_indices

    @settings(max_examples=40, deadline=None)
    @given(b=st.integers(1, 9), t_len=st.integers(3, 70), k_frac=st.floats(0.05, 0.9), nn=st.integers(1, 17),
           seed=st.integers(0, 2 ** 31 - 1))
    def check(b, t_len, k_frac, nn, seed):
        k = min(t_len - 2, max(1, int(k_frac * t_len)))
        w = t_len - k
        n = b * nn * w
        s = cpc2_amd.criterion.NegativeSampler()
        s.seed(seed)
        mt = MT19937(seed)
        torch.manual_seed(seed)
        for _step in range(2):
            batch_idx = torch.randint(low=0, high=b, size=(n,))                       # criterion.py:247-251
            seq_idx = torch.randint(low=1, high=t_len, size=(n,))                     # :254-256
            base = torch.arange(0, w).expand(b, nn, w).contiguous().view(-1)          # :258-262
            ref = torch.remainder(seq_idx + base, t_len) + batch_idx * t_len          # :264-266
            ext, bi, si = s.sample_host(b, t_len, w, nn, want_parts=True, time_major=False)
            assert np.array_equal(bi.numpy(), batch_idx.numpy()) and np.array_equal(si.numpy(), seq_idx.numpy())
            assert np.array_equal(ext.numpy().astype(np.int64), ref.numpy())
            o_bi, o_si, o_ext = negative_indices(mt, b, t_len, w, nn)
            assert np.array_equal(np.asarray(o_ext, dtype=np.int64), ref.numpy())
        # the kernels' time-major layout holds the same indices, negatives of one (b, t) contiguous
        s.seed(seed)
        tm = s.sample_host(b, t_len, w, nn, time_major=True).numpy().astype(np.int64)
        torch.manual_seed(seed)
        batch_idx = torch.randint(low=0, high=b, size=(n,))
        seq_idx = torch.randint(low=1, high=t_len, size=(n,))
        base = torch.arange(0, w).expand(b, nn, w).contiguous().view(-1)
        ref = (torch.remainder(seq_idx + base, t_len) + batch_idx * t_len).view(b, nn, w)
        assert np.array_equal(tm.reshape(b, w, nn), ref.permute(0, 2, 1).numpy())

    check()
