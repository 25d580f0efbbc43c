"""Does any kernel of a training step read memory it (or a predecessor) did not write?  The caching allocator's free blocks are
filled with NaN before every step; a step that reads uninitialised scratch then turns NaN (or changes)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import cpc2_amd
from cpc2_amd.train import buildOptimizer, cpcStep
from oracle import synth
DEV = torch.device("cuda:0")
HIDDEN = int(sys.argv[1]) if len(sys.argv) > 1 else 64
B, K, NNEG = 2, 12, 16
def build():
    mp = synth.encoder_params(HIDDEN, 21); mp.update(synth.gru_params(HIDDEN, HIDDEN, 1, 22))
    model = cpc2_amd.CPCModel(cpc2_amd.CPCEncoder(HIDDEN), cpc2_amd.CPCAR(HIDDEN, HIDDEN, False, 1)); model.load_state_dict(mp)
    crit = cpc2_amd.CPCUnsupersivedCriterion(K, HIDDEN, HIDDEN, NNEG, rnnMode="linear", sizeInputSeq=128)
    crit.load_state_dict(synth.predictor_params(K, HIDDEN, HIDDEN, 23))
    model, crit = model.to(DEV), crit.to(DEV)
    return model, crit, buildOptimizer(model, crit, lr=1e-3)
def poison(value):
    blocks = [torch.full((n,), value, device=DEV) for n in (1 << 26, 1 << 24, 1 << 24, 1 << 22, 1 << 22, 1 << 20, 1 << 20, 1 << 18, 1 << 16)]
    torch.cuda.synchronize(); del blocks
def run(value):
    torch.cuda.empty_cache()
    model, crit, opt = build()
    crit.seed(1234)
    x = synth.audio_windows(B, 20480, 100).to(DEV); label = torch.zeros(B, dtype=torch.long, device=DEV)
    grads = []
    for _ in range(2):
        if value is not None: poison(value)
        tot, ls, _ = cpcStep(x, x, label, model, crit); tot.backward()
        grads.append(opt.flat_grad.clone()); opt.step(); opt.zero_grad()
    torch.cuda.synchronize()
    return opt.flat.clone(), grads
ref, gref = run(None)
for val in (float("nan"), 1e30, 0.0):
    got, g = run(val)
    bad = ~torch.isfinite(got)
    d = (got - ref).abs()
    gd = (g[0] - gref[0]).abs()
    nz = torch.nonzero((gd > 0) | ~torch.isfinite(g[0])).view(-1)
    print(f"poison {val}: non-finite params {int(bad.sum())}; max |diff| {float(torch.nan_to_num(d, nan=1e9).max()):.3e}; step-1 gradient elements that differ {nz.numel()}"
          + (f" in [{int(nz.min())}, {int(nz.max())}]" if nz.numel() else ""))
