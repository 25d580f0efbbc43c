"""Many optimisation steps on REAL audio, HIP path against the CPU oracle on identical batches and negative streams:
the reference's cpc/test_data fixture, CPC-small, deterministic batches (sorted files, sequential windows), constant
learning rate.  Prints the epoch means of both and the largest relative difference per epoch.
   python tools/fixture_parity.py [epochs] [lr] [default]      ("default": torch's default initialisation under seed 0
                                                                 instead of the synthetic parameters)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import cpc2_amd
from cpc2_amd.dataset import AudioBatchData, filterSeqs, findAllSeqs
from cpc2_amd.train import buildOptimizer, cpcStep
from oracle import cpc_oracle as O, synth
from oracle.mt19937 import MT19937

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DB = os.path.join(ROOT, "tests", "golden", "test_db")
epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 8
lr = float(sys.argv[2]) if len(sys.argv) > 2 else 2e-4
hidden, k, nn, b = 256, 12, 128, 8
dev = torch.device("cuda:0")
seqs, speakers = findAllSeqs(DB, extension=".flac")
seqs = sorted(filterSeqs(os.path.join(ROOT, "tests", "golden", "seq_list.txt"), seqs), key=lambda s: s[1])
data = AudioBatchData(DB, 20480, seqs, None, len(speakers), device=dev)
batches = [seq[:, 0].contiguous() for seq, _ in data.getDataLoader(b, "sequential", False)]
print(f"{len(batches)} batches of {b} windows per epoch, lr {lr}")

torch.manual_seed(0)
model = cpc2_amd.CPCModel(cpc2_amd.CPCEncoder(hidden), cpc2_amd.CPCAR(hidden, hidden, False, 1))
crit = cpc2_amd.CPCUnsupersivedCriterion(k, hidden, hidden, nn, rnnMode="linear", sizeInputSeq=128)
if len(sys.argv) > 3 and sys.argv[3] == "default":
    mp = {n: v.detach().clone() for n, v in model.state_dict().items()}
    cp = {n: v.detach().clone() for n, v in crit.state_dict().items()}
else:
    mp = synth.encoder_params(hidden, 21)
    mp.update(synth.gru_params(hidden, hidden, 1, 22))
    cp = synth.predictor_params(k, hidden, hidden, 23)
    model.load_state_dict(mp)
    crit.load_state_dict(cp)
model, crit = model.to(dev), crit.to(dev)
opt = buildOptimizer(model, crit, lr=lr)
crit.seed(99)

torch.set_num_threads(min(16, os.cpu_count() or 1))
params = {n: v.clone().requires_grad_(True) for n, v in list(cp.items()) + list(mp.items())}
adam = O.Adam({n: v.data for n, v in params.items()}, lr=lr)
mt = MT19937(99)

for epoch in range(epochs):
    t0 = time.time()
    got, ref, worst = [], [], 0.0
    for x in batches:
        label = torch.zeros(x.shape[0], dtype=torch.long, device=dev)
        tot, losses, _ = cpcStep(x, x, label, model, crit, dedup=True)
        tot.backward()
        opt.step()
        opt.zero_grad()
        xc = x.cpu()
        rtot, rlosses, _ = O.train_step_loss(xc, xc, {n: params[n] for n in mp}, {n: params[n] for n in cp}, mt, k, nn)
        grads = torch.autograd.grad(rtot, list(params.values()))
        adam.step(dict(zip(params, grads)))
        g, r = losses.detach().cpu().view(-1), rlosses.detach().view(-1)
        worst = max(worst, float(((g - r).abs() / r.abs()).max()))
        got.append(float(g.mean()))
        ref.append(float(r.mean()))
    print(f"epoch {epoch:2d}: HIP {np.mean(got):.5f}  oracle {np.mean(ref):.5f}  worst relative difference of a step's losses {worst:.2e}"
          f"   ({time.time() - t0:.0f} s)", flush=True)
