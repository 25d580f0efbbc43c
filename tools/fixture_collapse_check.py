"""Does the oracle follow the HIP path through the loss spike of tools/train_fixture.py (epoch 3, when the ramp reaches the
full learning rate)?  Same initialisation (torch default, seed 0), same uniform sampler with random offsets, same ramp,
identical batches and negative streams for both.   python tools/fixture_collapse_check.py [epochs]"""
import os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import cpc2_amd
from cpc2_amd.dataset import AudioBatchData, filterSeqs, findAllSeqs
from cpc2_amd.train import buildOptimizer, cpcStep
from oracle import cpc_oracle as O
from oracle.mt19937 import MT19937

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DB = os.path.join(ROOT, "tests", "golden", "test_db")
epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 6
hidden, k, nn, b, lr, ramp = 256, 12, 128, 8, 2e-4, 3
dev = torch.device("cuda:0")
torch.manual_seed(0); random.seed(0); np.random.seed(0)
seqs, speakers = findAllSeqs(DB, extension=".flac")
seqs = sorted(filterSeqs(os.path.join(ROOT, "tests", "golden", "seq_list.txt"), seqs), key=lambda s: s[1])
data = AudioBatchData(DB, 20480, seqs, None, len(speakers), device=dev)
model = cpc2_amd.CPCModel(cpc2_amd.CPCEncoder(hidden), cpc2_amd.CPCAR(hidden, hidden, False, 1))
crit = cpc2_amd.CPCUnsupersivedCriterion(k, hidden, hidden, nn, rnnMode="linear", sizeInputSeq=128)
mp = {n: v.detach().clone() for n, v in model.state_dict().items()}
cp = {n: v.detach().clone() for n, v in crit.state_dict().items()}
model, crit = model.to(dev), crit.to(dev)
opt = buildOptimizer(model, crit, lr=lr)
crit.seed(7)
torch.set_num_threads(min(16, os.cpu_count() or 1))
params = {n: v.clone().requires_grad_(True) for n, v in list(cp.items()) + list(mp.items())}
adam = O.Adam({n: v.data for n, v in params.items()}, lr=lr)
mt = MT19937(7)
for epoch in range(epochs):
    rate = lr * min(1.0, (epoch + 1) / ramp)
    opt.param_groups[0]["lr"] = rate
    adam.lr = rate
    got, ref, worst = [], [], 0.0
    t0 = time.time()
    for seq, label in data.getDataLoader(b, "uniform", True):
        past, future = seq[:, 0].contiguous(), seq[:, 1].contiguous()
        tot, losses, _ = cpcStep(past, future, label, model, crit)
        tot.backward(); opt.step(); opt.zero_grad()
        rtot, rlosses, _ = O.train_step_loss(past.cpu(), future.cpu(), {n: params[n] for n in mp}, {n: params[n] for n in cp}, mt, k, nn)
        grads = torch.autograd.grad(rtot, list(params.values()))
        adam.step(dict(zip(params, grads)))
        g, r = losses.detach().cpu().view(-1), rlosses.detach().view(-1)
        worst = max(worst, float(((g - r).abs() / r.abs()).max()))
        got.append(g.numpy()); ref.append(r.numpy())
    got, ref = np.mean(got, axis=0), np.mean(ref, axis=0)
    print(f"epoch {epoch}: lr {rate:.2e}  HIP mean {got.mean():.4f} (k=1 {got[0]:.4f})  oracle mean {ref.mean():.4f} (k=1 {ref[0]:.4f})  "
          f"worst step difference {worst:.1e}  ({time.time() - t0:.0f} s)", flush=True)
