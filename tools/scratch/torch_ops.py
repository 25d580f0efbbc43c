"""aten ops that launch device copies / fills in one training step, with their Python call sites."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from torch.profiler import profile, ProfilerActivity
class A: pass
args = A(); args.batch = 64; args.dedup = False; args.no_prof = True
dev = torch.device("cuda:0")
from cpc2_amd.train import DataParallelContext, cpcStep
cfg = bench.CONFIGS["small"]
model, crit, opt = bench.build(cfg, dev)
dp = DataParallelContext(opt, early_params=list(crit.parameters()) + list(model.gAR.parameters()))
crit.seed(1234); crit.sampler.prefetch = True
x = (0.05 * torch.randn(64, 1, bench.WINDOW)).to(dev)
label = torch.zeros(64, dtype=torch.long, device=dev)
def step():
    tot, losses, _ = cpcStep(x, x, label, model, crit, dp=dp)
    tot.backward(); dp.reduce_and_step(); opt.zero_grad()
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(); torch.cuda.synchronize()
from collections import Counter
c = Counter()
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::fill_", "aten::zero_", "aten::cat", "aten::contiguous", "aten::clone", "aten::add_", "aten::sum", "aten::ones_like", "aten::zeros", "aten::empty_like", "aten::mul"):
        st = [s for s in (ev.stack or []) if "cpc2_amd" in s or "bench.py" in s or "torch_ops" in s]
        c[(ev.name, st[0] if st else "?")] += 1
for k, v in sorted(c.items(), key=lambda kv: -kv[1]): print(v, k)
