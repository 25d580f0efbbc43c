"""Per-step wall time of the headline configuration over the first 80 steps of a process (does the step time settle, and when?)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from cpc2_amd.train import DataParallelContext, cpcStep
dev = torch.device("cuda:0")
cfg = bench.CONFIGS["small"]
model, crit, opt = bench.build(cfg, dev)
dp = DataParallelContext(opt)
crit.seed(1234); crit.sampler.prefetch = True
x = (0.05 * torch.randn(64, 1, bench.WINDOW, generator=torch.Generator().manual_seed(1000))).to(dev)
label = torch.zeros(64, dtype=torch.long, device=dev)
ts = []
for i in range(80):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    tot, l, a = cpcStep(x, x, label, model, crit, dp=dp); tot.backward(); dp.reduce_and_step(); opt.zero_grad()
    torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
print("per-step ms (synchronised every step):", " ".join(f"{t:.2f}" for t in ts))
# and unsynchronised blocks of 10
for blk in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(10):
        tot, l, a = cpcStep(x, x, label, model, crit, dp=dp); tot.backward(); dp.reduce_and_step(); opt.zero_grad()
    torch.cuda.synchronize(); print(f"block {blk}: {100 * (time.perf_counter() - t0):.3f} ms/step")
