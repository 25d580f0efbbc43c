"""Which torch operators (not library kernels) still launch device work inside one training step?  torch.profiler, one step."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from torch.profiler import profile, ProfilerActivity
from cpc2_amd.train import DataParallelContext, backward, cpcStep
cfg = bench.CONFIGS["small"]
dev = torch.device("cuda:0")
model, crit, opt = bench.build(cfg, dev)
dp = DataParallelContext(opt)
crit.seed(1); crit.sampler.prefetch = True
x = (0.05 * torch.randn(64, 1, bench.WINDOW)).to(dev)
label = torch.zeros(64, dtype=torch.long, device=dev)
def step():
    tot, losses, acc = cpcStep(x, x, label, model, crit)
    backward(tot); dp.reduce_and_step(); opt.zero_grad()
for _ in range(5): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
for ev in prof.key_averages(group_by_stack_n=4):
    if ev.device_time_total > 0 and ev.key.startswith("aten::"):
        print(ev.key, "device us %.1f" % ev.device_time_total, "count", ev.count)
        for fr in ev.stack[:4]:
            print("      ", fr)
