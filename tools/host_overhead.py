"""Host time to ENQUEUE one training step (Python + ctypes + HIP launches) against the GPU time of the step:
   python tools/host_overhead.py [config]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from cpc2_amd.train import DataParallelContext, cpcStep

cfg = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "small"]
dev = torch.device("cuda:0")
model, crit, opt = bench.build(cfg, dev)
dp = DataParallelContext(opt)
crit.seed(1)
crit.sampler.prefetch = True
x = (0.05 * torch.randn(64, 1, bench.WINDOW)).to(dev)
label = torch.zeros(64, dtype=torch.long, device=dev)

def step():
    tot, losses, acc = cpcStep(x, x, label, model, crit)
    tot.backward(); dp.reduce_and_step(); opt.zero_grad()

for _ in range(5): step()
torch.cuda.synchronize()
n = 30
t0 = time.perf_counter()
for _ in range(n): step()
t_enq = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"enqueue {1e3 * t_enq / n:.2f} ms/step (host), complete {1e3 * t_all / n:.2f} ms/step (GPU-bound when larger)")
# pure host cost: start each step on an idle GPU, stop the clock when the Python call returns
host = 0.0
for _ in range(n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step()
    host += time.perf_counter() - t0
torch.cuda.synchronize()
print(f"host-only {1e3 * host / n:.2f} ms/step (step enqueued on an idle GPU)")
