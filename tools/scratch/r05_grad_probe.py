"""At the bench workload (b = 64): along ONE training trajectory (default form), the gradient of cpcStep's default form against
strict=True on the same parameters and the same negative indices, every 20 steps.  python tools/scratch/r05_grad_probe.py [steps] [config]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from cpc2_amd.train import backward, cpcStep
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 61
cfg = bench.CONFIGS[sys.argv[2] if len(sys.argv) > 2 else "small"]
lead_strict = len(sys.argv) > 3 and sys.argv[3] == "strict"      # the trajectory is the STRICT form's, the default form is evaluated beside it
every = int(sys.argv[4]) if len(sys.argv) > 4 else 20
dev = torch.device("cuda:0")
mA, cA, oA = bench.build(cfg, dev)
mB, cB, oB = bench.build(cfg, dev)
g = torch.Generator().manual_seed(1000)
x = (0.05 * torch.randn(64, 1, bench.WINDOW, generator=g)).to(dev)
label = torch.zeros(64, dtype=torch.long, device=dev)
names = [n for n, _ in list(cA.named_parameters()) + list(mA.named_parameters())]
for step in range(steps):
    cA.seed(5000 + step)
    torch.manual_seed(step)                  # (a transformer layer draws its dropout seed from torch's CPU generator)
    tot, lA, _ = cpcStep(x, x, label, mA, cA, strict=lead_strict)
    backward(tot)
    if step % every == 0:
        oB.flat.copy_(oA.flat)
        cB.seed(5000 + step)
        torch.manual_seed(step)
        totB, lB, _ = cpcStep(x, x, label, mB, cB, strict=not lead_strict)
        backward(totB)
        torch.cuda.synchronize()
        worst, wname = 0.0, ""
        for name, p, off in zip(names, oA.params, oA.offsets):
            n = p.numel()
            a, b = oA.flat_grad[off:off + n].double(), oB.flat_grad[off:off + n].double()
            e = float((a - b).abs().max() / (b.abs().max() + 1e-30))
            if e > worst:
                worst, wname = e, name
        print("step %3d: loss (trajectory form) %.5f (other form) %.5f (max rel diff %.2e); worst gradient difference %.2e of its tensor's scale (%s); |grad| %.3e"
              % (step, float(lA.mean()), float(lB.mean()), float(((lA - lB).abs() / lB.abs()).max()), worst, wname, float(oB.flat_grad.abs().max())))
        oB.zero_grad()
    oA.step(); oA.zero_grad()
