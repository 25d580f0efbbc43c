"""Loss curves of cpcStep's default form against strict=True on the bench workload (same seeds): python tools/scratch/r05_curve_probe.py [steps] [config]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from cpc2_amd.train import DataParallelContext, backward, cpcStep
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
cfg = bench.CONFIGS[sys.argv[2] if len(sys.argv) > 2 else "small"]
dev = torch.device("cuda:0")
curves = {}
for form in ("default", "strict", "default_nodefer"):
    if form == "default_nodefer":
        os.environ["CPC_NCE_NO_DEFER"] = "1"; os.environ["CPC_NO_GRAD_TAIL"] = "1"
    model, crit, opt = bench.build(cfg, dev)
    dp = DataParallelContext(opt)
    crit.seed(1234)
    g = torch.Generator().manual_seed(1000)
    x = (0.05 * torch.randn(64, 1, bench.WINDOW, generator=g)).to(dev)
    label = torch.zeros(64, dtype=torch.long, device=dev)
    out = []
    for step in range(steps):
        tot, losses, acc = cpcStep(x, x, label, model, crit, dp=dp, strict=form == "strict")
        backward(tot); dp.reduce_and_step(); opt.zero_grad()
        if step % 25 == 0 or step == steps - 1:
            out.append((step, float(losses.mean())))
    curves[form] = out
    os.environ.pop("CPC_NCE_NO_DEFER", None); os.environ.pop("CPC_NO_GRAD_TAIL", None)
for i in range(len(curves["default"])):
    print("step %4d   default %.4f   strict %.4f   default without deferrals %.4f" % (curves["default"][i][0], curves["default"][i][1], curves["strict"][i][1], curves["default_nodefer"][i][1]))
