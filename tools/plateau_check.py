#!/usr/bin/env python3
"""Does cpcStep's default form (context network on the b context windows, W frames) LEARN from its own trajectory at b = 64?
Round 5's soak left it on the ln(129) plateau after 2000 steps while the strict form fell after ~200 (white-noise windows: the step at
which Adam leaves that plateau depends on rounding-sized differences).  Both forms start here from the same parameters with the same
negative stream and run until the mean loss is below 4.5 (or --max-steps); the loss is printed every --every steps.
    python tools/plateau_check.py [--max-steps 6000] [--seeds 0,1]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cpc2_amd
from cpc2_amd.train import backward, buildOptimizer, cpcStep

ap = argparse.ArgumentParser()
ap.add_argument("--max-steps", type=int, default=6000)
ap.add_argument("--every", type=int, default=250)
ap.add_argument("--seeds", default="0,1")
ap.add_argument("--batch", type=int, default=64)
args = ap.parse_args()
dev = torch.device("cuda:0")
for seed in (int(s) for s in args.seeds.split(",")):
    for form in ("default", "strict"):
        torch.manual_seed(seed)
        model = cpc2_amd.CPCModel(cpc2_amd.CPCEncoder(256, "layerNorm"), cpc2_amd.CPCAR(256, 256, False, 1, mode="GRU")).to(dev)
        crit = cpc2_amd.CPCUnsupersivedCriterion(12, 256, 256, 128, rnnMode="linear", sizeInputSeq=128).to(dev)
        opt = buildOptimizer(model, crit, lr=2e-4)
        crit.seed(1234 + seed)
        crit.sampler.prefetch = True
        g = torch.Generator().manual_seed(1000 + seed)
        x = (0.05 * torch.randn(args.batch, 1, 20480, generator=g)).to(dev)
        label = torch.zeros(args.batch, dtype=torch.long, device=dev)
        trace, fell = [], None
        for step in range(args.max_steps):
            tot, losses, _acc = cpcStep(x, x, label, model, crit, strict=(form == "strict"))
            backward(tot)
            opt.step()
            opt.zero_grad()
            if (step + 1) % args.every == 0:
                mean = float(losses.detach().mean())
                trace.append(f"{step + 1}: {mean:.4f}")
                if mean < 4.5:
                    fell = step + 1
                    break
        print(f"seed {seed} {form:8s} " + ("left the plateau by step %d" % fell if fell else "still above 4.5 after %d steps" % args.max_steps) + " | " + "  ".join(trace), flush=True)
