"""criterion forward + backward at the bench shape, timed: python tools/scratch/nce_probe.py [hidden] [negatives] [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import cpc2_amd
from oracle import synth
H = int(sys.argv[1]) if len(sys.argv) > 1 else 256
NN = int(sys.argv[2]) if len(sys.argv) > 2 else 128
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
dev = "cuda:0"
crit = cpc2_amd.CPCUnsupersivedCriterion(12, H, H, NN, rnnMode="linear", sizeInputSeq=999).to(dev)
c = synth.features((64, 128, H), 1).to(dev).requires_grad_(True)
z = synth.features((64, 128, H), 2, relu=True).to(dev).requires_grad_(True)
def step():
    losses, acc = crit(c, z, None)
    e0.record()
    losses.sum().backward()
    e1.record()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(2): step()
torch.cuda.synchronize()
ts = []
for _ in range(reps):
    step(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
print("backward (criterion, incl. predictor GEMMs) ms:", sorted(ts)[len(ts)//2])
