"""End-to-end run on the reference's own cpc/test_data fixture (kept under tests/golden/test_db): FLAC files -> HBM-resident
feeder -> CPC-small (H=256, GRU, 12 linear predictors, 128 negatives) -> train.run with a ramp + step schedule ->
checkpoints in the reference layout.  Prints one line per epoch.
   python tools/train_fixture.py [epochs] [out_dir]"""
import json
import os
import random
import sys
import tempfile
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import cpc2_amd
from cpc2_amd.dataset import AudioBatchData, filterSeqs, findAllSeqs
from cpc2_amd.train import buildOptimizer, buildScheduler, getAR, getCriterion, getEncoder, run

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DB = os.path.join(ROOT, "tests", "golden", "test_db")
SEQS = os.path.join(ROOT, "tests", "golden", "seq_list.txt")
epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 20
out_dir = sys.argv[2] if len(sys.argv) > 2 else tempfile.mkdtemp(prefix="cpc_fixture_")

args = types.SimpleNamespace(hiddenEncoder=256, hiddenGar=256, nPredicts=12, negativeSamplingExt=128, sizeWindow=20480,
                             samplingType="uniform", cpc_mode=None, encoder_type="cpc", normMode="layerNorm", arMode="GRU",
                             nLevelsGRU=1, rnnMode="linear", dropout=False, abspos=False)
torch.manual_seed(0)
random.seed(0)
np.random.seed(0)
dev = torch.device("cuda:0")
seqs, speakers = findAllSeqs(DB, extension=".flac")
seqs = sorted(filterSeqs(SEQS, seqs), key=lambda sq: sq[1])          # os.walk order differs from box to box
data = AudioBatchData(DB, args.sizeWindow, seqs, None, len(speakers), device=dev)
model = cpc2_amd.CPCModel(getEncoder(args), getAR(args)).to(dev)
crit = getCriterion(args, model.gEncoder.DOWNSAMPLING).to(dev)
opt = buildOptimizer(model, crit, lr=2e-4)
sched = buildScheduler(opt, schedulerStep=-1, schedulerRamp=3)
os.makedirs(out_dir, exist_ok=True)
ckpt = os.path.join(out_dir, "checkpoint")
with open(ckpt + "_args.json", "w") as fh:
    json.dump(vars(args), fh, indent=2)
logs = {"epoch": [], "iter": [], "saveStep": max(1, epochs // 2), "logging_step": 10 ** 9}
print(f"{len(seqs)} sequences, {len(data)} windows = {len(data) * 1.28:.0f} s of audio, {len(speakers)} speakers; checkpoints in {out_dir}")
run(data, data, 8, args.samplingType, model, crit, epochs, ckpt, opt, sched, logs)
for e in logs["epoch"]:
    print(f"epoch {e:3d}: train loss {np.mean(logs['locLoss_train'][e]):.4f} acc {np.mean(logs['locAcc_train'][e]):.4f} | "
          f"val loss {np.mean(logs['locLoss_val'][e]):.4f} acc {np.mean(logs['locAcc_val'][e]):.4f}   "
          f"(k=1: {logs['locLoss_val'][e][0]:.3f} / {logs['locAcc_val'][e][0]:.3f}, k=12: {logs['locLoss_val'][e][-1]:.3f} / {logs['locAcc_val'][e][-1]:.3f})")
first, last = np.mean(logs["locLoss_val"][0]), np.mean(logs["locLoss_val"][-1])
print(f"validation loss {first:.4f} -> {last:.4f} (chance: ln 129 = {np.log(129):.4f})")
assert last < first
