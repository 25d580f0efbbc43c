#!/usr/bin/env python3
"""Several epochs of the feeder loop in one process: ms per step per epoch (does anything accumulate?).  tools/feeder_epochs.py [epochs] [ring_clear]"""
import os, sys, time, tempfile, shutil, random, io, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from cpc2_amd.dataset import AudioBatchData, findAllSeqs
from cpc2_amd.train import DataParallelContext, trainStep
from cpc2_amd import _lib
dev = torch.device("cuda:0")
cfg = bench.CONFIGS["small"]
model, crit, opt = bench.build(cfg, dev)
dp = DataParallelContext(opt)
crit.seed(1234)
tmp = tempfile.mkdtemp(prefix="cpc_feeder_", dir="/dev/shm")
try:
    bench.write_synthetic_corpus(tmp, 200 * 64, seed=7)
    random.seed(11)
    seqs, speakers = findAllSeqs(tmp, extension=".wav")
    data = AudioBatchData(tmp, bench.WINDOW, seqs, None, len(speakers), device=dev)
    for ep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
        loader = data.getDataLoader(64, "samespeaker", True)
        _lib.HOST_WAITS.clear()
        with contextlib.redirect_stdout(io.StringIO()):
            t0 = time.perf_counter()
            logs = trainStep(loader, model, crit, opt, None, 1000, dp=dp)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        print(f"epoch {ep}: {1e3 * dt / logs['iter']:.3f} ms per step over {logs['iter']} steps, loss {float(logs['locLoss_train'].mean()):.4f}, "
              f"allocated {torch.cuda.memory_allocated() >> 20} MB reserved {torch.cuda.memory_reserved() >> 20} MB, waits { {k: round(1e3 * v / logs['iter'], 2) for k, v in _lib.HOST_WAITS.items()} }", flush=True)
finally:
    shutil.rmtree(tmp, ignore_errors=True)
