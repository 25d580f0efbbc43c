"""criterion forward only at the bench shape, timed with the library's own events: python tools/scratch/fwd_probe.py [hidden] [negatives]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ctypes
import torch
import cpc2_amd
from cpc2_amd import _lib
from oracle import synth
H = int(sys.argv[1]) if len(sys.argv) > 1 else 256
NN = int(sys.argv[2]) if len(sys.argv) > 2 else 128
dev = "cuda:0"
lib = _lib.load()
crit = cpc2_amd.CPCUnsupersivedCriterion(12, H, H, NN, rnnMode="linear", sizeInputSeq=999).to(dev)
c = synth.features((64, 128, H), 1).to(dev)
z = synth.features((64, 128, H), 2, relu=True).to(dev)
with torch.no_grad():
    for _ in range(3): crit(c, z, None)
    torch.cuda.synchronize()
    lib.cpc_prof_enable(1)
    tot, cnt = ctypes.c_double(), ctypes.c_long()
    lib.cpc_prof_read(b"infonce_fwd", ctypes.byref(tot), ctypes.byref(cnt))      # reading clears
    for _ in range(10): crit(c, z, None)
    torch.cuda.synchronize()
    lib.cpc_prof_read(b"infonce_fwd", ctypes.byref(tot), ctypes.byref(cnt))
    lib.cpc_prof_enable(0)
print(f"infonce_fwd: {tot.value / max(cnt.value, 1) * 1e3:.1f} us per launch ({cnt.value} launches)  env {os.environ.get('CPC_NCE_RING_DBG')}")
