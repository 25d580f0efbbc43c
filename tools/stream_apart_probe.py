#!/usr/bin/env python3
"""Which hardware queue the library's streams get in a crowded process, and what cpc_streams_overlap says about every pair.
    rocprofv3 --kernel-trace --output-format csv -d OUT -- python3 tools/stream_apart_probe.py ; python3 tools/show_queues.py OUT"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cpc2_amd import _lib
lib = _lib.load()
dev = torch.device("cuda:0")
x = torch.zeros(8, device=dev); torch.cuda.synchronize()
cur = _lib.stream_ptr(dev)
names, streams = ["null"], [cur]
side = ctypes.c_void_p()
_lib.check(lib.cpc_side_stream(cur, ctypes.byref(side)), "side"); names.append("side"); streams.append(side)
crowd = [torch.cuda.Stream(dev) for _ in range(int(os.environ.get("CROWD", "40")))]
for i in range(int(os.environ.get("MADE", "3"))):
    raw = ctypes.c_void_p()
    _lib.check(lib.cpc_stream_create_apart((ctypes.c_void_p * 1)(cur.value), 1, ctypes.byref(raw)), "apart")
    names.append(f"made{i}"); streams.append(raw)
for i, s in enumerate(crowd[:6]):
    names.append(f"pool{i}"); streams.append(ctypes.c_void_p(s.cuda_stream))
print("failures", lib.cpc_stream_apart_failures())
print("spin on row, tag on column: 1 = beside, 0 = behind")
print(" " * 8 + " ".join(f"{n:>6}" for n in names))
for a, na in zip(streams, names):
    print(f"{na:>8}" + " ".join(f"{lib.cpc_streams_overlap(a, b) if a.value != b.value else -1:>6}" for b in streams))
torch.cuda.synchronize()
