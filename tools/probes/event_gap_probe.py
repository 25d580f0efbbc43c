#!/usr/bin/env python3
"""What a cross-stream primitive costs the stream it is issued on, between two kernels (GPU time per iteration, host far ahead):
   nothing / event record (no timing) / event record (timing) / wait for an event of another stream that fired long ago /
   wait for an event recorded on the SAME stream.  python tools/probes/event_gap_probe.py"""
import time
import torch
dev = torch.device("cuda:0")
x = torch.zeros(8 << 20, device=dev)
other = torch.cuda.Stream(dev)
old = torch.cuda.Event()
with torch.cuda.stream(other):
    x[:16].add_(1)
    old.record()
torch.cuda.synchronize()


def run(kind, n=400):
    evs = []
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        x.mul_(1.0001)
        if kind == "record":
            e = torch.cuda.Event(); e.record(); evs.append(e)
        elif kind == "record_timing":
            e = torch.cuda.Event(enable_timing=True); e.record(); evs.append(e)
        elif kind == "wait_old_other":
            torch.cuda.current_stream(dev).wait_event(old)
        elif kind == "wait_fresh_other":
            with torch.cuda.stream(other):
                e = torch.cuda.Event(); e.record()
            torch.cuda.current_stream(dev).wait_event(e); evs.append(e)
        elif kind == "query_then_skip":
            if not old.query():
                torch.cuda.current_stream(dev).wait_event(old)
        x.mul_(0.9999)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n


for kind in ("nothing", "record", "record_timing", "wait_old_other", "wait_fresh_other", "query_then_skip", "nothing"):
    run(kind, 50)
    print(f"{kind:18s} {run(kind):7.2f} us per iteration (two kernels)")
