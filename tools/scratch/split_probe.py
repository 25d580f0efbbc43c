"""cpc_split_planes and the two weight-gradient families alone at CPC-large's context-network shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cpc2_amd import _lib
lib = _lib.load()
dev = torch.device("cuda:0")
st = _lib.stream_ptr(dev)
def timeit(fn, reps=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
for rows, cols in ((16384, 1536), (16384, 512), (8192, 6144)):
    x = torch.randn(rows, cols, device=dev)
    plane = rows * cols
    out = torch.empty(3 * plane, dtype=torch.int16, device=dev)
    us = timeit(lambda: _lib.check(lib.cpc_split_planes(_lib.ptr(x), cols, rows, cols, _lib.ptr(out), plane, 0, rows, st)))
    print(f"split {rows} x {cols}: {us:.1f} us = {rows * cols * 10 / us / 1e6:.2f} TB/s")
for r, m, n in ((16384, 1536, 512), (8192, 6144, 512)):
    a, b = torch.randn(r, m, device=dev), torch.randn(r, n, device=dev)
    c = torch.empty(m, n, device=dev)
    nb = lib.cpc_gemm_tn_scratch_bytes(m, n, r)
    sc = torch.empty(nb, dtype=torch.uint8, device=dev)
    us = timeit(lambda: _lib.check(lib.cpc_gemm_tn(_lib.ptr(a), m, _lib.ptr(b), n, _lib.ptr(c), n, m, n, r, _lib.ptr(sc), nb, st)))
    print(f"gemm_tn {r} x {m} x {n} ({'planes' if not os.environ.get('CPC_TN_NO_PLANES') else 'split in kernel'}): {us:.1f} us = {2.0 * r * m * n / us / 1e6:.0f} TFLOP/s")
