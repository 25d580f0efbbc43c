#!/usr/bin/env python3
"""Tile / K-split sweep of the split-in-kernel NT product on given shapes, each arm in its own process (the switches are read once):
    python tools/x6_sweep.py "7424x256x3072,7424x256x768,8192x256x2048,8192x256x256"
Arms: kernel family (old two-barrier / pipelined) x tile (64x128, 128x128, 128x256) x splits.  us per launch, alone, random data.
(A K split through cpc_gemm_nt has no slab scratch: partial products are added with atomics into a zeroed C -- the memset is in.)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 2 and sys.argv[1] == "--one-tn":
    sys.path.insert(0, ROOT)
    import torch
    from cpc2_amd import _lib
    lib = _lib.load()
    dev = torch.device("cuda:0")
    out = []
    for shp in sys.argv[2].split(","):
        m, n, r = (int(v) for v in shp.split("x"))
        a = torch.randn(r, m, device=dev); b = torch.randn(r, n, device=dev) * 0.05; c = torch.empty(m, n, device=dev)
        nb = lib.cpc_gemm_tn_scratch_bytes(m, n, r)
        sc = torch.empty(nb, dtype=torch.uint8, device=dev)
        run = lambda: _lib.check(lib.cpc_gemm_tn(_lib.ptr(a), m, _lib.ptr(b), n, _lib.ptr(c), n, m, n, r, _lib.ptr(sc), nb, _lib.stream_ptr(dev)))
        for _ in range(3): run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(20): run()
        e1.record(); torch.cuda.synchronize()
        out.append(1e3 * e0.elapsed_time(e1) / 20)
    print(json.dumps(out))
    sys.exit(0)
if len(sys.argv) > 2 and sys.argv[1] == "--one":
    sys.path.insert(0, ROOT)
    import torch
    from cpc2_amd import _lib
    lib = _lib.load()
    dev = torch.device("cuda:0")
    out = []
    for shp in sys.argv[2].split(","):
        m, n, k = (int(v) for v in shp.split("x"))
        a = torch.randn(m, k, device=dev); b = torch.randn(n, k, device=dev) * 0.05; c = torch.empty(m, n, device=dev)
        run = lambda: _lib.check(lib.cpc_gemm_nt(_lib.ptr(a), k, _lib.ptr(b), k, _lib.ptr(c), n, None, m, n, k, _lib.stream_ptr(dev)))
        for _ in range(3): run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(20): run()
        e1.record(); torch.cuda.synchronize()
        out.append(1e3 * e0.elapsed_time(e1) / 20)
    print(json.dumps(out))
    sys.exit(0)
if sys.argv[1] == "--tn":
    # TN: M x N over R rows, row split sweep (slab reduce included)
    names = sys.argv[2].split(",")
    print("| kernel | row splits | " + " | ".join(names) + " |")
    print("|---|---|" + "---|" * len(names))
    for fam in ("old", "pipelined"):
        for splits in (0, 8, 16, 24, 32, 48, 64, 96):
            env = dict(os.environ)
            if splits:
                env["CPC_GEMM_TN_SPLITS"] = str(splits)
            if fam == "old":
                env["CPC_GEMM_X6_OLD"] = "1"
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--one-tn", sys.argv[2]], env=env, capture_output=True, text=True)
            if r.returncode != 0:
                print(f"| {fam} | {splits} | failed: {r.stderr[-200:]} |"); continue
            us = json.loads(r.stdout.strip().splitlines()[-1])
            cells = []
            for shp, t in zip(names, us):
                m, n, k = (int(v) for v in shp.split("x"))
                cells.append(f"{t:.1f} us ({2.0 * m * n * k / t / 1e6 / (2500 / 6):.2f})")
            print(f"| {fam} | {splits or 'default'} | " + " | ".join(cells) + " |", flush=True)
    sys.exit(0)
shapes = sys.argv[1]
names = shapes.split(",")
print("| kernel | tile | splits | " + " | ".join(names) + " |")
print("|---|---|---|" + "---|" * len(names))
for fam in ("old", "pipelined"):
    for tile in ("1,2", "2,2", "2,4"):
        for splits in (1, 2, 4, 8):
            env = dict(os.environ, CPC_GEMM_TILE=tile, CPC_GEMM_SPLITS=str(splits))
            if fam == "old":
                env["CPC_GEMM_X6_OLD"] = "1"
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--one", shapes], env=env, capture_output=True, text=True)
            if r.returncode != 0:
                print(f"| {fam} | {tile} | {splits} | failed: {r.stderr[-200:]} |"); continue
            us = json.loads(r.stdout.strip().splitlines()[-1])
            cells = []
            for shp, t in zip(names, us):
                m, n, k = (int(v) for v in shp.split("x"))
                cells.append(f"{t:.1f} us ({2.0 * m * n * k / t / 1e6 / (2500 / 6):.2f})")
            mi, nj = tile.split(",")
            print(f"| {fam} | {64 * int(mi)}x{64 * int(nj)} | {splits} | " + " | ".join(cells) + " |", flush=True)
