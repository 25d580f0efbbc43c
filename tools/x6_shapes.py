#!/usr/bin/env python3
"""Per-shape table of the split-in-kernel GEMM launches (gemm_nt_x6_kernel / gemm_tn_x6_kernel) of one training step.

    python tools/x6_shapes.py [--configs small,transformer,large] [--reps 30]

Step 1 logs every launch of one step per configuration (CPC_GEMM_LOG=1: M, N, K / R, leading dimensions, tile, grid, K / row split).
Step 2 replays every distinct shape ALONE through the C entries (cpc_gemm_nt / cpc_gemm_tn) on random data, events around `reps`
launches: us, TFLOP/s of algorithmic f32 flops, fraction of 416.7 (2500 / 6).  The TN time includes the slab reduce (as the step's
class timer does).  Prints markdown."""
import argparse, collections, json, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PEAK = 2500.0 / 6.0
ap = argparse.ArgumentParser()
ap.add_argument("--configs", default="small,transformer,large")
ap.add_argument("--reps", type=int, default=30)
ap.add_argument("--replay", default=None, help=argparse.SUPPRESS)
args = ap.parse_args()

if args.replay is None:
    shapes = collections.OrderedDict()
    for cfg in args.configs.split(","):
        env = dict(os.environ, CPC_GEMM_LOG="1")
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", cfg, "--steps", "1", "--warmup", "1", "--cpu-seconds", "0",
                            "--also=", "--no-prof"], env=env, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [l for l in r.stderr.splitlines() if l.startswith("cpc_gemm ")]
        lines = lines[len(lines) // 2:]                       # the second of the two steps
        for l in lines:
            if "kernel=x6" not in l:
                continue
            key = re.sub(r" ld[abc]=\d+", "", l)
            shapes.setdefault(key, {"line": l, "per_step": collections.Counter()})["per_step"][cfg] += 1
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--reps", str(args.reps), "--replay", json.dumps([v["line"] for v in shapes.values()])],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    times = json.loads(r.stdout.strip().splitlines()[-1])
    print("| kind | M x N x K (TN: M x N over R rows) | tile, grid, split | launches per step | us alone | TFLOP/s | of 416.7 |")
    print("|---|---|---|---|---|---|---|")
    tot = collections.Counter()
    for (key, v), us in zip(shapes.items(), times):
        f = dict(kv.split("=") for kv in v["line"].split()[2:])
        kind = v["line"].split()[1]
        m, n, k = int(f["M"]), int(f["N"]), int(f.get("K", f.get("R")))
        fl = 2.0 * m * n * k
        tf = fl / us / 1e6
        per = ", ".join(f"{c} x{cnt}" for c, cnt in v["per_step"].items())
        extra = f"{f['tile']}, {f['grid']} wg, split {f['splits']}" + (f", epi {f['epi']}" if f.get("epi", "0") != "0" else "")
        print(f"| {kind} | {m} x {n} x {k} | {extra} | {per} | {us:.1f} | {tf:.0f} | {tf / PEAK:.3f} |")
        for c, cnt in v["per_step"].items():
            tot[(c, kind)] += cnt * us
            tot[(c, kind, "fl")] += cnt * fl
    print()
    for c in args.configs.split(","):
        for kind in ("nt", "tn"):
            if tot[(c, kind)]:
                print(f"* {c} {kind.upper()}: {tot[(c, kind)]:.0f} us per step alone, {tot[(c, kind, 'fl')] / tot[(c, kind)] / 1e6:.0f} TFLOP/s = "
                      f"{tot[(c, kind, 'fl')] / tot[(c, kind)] / 1e6 / PEAK:.3f} of 416.7")
    sys.exit(0)

import torch
from cpc2_amd import _lib
lib = _lib.load()
dev = torch.device("cuda:0")
out = []
for line in json.loads(args.replay):
    f = dict(kv.split("=") for kv in line.split()[2:])
    kind = line.split()[1]
    g = torch.Generator(device=dev).manual_seed(1)
    if kind == "nt":
        m, n, k, lda, ldb, ldc = (int(f[x]) for x in ("M", "N", "K", "lda", "ldb", "ldc"))
        a = torch.randn(m * lda + k, device=dev, generator=g)
        b = torch.randn(n * ldb + k, device=dev, generator=g)
        c = torch.empty(m * max(ldc, n), device=dev)
        run = lambda: _lib.check(lib.cpc_gemm_nt(_lib.ptr(a), lda, _lib.ptr(b), ldb, _lib.ptr(c), max(ldc, n), None, m, n, k, _lib.stream_ptr(dev)))
    else:
        m, n, r, lda, ldb, ldc = (int(f[x]) for x in ("M", "N", "R", "lda", "ldb", "ldc"))
        a = torch.randn(r * lda + m, device=dev, generator=g)
        b = torch.randn(r * ldb + n, device=dev, generator=g)
        c = torch.empty(m * max(ldc, n), device=dev)
        nb = lib.cpc_gemm_tn_scratch_bytes(m, n, r)
        sc = torch.empty(nb, dtype=torch.uint8, device=dev)
        run = lambda: _lib.check(lib.cpc_gemm_tn(_lib.ptr(a), lda, _lib.ptr(b), ldb, _lib.ptr(c), max(ldc, n), m, n, r, _lib.ptr(sc), nb, _lib.stream_ptr(dev)))
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(args.reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    out.append(1e3 * e0.elapsed_time(e1) / args.reps)
print(json.dumps(out))
