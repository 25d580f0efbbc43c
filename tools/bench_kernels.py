#!/usr/bin/env python3
"""One-line view of a bench.py JSON line:  tools/bench_kernels.py <json file> [substring of the kernel classes to show]"""
import json
import sys

d = json.load(open(sys.argv[1]))
pat = sys.argv[2] if len(sys.argv) > 2 else ""
print(sys.argv[1], d["ms_per_step"], "ms/step", {k: round(v["ms_per_step"], 3) for k, v in d["kernels"].items() if pat in k})
