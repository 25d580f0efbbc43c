#!/usr/bin/env python3
"""Per-step view of a rocprofv3 kernel_stats.csv:  tools/show_stats.py <csv> <steps incl. warm-up> [rows]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 25
total = sum(float(r["TotalDurationNs"]) for r in rows) / 1e6
print(f"kernel time {total / steps:.3f} ms/step over {steps} steps")
for r in rows[:top]:
    print(f"{r['Name'][:84]:84s} {int(r['Calls']) / steps:7.1f}/step {float(r['TotalDurationNs']) / 1e6 / steps:8.3f} ms/step "
          f"{float(r['AverageNs']) / 1e3:9.1f} us")
