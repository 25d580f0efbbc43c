#!/usr/bin/env python3
"""Which HIP calls sit between two kernels of the training stream that are separated by an idle gap?
   python tools/trace_joins.py DIR   (DIR: rocprofv3 --hip-trace --kernel-trace --output-format csv)"""
import csv, glob, sys, collections
root = sys.argv[1]
kern, api = [], []
for f in glob.glob(root + "/**/*kernel_trace.csv", recursive=True):
    kern += list(csv.DictReader(open(f)))
for f in glob.glob(root + "/**/*hip_api_trace.csv", recursive=True):
    api += list(csv.DictReader(open(f)))
kern.sort(key=lambda r: int(r["Start_Timestamp"]))
api.sort(key=lambda r: int(r["Start_Timestamp"]))
by_corr = {r["Correlation_Id"]: r for r in api}
adam = [i for i, r in enumerate(kern) if "adam_kernel" in r["Kernel_Name"]]
main_q = collections.Counter(kern[i]["Queue_Id"] for i in adam).most_common(1)[0][0]
seg = kern[adam[-2] + 1: adam[-1] + 1]                     # the last whole step
main = [r for r in seg if r["Queue_Id"] == main_q]
name = lambda r: r["Kernel_Name"].replace("cpc::", "").replace("void ", "").split("(")[0][:40]
prev = kern[adam[-2]]
tid = by_corr[prev["Correlation_Id"]]["Thread_Id"] if prev["Correlation_Id"] in by_corr else None
print("gap_us  kernel after the gap                       HIP calls of the launching thread between the two launches")
for r in main:
    gap = (int(r["Start_Timestamp"]) - int(prev["End_Timestamp"])) / 1e3
    if gap > 3.0 and prev["Correlation_Id"] in by_corr and r["Correlation_Id"] in by_corr:
        a0, a1 = int(by_corr[prev["Correlation_Id"]]["End_Timestamp"]), int(by_corr[r["Correlation_Id"]]["Start_Timestamp"])
        calls = [c["Function"] for c in api if a0 <= int(c["Start_Timestamp"]) <= a1 and (tid is None or c["Thread_Id"] == tid)]
        cnt = collections.Counter(calls)
        print(f"{gap:6.1f}  {name(prev):>30s} -> {name(r):40s} " + ", ".join(f"{k} x{v}" if v > 1 else k for k, v in cnt.items()))
    if int(r["End_Timestamp"]) > int(prev["End_Timestamp"]):
        prev = r
