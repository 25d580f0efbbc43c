#!/usr/bin/env python3
"""Per-step busy / idle of the training stream's queue from a rocprofv3 --kernel-trace csv dir: steps are delimited by adam_kernel.
   python tools/trace_gaps.py DIR [first_step last_step]"""
import csv, glob, sys, collections
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
main_q = collections.Counter(rows[i]["Queue_Id"] for i in adam).most_common(1)[0][0]
lo = int(sys.argv[2]) if len(sys.argv) > 2 else 0
hi = int(sys.argv[3]) if len(sys.argv) > 3 else len(adam) - 1
print("step  span_us  main_busy_us  main_idle_us  biggest gaps (us, after kernel)   #kernels   batch(grid of conv0)")
for s in range(max(lo, 1), min(hi, len(adam) - 1) + 1):
    seg = rows[adam[s - 1] + 1: adam[s] + 1]
    t0 = int(rows[adam[s - 1]]["End_Timestamp"]); t1 = int(rows[adam[s]]["End_Timestamp"])
    main = [r for r in seg if r["Queue_Id"] == main_q]
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in main) / 1e3
    gaps, prev_end, prev_name = [], t0, "adam"
    for r in main:
        g = (int(r["Start_Timestamp"]) - prev_end) / 1e3
        if g > 8:
            gaps.append((round(g), prev_name[:28] + "->" + r["Kernel_Name"].replace("cpc::", "").replace("void ", "")[:24]))
        prev_end = max(prev_end, int(r["End_Timestamp"])); prev_name = r["Kernel_Name"].replace("cpc::", "").replace("void ", "")
    c0 = [r for r in seg if "conv0_bwd" in r["Kernel_Name"]]
    print(f"{s:4d} {(t1 - t0) / 1e3:8.0f} {busy:10.0f} {(t1 - t0) / 1e3 - busy:10.0f}   {sorted(gaps, reverse=True)[:4]}   {len(seg)}")
