"""Print the last step of a rocprofv3 kernel trace (tools/trace_step.sh): python tools/show_trace.py <dir>"""
import csv, glob, sys
d = sys.argv[1]
rows = list(csv.DictReader(open(glob.glob(d + "/*/*kernel_trace.csv")[0])))
copies = []
for f in glob.glob(d + "/*/*memory_copy_trace.csv"):
    copies = list(csv.DictReader(open(f)))
idx = [i for i, r in enumerate(rows) if "adam" in r["Kernel_Name"]]
seg = rows[idx[-2] + 1: idx[-1] + 1]
t0, t1 = int(seg[0]["Start_Timestamp"]), int(seg[-1]["End_Timestamp"])
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("cpc::", "").replace("void ", "")[:44],
       f"grid {int(r['Grid_Size_X']) // int(r['Workgroup_Size_X'])}x{int(r['Grid_Size_Y']) // int(r['Workgroup_Size_Y'])}x{int(r['Grid_Size_Z']) // int(r['Workgroup_Size_Z'])}") for r in seg]
for c in copies:
    s, e = int(c["Start_Timestamp"]), int(c["End_Timestamp"])
    if t0 <= s <= t1:
        ev.append((s, e, "COPY " + c.get("Direction", ""), c.get("Bytes", "") if "Bytes" in c else ""))
ev.sort()
tot, prev_end = 0.0, t0
agg = {}
gaps = []
for s, e, name, extra in ev:
    d_us = (e - s) / 1e3
    gap = (s - prev_end) / 1e3
    if not name.startswith("COPY"):
        gaps.append(max(gap, 0.0))
        tot += d_us
        prev_end = max(prev_end, e)
        agg[name] = agg.get(name, 0) + d_us
    print(f"{(s - t0) / 1e3:9.1f} {d_us:8.1f} {'gap %6.1f' % gap if gap > 3 else '          '} {name:44s} {extra}")
print("kernel sum %.1f us, span %.1f us" % (tot, (t1 - t0) / 1e3))
print("launches %d; idle between consecutive kernels: total %.1f us (gaps <= 3 us: %d, sum %.1f; 3-10 us: %d, sum %.1f; > 10 us: %d, sum %.1f)" % (
    len(gaps), sum(gaps), sum(g <= 3 for g in gaps), sum(g for g in gaps if g <= 3), sum(3 < g <= 10 for g in gaps),
    sum(g for g in gaps if 3 < g <= 10), sum(g > 10 for g in gaps), sum(g for g in gaps if g > 10)))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1]):
    print("  %8.1f  %s" % (v, k))
