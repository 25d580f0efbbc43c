#!/usr/bin/env python3
"""Which HSA queue every stream's kernels ran on, from a rocprofv3 --kernel-trace directory (csv or the rocpd sqlite db)."""
import collections, csv, glob, os, sqlite3, sys

def rows_from(path):
    files = glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True)
    if files:
        for f in files:
            for r in csv.DictReader(open(f)):
                yield r["Queue_Id"], r["Stream_Id"], r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        return
    for f in glob.glob(os.path.join(path, "**", "*.db"), recursive=True):
        db = sqlite3.connect(f)
        tabs = [t[0] for t in db.execute("select name from sqlite_master where type in ('table','view')")]
        view = "kernels" if "kernels" in tabs else None
        if view is None:
            print("tables:", tabs); return
        cols = [c[1] for c in db.execute(f"pragma table_info({view})")]
        q = "queue_id" if "queue_id" in cols else "queue"
        s = "stream_id" if "stream_id" in cols else "stream"
        for r in db.execute(f"select {q}, {s}, name, start, end from {view}"):
            yield r

agg = collections.defaultdict(lambda: [collections.Counter(), 0, 0])
for q, s, name, a, b in rows_from(sys.argv[1]):
    e = agg[(str(q), str(s))]
    e[0][name.split("(")[0][-48:]] += 1
    e[1] += 1
    e[2] += b - a
for (q, s), (names, n, busy) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"queue {q:>3} stream {s:>3}: {n:6d} dispatches {busy / 1e6:9.2f} ms busy  ", ", ".join(f"{k} x{v}" for k, v in names.most_common(4)))
