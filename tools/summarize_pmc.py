#!/usr/bin/env python3
"""Per-kernel summary of a tools/gpu_profile.sh run: time share, HBM bytes per launch (FETCH_SIZE doubled for
wide coalesced reads as MI355X_MICROARCH.md prescribes; WRITE_SIZE as is; both counted in KiB), MFMA-busy share.
   python tools/summarize_pmc.py gpurun_out/prof_<tag> [steps] [config]  > profiles/<name>.md
Also merges the HBM bytes per launch of every kernel into profiles/pmc_traffic.json under "<kernel>:<config>" -- the
table bench.py reads its roofline `traffic` from (so that the number in the JSON line is the one of a committed profile)."""
import csv, glob, json, os, sys, collections
root = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
config = sys.argv[3] if len(sys.argv) > 3 else "small"
def one(pattern):
    f = glob.glob(f"{root}/{pattern}")
    return f[0] if f else None
def short(n):
    n = n.replace("cpc::", "").replace("void ", "")
    return n.split("(")[0][:44]
stats = list(csv.DictReader(open(one("stats/*/*kernel_stats.csv"))))
def agg(path, counter):
    d = collections.defaultdict(list)
    if not path: return d
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            d[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return d
fetch = agg(one("fetch/*/*counter_collection.csv"), "FETCH_SIZE")
write = agg(one("write/*/*counter_collection.csv"), "WRITE_SIZE")
mfma = agg(one("mfma/*/*counter_collection.csv"), "SQ_VALU_MFMA_BUSY_CYCLES")
gui = agg(one("mfma/*/*counter_collection.csv"), "GRBM_GUI_ACTIVE")
tot = sum(float(r["TotalDurationNs"]) for r in stats)
print(f"| kernel | launches/step | ms/step | avg us | % GPU time | HBM read MB/launch (2x FETCH_SIZE) | HBM write MB/launch | achieved HBM GB/s | MFMA busy % |")
print("|---|---|---|---|---|---|---|---|---|")
for r in stats[:18]:
    n = short(r["Name"]); calls = int(r["Calls"]); avg = float(r["AverageNs"])
    fr = 2 * 1024 * sum(fetch[n]) / max(len(fetch[n]), 1) if n in fetch else None
    wr = 1024 * sum(write[n]) / max(len(write[n]), 1) if n in write else None
    bw = (fr + wr) / avg if fr is not None and wr is not None else None       # bytes/ns = GB/s
    mu = None
    if n in mfma and n in gui and sum(gui[n]) > 0:
        mu = 100.0 * sum(mfma[n]) / (sum(gui[n]) / 8 * 1024)     # GUI_ACTIVE summed over 8 XCDs; 1024 SIMDs
    f = lambda v, fmt: (fmt % v) if v is not None else "-"
    print(f"| `{n}` | {calls/steps:.1f} | {float(r['TotalDurationNs'])/1e6/steps:.3f} | {avg/1e3:.1f} | {float(r['Percentage']):.1f} | "
          f"{f(fr/1e6 if fr is not None else None, '%.1f')} | {f(wr/1e6 if wr is not None else None, '%.1f')} | {f(bw, '%.0f')} | {f(mu, '%.0f')} |")
# the table bench.py reads
here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = os.path.join(here, "profiles", "pmc_traffic.json")
table = json.load(open(path)) if os.path.exists(path) else {"note": "HBM bytes per launch = 1024 * (2 * FETCH_SIZE + WRITE_SIZE), mean over the launches of a profiled bench.py run (rocprofv3 --pmc, separate passes)", "kernels": {}}
for r in stats:
    full = r["Name"].replace("cpc::", "").replace("void ", "").split("(")[0]
    n = short(r["Name"])
    if n in fetch and n in write:
        fr = 2 * 1024 * sum(fetch[n]) / len(fetch[n]); wr = 1024 * sum(write[n]) / len(write[n])
        table["kernels"][f"{full}:{config}"] = {"bytes_per_launch": round(fr + wr), "read": round(fr), "write": round(wr),
                                                "launches_per_step": int(r["Calls"]) / steps,
                                                "source": os.path.basename(root.rstrip("/"))}
json.dump(table, open(path, "w"), indent=1, sort_keys=True)
print(f"\ntotal GPU time per step: {tot/1e6/steps:.3f} ms ({steps} steps incl. warm-up)")
