"""Per-shape table of the plane-fed GEMM launches of one training step (the 8 NT and 4 TN launches of the encoder):

    python tools/planes_shapes.py <kernel-trace dir of tools/trace_step.sh> [stamps log] [--hidden 256 --windows 128]

The launches of the LAST step of the trace are matched to their shapes by order (forward conv1..4, then backward per layer 4..1:
weight gradient, backward data); the algorithmic FLOPs are the convolutions' own (2 * rows * k * H * H, junk rows not counted).
`stamps log`: stderr of `CPC_PLANES_DBG=8 python bench.py --steps 1 --warmup 1 ...` -- one "planes stamps" line per NT launch with
the in-kernel clock (d s_memtime / d s_memrealtime), loop cycles per tile, prologue and epilogue; the last step's eight are used.
Prints a markdown table (profiles/r04_planes_shapes.md)."""
import argparse
import csv
import glob
import re

CONV = ((10, 5, 3), (8, 4, 2), (4, 2, 1), (4, 2, 1), (4, 2, 1))
PEAK = 2500.0 / 6.0

ap = argparse.ArgumentParser()
ap.add_argument("trace")
ap.add_argument("stamps", nargs="?")
ap.add_argument("--hidden", type=int, default=256)
ap.add_argument("--windows", type=int, default=128, help="windows through the encoder (2b with reference semantics)")
ap.add_argument("--label", default="")
args = ap.parse_args()

lens = [20480]
for k, s, p in CONV:
    lens.append((lens[-1] + 2 * p - k) // s + 1)
H, N = args.hidden, args.windows
flops = {i: 2.0 * N * lens[i + 1] * CONV[i][0] * H * H for i in range(1, 5)}

rows = list(csv.DictReader(open(glob.glob(args.trace + "/*/*kernel_trace.csv")[0])))
adam = [i for i, r in enumerate(rows) if "adam" in r["Kernel_Name"]]
seg = rows[adam[-2] + 1: adam[-1] + 1]
planes = [r for r in seg if "gemm_planes_kernel" in r["Kernel_Name"]]


def kind(r):
    m = re.search(r"gemm_planes_kernel<(\d+), (true|false), (\d+), (true|false)(?:, (true|false))?>", r["Kernel_Name"])
    tn, norm, pair = m.group(2) == "true", m.group(4) == "true", m.group(5) == "true"
    return tn, norm, pair


nt = [r for r in planes if not kind(r)[0]]
tn = [r for r in planes if kind(r)[0]]
assert len(nt) == 8 and len(tn) == 4, (len(nt), len(tn))
names_nt = [f"conv{i} forward" for i in (1, 2, 3, 4)] + [f"conv{i} backward data" for i in (4, 3, 2, 1)]
layer_nt = [1, 2, 3, 4, 4, 3, 2, 1]
names_tn = [f"conv{i} weight gradient" for i in (4, 3, 2, 1)]
layer_tn = [4, 3, 2, 1]

stamps = []
if args.stamps:
    for line in open(args.stamps):
        m = re.search(r"planes stamps: (\d+) tiles, loop (\d+) cycles = ([\d.]+) us per tile, clock ([\d.]+) GHz; prologue ([\d.]+) us, epilogue ([\d.]+) us", line)
        if m:
            stamps.append(m.groups())
    stamps = stamps[-8:]


def row(name, r, layer, st=None):
    us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tf = flops[layer] / us / 1e6
    g = [int(r[f"Grid_Size_{a}"]) // int(r[f"Workgroup_Size_{a}"]) for a in "XYZ"]
    tnk, norm, pair = kind(r)
    form = ("TN" if tnk else "NT") + (" pair" if pair else "") + (" + norm epilogue" if norm else "")
    if st and float(st[3]) > 3.0:            # a K-split launch: the stamps are per workgroup, not per tile
        extra = f" (K split: stamps not per tile) | | {st[4]} / {st[5]} |"
    else:
        extra = f" {st[3]} | {int(st[1]) / 1e3:.0f} k | {st[4]} / {st[5]} |" if st else " | | |"
    return f"| {name} | {form} | {g[0]}x{g[1]}x{g[2]} | {flops[layer] / 1e9:.1f} | {us:.1f} | {tf:.0f} | {tf / PEAK:.3f} |" + extra


print(f"| launch | form | grid | GFLOP | us | TFLOP/s | of {PEAK:.1f} | in-kernel clock (GHz) | loop cycles / tile | prologue / epilogue (us) |")
print("|---|---|---|---|---|---|---|---|---|---|")
tot_us = tot_fl = 0.0
for i, (name, r, layer) in enumerate(zip(names_nt, nt, layer_nt)):
    print(row(name, r, layer, stamps[i] if len(stamps) == 8 else None))
    tot_us += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot_fl += flops[layer]
print(f"| **NT, 8 launches** | | | {tot_fl / 1e9:.1f} | {tot_us:.1f} | {tot_fl / tot_us / 1e6:.0f} | **{tot_fl / tot_us / 1e6 / PEAK:.3f}** | | | |")
tu = tf_ = 0.0
for name, r, layer in zip(names_tn, tn, layer_tn):
    print(row(name, r, layer))
    tu += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tf_ += flops[layer]
print(f"| **TN, 4 launches** | | | {tf_ / 1e9:.1f} | {tu:.1f} | {tf_ / tu / 1e6:.0f} | **{tf_ / tu / 1e6 / PEAK:.3f}** | | | |")
