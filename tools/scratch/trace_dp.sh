OUT=$GRAFT_REPO_ROOT/gpurun_out/kt_dp
cd /tmp && export TMPDIR=/tmp
export CPC_BENCH_FORCE_DIST=1
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 16 --warmup 4 --cpu-seconds 0 --no-prof --also "" > $OUT.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob
rows=list(csv.DictReader(open(glob.glob("gpurun_out/kt_dp/*/*kernel_trace.csv")[0])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
ad=[i for i,r in enumerate(rows) if "adam" in r["Kernel_Name"]]
print("steps", len(ad))
for a,b in zip(ad[:-1],ad[1:]):
    seg=rows[a+1:b+1]
    t0=int(rows[a]["End_Timestamp"]); t1=int(rows[b]["End_Timestamp"])
    busy=0; cur=t0; gaps=[]
    prev_end=t0
    for r in seg:
        s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
        if s>prev_end+20000: gaps.append((round((s-prev_end)/1e3), rows[rows.index(r)-1]["Kernel_Name"][:30], r["Kernel_Name"][:30]))
        prev_end=max(prev_end,e)
    names=set(r["Kernel_Name"].split("(")[0][:40] for r in seg)
    print(round((t1-t0)/1e3), "us; gaps>20us:", gaps[:6], [n for n in names if "nccl" in n.lower() or "rccl" in n.lower()][:3])
PY
