"""Run tests/dp_job.py 'single' several times and a 2-rank pair; print where results differ."""
import os, subprocess, sys, tempfile, torch, socket
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
job = os.path.join(ROOT, "tests", "dp_job.py")
tmp = tempfile.mkdtemp()
def port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return str(p)
env = dict(os.environ, PYTHONPATH=ROOT, CPC_DP_JOB_DUMP="1")
outs = []
for i in range(3):
    o = os.path.join(tmp, f"s{i}.pt")
    subprocess.check_call([sys.executable, job, "single", "0", "1", port(), o], env=env, stdout=subprocess.DEVNULL, stderr=subprocess.STDOUT)
    outs.append(torch.load(o))
for i in range(1, 3):
    d = (outs[i]["flat"] - outs[0]["flat"]).abs()
    print(f"single run {i} vs 0: max diff {float(d.max()):.3e}, elements differing {int((d > 0).sum())} of {d.numel()}, losses equal {torch.equal(outs[i]['losses'], outs[0]['losses'])}")
for rep in range(int(os.environ.get('REPS', 8))):
    p = port()
    procs = []
    for r in range(2):
        o = os.path.join(tmp, f"r{rep}_{r}.pt")
        procs.append((subprocess.Popen([sys.executable, job, "ranks", str(r), "2", p, o], env=env, stdout=open(o + ".log", "w"), stderr=subprocess.STDOUT), o))
    res = []
    for pr, o in procs:
        if pr.wait() != 0: print(open(o + '.log').read()[-1500:])
        res.append(torch.load(o))
    d = (res[0]["flat"] - outs[0]["flat"]).abs()
    idx = int(d.argmax())
    for rk in range(2):
        gk = torch.load(os.path.join(tmp, f"r{rep}_{rk}.pt.grad"))
        if rep == 0:
            globals()[f"gref{rk}"] = gk
        else:
            dd = (gk - globals()[f"gref{rk}"]).abs()
            if float(dd.max()) > 0:
                nzz = torch.nonzero(dd > 0).view(-1)
                print(f"   rank {rk} LOCAL gradient of rep {rep} differs from rep 0's: {nzz.numel()} elements in [{int(nzz.min())}, {int(nzz.max())}], max {float(dd.max()):.3e}")
    both = torch.stack([res[0]["losses"], res[1]["losses"]], dim=1).reshape(outs[0]["losses"].shape)
    nz = torch.nonzero(d > 0).view(-1)
    print(f"ranks rep {rep}: r0==r1 {torch.equal(res[0]['flat'], res[1]['flat'])}; vs single max diff {float(d.max()):.3e} at {idx}; differing {nz.numel()} in [{int(nz.min()) if nz.numel() else -1}, {int(nz.max()) if nz.numel() else -1}]; losses equal {torch.equal(both, outs[0]['losses'])} (max {float((both - outs[0]['losses']).abs().max()):.2e})")
# ---- the DDP arrangement
for rep in range(int(os.environ.get('REPS_DDP', 0))):
    p = port()
    procs = []
    for r in range(2):
        o = os.path.join(tmp, f"d{rep}_{r}.pt")
        procs.append((subprocess.Popen([sys.executable, job, "ddp", str(r), "2", p, o], env=env, stdout=open(o + ".log", "w"), stderr=subprocess.STDOUT), o))
    res = []
    for pr, o in procs:
        if pr.wait() != 0: print(open(o + '.log').read()[-1500:])
        res.append(torch.load(o))
    d = (res[0]["flat"] - outs[0]["flat"]).abs()
    print(f"ddp rep {rep}: r0==r1 {torch.equal(res[0]['flat'], res[1]['flat'])}; vs single max diff {float(d.max()):.3e}")
