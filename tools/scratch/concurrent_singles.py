"""Two (or three) independent single-process training jobs at once on one GPU: do their results equal a solo run's, bit for bit?"""
import os, subprocess, sys, tempfile, torch, socket
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
job = os.path.join(ROOT, "tests", "dp_job.py")
tmp = tempfile.mkdtemp()
env = dict(os.environ, PYTHONPATH=ROOT)
solo = os.path.join(tmp, "solo.pt")
subprocess.check_call([sys.executable, job, "single", "0", "1", "0", solo], env=env, stdout=subprocess.DEVNULL, stderr=subprocess.STDOUT)
ref = torch.load(solo)
NP = int(os.environ.get("NPROC", 2))
bad = 0
for rep in range(int(os.environ.get("REPS", 12))):
    procs = []
    for i in range(NP):
        o = os.path.join(tmp, f"c{rep}_{i}.pt")
        procs.append((subprocess.Popen([sys.executable, job, "single", "0", "1", "0", o], env=env, stdout=subprocess.DEVNULL, stderr=subprocess.STDOUT), o))
    for pr, o in procs:
        pr.wait()
        got = torch.load(o)
        d = (got["flat"] - ref["flat"]).abs()
        if float(d.max()) > 0:
            nz = torch.nonzero(d > 0).view(-1)
            bad += 1
            print(f"rep {rep}: a concurrent single differs from the solo run: {nz.numel()} elements in [{int(nz.min())}, {int(nz.max())}], max {float(d.max()):.3e}; losses equal {torch.equal(got['losses'], ref['losses'])}")
print("concurrent runs that differ:", bad)
