"""A bounded number of fresh runs of the self-diagnosing data-parallel jobs (tests/dp_job.py), started exactly as
tests/conftest.py starts them, each compared with tests/test_dp_gpu.py's own checks.  One line per run; the records of any run
with a finding are kept under <out>/run<i>/ so that the finding can be read again later.

    python tools/dp_pair_repeat.py [runs=10] [out=gpurun_out/dp_pair_repeat]

This process never touches the GPU (torch is used for torch.load and host arithmetic only)."""
import json
import os
import shutil
import socket
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return str(p)


def main():
    runs = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    out_dir = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", "dp_pair_repeat")
    os.makedirs(out_dir, exist_ok=True)
    import torch
    import test_dp_gpu as T
    job = os.path.join(ROOT, "tests", "dp_job.py")
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
    log = open(os.path.join(out_dir, "summary.jsonl"), "a")
    findings = 0
    for i in range(runs):
        tmp = tempfile.mkdtemp(prefix="cpc_dp_rep_")
        p0, p1, p2 = free_port(), free_port(), free_port()
        groups = [{"rank0": ("ranks", 0, 2, p0), "rank1": ("ranks", 1, 2, p0), "single": ("single", 0, 1, p0), "nccl": ("nccl", 0, 1, p1)},
                  {"ddp0": ("ddp", 0, 2, p2), "ddp1": ("ddp", 1, 2, p2)}]
        res, rcs = {}, {}
        for group in groups:
            procs = {}
            for name, (mode, rank, world, port) in group.items():
                o = os.path.join(tmp, name + ".pt")
                procs[name] = (subprocess.Popen([sys.executable, job, mode, str(rank), str(world), port, o], env=env,
                                                stdout=open(os.path.join(tmp, name + ".log"), "w"), stderr=subprocess.STDOUT), o)
            for name, (p, o) in procs.items():
                try:
                    rcs[name] = p.wait(timeout=300)
                except subprocess.TimeoutExpired:
                    p.kill()
                    rcs[name] = -9
                if rcs[name] == 0:
                    res[name] = torch.load(o)
        rec = {"run": i, "rc": rcs}
        if all(v == 0 for v in rcs.values()):
            try:
                for who in res:
                    T._verify_record(res[who], who)
                rec["records_ok"] = True
            except AssertionError as e:
                rec["records_ok"] = str(e)
            k, t = T._compare((res["rank0"], res["rank1"]), res["single"])
            kd, td = T._compare((res["ddp0"], res["ddp1"]), res["single"], scale=0.5)
            rec.update(kernel=k, transport=t, ddp_kernel=kd, ddp_transport=td,
                       ranks_equal=bool(torch.equal(res["rank0"]["flat"], res["rank1"]["flat"])),
                       nccl_pre_equals_rank0=bool(torch.equal(res["nccl"]["pre"][0], res["rank0"]["pre"][0])),
                       end_to_end_max=float((res["rank0"]["flat"] - res["single"]["flat"]).abs().max()),
                       ddp_end_to_end_max=float((res["ddp0"]["flat"] - res["single"]["flat"]).abs().max()))
        bad = any(rec.get(key) for key in ("kernel", "transport", "ddp_kernel", "ddp_transport")) or rec.get("records_ok") is not True
        if bad and "single" in res:
            findings += 1
            # which rank, which elements, what values: a small record instead of the 15 MB of tensors
            detail = {}
            single = res["single"]
            for pair, names_ in (("rank", ("rank0", "rank1")), ("ddp", ("ddp0", "ddp1"))):
                for step in range(single["pre"].shape[0]):
                    m0 = single["micro"][step][0]
                    m1 = single["micro"][step][1] - m0
                    for k, (nm, ref) in enumerate(zip(names_, (m0, m1))):
                        got = res[nm]["pre"][step]
                        d = (got - ref).abs()
                        tol = 4e-6 * float(single["pre"][step].abs().max())
                        idx = torch.nonzero(d > tol).view(-1)
                        if idx.numel():
                            per_param = {}
                            for j in idx.tolist():
                                nmp = T._where(single["names"], j)
                                per_param[nmp] = per_param.get(nmp, 0) + 1
                            detail[f"{nm}.step{step}"] = {
                                "count": int(idx.numel()), "first": int(idx.min()), "last": int(idx.max()), "per_param": per_param,
                                "max_abs_diff": float(d.max()), "ref_absmax": float(ref.abs().max()),
                                "idx": idx[:24].tolist(), "got": [float(v) for v in got[idx[:24]]], "ref": [float(v) for v in ref[idx[:24]]],
                                "ratio_got_over_ref_median": float((got[idx] / ref[idx]).median())}
            rec["detail"] = detail
        shutil.rmtree(tmp, ignore_errors=True)
        log.write(json.dumps(rec) + "\n")
        log.flush()
        print(json.dumps({k: v for k, v in rec.items() if k != "detail"})[:700], flush=True)
    print(f"runs {runs}, runs with a finding {findings}")


if __name__ == "__main__":
    main()
