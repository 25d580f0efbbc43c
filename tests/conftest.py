import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return load


# ---- multi-process GPU jobs --------------------------------------------------------------------------------------
# tests/test_dp_gpu.py compares a 2-rank data-parallel run of the real model with a single process doing the same work,
# and checks an RCCL process group of one rank.  Every rank is its own python process on cuda:0; they are started HERE,
# when the session starts -- before this process has made any GPU call -- and the tests only collect their results.
_DP = {}


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


DP_TESTS = ("test_two_ranks_equal_one_process_with_two_micro_batches", "test_rccl_process_group_of_one_rank",
            "test_reference_style_ddp_wrapping_with_flat_adam", "test_reference_ddp_arguments_do_not_defer_the_criterion_backward",
            "test_reference_ddp_wrapping_on_rccl_takes_the_streaming_recurrent_kernels")


def _dp_tests_selected(config):
    """Will this session run anything of tests/test_dp_gpu.py?  (Decided from the command line alone: the jobs must be
    started before collection imports anything that could touch the GPU.)"""
    markexpr = config.getoption("markexpr", "") or ""
    if "not gpu" in markexpr or os.environ.get("CPC_SKIP_DP_JOBS"):
        return False
    files = [a.split("::")[0] for a in config.args]
    if files and not any(os.path.isdir(f) or os.path.basename(f) == "test_dp_gpu.py" for f in files):
        return False
    explicit = [a.split("::", 1)[1] for a in config.args if "::" in a and os.path.basename(a.split("::")[0]) == "test_dp_gpu.py"]
    if any("::" in a for a in config.args) and not explicit and not any(os.path.isdir(f) for f in files):
        return False
    keyword = config.getoption("keyword", "") or ""
    if keyword:
        try:
            from _pytest.mark.expression import Expression
            expr = Expression.compile(keyword)
            return any(expr.evaluate(lambda word, name=name: word in name or word in "test_dp_gpu.py") for name in DP_TESTS)
        except Exception:
            return True
    return True


def pytest_sessionstart(session):
    import subprocess
    import tempfile
    if not _dp_tests_selected(session.config):
        return
    try:
        import torch
        if torch.cuda.device_count() < 1:      # (counting devices does not initialise the GPU)
            return
    except Exception:
        return
    tmp = tempfile.mkdtemp(prefix="cpc_dp_")
    job = os.path.join(ROOT, "tests", "dp_job.py")
    port, port1, port2, port3 = str(_free_port()), str(_free_port()), str(_free_port()), str(_free_port())
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = {}
    # groups, one after the other (a GPU box allows six processes on its card at once, this session included): the ranks of
    # a group are started together by a launcher process that never touches the GPU itself
    groups = [{"rank0": ("ranks", 0, 2, port), "rank1": ("ranks", 1, 2, port), "single": ("single", 0, 1, port),
               "nccl": ("nccl", 0, 1, port1)},
              {"ddpnccl": ("ddpnccl", 0, 1, str(_free_port()))},
              {"ddp0": ("ddp", 0, 2, port2), "ddp1": ("ddp", 1, 2, port2)},
              {"ddpref0": ("ddpref", 0, 2, port3), "ddpref1": ("ddpref", 1, 2, port3)}]
    plan = []
    for group in groups:
        cmds = []
        for name, (mode, rank, world, prt) in group.items():
            out = os.path.join(tmp, name + ".pt")
            log = os.path.join(tmp, name + ".log")
            cmds.append(([sys.executable, job, mode, str(rank), str(world), prt, out], log, os.path.join(tmp, name + ".rc")))
            procs[name] = (None, out, log)
        plan.append(cmds)
    launcher = ("import subprocess, sys, json\n"
                "plan = json.loads(sys.argv[1])\n"
                "for cmds in plan:\n"
                "    ps = [(subprocess.Popen(c, stdout=open(l, 'w'), stderr=subprocess.STDOUT), rc) for c, l, rc in cmds]\n"
                "    for p, rc in ps:\n"
                "        open(rc, 'w').write(str(p.wait()))\n")
    import json
    # its own session (= process group): a time-out can then stop the launcher AND the rank processes it started
    proc = subprocess.Popen([sys.executable, "-c", launcher, json.dumps(plan)], env=env, cwd=ROOT, start_new_session=True)
    for name in procs:
        procs[name] = (proc, procs[name][1], procs[name][2])
    _DP.update(procs)


@pytest.fixture(scope="session")
def dp_jobs():
    """name -> loaded result of tests/dp_job.py (waits for the processes started at session start)."""
    import signal
    import torch
    if not _DP:
        pytest.skip("data-parallel jobs were not started (no GPU, or -m 'not gpu')")
    results = {}
    launcher = next(iter(_DP.values()))[0]
    try:
        launcher.wait(timeout=600)
    except Exception:
        try:
            os.killpg(launcher.pid, signal.SIGKILL)         # exactly the process group created above: launcher + its ranks
        except OSError:
            pass
        launcher.wait()
        raise AssertionError("the data-parallel jobs did not finish:\n" + "\n".join(open(l).read()[-1500:] for _p, _o, l in _DP.values()))
    for name, (_proc, out, log) in _DP.items():
        rc_path = out[:-3] + ".rc"
        rc = int(open(rc_path).read()) if os.path.exists(rc_path) else -1
        assert rc == 0, f"dp job {name} failed ({rc}):\n" + open(log).read()[-3000:]
        results[name] = torch.load(out)
    keep = os.environ.get("CPC_DP_KEEP")                    # diagnostics: keep the records (e.g. under gpurun_out/)
    if keep:
        import shutil
        os.makedirs(keep, exist_ok=True)
        for name, (_proc, out, log) in _DP.items():
            shutil.copy(out, keep)
            shutil.copy(log, keep)
    return results
