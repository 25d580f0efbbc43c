"""bench.py on the GPU box: the one-rank process-group mode (what every rank of an N-GPU run does, minus the wire) must cost what
its hooks and collectives cost, not more.  A lowest-priority side stream once made every kernel ~45 % slower as soon as RCCL was
initialised in the process (profiles/r03_dist_priority_bisect.txt); nothing but a timing run shows that kind of defect."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(env_extra):
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0", **env_extra)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "16", "--warmup", "6", "--cpu-seconds", "0",
                        "--also", "", "--no-prof"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_one_rank_process_group_step_costs_about_a_plain_step():
    plain = _bench({})
    dist = _bench({"CPC_BENCH_FORCE_DIST": "1"})
    assert dist["comm"]["process_group"] == "nccl" and plain["comm"]["process_group"] is None
    ratio = dist["ms_per_step"] / plain["ms_per_step"]
    if ratio >= 1.2:
        # one 16-step run on a box that has just come up can be off by itself (seen once: 6.31 against 5.23 ms, then 5.17 against
        # 5.09 on the next box); the defect this test exists for does not go away on a second measurement
        plain = _bench({})
        dist = _bench({"CPC_BENCH_FORCE_DIST": "1"})
        ratio = dist["ms_per_step"] / plain["ms_per_step"]
    # measured 1.02-1.03 (hooks + two collectives of one rank); the defect this guards against measured 1.45
    assert ratio < 1.2, f"process-group mode {dist['ms_per_step']} ms per step against {plain['ms_per_step']} plain"
    # the fields an 8-GPU run will be read by (there is no multi-GPU node to measure a scaling curve on): what the exchange holds
    # the compute stream for per step -- with one rank the wire costs nothing, so this is the floor of the collectives themselves
    comm = dist["comm"]
    assert comm["early_bytes"] > 0 and comm["late_bytes"] > 0 and comm["early_bytes"] + comm["late_bytes"] == comm["gradient_bytes"]
    assert comm["exposed_ms_per_step"] is not None and 0 <= comm["exposed_ms_per_step"] < 0.2, comm
    assert comm["rank_ms_per_step_min"] <= comm["rank_ms_per_step_max"] <= dist["ms_per_step"] * 1.01
    assert plain["comm"]["exposed_ms_per_step"] is None
