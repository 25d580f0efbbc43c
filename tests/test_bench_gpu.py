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
                        "--also", ""], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = r.stdout.strip().splitlines()[-1]
    # every line this module measures is kept where gpurun merges it back from: a failure is then read from the record, not re-run
    keep = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(keep, exist_ok=True)
        tag = "_".join(f"{k[4:].lower()}{v}" for k, v in sorted(env_extra.items()) if k != "CPC_BENCH_NOTE") or "plain"
        if "CPC_BENCH_NOTE" in env_extra:
            tag += "_warming"
        with open(os.path.join(keep, f"test_bench_gpu_{tag}.jsonl"), "a") as fh:
            fh.write(line + "\n")
    except OSError:
        pass
    return json.loads(line)


_ALONE = ("conv0_fwd", "conv0_bwd", "gru_fwd", "infonce_fwd", "infonce_bwd")


def _same_box_conditions(a, b, what):
    """The two runs are separate processes on one GPU of a shared host.  Neither mode can change how long a KERNEL takes, so kernel
    times that differ between them mean the box changed under the measurement (seen twice in ~60 runs: every kernel 7-37 % slower
    for a whole run, the latency- and memory-bound ones most: `profiles/r06_process_group_queues.md` section 5) and a step-time
    comparison of the two says nothing about the library.  Compared on the five timed classes that run with nothing beside them
    (conv0 both ways, the recurrent forward, the criterion's two kernels: 1.00-1.03 ms per step together, +-1.5 % run to run on a
    quiet box, +20 % in the disturbed runs; the GEMM classes' times move by 5-10 % with what happens to run beside them and are
    not used).  The structural assertions (stream placement, hold times, host CPU time) do not depend on this and are always made."""
    ka, kb = (sum(r["kernels"][n]["ms_per_step"] for n in _ALONE) for r in (a, b))
    if abs(ka / kb - 1.0) <= 0.04:
        return True
    import warnings
    msg = (f"{what}: the kernels that run alone took {ka:.3f} against {kb:.3f} ms per step -- the box changed under the "
           f"measurement; step times not compared ({a['host']['step_ms_median']:.3f} against {b['host']['step_ms_median']:.3f} ms)")
    warnings.warn(msg)
    try:
        with open(os.path.join(ROOT, "gpurun_out", "test_bench_gpu_not_compared.txt"), "a") as fh:
            fh.write(msg + "\n")
    except OSError:
        pass
    return False


def test_one_rank_process_group_step_costs_about_a_plain_step():
    # This module is the first thing that runs on a fresh box: the first bench process pages everything in (torch, the 42 MB of
    # code objects, kernels loaded at their first launch) and has been seen at 23 ms per step with single steps of 90 ms, the
    # host descheduled for 17 ms per step (thread CPU 1.0 ms).  One discarded run first; its line is kept with the others.
    _bench({"CPC_BENCH_NOTE": "cache warming run of tests/test_bench_gpu.py (discarded)"})
    plain = _bench({})
    dist = _bench({"CPC_BENCH_FORCE_DIST": "1"})
    assert dist["comm"]["process_group"] == "nccl" and plain["comm"]["process_group"] is None
    # Round 5 recorded 5.86 against 4.9 ms per step here once, every kernel at its usual time (profiles/r05_bench_nccl1.json), and
    # this test re-measured before failing.  The cause (profiles/r06_process_group_queues.md): the library's side stream shared
    # the training stream's HARDWARE QUEUE -- assigned by use count at creation, crowded by the process group's streams -- so the
    # deferred backward work (1.1 ms per step) ran in front of the backward pass instead of beside it.  The library's streams are
    # now made on queues that were observed to run beside the caller's, and the line says so: no second measurement.
    for rec in (plain, dist):
        host = rec["host"]
        assert host["side_stream_runs_beside_training_stream"] is True, host
        assert host["streams_handed_out_untested"] == 0, host
        # the side stream's work ran BESIDE the backward pass: the training stream stood still ~0.02 ms per step at its joins
        # (its whole length, ~1.1 ms, when it does not)
        assert host["training_stream_held_by_side_stream_ms_per_step"] < 0.3, host
    # The box is one GPU of a shared host: a step of the 16 now and then takes 2-3 ms longer (recorded: `step_ms_max`,
    # `step_ms_max_index`, `gpu_clock.other_gpus_busy_max`), and that is not what this test is about -- the modes are compared on the
    # MEDIAN step; measured 1.01-1.03 (hooks + two collectives of one rank)
    if _same_box_conditions(dist, plain, "one RCCL rank against no process group"):
        ratio = dist["host"]["step_ms_median"] / plain["host"]["step_ms_median"]
        assert ratio < 1.1, f"process-group mode {dist['host']} against plain {plain['host']}"
        # ... and on what a step takes BEYOND its kernels (gaps on the training stream: what a queue collision or a starved host
        # adds while every kernel keeps its time -- round 5's 5.86 ms run had +1 ms of it); measured 0.05-0.15 ms for the hooks
        beyond = [r["host"]["step_ms_median"] - r["host"]["timed_kernel_classes_ms_per_step"] for r in (dist, plain)]
        # (the GEMM classes' own times move by +-0.2 ms with what runs beside them: the bar is half the defect's size)
        assert beyond[0] - beyond[1] < 0.5, (beyond, dist["host"], plain["host"])
    # the fields an 8-GPU run will be read by (there is no multi-GPU node to measure a scaling curve on): what the exchange holds
    # the compute stream for per step -- with one rank the wire costs nothing, so this is the floor of the collectives themselves
    comm = dist["comm"]
    assert comm["early_bytes"] > 0 and comm["late_bytes"] > 0 and comm["early_bytes"] + comm["late_bytes"] == comm["gradient_bytes"]
    assert comm["exposed_ms_per_step"] is not None and 0 <= comm["exposed_ms_per_step"] < 0.2, comm
    assert comm["rank_ms_per_step_min"] <= comm["rank_ms_per_step_max"] <= dist["ms_per_step"] * 1.01
    assert plain["comm"]["exposed_ms_per_step"] is None
    # what a step costs the host: far below the step itself, so that eight ranks on one host do not starve their GPUs
    # (CPU time of the training thread: wall time inside step() also counts whatever descheduled the thread on a shared host)
    for rec in (plain, dist):
        assert rec["host"]["thread_cpu_ms_per_step"] < 0.6 * rec["host"]["step_ms_median"], rec["host"]


def test_a_rank_on_two_cores_keeps_its_gpu_fed():
    """Eight ranks share one host: a rank gets a couple of cores.  Pinned to two (process-wide, before any GPU call) the step
    costs what it costs on sixteen."""
    free = _bench({"CPC_BENCH_FORCE_DIST": "1"})
    two = _bench({"CPC_BENCH_FORCE_DIST": "1", "CPC_BENCH_PIN_CORES": "2"})
    assert two["host"]["pinned"] and two["host"]["cores"] == 2
    if _same_box_conditions(two, free, "two cores against sixteen"):
        assert two["host"]["step_ms_median"] < 1.05 * free["host"]["step_ms_median"], (two["host"], free["host"])
    assert two["host"]["thread_cpu_ms_per_step"] < 0.6 * two["host"]["step_ms_median"], two["host"]


def test_the_drivers_launch_of_two_ranks_prints_one_line():
    """`python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2` exactly as the driver starts an N-GPU run, rehearsed
    with two gloo ranks SHARING this box's one GPU (plumbing, not a performance number): every collective of the bench is entered by
    every rank -- in round 6 a diagnostic all-reduce behind rank 0's early return hung this very command -- and rank 0 prints one
    line for the whole job."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0", CPC_BENCH_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2",
                        "--also", "large"], env=env, capture_output=True, text=True, timeout=420)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 4 and rec["scaling"] == "weak"
    assert rec["config"]["global_batch"] == 128 and rec["comm"]["world"] == 2 and rec["comm"]["process_group"] == "gloo"
    assert rec["value"] > 0 and abs(rec["value"] - 2 * 64 * 1.28 / (rec["ms_per_step"] * 1e-3)) < 0.01 * rec["value"]
    assert rec["comm"]["early_bytes"] + rec["comm"]["late_bytes"] == rec["comm"]["gradient_bytes"]
    assert "rccl_stream_runs_beside_training_stream" in rec["host"]
    assert [o["config"]["workload"][:9] for o in rec["other_configs"]] == ["CPC-large"] and "error" not in rec["other_configs"][0]
