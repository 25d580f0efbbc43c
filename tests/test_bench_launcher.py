"""`python bench.py --gpus N` starts its own N ranks (round-2 review item 2; the reference enters its multi-process path from its
own entry point too: cpc/train.py:291-295 -> cpc/distributed_training/distributed_mode.py:75-86,129-142).  CPU only: the
ranks rendezvous over gloo on 127.0.0.1, all-reduce one number and rank 0 prints it."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*flags, env=None):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], env=e, capture_output=True, text=True, timeout=300)


def test_bench_spawns_its_own_ranks():
    res = _run("--gpus", "3", "--rendezvous-only")
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout                         # ONE JSON line, from rank 0
    rec = json.loads(lines[0])
    assert rec == {"rendezvous": 3, "rank_sum": 6.0, "master": "127.0.0.1"}
    assert "started 3 ranks" in res.stderr


def test_bench_launcher_propagates_a_rank_failure():
    # no GPU here (and gloo asked for): every rank dies in torch.cuda.set_device -> the launcher must report non-zero
    import torch
    if torch.cuda.device_count() > 0:
        import pytest
        pytest.skip("needs a machine without a GPU")
    res = _run("--gpus", "2", "--steps", "1", "--warmup", "1", env={"CPC_BENCH_BACKEND": "gloo"})
    assert res.returncode != 0
    assert "stopping the others" in res.stderr


def test_bench_rejects_mismatched_world_size():
    res = _run("--gpus", "2", "--rendezvous-only", env={"WORLD_SIZE": "4", "RANK": "0"})
    assert res.returncode != 0 and "must agree" in res.stderr


def test_no_collective_behind_a_rank_zero_only_return():
    """Every collective of bench.py is entered by every rank.  Round 6 shipped -- for a few commits -- a diagnostic all-reduce behind
    `if rank != 0: return None`: invisible with one rank, a hang with two (tests/test_bench_gpu.py now launches two).  Static form of
    the same check, without a GPU: inside any function, no `dist.<collective>(...)` call sits on a later line than a return that only
    the non-zero ranks take."""
    import ast
    path = os.path.join(ROOT, "bench.py")
    tree = ast.parse(open(path).read())
    collectives = {"all_reduce", "barrier", "broadcast", "all_gather", "reduce", "reduce_scatter", "all_to_all", "gather", "scatter"}

    def rank_only_return_lines(fn):
        out = []
        for node in ast.walk(fn):
            if isinstance(node, ast.If) and isinstance(node.test, ast.Compare) and isinstance(node.test.left, ast.Name) \
                    and node.test.left.id == "rank" and any(isinstance(s, ast.Return) for s in node.body):
                out.append(node.lineno)
        return out

    found = []
    for fn in [n for n in ast.walk(tree) if isinstance(n, (ast.FunctionDef, ast.AsyncFunctionDef))]:
        cuts = rank_only_return_lines(fn)
        if not cuts:
            continue
        first = min(cuts)
        for node in ast.walk(fn):
            if isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute) and node.func.attr in collectives \
                    and isinstance(node.func.value, ast.Name) and node.func.value.id == "dist" and node.lineno > first:
                found.append((fn.name, node.lineno, node.func.attr))
    assert not found, f"collectives behind a rank-dependent return: {found}"
