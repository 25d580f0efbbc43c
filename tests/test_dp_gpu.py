"""Data parallelism on the real model: two ranks sharing cuda:0 over gloo against one process doing both
shards as micro-batches, and an RCCL (backend "nccl") process group of one rank.  The rank processes are started by
tests/conftest.py at session start (tests/dp_job.py); here their results are compared."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_two_ranks_equal_one_process_with_two_micro_batches(dp_jobs):
    r0, r1, single = dp_jobs["rank0"], dp_jobs["rank1"], dp_jobs["single"]
    assert r0["step_count"] == r1["step_count"] == single["step_count"] == 2
    # parameters were broadcast (rank 1 started from different ones) and stay identical on both ranks, bit for bit
    assert torch.equal(r0["flat"], r1["flat"])
    # the same update as one process that accumulates the two shards' gradients: identical up to fp32 reduction order
    err = float((r0["flat"] - single["flat"]).abs().max())
    scale = float(single["flat"].abs().max())
    assert err <= 2e-6 * scale, (err, scale)
    # each rank saw its own shard with its own negative stream: the single process' micro-batch losses, interleaved
    both = torch.stack([r0["losses"], r1["losses"]], dim=1).reshape(single["losses"].shape)
    assert torch.allclose(both, single["losses"], rtol=2e-5, atol=1e-6)
    assert not torch.allclose(r0["losses"], r1["losses"])


def test_rccl_process_group_of_one_rank(dp_jobs):
    """backend 'nccl' IS RCCL on ROCm: init, parameter broadcast, overlapped and blocking gradient all-reduce, Adam."""
    res = dp_jobs["nccl"]
    assert res["step_count"] == 3 and torch.isfinite(res["flat"]).all() and torch.isfinite(res["losses"]).all()
    # world 1: the same trajectory as rank 0's first two steps would have alone -- just check the steps moved the weights
    assert float(res["losses"][0].mean()) > 0


def test_reference_style_ddp_wrapping_with_flat_adam(dp_jobs):
    """cpc/train.py:523-527 as is: DistributedDataParallel around model and criterion, FlatAdam stepping the flat buffer
    the fused backward kernels write their gradients into.  Same update as the single process with two micro-batches."""
    d0, d1, single = dp_jobs["ddp0"], dp_jobs["ddp1"], dp_jobs["single"]
    assert d0["step_count"] == d1["step_count"] == 2
    assert torch.equal(d0["flat"], d1["flat"])
    err = float((d0["flat"] - single["flat"]).abs().max())
    scale = float(single["flat"].abs().max())
    assert err <= 2e-6 * scale, (err, scale)
    both = torch.stack([d0["losses"], d1["losses"]], dim=1).reshape(single["losses"].shape)
    assert torch.allclose(both, single["losses"], rtol=2e-5, atol=1e-6)
