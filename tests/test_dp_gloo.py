"""Data-parallel glue on CPU: world_size 2, gloo.  Exercises cpc2_amd.train.DataParallelContext
(one parameter broadcast at start, one flat-gradient all-reduce per step, 1/world folded into the
optimiser) with a stand-in optimiser whose step is the oracle's Adam -- the communication pattern is
what is under test here; the fused HIP Adam itself is covered by the GPU tests."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from cpc2_amd.train import DataParallelContext
from oracle.cpc_oracle import Adam


class HostFlatOptimizer:
    """Same surface as FlatAdam (flat, flat_grad, step(grad_scale)) on CPU tensors."""

    def __init__(self, flat):
        self.flat = flat
        self.flat_grad = torch.zeros_like(flat)
        self._adam = Adam({"p": self.flat}, lr=1e-2)
        self.last_scale = None

    def step(self, grad_scale=1.0):
        if getattr(self, "direct_grads", False):
            self._gather_stray_grads()
        self.last_scale = grad_scale
        self._adam.step({"p": self.flat_grad * grad_scale})


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(100 + rank)                       # ranks start from DIFFERENT parameters
    opt = HostFlatOptimizer(torch.randn(1000))
    dp = DataParallelContext(opt)                       # broadcast from rank 0
    start = opt.flat.clone()
    g = torch.Generator().manual_seed(7 + rank)         # each rank has its own shard -> its own gradient
    grads = []
    for _ in range(3):
        opt.flat_grad.copy_(torch.randn(1000, generator=g))
        grads.append(opt.flat_grad.clone())
        dp.reduce_and_step()
    out[rank] = (start, opt.flat.clone(), opt.last_scale, grads)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_average_matches_single_process():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    (s0, p0, sc0, g0), (s1, p1, sc1, g1) = out[0], out[1]
    assert torch.equal(s0, s1), "parameters were not broadcast from rank 0"
    assert torch.equal(p0, p1), "ranks diverged after all-reduce + step"
    assert sc0 == sc1 == 0.5
    # single-process equivalent: the same three steps on the averaged per-rank gradients
    ref = s0.clone()
    adam = Adam({"p": ref}, lr=1e-2)
    for a, b in zip(g0, g1):
        adam.step({"p": (a + b) * 0.5})
    assert torch.allclose(p0, ref, atol=1e-6)


def test_world_size_one_needs_no_process_group():
    opt = HostFlatOptimizer(torch.ones(10))
    dp = DataParallelContext(opt)
    opt.flat_grad.fill_(1.0)
    dp.reduce_and_step()
    assert dp.world == 1 and opt.last_scale == 1.0


# ---- the overlapped form: early slices reduced from an autograd hook, the rest in reduce_and_step -----------------------
class HostParamOptimizer(HostFlatOptimizer):
    """+ the parameter bookkeeping of FlatAdam (params, offsets, direct_grads, _gather_stray_grads) on CPU tensors."""

    def __init__(self, params):
        self.params = list(params)
        self.offsets, off = [], 0
        for p in self.params:
            self.offsets.append(off)
            off += p.numel()
        flat = torch.cat([p.data.reshape(-1) for p in self.params]).clone()
        super().__init__(flat)
        for p, o in zip(self.params, self.offsets):
            p.data = self.flat[o:o + p.numel()].view(p.shape)
        self.direct_grads = True

    def _gather_stray_grads(self, ranges=None):
        for p, o in zip(self.params, self.offsets):
            if ranges is not None and not any(lo <= o < hi for lo, hi in ranges):
                continue
            if p.grad is not None:
                self.flat_grad[o:o + p.numel()].copy_(p.grad.reshape(-1))

    def zero_grad(self):
        self.flat_grad.zero_()
        for p in self.params:
            p.grad = None


def _overlap_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(5)                                # same start on every rank (and rank 0's is broadcast anyway)
    head = torch.nn.Parameter(torch.randn(6, 4))        # "criterion": first in the flat buffer
    enc = torch.nn.Parameter(torch.randn(4, 3))         # "encoder"
    ctx = torch.nn.Parameter(torch.randn(4, 4))         # "context network": last
    opt = HostParamOptimizer([head, enc, ctx])
    dp = DataParallelContext(opt, early_params=[head, ctx])
    assert dp.early == [(0, 24), (36, 52)] and dp.late == [(24, 36)]
    g = torch.Generator().manual_seed(50 + rank)
    seen = []
    for _ in range(3):
        x = torch.randn(5, 3, generator=g)
        z = x @ enc.t()                                 # encoder output
        dp.attach(z)
        loss = ((torch.tanh(z @ ctx.t()) @ head.t()) ** 2).sum()
        loss.backward()
        assert dp._fired and len(dp._pending) == 2
        seen.append(torch.cat([p.grad.reshape(-1) for p in opt.params]).clone())
        dp.reduce_and_step()
        opt.zero_grad()
    out[rank] = (opt.flat.clone(), seen)
    dist.barrier()
    dist.destroy_process_group()


def test_overlapped_early_reduction_equals_one_blocking_all_reduce():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_overlap_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    (p0, g0), (p1, g1) = out[0], out[1]
    assert torch.equal(p0, p1)
    # replay on one process: the averaged gradients of both ranks, the same Adam
    torch.manual_seed(5)
    start = torch.cat([torch.randn(6, 4).reshape(-1), torch.randn(4, 3).reshape(-1), torch.randn(4, 4).reshape(-1)])
    adam = Adam({"p": start}, lr=1e-2)
    # (gradients depend on the parameters, which are identical on both ranks at every step: the recorded ones are valid)
    for a, b in zip(g0, g1):
        adam.step({"p": (a + b) * 0.5})
    assert torch.allclose(p0, start, atol=1e-6)
