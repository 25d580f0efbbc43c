"""One rank of the data-parallel equivalence job (tests/test_dp_gpu.py); started as a fresh process BEFORE the test
session touches the GPU.

    python tests/dp_job.py <mode> <rank> <world> <port> <out.pt>

mode "ranks":  `world` processes share cuda:0, backend gloo; rank r trains on ITS shard of the windows with ITS negative
               stream (the real model, FlatAdam, DataParallelContext with the overlapped early reduction) for 2 steps.
mode "single": one process does the same 2 steps on the union, shard by shard (two micro-batches whose gradients add up,
               Adam with grad_scale 1/2): SURVEY 8(e)'s equivalence.
mode "nccl":   world_size 1 over RCCL: init, broadcast, overlapped + blocking all-reduce through DataParallelContext.
mode "ddp":    the reference's own arrangement (cpc/train.py:523-527): model and criterion wrapped in
               torch.nn.parallel.DistributedDataParallel (gloo, two ranks on cuda:0), FlatAdam as the optimiser -- DDP's
               buckets average the gradients that the fused backward kernels wrote into the flat buffer.
mode "ddpnccl": that arrangement on an RCCL process group of ONE rank at hidden 256 (where the bare model takes the cooperative GRU
               kernels): cpcStep must switch the process to the streaming recurrent kernels (DDP all-reduces the criterion's
               bucket -- an RCCL kernel -- while the recurrent backward runs), the asynchronous error word stays clean, and the
               update is the bare model's under the same policy.
mode "ddpref": the same with the reference's own DDP arguments (find_unused_parameters left False: the wrappers then hand
               tensor attributes through, which is how a criterion that keyed its deferred backward on one would have let
               DDP's reducer read predictor gradients that a side stream had not written yet -- round-3 advisor finding).

Self-diagnosing (round-2 review): every rank keeps, per step, an ON-STREAM clone of its own gradient taken right before each
all-reduce (DataParallelContext.trace, no device-wide synchronisation anywhere near it) and of the buffer of sums after the
last one; the single process keeps its accumulated gradient per step.  Every record travels with a checksum computed on the
device, so a bad device-to-host copy of the record itself cannot pass for a finding.  tests/test_dp_gpu.py compares them.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
mode, rank, world, port, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]

import torch                                    # noqa: E402
import torch.distributed as dist                # noqa: E402

import cpc2_amd                                 # noqa: E402
from cpc2_amd.criterion import NegativeSampler  # noqa: E402
from cpc2_amd.train import DataParallelContext, buildOptimizer, cpcStep  # noqa: E402
from oracle import synth                        # noqa: E402

DEV = torch.device("cuda:0")
HIDDEN, B, K, NNEG, STEPS, SHARDS = 64, 2, 12, 16, 2, 2
if mode == "ddpnccl":
    HIDDEN, NNEG = 256, 32


def build():
    mp = synth.encoder_params(HIDDEN, 21)
    mp.update(synth.gru_params(HIDDEN, HIDDEN, 1, 22))
    model = cpc2_amd.CPCModel(cpc2_amd.CPCEncoder(HIDDEN), cpc2_amd.CPCAR(HIDDEN, HIDDEN, False, 1))
    model.load_state_dict(mp)
    crit = cpc2_amd.CPCUnsupersivedCriterion(K, HIDDEN, HIDDEN, NNEG, rnnMode="linear", sizeInputSeq=128)
    crit.load_state_dict(synth.predictor_params(K, HIDDEN, HIDDEN, 23))
    model, crit = model.to(DEV), crit.to(DEV)
    return model, crit, buildOptimizer(model, crit, lr=1e-3)


def shard(r):
    return synth.audio_windows(B, 20480, 100 + r).to(DEV)


def checksum(t):
    """64-bit sum of the fp32 bit patterns, computed on the device."""
    return int(t.contiguous().view(torch.int32).sum(dtype=torch.int64).item())


class Tape:
    """Per step: `pre` (this rank's gradient, assembled from the slices seen right before each all-reduce) and `post` (the
    sums), as on-stream device copies."""

    def __init__(self, n):
        self.n, self.pre, self.post = n, [], []

    def begin(self):
        self.pre.append(torch.full((self.n,), float("nan"), device=DEV))
        self.post.append(None)

    def __call__(self, kind, lo, hi, view):
        if kind == "pre":
            self.pre[-1][lo:hi].copy_(view)
        else:
            self.post[-1] = view.clone()

    def export(self):
        torch.cuda.synchronize()
        out = {}
        for name, seq in (("pre", self.pre), ("post", self.post)):
            out[name] = torch.stack([t.cpu() for t in seq])
            out[name + "_sum"] = [checksum(t) for t in seq]
        return out


label = torch.zeros(B, dtype=torch.long, device=DEV)
model, crit, opt = build()
names = [(off, n) for off, (n, _p) in zip(opt.offsets, list(crit.named_parameters()) + list(model.named_parameters()))]
tape = Tape(opt.flat_grad.numel())
losses, after_backward = [], []
if mode in ("ranks", "nccl"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("nccl" if mode == "nccl" else "gloo", rank=rank, world_size=world)
    if rank != 0:
        opt.flat.mul_(1.5)                     # the broadcast must bring rank 0's parameters
    overlap = not os.environ.get("CPC_DP_JOB_NO_OVERLAP")        # (diagnostics: one blocking all-reduce)
    dp = DataParallelContext.for_modules(opt, model, crit, overlap=overlap, trace=tape)
    assert not overlap or (dp.early and dp.late), (dp.early, dp.late)
    crit.seed(1234 + rank)
    x = shard(rank)
    for _ in range(STEPS):
        tape.begin()
        tot, ls, _acc = cpcStep(x, x, label, model, crit, dp=dp)
        tot.backward()
        assert not overlap or (dp._fired and len(dp._pending) == len(dp.early))       # the early slices are on their way
        if dp.late:
            lo, hi = dp.late[0][0], dp.late[-1][1]
            after_backward.append(opt.flat_grad[lo:hi].clone())      # the encoder's slice when backward has returned (on stream)
        dp.reduce_and_step()
        opt.zero_grad()
        losses.append(ls.detach().cpu())
    if mode == "nccl":                          # and the blocking form
        dp2 = DataParallelContext(opt, overlap=False, trace=tape)
        tape.begin()
        tot, ls, _acc = cpcStep(x, x, label, model, crit)
        tot.backward()
        dp2.reduce_and_step()
        opt.zero_grad()
    torch.cuda.synchronize()
    dist.barrier()
    dist.destroy_process_group()
elif mode in ("ddp", "ddpref"):
    from torch.distributed.algorithms.ddp_comm_hooks import default_hooks
    from torch.nn.parallel import DistributedDataParallel as DDP
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    if rank != 0:
        opt.flat.mul_(1.5)                     # DDP's constructor broadcasts rank 0's parameters (into the flat buffer's views)
    unused = mode == "ddp"
    ddp_model = DDP(model, device_ids=[0], find_unused_parameters=unused)
    ddp_crit = DDP(crit, device_ids=[0], find_unused_parameters=unused)
    home = {id(p): off for p, off in zip(opt.params, opt.offsets)}

    def tap_then_allreduce(_state, bucket):
        # what DDP does by default (divide by the world size, all-reduce the bucket), after an on-stream copy of this rank's
        # gradients as the bucket holds them
        for g, p in zip(bucket.gradients(), bucket.parameters()):
            off = home[id(p)]
            tape.pre[-1][off:off + p.numel()].copy_(g.reshape(-1))
        return default_hooks.allreduce_hook(None, bucket)

    ddp_model.register_comm_hook(None, tap_then_allreduce)
    ddp_crit.register_comm_hook(None, tap_then_allreduce)
    crit.seed(1234 + rank)
    x = shard(rank)
    for _ in range(STEPS):
        tape.begin()
        tot, ls, _acc = cpcStep(x, x, label, ddp_model, ddp_crit)
        tot.backward()
        from cpc2_amd import criterion as _cm
        assert not _cm._deferred, "a criterion inside DistributedDataParallel must not defer its backward"
        opt._gather_stray_grads()
        tape("post", 0, opt.flat_grad.numel(), opt.flat_grad)         # DDP's averages, back in the parameters' gradients
        opt.step()
        opt.zero_grad()
        losses.append(ls.detach().cpu())
    torch.cuda.synchronize()
    dist.barrier()
    dist.destroy_process_group()
elif mode == "ddpnccl":
    from torch.nn.parallel import DistributedDataParallel as DDP
    from cpc2_amd import _lib
    lib = _lib.load()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world))
    x = shard(rank)
    # (a) the bare model under the streaming policy, no process group: the update the wrapped run must reproduce
    lib.cpc_coop_set_policy(1)
    crit.seed(1234)
    bare = []
    for _ in range(STEPS):
        tot, ls, _acc = cpcStep(x, x, label, model, crit, strict=True)
        tot.backward()
        opt.step()
        opt.zero_grad()
        bare.append(ls.detach().cpu())
    bare_flat = opt.flat.detach().clone()
    assert lib.cpc_coop_launches() == 0
    # (b) the bare model under the default policy takes the cooperative kernels at this width
    lib.cpc_coop_set_policy(0)
    model, crit, opt = build()
    tot, _ls, _acc = cpcStep(x, x, label, model, crit)
    tot.backward()
    opt.zero_grad()
    coop_bare = lib.cpc_coop_launches()
    assert coop_bare >= 2, coop_bare
    # (c) the reference's arrangement on RCCL: fresh modules, DDP around both
    model, crit, opt = build()
    dist.init_process_group("nccl", rank=rank, world_size=world)
    ddp_model = DDP(model, device_ids=[0])
    ddp_crit = DDP(crit, device_ids=[0])
    crit.seed(1234)
    for _ in range(STEPS):
        tape.begin()
        tot, ls, _acc = cpcStep(x, x, label, ddp_model, ddp_crit)
        tot.backward()
        opt._gather_stray_grads()
        tape("pre", 0, opt.flat_grad.numel(), opt.flat_grad)
        tape("post", 0, opt.flat_grad.numel(), opt.flat_grad)
        opt.step()
        opt.zero_grad()
        losses.append(ls.detach().cpu())
    _lib.check(lib.cpc_async_error_check(_lib.stream_ptr(DEV)), "async error check")
    extra = {"policy_after": int(lib.cpc_coop_set_policy(-1)), "coop_launches_wrapped": int(lib.cpc_coop_launches() - coop_bare),
             "coop_launches_bare": int(coop_bare), "bare_losses": torch.stack(bare), "bare_flat": bare_flat.cpu()}
    torch.cuda.synchronize()
    dist.barrier()
    dist.destroy_process_group()
else:
    samplers, micro = [], []
    for r in range(SHARDS):
        smp = NegativeSampler()
        smp.seed(1234 + r)
        samplers.append(smp)
    for _ in range(STEPS):
        tape.begin()
        for r in range(SHARDS):
            crit.sampler = samplers[r]
            x = shard(r)
            tot, ls, _acc = cpcStep(x, x, label, model, crit)
            tot.backward()
            losses.append(ls.detach().cpu())
            opt._gather_stray_grads()
            micro.append(opt.flat_grad.clone())                   # accumulated after shard r (on stream)
        opt._gather_stray_grads()
        tape("pre", 0, opt.flat_grad.numel(), opt.flat_grad)          # the accumulated gradient of both shards
        tape("post", 0, opt.flat_grad.numel(), opt.flat_grad)
        opt.step(grad_scale=1.0 / SHARDS)
        opt.zero_grad()
    torch.cuda.synchronize()
result = {"flat": opt.flat.detach().cpu(), "flat_sum": checksum(opt.flat), "losses": torch.stack(losses),
          "step_count": opt.step_count, "names": names, "mode": mode}
result.update(tape.export())
if mode == "ddpnccl":
    result.update(extra)
if mode == "single":
    result["micro"] = torch.stack([m.cpu() for m in micro]).view(STEPS, SHARDS, -1)     # [step][shard]: accumulated so far
if after_backward:
    result["after_backward"] = torch.stack([t.cpu() for t in after_backward])
torch.save(result, out)
print("dp_job done", mode, rank, flush=True)
