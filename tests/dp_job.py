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


label = torch.zeros(B, dtype=torch.long, device=DEV)
model, crit, opt = build()
losses = []
if mode in ("ranks", "nccl"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("nccl" if mode == "nccl" else "gloo", rank=rank, world_size=world)
    if rank != 0:
        opt.flat.mul_(1.5)                     # the broadcast must bring rank 0's parameters
    overlap = not os.environ.get("CPC_DP_JOB_NO_OVERLAP")        # (diagnostics, tools/scratch/dp_diag.py: one blocking all-reduce)
    dp = DataParallelContext(opt, early_params=list(crit.parameters()) + list(model.gAR.parameters()), overlap=overlap)
    assert not overlap or (dp.early and dp.late), (dp.early, dp.late)
    crit.seed(1234 + rank)
    x = shard(rank)
    for _ in range(STEPS):
        tot, ls, _acc = cpcStep(x, x, label, model, crit, dp=dp)
        tot.backward()
        assert not overlap or (dp._fired and len(dp._pending) == len(dp.early))       # the early slices are on their way
        if os.environ.get("CPC_DP_JOB_DUMP") and not losses:                          # (diagnostics: this rank's own gradient, step 1)
            torch.cuda.synchronize()
            torch.save(opt.flat_grad.detach().cpu(), out + ".grad")
        dp.reduce_and_step()
        opt.zero_grad()
        losses.append(ls.detach().cpu())
    if mode == "nccl":                          # and the blocking form
        dp2 = DataParallelContext(opt, overlap=False)
        tot, ls, _acc = cpcStep(x, x, label, model, crit)
        tot.backward()
        dp2.reduce_and_step()
        opt.zero_grad()
    torch.cuda.synchronize()
    dist.barrier()
    dist.destroy_process_group()
elif mode == "ddp":
    from torch.nn.parallel import DistributedDataParallel as DDP
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    if rank != 0:
        opt.flat.mul_(1.5)                     # DDP's constructor broadcasts rank 0's parameters (into the flat buffer's views)
    ddp_model = DDP(model, device_ids=[0], find_unused_parameters=True)
    ddp_crit = DDP(crit, device_ids=[0], find_unused_parameters=True)
    crit.seed(1234 + rank)
    x = shard(rank)
    for _ in range(STEPS):
        tot, ls, _acc = cpcStep(x, x, label, ddp_model, ddp_crit)
        tot.backward()
        opt.step()
        opt.zero_grad()
        losses.append(ls.detach().cpu())
    torch.cuda.synchronize()
    dist.barrier()
    dist.destroy_process_group()
else:
    samplers = []
    for r in range(SHARDS):
        smp = NegativeSampler()
        smp.seed(1234 + r)
        samplers.append(smp)
    for _ in range(STEPS):
        for r in range(SHARDS):
            crit.sampler = samplers[r]
            x = shard(r)
            tot, ls, _acc = cpcStep(x, x, label, model, crit)
            tot.backward()
            losses.append(ls.detach().cpu())
        opt.step(grad_scale=1.0 / SHARDS)
        opt.zero_grad()
    torch.cuda.synchronize()
torch.save({"flat": opt.flat.detach().cpu(), "losses": torch.stack(losses), "step_count": opt.step_count}, out)
print("dp_job done", mode, rank, flush=True)
