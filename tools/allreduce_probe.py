"""Cost of one flat-gradient all-reduce as train.DataParallelContext issues it (RCCL through torch.distributed), on
however many ranks torchrun started:   python -m torch.distributed.run --nproc-per-node N tools/allreduce_probe.py"""
import os
import time

import torch
import torch.distributed as dist

rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", 0)) % torch.cuda.device_count())
dist.init_process_group("nccl", init_method="env://", world_size=world, rank=rank)
for mb in (1, 10, 46):
    buf = torch.zeros(mb * 262144, device="cuda")
    work = torch.randn(4096, 4096, device="cuda")
    for _ in range(5):
        dist.all_reduce(buf)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        dist.all_reduce(buf)
    torch.cuda.synchronize()
    alone = (time.perf_counter() - t0) / 50
    # between two kernels of the training stream, as in a step
    t0 = time.perf_counter()
    for _ in range(50):
        work @ work
        dist.all_reduce(buf)
        work @ work
    torch.cuda.synchronize()
    mixed = (time.perf_counter() - t0) / 50
    t0 = time.perf_counter()
    for _ in range(50):
        work @ work
        work @ work
    torch.cuda.synchronize()
    base = (time.perf_counter() - t0) / 50
    if rank == 0:
        print(f"{mb:3d} MB x {world} ranks: back to back {alone * 1e6:7.1f} us, inside a stream of kernels +{(mixed - base) * 1e6:7.1f} us")
dist.destroy_process_group()
