"""One-process-per-GPU plumbing for the multi-GPU benchmark / serving path.

Generations are independent units: ranks shard the request stream with NO data-path collective;
the only collectives are the barrier around the timed region and the max-over-ranks reduction of the wall time
(RCCL over xGMI on the GPU box — backend "nccl" IS RCCL on ROCm; gloo in the CPU tests).
"""
import os

import torch
import torch.distributed as dist


class Ctx:
    def __init__(self, rank, world, local, backend):
        self.rank, self.world, self.local, self.backend = rank, world, local, backend


def init(backend="nccl"):
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        kw = {}
        if backend == "nccl":
            kw["device_id"] = torch.device("cuda", local)
        dist.init_process_group(backend, **kw)
    return Ctx(rank, world, local, backend)


def barrier(ctx):
    if ctx.world > 1:
        dist.barrier()


def max_over_ranks(ctx, seconds):
    """Wall time of the slowest rank (the job finishes when the last rank does)."""
    if ctx.world == 1:
        return float(seconds)
    dev = torch.device("cuda", ctx.local) if ctx.backend == "nccl" else torch.device("cpu")
    t = torch.tensor([seconds], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def shard_seeds(ctx, steps):
    """Request ids handled by this rank: round-robin over the global stream (rank r takes r, r+W, r+2W, …)."""
    return [ctx.rank + i * ctx.world for i in range(steps)]


def total_units(ctx, steps_per_rank):
    return steps_per_rank * ctx.world


def finalize(ctx):
    if ctx.world > 1:
        dist.destroy_process_group()
