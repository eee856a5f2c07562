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
            # one process per GPU: LOCAL_RANK must name a device this process can see — say so before RCCL fails obscurely
            n = torch.cuda.device_count()
            if not 0 <= local < n:
                raise RuntimeError(f"rank {rank}: LOCAL_RANK={local} but this process sees {n} GPU(s) "
                                   f"(HIP_VISIBLE_DEVICES={os.environ.get('HIP_VISIBLE_DEVICES')}, "
                                   f"ROCR_VISIBLE_DEVICES={os.environ.get('ROCR_VISIBLE_DEVICES')}): launch one rank per visible GPU")
            torch.cuda.set_device(local)
            kw["device_id"] = torch.device("cuda", local)
        dist.init_process_group(backend, **kw)
    return Ctx(rank, world, local, backend)


def device_identity(ctx):
    """What this rank runs on: (rank, local rank, host, device name, device uuid / PCI bus id) — a GPU rank's identity is unique per
    physical GPU, so two ranks bound to one device show up as duplicates in ranks_seen()."""
    import socket
    ident = {"rank": ctx.rank, "local_rank": ctx.local, "host": socket.gethostname(), "pid": os.getpid()}
    if ctx.backend == "nccl" and torch.cuda.is_available():
        pr = torch.cuda.get_device_properties(ctx.local)
        uuid = getattr(pr, "uuid", None)
        ident.update(device=pr.name, device_id=str(uuid) if uuid is not None else f"pci:{getattr(pr, 'pci_bus_id', '?')}:{ctx.local}",
                     cus=getattr(pr, "multi_processor_count", None))
    else:
        ident.update(device="cpu", device_id=f"cpu:{ident['host']}:{ident['pid']}")
    return ident


def ranks_seen(ctx, strict_devices=False):
    """All-gather of every rank's device_identity() over the job's process group: the proof in the bench line that the collective
    backend really connected `world` ranks, and on which device each sits. Raises when ranks are missing / duplicated. Two ranks that
    report the same device id are flagged ("shared_device": true on those entries + a warning on stderr) — fatal only with
    strict_devices: a runtime that reports no per-device uuid must not stop a correct 8-GPU run."""
    me = device_identity(ctx)
    if ctx.world == 1:
        return [me]
    got = [None] * ctx.world
    dist.all_gather_object(got, me)
    if sorted(g["rank"] for g in got) != list(range(ctx.world)):
        raise RuntimeError(f"ranks_seen: expected ranks 0..{ctx.world - 1}, the group returned {[g['rank'] for g in got]}")
    ids = [(g["host"], g["device_id"]) for g in got]
    if len(set(ids)) != ctx.world:
        msg = (f"ranks_seen: {ctx.world} ranks but only {len(set(ids))} distinct device ids: {ids} — every rank needs its own GPU "
               f"(check LOCAL_RANK / HIP_VISIBLE_DEVICES of the launcher)")
        if strict_devices:
            raise RuntimeError(msg)
        for g in got:
            g["shared_device"] = ids.count((g["host"], g["device_id"])) > 1
        if ctx.rank == 0:
            import sys
            print("WARNING " + msg, file=sys.stderr, flush=True)
    return got


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
