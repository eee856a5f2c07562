"""Latency of the one-shot all-reduce (csrc/comm.hip) with N processes sharing GPU 0 — the protocol cost (launch + publish + flag
round trip + rank-ordered reduce) without a fabric hop; xGMI adds its link latency per peer on a multi-GPU node (unmeasured here).
    python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 tools/bench_oneshot_allreduce.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from seedx_amd.parallel import IpcComm

torch.cuda.set_device(0)
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
comm = IpcComm(None, cap_floats=131072)
for n, what in ((5120, "1 sequence x 5120 fp32 = 20 KB (batch-1 decode)"), (16 * 5120, "16 x 5120 fp32 = 320 KB (lock-step batch 16)"), (131072, "512 KB")):
    t = torch.randn(n, device="cuda")
    for _ in range(20):
        comm.all_reduce(t)
    torch.cuda.synchronize()
    dist.barrier()
    # eager launches
    t0 = time.perf_counter()
    for _ in range(200):
        comm.all_reduce(t)
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / 200 * 1e6
    # 80 all-reduces (one decode token of the 40-layer model) as one graph replay
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        comm.all_reduce(t)
    torch.cuda.synchronize()
    dist.barrier()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        for _ in range(80):
            comm.all_reduce(t)
    g.replay()
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(10):
        g.replay()
    torch.cuda.synchronize()
    graphed = (time.perf_counter() - t0) / 800 * 1e6
    comm.check()
    if rank == 0:
        print(f"{what}: {eager:.1f} us per eager call, {graphed:.1f} us per call inside a graph of 80 ({world} ranks on one GPU)", flush=True)
dist.barrier()
comm.close()
dist.destroy_process_group()
