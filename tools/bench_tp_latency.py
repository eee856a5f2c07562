"""Latency mode: ONE generation spread over the GPUs of a node (north-star's tensor-parallel form).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port 29533 \
        tools/bench_tp_latency.py --steps 3

  * Llama decoder: Megatron TP over all N ranks (seedx_amd.parallel / llama.py): 2 fp32 all-reduces per layer over RCCL
  * SDXL UNet: CFG-parallel on ranks 0 and 1 (one eps all-gather per denoise step); ranks >= 2 idle in that phase
  * ViT + resamplers: replicated (8.4 TFLOP, < 1 % of a generation)

NOT the headline metric: replicas (bench.py) give the higher gens/s because generations are independent; this mode
trades throughput for single-request latency. The sharded code path is validated on one GPU with virtual ranks
(tests/test_tensor_parallel_gpu.py); this launcher itself could not be exercised on the single-GPU boxes of this pool.
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--unet-steps", type=int, default=50)
    ap.add_argument("--text-tokens", type=int, default=61)
    a = ap.parse_args()
    import torch.distributed as dist
    from seedx_amd import dist_utils as du
    from seedx_amd.parallel import Comm, TorchDistComm
    ctx = du.init("nccl")
    torch.cuda.set_device(ctx.local)
    dev = torch.device("cuda", ctx.local)
    llm_comm = TorchDistComm() if ctx.world > 1 else Comm()
    cfg_group = dist.new_group([0, 1]) if ctx.world >= 2 else None          # collective call: every rank executes it
    cfg_comm = TorchDistComm(cfg_group) if (ctx.world >= 2 and ctx.rank < 2) else None
    bench.BATCH = 1
    tok = bench.BenchTokenizer()
    with torch.no_grad():
        vit, agent, adapter = bench.build_models(dev, torch.bfloat16, llm_comm=llm_comm, cfg_comm=cfg_comm)
        inp = bench.make_inputs(dev)

        def one(seed):
            feats = bench.front_half(vit, agent, tok, inp, a.text_tokens, dev)
            if ctx.world < 2 or ctx.rank < 2:
                bench.back_half(adapter, feats, a.unet_steps, seed)
        for i in range(a.warmup):
            one(100 + i)
        torch.cuda.synchronize(); du.barrier(ctx); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(a.steps):
            one(i)
        torch.cuda.synchronize(); du.barrier(ctx); torch.cuda.synchronize()
        dt = du.max_over_ranks(ctx, time.perf_counter() - t0)
    if ctx.rank == 0:
        print(json.dumps({"metric": "single-request latency (img-in -> txt + 1024px latents)", "value": dt / a.steps,
                          "unit": "s/generation", "n_gpus": ctx.world, "steps": a.steps, "warmup": a.warmup,
                          "higher_is_better": False, "scaling": "strong", "dtype": "bf16",
                          "config": {"workload": "as bench.py, batch 1", "parallelism":
                                     "llama tp%d (RCCL all-reduce), unet cfg-parallel x%d" % (ctx.world, min(ctx.world, 2))}}))
    du.finalize(ctx)


if __name__ == "__main__":
    main()
