"""Latency mode: ONE generation spread over the GPUs of a node (the north-star's sharded form).

    python tools/bench_tp_latency.py --gpus N [--steps 3] [--config 0|4]     (starts the N ranks itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port 29533 \
        tools/bench_tp_latency.py --steps 3                                   (or under an external launcher)

  * Llama decoder: Megatron tensor parallelism over all N ranks (seedx_amd.parallel / llama.py): heads, FFN rows and
    vocabulary split, 2 fp32 all-reduces per layer. `--llm-comm ipc` (default): the one-shot all-reduce over hipIpc peer
    buffers (csrc/comm.hip, IpcComm) — one kernel launch per collective, so the decode step is captured into a HIP graph
    WITH its 80 all-reduces; the 1.7-M-float prefill payloads exceed its staging and go to RCCL. `--llm-comm rccl`: every
    all-reduce an RCCL call, decode step eager (RCCL calls are not captured on this stack).
  * SDXL UNet: pixel-row sharding over all N ranks (seedx_amd/seqpar.py): weights replicated, one K|V all-gather per
    self-attention overlapped with the Q projection on RCCL's side stream, conv halo rows, GroupNorm statistics
    all-reduce; `--unet cfg` selects the older CFG-parallel split (2 ranks busy) instead. `--config 4` runs the edit
    pipeline (Bc = 3, 8 input channels) — BASELINE config 4 at TP = N.
  * ViT + resamplers + VAE: replicated (< 2 % of a generation)

NOT the headline metric: replicas (bench.py) give the higher gens/s because generations are independent; this mode
trades throughput for single-request latency. The sharded code paths are validated on one GPU with virtual ranks
(tests/test_tensor_parallel_gpu.py) and on CPU over gloo (tests/test_cpu_suite.py); this pool's boxes have ONE GPU, so the
RCCL / xGMI timing of this launcher has not been measured. `--share-gpu` runs the N ranks as N processes on GPU 0 over a
gloo bootstrap (IpcComm for the decoder, gloo for everything else): a functional check of the launcher and of the graph-replayed
tensor-parallel decode on the 1-GPU pool, not a performance number (the ranks time-slice one GPU).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--unet-steps", type=int, default=50)
    ap.add_argument("--text-tokens", type=int, default=61)
    ap.add_argument("--config", type=int, default=0, choices=[0, 4])
    ap.add_argument("--unet", default="rows", choices=["rows", "cfg"])
    ap.add_argument("--llm-comm", default="ipc", choices=["ipc", "rccl"])
    ap.add_argument("--unet-comm", default="rccl", choices=["ipc", "rccl"],
                    help="ipc: the conv halo rows of the row-sharded UNet go through the one-shot kernel (K|V gathers and fp64 "
                         "GroupNorm sums stay on the process group)")
    ap.add_argument("--share-gpu", action="store_true", help="all ranks on GPU 0, gloo bootstrap (functional check on a 1-GPU box)")
    a = ap.parse_args()
    if a.gpus and a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr",
               "127.0.0.1", "--master-port", str(bench._free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.run(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")).returncode)
    import torch.distributed as dist
    from seedx_amd import dist_utils as du
    from seedx_amd.parallel import Comm, IpcComm, TorchDistComm
    if a.share_gpu:
        os.environ["LOCAL_RANK"] = "0"
    ctx = du.init("gloo" if a.share_gpu else "nccl")
    if a.gpus and ctx.world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but {ctx.world} rank(s) came up")
    torch.cuda.set_device(ctx.local)
    dev = torch.device("cuda", ctx.local)
    multi = ctx.world > 1
    llm_comm = (IpcComm(None, device=dev) if a.llm_comm == "ipc" else TorchDistComm()) if multi else Comm()
    unet_comm = cfg_comm = None
    edit = a.config == 4
    if multi and a.unet == "rows":
        unet_comm = IpcComm(None, device=dev, graph_safe=False) if a.unet_comm == "ipc" else TorchDistComm()
    elif multi:
        nb = 3 if edit else 2
        assert ctx.world >= nb, "CFG-parallel needs one rank per guidance branch"
        grp = dist.new_group(list(range(nb)))                                  # collective call: every rank executes it
        cfg_comm = TorchDistComm(grp) if ctx.rank < nb else None
    # the image's crops are split over the ranks for the ViT (SURVEY.md §8e "preferred"): features all-gathered on the bootstrap group
    vit_comm = TorchDistComm() if multi else None
    bench.BATCH = 1
    tok = bench.BenchTokenizer()
    with torch.no_grad():
        vit, agent, adapter = bench.build_models(dev, torch.bfloat16, llm_comm=llm_comm, cfg_comm=cfg_comm,
                                                 unet_comm=unet_comm, edit=edit)
        inp = bench.make_inputs(dev, extra_text=16 if edit else 0)
        src = torch.randn(1, 4, 128, 128, generator=torch.Generator().manual_seed(7))
        in_back = a.unet == "rows" or not multi or cfg_comm is not None

        def one(seed):
            feats = bench.front_half(vit, agent, tok, inp, 8 if edit else a.text_tokens, dev, vit_comm=vit_comm)
            if in_back:
                bench.back_half(adapter, feats, a.unet_steps, seed, **({"image_latents": src} if edit else {}))
        for i in range(a.warmup):
            one(100 + i)
        torch.cuda.synchronize(); du.barrier(ctx); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(a.steps):
            one(i)
        torch.cuda.synchronize(); du.barrier(ctx); torch.cuda.synchronize()
        dt = du.max_over_ranks(ctx, time.perf_counter() - t0)
        # the decoder's decode step on its own: graph replay with the collectives inside (ipc) or eager (rccl)
        llm = agent.llm
        ids = torch.full((1, 130), -1, dtype=torch.int32, device=dev)
        hid = torch.zeros((1, 130, llm.config.hidden_size), device=dev)
        img_ids = torch.arange(llm.V - 200, llm.V - 134, dtype=torch.int32, device=dev)
        graphed = bool(getattr(llm_comm, "graph_safe", True))
        llm._P["cur"].fill_(5)
        llm._P["step"].zero_()                   # rows of ids / hid; the KV position continues from the last generation (~500 keys)
        for _ in range(3):
            llm.decode_step(img_ids, ids, hid, use_graph=graphed)
        torch.cuda.synchronize(); du.barrier(ctx)
        t0 = time.perf_counter()
        for _ in range(100):
            llm.decode_step(img_ids, ids, hid, use_graph=graphed)
        torch.cuda.synchronize()
        tok_ms = du.max_over_ranks(ctx, time.perf_counter() - t0) * 10.0
        if isinstance(llm_comm, IpcComm):
            llm_comm.check()
    if ctx.rank == 0:
        print(json.dumps({"metric": "single-request latency (%s)" % ("edit: img+instruction -> edited 1024px image" if edit
                                                                     else "img-in -> txt + 1024px image"),
                          "value": dt / a.steps, "unit": "s/generation", "n_gpus": ctx.world, "steps": a.steps,
                          "warmup": a.warmup, "higher_is_better": False, "scaling": "strong", "dtype": "bf16",
                          "llm_decode_ms_per_token": tok_ms, "llm_decode_graph_replayed": graphed and llm._graph is not None,
                          "share_gpu": bool(a.share_gpu),
                          "config": {"workload": "as bench.py --config %d, batch 1" % a.config, "parallelism":
                                     "llama tp%d (%s), unet %s" % (ctx.world, "one-shot IPC all-reduce in the decode graph, RCCL for prefill"
                                                                   if a.llm_comm == "ipc" else "RCCL all-reduce", "pixel-row sharded x%d (K|V all-gather)"
                                                                                 % ctx.world if a.unet == "rows" else "cfg-parallel")}}))
    du.finalize(ctx)


if __name__ == "__main__":
    main()
