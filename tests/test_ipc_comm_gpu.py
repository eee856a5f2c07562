"""One-shot all-reduce / all-gather over hipIpc-mapped peer buffers (csrc/comm.hip, seedx_amd.parallel.IpcComm) with 2, 4 and 8
PROCESSES ON ONE GPU — the only way the 1-GPU pool can exercise the real cross-process protocol (IPC handle exchange, epoch
flags sized by the world, system-scope visibility, HIP-graph capture of the collective) at the node's rank counts.
Multi-GPU timing over xGMI remains unmeasured.
  * all-reduce: bit-identical to the rank-ordered sum (= ThreadComm's `parts[0] + parts[1]`) for payloads from 4 B to the
    staging capacity, 300 back-to-back epochs (slot reuse), then captured into a HIP graph and replayed
  * all-gather through the same staging (fp32, and 16-bit payloads as 32-bit words)
  * the row-sharded UNet's conv halo rows as a NEIGHBOUR-ONLY exchange (mode 1: a rank signals / waits for rank ± 1) and
    seqpar.with_halo's one-launch slab packing
  * tensor-parallel Llama (tp = 2 / 4 / 8: 8 heads → 4 / 2 / 1 per rank): prefill + graph-replayed decode steps with their all-reduces INSIDE the graph; logits and
    greedy ids equal the single-rank run's tokens and agree with the oracle"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
sys.path.insert(0, os.environ["SX_ROOT"])
import torch
import torch.distributed as dist
from seedx_amd.parallel import Comm, IpcComm

torch.cuda.set_device(0)                                   # both ranks share the one GPU
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
dev = torch.device("cuda:0")
# bounded polls (≈ 1 s): if the processes' kernels were not co-scheduled on the shared GPU the test fails instead of hanging the box
comm = IpcComm(None, cap_floats=131072, device=dev, max_spin=1 << 20)
assert comm.graph_safe and comm.world == world == int(os.environ["SX_WORLD"])
EPOCHS = 300 if world == 2 else 105

def payload(r, it, n):
    g = torch.Generator().manual_seed(1000 * it + r)
    return torch.randn(n, generator=g)

# ---- all-reduce, many epochs, many sizes ------------------------------------------------------------------------------
sizes = [1, 7, 4098, 5120, 8191, 16 * 5120, 131072]
for it in range(EPOCHS):
    n = sizes[it % len(sizes)]
    parts = [payload(r, it, n) for r in range(world)]
    t = parts[rank].to(dev)
    comm.all_reduce(t)
    exp = parts[0].to(dev)
    for q in parts[1:]:
        exp = exp + q.to(dev)                              # rank order, like ThreadComm.all_reduce
    assert torch.equal(t, exp), f"rank {rank} it {it} n {n}: max diff {(t - exp).abs().max().item()}"
comm.check()

# ---- all-gather ---------------------------------------------------------------------------------------------------------
for it in range(20):
    parts = [payload(r, 7000 + it, 4000).view(8, 500) for r in range(world)]
    out = comm.all_gather(parts[rank].to(dev))
    assert out.shape == (world, 8, 500) and torch.equal(out.cpu(), torch.stack(parts))

# 16-bit payloads (the row-sharded UNet's conv halo rows) travel as 32-bit words through the same kernel
for it in range(10):
    parts = [payload(r, 8000 + it, 2 * 2 * 64 * 320).view(2, 2, 64, 320).to(torch.bfloat16) for r in range(world)]
    out = comm.all_gather(parts[rank].to(dev))
    assert out.dtype == torch.bfloat16 and out.shape == (world, 2, 2, 64, 320) and torch.equal(out.cpu(), torch.stack(parts))

# ---- neighbour-only halo exchange (sx_oneshot_args.mode 1) + the fused slab packer (sx_halo_pack) ---------------------------
from seedx_amd import seqpar
def edge_rows(r, it, B, W, C):
    g = torch.Generator().manual_seed(50000 + 100 * it + r)
    return torch.randn(2, B, W, C, generator=g).to(torch.bfloat16)            # (first row, last row) of rank r's slab
for it in range(40):
    B, W, C = ((2, 64, 320), (3, 32, 640), (1, 128, 320), (2, 16, 1280))[it % 4]
    mine = edge_rows(rank, it, B, W, C).to(dev)
    got = comm.halo_exchange(mine)
    exp = torch.zeros_like(mine)
    if rank > 0:
        exp[0] = edge_rows(rank - 1, it, B, W, C)[1].to(dev)
    if rank < world - 1:
        exp[1] = edge_rows(rank + 1, it, B, W, C)[0].to(dev)
    assert torch.equal(got, exp), f"halo exchange rank {rank} it {it}"
assert comm._halo is not None and comm._halo.neighbor_only
comm._halo.check()
# with_halo: slab [B, Hl*W, C] -> [B, Hl + 2, W (+1), C] with the neighbours' rows / zero borders, in one packing launch
B, Hl, W, C = 2, 4, 16, 320
slab = lambda r: torch.randn(B, Hl * W, C, generator=torch.Generator().manual_seed(777 + r)).to(torch.bfloat16)
for left_col, bottom in ((False, True), (True, False)):
    out = seqpar.with_halo(slab(rank).to(dev), comm, Hl, W, left_col=left_col, bottom=bottom)
    off = 1 if left_col else 0
    exp = torch.zeros(B, Hl + 1 + (1 if bottom else 0), W + off, C, dtype=torch.bfloat16)
    exp[:, 1:Hl + 1, off:] = slab(rank).view(B, Hl, W, C)
    if rank > 0:
        exp[:, 0, off:] = slab(rank - 1).view(B, Hl, W, C)[:, -1]
    if bottom and rank < world - 1:
        exp[:, Hl + 1, off:] = slab(rank + 1).view(B, Hl, W, C)[:, 0]
    assert torch.equal(out.cpu(), exp), f"with_halo rank {rank} left_col={left_col}"

# ---- the collective inside a HIP graph -----------------------------------------------------------------------------------
buf = torch.zeros(16 * 5120, device=dev)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    comm.all_reduce(buf)                                   # warm-up on the capture stream
torch.cuda.synchronize()
dist.barrier()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s):
    comm.all_reduce(buf)
for it in range(10):
    parts = [payload(r, 9000 + it, buf.numel()) for r in range(world)]
    buf.copy_(parts[rank])
    g.replay()
    torch.cuda.synchronize()
    exp = parts[0]
    for q in parts[1:]:
        exp = exp + q
    assert torch.equal(buf.cpu(), exp), f"graph replay {it}"
comm.check()

# ---- tensor-parallel Llama over the one-shot collectives, decode step captured with its all-reduces -----------------------
from oracle import restated, weights
from seedx_amd.llama import LlamaForCausalLM
cfg = dict(hidden_size=1024, intermediate_size=1024, num_hidden_layers=3, num_attention_heads=8, vocab_size=500,
           rms_norm_eps=1e-5, max_position_embeddings=512)
dt = torch.float16
sd = weights.llama_sd(cfg)
xe = torch.randn(1, 21, cfg["hidden_size"], generator=torch.Generator().manual_seed(3)) * 0.5
lref, _, href = restated.llama_forward(sd, cfg, xe, table_dtype=dt)
img_ids = torch.arange(400, 466, dtype=torch.int32, device=dev)

def run(c, use_graph):
    llm = LlamaForCausalLM(dict(cfg), max_cache_len=64, max_batch=1, comm=c)
    llm.load_state_dict(dict(sd))
    llm.eval().to(dev, dt)
    out = llm(inputs_embeds=xe.to(dev), output_hidden_states=True)
    ids = torch.full((1, 8), -1, dtype=torch.int32, device=dev)
    hid = torch.zeros((1, 8, cfg["hidden_size"]), device=dev)
    llm._P["cur"].fill_(int(out["logits"][0, -1].argmax()))
    for _ in range(5):
        llm.decode_step(img_ids, ids, hid, use_graph=use_graph)
    torch.cuda.synchronize()
    return out["logits"][0, -1].float().cpu(), ids.cpu(), llm

single_logits, single_ids, _ = run(Comm(), True)
dist.barrier()
tp_logits, tp_ids, llm = run(comm, True)
comm.check()
assert llm._graph is not None, "the TP decode step must have been captured (IpcComm.graph_safe)"
rel = ((tp_logits[:500] - lref[0, -1]).norm() / lref[0, -1].norm()).item()
assert rel < 3e-3, rel
assert torch.equal(tp_ids, single_ids), (tp_ids, single_ids)
every = [None] * world
dist.all_gather_object(every, (tp_logits, tp_ids))
assert all(torch.equal(every[0][0], e[0]) and torch.equal(every[0][1], e[1]) for e in every[1:]), "ranks disagree"
dist.barrier()
comm.close()
print(f"RANK {rank} OK logits rel-L2 vs oracle {rel:.2e}", flush=True)
dist.destroy_process_group()
'''


@pytest.mark.parametrize("world", [2, 4, 8])
def test_oneshot_collectives_n_processes_on_one_gpu(tmp_path, world):
    w = tmp_path / "ipc_worker.py"
    w.write_text(WORKER)
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, SX_ROOT=ROOT, SX_WORLD=str(world), OMP_NUM_THREADS="2", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
               PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), str(w)], env=env, capture_output=True, text=True, timeout=600)
    print(r.stdout[-3000:])
    assert r.returncode == 0, r.stdout[-3000:] + "\n" + r.stderr[-6000:]
    assert all(f"RANK {k} OK" in r.stdout for k in range(world))
