"""Run-to-run reproducibility of the UNet forward / denoise loop at the bench batch: the same inputs twice (and with the fused
statistics switched off) → rel-L2 and bit equality of the outputs. A last-bit statistics difference shows as ~1e-6; anything near
1e-3 is a race or an uninitialised read."""
import argparse
import os
import sys

ROOT = os.environ.get("SX_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--dtype", default="fp16")
a = ap.parse_args()
os.environ["SX_LN_FOLD"] = "1"
import bench
from seedx_amd import ops
from seedx_amd import unet as unet_mod
bench.BATCH, bench.USE_VAE = a.batch, False
dt = torch.float16 if a.dtype == "fp16" else torch.bfloat16
dev = torch.device("cuda:0")


def rel(x, y):
    return ((x.float() - y.float()).norm() / y.float().norm()).item()


with torch.no_grad():
    _, _, adapter = bench.build_models(dev, dt, need=("adapter",))
    unet = adapter.unet
    unet._pack()
    B = 2 * a.batch
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, 128 * 128, 4, generator=g).to(dev)
    ehs = torch.randn(B, 77, unet.cfg["cross_attention_dim"], generator=g).to(dev)
    pooled = torch.randn(B, 1280, generator=g).to(dev)
    tid = torch.tensor([[1024., 1024., 0., 0., 1024., 1024.]]).repeat(B, 1).to(dev)
    ts = torch.tensor([500.0], device=dev)
    ctx = unet.prepare_context(ehs)

    def fwd():
        temb = unet.time_embeddings(ts, None, pooled, tid, B)
        return unet.forward_nhwc(x, temb, ctx, B, 128, 128).clone()

    for name, gn, ln in (("GN_FUSE=1 LN_FOLD=1", True, True), ("GN_FUSE=0 LN_FOLD=1", False, True), ("GN_FUSE=1 LN_FOLD=0", True, False),
                         ("GN_FUSE=0 LN_FOLD=0", False, False)):
        ops.GN_FUSE, unet_mod.LN_FOLD = gn, ln
        outs = [fwd() for _ in range(3)]
        torch.cuda.synchronize()
        print(f"one forward, {name}: run 2 vs 1 rel-L2 {rel(outs[1], outs[0]):.3e} equal={torch.equal(outs[1], outs[0])}; "
              f"run 3 vs 1 {rel(outs[2], outs[0]):.3e} equal={torch.equal(outs[2], outs[0])}", flush=True)
        if name.startswith("GN_FUSE=1 LN_FOLD=1"):
            base = outs[0]
        else:
            print(f"    vs the default setting: rel-L2 {rel(outs[0], base):.3e}", flush=True)
    ops.GN_FUSE, unet_mod.LN_FOLD = True, True
    feats = torch.randn(a.batch, 64, 4096, device=dev).to(dt)
    for chains in (1, 2):
        adapter._loop.chains = chains
        adapter._loop._graph = None
        lats = []
        for r in range(3):
            lats.append(adapter.generate(image_embeds=feats, num_inference_steps=a.steps, seed=1, output_type="latent").float().clone())
        torch.cuda.synchronize()
        print(f"{a.steps}-step loop, chains={chains}: run 2 vs 1 rel-L2 {rel(lats[1], lats[0]):.3e} equal={torch.equal(lats[1], lats[0])}; "
              f"run 3 vs 2 {rel(lats[2], lats[1]):.3e}", flush=True)
    for gn, ln in ((False, False),):
        ops.GN_FUSE, unet_mod.LN_FOLD = gn, ln
        adapter._loop._graph = None
        lats = [adapter.generate(image_embeds=feats, num_inference_steps=a.steps, seed=1, output_type="latent").float().clone() for _ in range(2)]
        print(f"{a.steps}-step loop, chains=2, GN_FUSE=0 LN_FOLD=0: run 2 vs 1 rel-L2 {rel(lats[1], lats[0]):.3e} equal={torch.equal(lats[1], lats[0])}", flush=True)
