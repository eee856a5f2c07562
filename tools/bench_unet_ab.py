"""Same-process A/B of the graph-replayed 50-step UNet loop at the bench batch, interleaved rounds: fused GroupNorm statistics on / off
(--what gn: ops.GN_FUSE) or LayerNorms folded into their neighbour GEMMs on / off (--what ln: unet.LN_FOLD; the process runs with
SX_LN_FOLD=1 so that the folded weight copies exist). Prints ms per 50-step loop (16 generations, CFG-2, two kernel chains) and, for ln,
the rel-L2 difference of the final latents between the two settings."""
import argparse
import os
import sys
import time

ROOT = os.environ.get("SX_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--steps", type=int, default=50)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--dtype", default="fp16")
ap.add_argument("--what", default="gn", choices=("gn", "ln"))
a = ap.parse_args()
if a.what == "ln":
    os.environ["SX_LN_FOLD"] = "1"
import bench
from seedx_amd import ops
from seedx_amd import unet as unet_mod
bench.BATCH, bench.USE_VAE = a.batch, False
dt = torch.float16 if a.dtype == "fp16" else torch.bfloat16
dev = torch.device("cuda:0")
with torch.no_grad():
    _, _, adapter = bench.build_models(dev, dt, need=("adapter",))
    adapter._loop.chains = 2
    feats = torch.randn(a.batch, 64, 4096, device=dev).to(dt)
    res = {0: [], 1: []}
    last = {}
    for r in range(a.rounds + 1):
        for fuse in (1, 0):
            if a.what == "gn":
                ops.GN_FUSE = bool(fuse)
            else:
                unet_mod.LN_FOLD = bool(fuse)
            adapter._loop._graph = None                      # re-capture with the other setting
            adapter.generate(image_embeds=feats, num_inference_steps=2, seed=1, output_type="latent")
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            lat = adapter.generate(image_embeds=feats, num_inference_steps=a.steps, seed=1, output_type="latent")
            torch.cuda.synchronize()
            last[fuse] = lat.detach().float().clone()
            if r:
                res[fuse].append((time.perf_counter() - t0) * 1e3)
    for fuse in (1, 0):
        v = sorted(res[fuse])
        print(f"{'GN_FUSE' if a.what == 'gn' else 'LN_FOLD'}={fuse}: median {v[len(v) // 2]:.1f} ms per {a.steps}-step loop (all: {', '.join('%.1f' % x for x in v)})")
    if a.what == "ln":
        x, y = [torch.as_tensor(last[k]).float() for k in (1, 0)]
        print(f"final latents, folded vs separate LayerNorm launches: rel-L2 {((x - y).norm() / y.norm()).item():.3e}")
