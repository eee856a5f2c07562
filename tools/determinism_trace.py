"""Where do two runs of the SAME UNet forward part? The fp32 residual stream at every progress mark (after every resnet, in the
middle and at the end of every transformer layer) of two eager forwards on identical inputs, rel-L2 per mark."""
import argparse
import os
import sys

ROOT = os.environ.get("SX_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--dtype", default="fp16")
ap.add_argument("--gn-fuse", type=int, default=1)
ap.add_argument("--ln-fold", type=int, default=1)
a = ap.parse_args()
os.environ["SX_LN_FOLD"] = "1"
import bench
from seedx_amd import ops
from seedx_amd import unet as unet_mod
bench.BATCH, bench.USE_VAE = a.batch, False
dt = torch.float16 if a.dtype == "fp16" else torch.bfloat16
dev = torch.device("cuda:0")
ops.GN_FUSE, unet_mod.LN_FOLD = bool(a.gn_fuse), bool(a.ln_fold)


def rel(x, y):
    return ((x - y).norm() / (y.norm() + 1e-30)).item()


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
    traces, outs = [], []
    for r in range(2):
        unet._trace = []
        temb = unet.time_embeddings(ts, None, pooled, tid, B)
        outs.append(unet.forward_nhwc(x, temb, ctx, B, 128, 128).float().clone())
        traces.append(unet._trace)
        unet._trace = None
    torch.cuda.synchronize()
    print(f"GN_FUSE={a.gn_fuse} LN_FOLD={a.ln_fold} batch {a.batch}: output rel-L2 run 2 vs 1: {rel(outs[1], outs[0]):.3e}")
    prev = 0.0
    for i, (u, v) in enumerate(zip(*traces)):
        d = rel(v, u)
        mx = (v - u).abs().max().item()
        print(f"mark {i + 1:3d} {tuple(u.shape)}: rel-L2 {d:.3e}  max abs diff {mx:.3e}  |x|max {u.abs().max().item():.3e} rms {u.pow(2).mean().sqrt().item():.3e}"
              f"{'   <-- x%.1f' % (d / prev) if prev > 0 and d > 3 * prev else ''}")
        prev = d
