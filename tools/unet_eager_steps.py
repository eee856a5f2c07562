"""N eager (un-graphed) UNet CFG steps at a given batch — a rocprofv3 --kernel-trace --stats target."""
import argparse
import os
import sys

ROOT = os.environ.get("SX_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--steps", type=int, default=3)
a = ap.parse_args()
import bench
bench.BATCH = a.batch
bench.USE_VAE = False
dev = torch.device("cuda:0")
with torch.no_grad():
    try:
        _, _, adapter = bench.build_models(dev, torch.bfloat16, need=("adapter",))
    except TypeError:
        _, _, adapter = bench.build_models(dev, torch.bfloat16)
    adapter.use_graph = False
    adapter._loop.use_graph = False
    feats = torch.randn(a.batch, 64, 4096, device=dev).bfloat16()
    adapter.generate(image_embeds=feats, num_inference_steps=a.steps, seed=1, output_type="latent")
    torch.cuda.synchronize()
