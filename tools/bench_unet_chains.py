"""Same-process A/B of the graph-replayed UNet loop at the bench batch: number of concurrent kernel chains x start stagger
(detokenizer._DenoiseLoop.chains / .stagger), interleaved rounds. Prints ms per loop and the final-latents difference against the
first setting (the chains only re-partition independent samples: latents must agree to the fused-statistics noise)."""
import argparse
import os
import sys
import time

ROOT = os.environ.get("SX_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--rounds", type=int, default=2)
ap.add_argument("--dtype", default="fp16")
ap.add_argument("--settings", default="2:0,2:1,2:3,2:5,4:0,4:3,1:0", help="comma list of chains:stagger")
a = ap.parse_args()
import bench
bench.BATCH, bench.USE_VAE = a.batch, False
dt = torch.float16 if a.dtype == "fp16" else torch.bfloat16
dev = torch.device("cuda:0")
settings = [tuple(int(v) for v in s.split(":")) for s in a.settings.split(",")]
with torch.no_grad():
    _, _, adapter = bench.build_models(dev, dt, need=("adapter",))
    feats = torch.randn(a.batch, 64, 4096, device=dev).to(dt)
    res = {s: [] for s in settings}
    last = {}
    for r in range(a.rounds + 1):
        for s in settings:
            adapter._loop.chains, adapter._loop.stagger = s
            adapter._loop._graph = None
            adapter.generate(image_embeds=feats, num_inference_steps=2, seed=1, output_type="latent")
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            lat = adapter.generate(image_embeds=feats, num_inference_steps=a.steps, seed=1, output_type="latent")
            torch.cuda.synchronize()
            last[s] = lat.detach().float().clone()
            if r:
                res[s].append((time.perf_counter() - t0) * 1e3)
    base = last[settings[0]]
    for s in settings:
        v = sorted(res[s])
        d = ((last[s] - base).norm() / base.norm()).item()
        print(f"chains={s[0]} stagger={s[1]}: median {v[len(v) // 2]:.1f} ms per {a.steps}-step loop (all: {', '.join('%.1f' % x for x in v)}); "
              f"latents vs first setting rel-L2 {d:.2e}", flush=True)
