"""Time N graph-replayed UNet CFG steps at a given batch, plus a box calibrator (8192^3 GEMM via torch/hipBLASLt and an
HBM copy) so numbers from different gpurun boxes can be compared.  python tools/unet_step_time.py [--batch 16] [--steps 10]"""
import argparse
import os
import sys
import time

ROOT = os.environ.get("SX_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch


def calib(dev):
    a = torch.randn(8192, 8192, device=dev).bfloat16()
    b = torch.randn(8192, 8192, device=dev).bfloat16()
    for _ in range(3):
        a @ b
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        a @ b
    torch.cuda.synchronize()
    tf = 20 * 2 * 8192 ** 3 / (time.perf_counter() - t0) / 1e12
    x = torch.empty(1 << 28, dtype=torch.float32, device=dev)
    y = torch.empty_like(x)
    y.copy_(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        y.copy_(x)
    torch.cuda.synchronize()
    gbs = 10 * 2 * x.numel() * 4 / (time.perf_counter() - t0) / 1e9
    return tf, gbs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--steps", type=int, default=10)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    import bench
    bench.BATCH = a.batch
    bench.USE_VAE = False
    tf, gbs = calib(dev)
    with torch.no_grad():
        try:
            _, _, adapter = bench.build_models(dev, torch.bfloat16, need=("adapter",))
        except TypeError:
            _, _, adapter = bench.build_models(dev, torch.bfloat16)
        if os.environ.get("SX_CHAINS") and hasattr(adapter._loop, "chains"):
            adapter._loop.chains = int(os.environ["SX_CHAINS"])
        feats = torch.randn(a.batch, 64, 4096, device=dev).bfloat16()
        res = {}
        for steps in (a.steps, 4 * a.steps):
            kw = dict(image_embeds=feats, num_inference_steps=steps, seed=1, output_type="latent")
            adapter.generate(**kw)
            ts = []
            for _ in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                adapter.generate(**kw)
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
            res[steps] = min(ts)
        slope = (res[4 * a.steps] - res[a.steps]) / (3 * a.steps)
        fixed = res[a.steps] - slope * a.steps
    tf2, gbs2 = calib(dev)
    print("chains=%s " % os.environ.get("SX_CHAINS", "default"), end="")
    print("%s: UNet CFG step (batch %d x2): %.2f ms per step + %.1f ms fixed per generate() | box: hipBLASLt 8192^3 %.0f/%.0f TF, "
          "copy %.0f/%.0f GB/s" % (ROOT, a.batch, slope * 1e3, fixed * 1e3, tf, tf2, gbs, gbs2))


if __name__ == "__main__":
    main()
