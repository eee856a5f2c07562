"""SDXL VAE decode (1024 px, fp32-grade mode) with every GEMM forced onto one tile config (sx_gemm_force_tile) against the
cost model's own choice (-1): a check of the tile model on the VAE's M = 16k..1M, tripled-K shapes."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import restated_vae as rv          # seeded synthetic weights only
from seedx_amd import _lib
from seedx_amd.vae import AutoencoderKL

dev = torch.device("cuda:0")
lib = _lib.load()
A = rv.FULL_VAE
vae = AutoencoderKL(block_out_channels=A["block_out_channels"], layers_per_block=A["layers_per_block"])
vae.load_state_dict(rv.vae_sd(A, device=dev))
vae.to(dev, torch.float16, precision=sys.argv[1] if len(sys.argv) > 1 else "fp32")
z = torch.randn(1, 4, 128, 128, generator=torch.Generator().manual_seed(1)).to(dev)
names = {-1: "auto", 0: "128x128", 1: "128x80", 2: "64x128", 3: "64x64", 4: "256x256", 5: "256x320", 6: "256x160"}
for cfg in (-1, 0, 2, 4, 6, 5, 1, 3):
    lib.sx_gemm_force_tile(cfg)
    try:
        for _ in range(2):
            vae.decode(z)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(4):
            vae.decode(z)
        torch.cuda.synchronize()
        print("tile %-8s %7.2f ms / image" % (names[cfg], (time.perf_counter() - t0) / 4 * 1e3))
    except Exception as ex:
        print("tile %-8s failed: %s" % (names[cfg], str(ex)[:80]))
lib.sx_gemm_force_tile(-1)
