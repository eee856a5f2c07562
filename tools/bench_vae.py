"""SDXL VAE decode of one 128x128 latent → 1024 px image: the fp32-grade mode (two bf16 planes per operand, three MFMA
passes; what fp16 + force_upcast selects, like the reference's upcast_vae) against single 16-bit operands ("fast")."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import restated_vae as rv          # weights generator only (seeded synthetic state dict)
from seedx_amd.vae import AutoencoderKL

dev = torch.device("cuda:0")
A = rv.FULL_VAE
sd = rv.vae_sd(A, device=dev)
z = torch.randn(1, 4, 128, 128, generator=torch.Generator().manual_seed(1)).to(dev)
outs = {}
for dt, prec in ((torch.float16, "fp32"), (torch.float16, "fast"), (torch.bfloat16, "fast")):
    vae = AutoencoderKL(block_out_channels=A["block_out_channels"], layers_per_block=A["layers_per_block"])
    vae.load_state_dict(dict(sd))
    vae.to(dev, dt, precision=prec)
    for _ in range(2):
        out = vae.decode(z, return_dict=False)[0]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        out = vae.decode(z, return_dict=False)[0]
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    outs[(dt, prec)] = out
    print(f"VAE decode 1024 px, {dt} precision={prec}: {ms:7.2f} ms / image")
    del vae
ref = outs[(torch.float16, "fp32")]
for k, v in outs.items():
    print(k, "rel-L2 vs fp32-grade: %.2e" % ((v - ref).norm() / ref.norm()).item())
