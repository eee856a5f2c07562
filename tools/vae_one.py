"""One SDXL VAE decode (1024 px) in the mode given by argv[1] ("fp32" | "fast") — a target for rocprofv3 --kernel-trace --stats."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import restated_vae as rv          # seeded synthetic weights only
from seedx_amd.vae import AutoencoderKL

dev = torch.device("cuda:0")
A = rv.FULL_VAE
vae = AutoencoderKL(block_out_channels=A["block_out_channels"], layers_per_block=A["layers_per_block"])
vae.load_state_dict(rv.vae_sd(A, device=dev))
vae.to(dev, torch.float16, precision=sys.argv[1] if len(sys.argv) > 1 else "auto")
z = torch.randn(1, 4, 128, 128, generator=torch.Generator().manual_seed(1)).to(dev)
for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 4):
    out = vae.decode(z, return_dict=False)[0]
torch.cuda.synchronize()
print("ok", out.shape)
