"""Flash-attention kernel on the UNet / ViT / LLM shapes (GPU box only)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from seedx_amd import ops
from tools.bench_gemm_tiles import timeit


def main():
    dev = torch.device("cuda:0")
    dt = torch.bfloat16
    for name, B, H, S, Skv, D, causal in [("unet 64^2 b8", 16, 10, 4096, 4096, 64, False), ("unet 32^2 b8", 16, 20, 1024, 1024, 64, False),
                                          ("unet 64^2 b1", 2, 10, 4096, 4096, 64, False), ("unet 32^2 b1", 2, 20, 1024, 1024, 64, False),
                                          ("vit 2 crops", 2, 16, 1024, 1024, 104, False), ("llm prefill", 1, 40, 165, 165, 128, True),
                                          ("llm 8x2048", 1, 40, 2048, 2048, 128, True)]:
        q = torch.randn(B, S, H, D, device=dev).to(dt)
        k = torch.randn(B, Skv, H, D, device=dev).to(dt)
        v = torch.randn(B, Skv, H, D, device=dev).to(dt)
        t = timeit(lambda: ops.attention(q, k, v, D ** -0.5, causal=causal), iters=20)
        fl = 4.0 * B * H * S * Skv * D * (0.5 if causal else 1.0)
        print("%-14s B%2d H%2d S%5d D%3d causal%d: %8.1f us  %6.0f TF" % (name, B, H, S, D, causal, t * 1e6, fl / t / 1e12), flush=True)


if __name__ == "__main__":
    main()
