"""Eager launches of the UNet self-attention shapes for rocprofv3 --pmc passes."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from seedx_amd import ops

dev = torch.device("cuda:0")
dt = torch.bfloat16
for B, H, S, D in [(16, 10, 4096, 64), (16, 20, 1024, 64)]:
    q = torch.randn(B, S, H, D, device=dev).to(dt)
    k = torch.randn(B, S, H, D, device=dev).to(dt)
    v = torch.randn(B, S, H, D, device=dev).to(dt)
    for _ in range(4):
        ops.attention(q, k, v, D ** -0.5, causal=False)
torch.cuda.synchronize()
