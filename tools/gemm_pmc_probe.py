"""Eager launches of the dominant GEMM / conv shapes of one UNet step (for rocprofv3 --pmc passes)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from seedx_amd import ops

dev = torch.device("cuda:0")
dt = torch.bfloat16
shapes = [(2048, 20480, 1280, True), (2048, 3840, 1280, False), (2048, 1280, 1280, False), (2048, 1280, 5120, False),
          (8192, 10240, 640, True), (8192, 640, 2560, False)]
for M, N, K, glu in shapes:
    a = torch.randn(M, K, device=dev).to(dt)
    w = (torch.randn(N, K, device=dev) * 0.05).to(dt)
    for _ in range(5):
        ops.gemm(a, w, act="gelu" if glu else None, glu=glu)
x = torch.randn(2, 32, 32, 1280, device=dev).to(dt)
wc = (torch.randn(1280, 9 * 1280, device=dev) * 0.02).to(dt)
for _ in range(5):
    ops.conv3x3(x, wc, out_dtype=torch.float32)
torch.cuda.synchronize()
print("probe done")
