"""GroupNorm (+SiLU) on the UNet's shapes at CFG batch 32: time and effective GB/s of the statistics + apply pair."""
import os
import sys
import time

ROOT = os.environ.get("SX_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from seedx_amd import ops

dev = torch.device("cuda:0")
print(ROOT)
for B, HW, C in ((32, 1024, 1280), (32, 4096, 640), (32, 16384, 320), (32, 1024, 2560), (32, 4096, 960), (32, 4096, 1280)):
    x = torch.randn(B, HW, C, device=dev)
    g, b = torch.randn(C, device=dev), torch.randn(C, device=dev)
    for _ in range(3):
        ops.groupnorm(x, g, b, 32, 1e-5, True, torch.bfloat16)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        ops.groupnorm(x, g, b, 32, 1e-5, True, torch.bfloat16)
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / 20
    print("B%d HW%5d C%4d: %7.1f us  %5.0f GB/s (2 reads + 1 16-bit write)" % (B, HW, C, t * 1e6, x.numel() * 10 / t / 1e9))
