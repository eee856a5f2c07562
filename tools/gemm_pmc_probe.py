"""Eager launches of the dominant GEMM / conv shapes of one batch-8 UNet step (for rocprofv3 --pmc passes).

Each shape is launched 5x; the kernel-trace / counter CSVs identify them by dispatch order. Algorithmic HBM bytes per
launch (operands read once + output written once) are printed so tools/pmc_summary.py can set them beside FETCH_SIZE /
WRITE_SIZE."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from seedx_amd import ops

dev = torch.device("cuda:0")
dt = torch.bfloat16
shapes = [(16384, 20480, 1280, True), (16384, 3840, 1280, False), (16384, 1280, 1280, False), (16384, 1280, 5120, False),
          (65536, 10240, 640, True), (65536, 640, 2560, False)]
alg = []
for M, N, K, glu in shapes:
    a = torch.randn(M, K, device=dev).to(dt)
    w = (torch.randn(N, K, device=dev) * 0.05).to(dt)
    for _ in range(5):
        ops.gemm(a, w, act="gelu" if glu else None, glu=glu)
    alg.append({"kind": "linear", "M": M, "N": N, "K": K, "glu": glu,
                "alg_bytes": 2 * (M * K + N * K + M * (N // 2 if glu else N))})
for B, H, Cin, Cout in [(16, 32, 1280, 1280), (16, 64, 640, 640), (16, 128, 320, 320)]:
    x = torch.randn(B, H, H, Cin, device=dev).to(dt)
    wc = (torch.randn(Cout, 9 * Cin, device=dev) * 0.02).to(dt)
    for _ in range(5):
        ops.conv3x3(x, wc, out_dtype=torch.float32)
    alg.append({"kind": "conv3x3", "M": B * H * H, "N": Cout, "K": 9 * Cin,
                "alg_bytes": 2 * B * H * H * Cin + 2 * Cout * 9 * Cin + 4 * B * H * H * Cout})
torch.cuda.synchronize()
print("ALG " + json.dumps(alg))
