"""GEMM tile sweep on the UNet shapes at batch 4 generations (CFG batch 8): M = 8192 / 32768."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from seedx_amd import _lib, ops
from tools.bench_gemm_tiles import timeit

dev = torch.device("cuda:0")
lib = _lib.load()
dt = torch.bfloat16
names = ["128x128/2", "128x80/3", "64x128/3", "64x64/3", "256x256-8w", "256x128-8w"]
shapes = [(8192, 1280, 1280), (8192, 1280, 5120), (8192, 3840, 1280), (8192, 20480, 1280), (32768, 640, 640), (32768, 640, 2560),
          (32768, 1920, 640), (32768, 10240, 640)]
for M, N, K in shapes:
    a = torch.randn(M, K, device=dev).to(dt)
    w = (torch.randn(N, K, device=dev) * 0.05).to(dt)
    glu = N in (20480, 10240)
    row = []
    for c in range(6):
        lib.sx_gemm_force_tile(c)
        t = timeit(lambda: ops.gemm(a, w, act="gelu" if glu else None, glu=glu), iters=10)
        row.append("%s %5.0f" % (names[c], 2 * M * N * K / t / 1e12))
    lib.sx_gemm_force_tile(-1)
    t = timeit(lambda: ops.gemm(a, w, act="gelu" if glu else None, glu=glu), iters=10)
    tr = timeit(lambda: torch.matmul(a, w.t()), iters=10)
    print("M%6d N%6d K%5d | %s | auto %5.0f | hipblaslt %5.0f" % (M, N, K, " | ".join(row), 2 * M * N * K / t / 1e12, 2 * M * N * K / tr / 1e12), flush=True)
