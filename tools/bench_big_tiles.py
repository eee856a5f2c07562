"""A/B of the 8-wave GEMM tiles: cfg 4 = 256x256, cfg 5 = 256x320, cfg 6 = 256x160 (N of the SDXL UNet is k*320).

Run on the GPU box: python tools/bench_big_tiles.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from seedx_amd import _lib, ops
from tools.bench_gemm_tiles import timeit


def main():
    dev = torch.device("cuda:0")
    lib = _lib.load()
    dt = torch.bfloat16
    shapes = [(8192, 8192, 8192), (16384, 1280, 1280), (16384, 3840, 1280), (16384, 1280, 5120),
              (65536, 640, 640), (65536, 1920, 640), (65536, 640, 2560), (8192, 1280, 1280), (8192, 3840, 1280),
              (8192, 1280, 5120), (32768, 640, 640), (32768, 1920, 640), (32768, 640, 2560), (4096, 1280, 1280),
              (4096, 1280, 5120), (16384, 640, 640)]
    for M, N, K in shapes:
        a = torch.randn(M, K, device=dev).to(dt)
        w = (torch.randn(N, K, device=dev) * 0.05).to(dt)
        ref = torch.matmul(a, w.t()).float()
        row = []
        for c in (4, 5, 6):
            lib.sx_gemm_force_tile(c)
            out = ops.gemm(a, w).float()
            err = ((out - ref).norm() / ref.norm()).item()
            t = timeit(lambda: ops.gemm(a, w))
            row.append("cfg%d %7.1fus %5.0fTF err %.1e" % (c, t * 1e6, 2 * M * N * K / t / 1e12, err))
        lib.sx_gemm_force_tile(-1)
        tr = timeit(lambda: torch.matmul(a, w.t()))
        print("M%6d N%6d K%6d | %s | hipblaslt %5.0fTF" % (M, N, K, " | ".join(row), 2 * M * N * K / tr / 1e12), flush=True)
    for B, H, Cin, Cout in [(8, 32, 1280, 1280), (8, 64, 640, 640), (8, 128, 320, 320), (16, 32, 2560, 1280)]:
        x = torch.randn(B, H, H, Cin, device=dev).to(dt)
        w = (torch.randn(Cout, 9 * Cin, device=dev) * 0.02).to(dt)
        row = []
        for c in (4, 5, 6):
            lib.sx_gemm_force_tile(c)
            t = timeit(lambda: ops.conv3x3(x, w))
            row.append("cfg%d %7.1fus %5.0fTF" % (c, t * 1e6, 2 * B * H * H * Cout * 9 * Cin / t / 1e12))
        lib.sx_gemm_force_tile(-1)
        print("conv B%d H%d Cin%d Cout%d | %s" % (B, H, Cin, Cout, " | ".join(row)), flush=True)


if __name__ == "__main__":
    main()
