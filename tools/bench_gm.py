"""In-XCD traversal group height (sx_gemm_force_tile(300+gm)) on multi-round GEMM / conv launches. GPU box only."""
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
    gms = (64, 1, 2, 4, 8, 16)
    for M, N, K, glu in [(16384, 10240, 1280, True), (65536, 5120, 640, True), (16384, 3840, 1280, False),
                         (8192, 8192, 8192, False), (16384, 1280, 5120, False), (65536, 640, 2560, False),
                         (2048, 20480, 1280, True), (2048, 3840, 1280, False)]:
        a = torch.randn(M, K, device=dev).to(dt)
        w = (torch.randn(N, K, device=dev) * 0.05).to(dt)
        row = []
        for gm in gms:
            lib.sx_gemm_force_tile(300 + gm)
            t = timeit(lambda: ops.gemm(a, w, act="gelu" if glu else None, glu=glu))
            row.append("gm%-2d %7.1fus %5.0fTF" % (gm, t * 1e6, 2 * M * N * K / t / 1e12))
        print("M%6d N%6d K%5d | %s" % (M, N, K, " | ".join(row)), flush=True)
    for B, H, Cin, Cout in [(16, 32, 1280, 1280), (16, 64, 640, 640), (16, 128, 320, 320), (16, 64, 1280, 640),
                            (2, 64, 640, 640), (2, 128, 320, 320)]:
        x = torch.randn(B, H, H, Cin, device=dev).to(dt)
        w = (torch.randn(Cout, 9 * Cin, device=dev) * 0.02).to(dt)
        row = []
        for gm in gms:
            lib.sx_gemm_force_tile(300 + gm)
            t = timeit(lambda: ops.conv3x3(x, w), iters=10, warm=2)
            row.append("gm%-2d %7.1fus %5.0fTF" % (gm, t * 1e6, 2 * B * H * H * Cout * 9 * Cin / t / 1e12))
        print("conv B%d H%d Cin%d Cout%d | %s" % (B, H, Cin, Cout, " | ".join(row)), flush=True)
    lib.sx_gemm_force_tile(300)


if __name__ == "__main__":
    main()
