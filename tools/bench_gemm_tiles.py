"""GEMM tile-config sweep on the UNet / ViT / LLM shapes (run on the GPU box)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from seedx_amd import _lib, ops


def timeit(fn, iters=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def main():
    dev = torch.device("cuda:0")
    lib = _lib.load()
    dt = torch.bfloat16
    names = ["128x128/2", "128x80/3", "64x128/3", "64x64/3", "256x256-8w", "256x128-8w"]
    shapes = [(2048, 1280, 1280), (2048, 1280, 5120), (2048, 3840, 1280), (2048, 10240, 1280), (8192, 640, 640),
              (8192, 640, 2560), (8192, 1920, 640), (8192, 5120, 640), (32768, 320, 320), (2048, 1664, 1664),
              (2048, 4992, 1664), (2048, 8192, 1664), (2048, 1664, 8192), (165, 15360, 5120), (165, 5120, 13824),
              (65, 5120, 5120), (4096, 4096, 4096)]
    for M, N, K in shapes:
        a = torch.randn(M, K, device=dev).to(dt)
        w = (torch.randn(N, K, device=dev) * 0.05).to(dt)
        row = []
        for c in range(6):
            lib.sx_gemm_force_tile(c)
            t = timeit(lambda: ops.gemm(a, w))
            row.append("%s %6.1fus %5.0fTF" % (names[c], t * 1e6, 2 * M * N * K / t / 1e12))
        lib.sx_gemm_force_tile(-1)
        t = timeit(lambda: ops.gemm(a, w))
        tr = timeit(lambda: torch.matmul(a, w.t()))
        print("M%6d N%6d K%6d | %s | auto %6.1fus %5.0fTF | hipblaslt %5.0fTF" % (
            M, N, K, " | ".join(row), t * 1e6, 2 * M * N * K / t / 1e12, 2 * M * N * K / tr / 1e12), flush=True)
    for B, H, Cin, Cout in [(2, 32, 1280, 1280), (2, 32, 2560, 1280), (2, 64, 640, 640), (2, 128, 320, 320)]:
        x = torch.randn(B, H, H, Cin, device=dev).to(dt)
        w = (torch.randn(Cout, 9 * Cin, device=dev) * 0.02).to(dt)
        row = []
        for c in range(6):
            lib.sx_gemm_force_tile(c)
            t = timeit(lambda: ops.conv3x3(x, w))
            row.append("%s %6.1fus %5.0fTF" % (names[c], t * 1e6, 2 * B * H * H * Cout * 9 * Cin / t / 1e12))
        lib.sx_gemm_force_tile(-1)
        print("conv B%d H%d Cin%d Cout%d | %s" % (B, H, Cin, Cout, " | ".join(row)), flush=True)
    for rows, cols in [(2048, 1664), (8192, 640), (2048, 1280), (165, 5120)]:
        x = torch.randn(rows, cols, device=dev)
        g = torch.ones(cols, device=dev)
        b = torch.zeros(cols, device=dev)
        t = timeit(lambda: ops.layernorm(x, g, b, 1e-6, dt), iters=50)
        print("layernorm %d x %d: %.1f us  %.0f GB/s" % (rows, cols, t * 1e6, rows * cols * 6 / t / 1e9))


if __name__ == "__main__":
    main()
