"""A/B of the 2-D XCD tile partition (sx_gemm_force_tile(100|101)) on the hot GEMM/conv shapes, interleaved rounds."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from seedx_amd import _lib, ops

dev = torch.device("cuda:0")
lib = _lib.load()
dt = torch.bfloat16


def timeit(fn, iters=20, warm=3):
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


shapes = [(2048, 1280, 1280), (2048, 1280, 5120), (2048, 3840, 1280), (2048, 20480, 1280), (8192, 640, 640), (8192, 640, 2560),
          (8192, 1920, 640), (8192, 10240, 640), (32768, 320, 320), (2048, 4992, 1664), (2048, 8192, 1664), (2048, 1664, 8192),
          (165, 15360, 5120), (4096, 4096, 4096), (8192, 8192, 8192)]
for M, N, K in shapes:
    a = torch.randn(M, K, device=dev).to(dt)
    w = (torch.randn(N, K, device=dev) * 0.05).to(dt)
    glu = N in (20480, 10240)
    res = {100: [], 101: []}
    for r in range(3):
        for mode in (100, 101):
            lib.sx_gemm_force_tile(mode)
            res[mode].append(timeit(lambda: ops.gemm(a, w, act="gelu" if glu else None, glu=glu)))
    fl = 2 * M * N * K
    print("M%6d N%6d K%6d | linear remap %6.1fus %5.0fTF | 2-D xcd %6.1fus %5.0fTF" % (
        M, N, K, min(res[100]) * 1e6, fl / min(res[100]) / 1e12, min(res[101]) * 1e6, fl / min(res[101]) / 1e12), flush=True)
for B, H, Cin, Cout in [(2, 32, 1280, 1280), (2, 32, 2560, 1280), (2, 64, 640, 640), (2, 128, 320, 320)]:
    x = torch.randn(B, H, H, Cin, device=dev).to(dt)
    w = (torch.randn(Cout, 9 * Cin, device=dev) * 0.02).to(dt)
    res = {100: [], 101: []}
    for r in range(3):
        for mode in (100, 101):
            lib.sx_gemm_force_tile(mode)
            res[mode].append(timeit(lambda: ops.conv3x3(x, w)))
    fl = 2 * B * H * H * Cout * 9 * Cin
    print("conv B%d H%3d Cin%5d Cout%5d | linear remap %6.1fus %5.0fTF | 2-D xcd %6.1fus %5.0fTF" % (
        B, H, Cin, Cout, min(res[100]) * 1e6, fl / min(res[100]) / 1e12, min(res[101]) * 1e6, fl / min(res[101]) / 1e12), flush=True)
lib.sx_gemm_force_tile(101)
