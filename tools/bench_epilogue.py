"""Fixed per-round cost of the GEMM (prologue + fused epilogue): K = 64 launches against K = 1280, bf16 vs fp32 output,
with and without bias/residual. Run on the GPU box: python tools/bench_epilogue.py"""
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
    for M, N in [(16384, 10240), (16384, 1280), (65536, 640)]:
        res = torch.randn(M, N, device=dev)
        bias = torch.randn(N, device=dev)
        for cfg in (0, 4, 5):
            lib.sx_gemm_force_tile(cfg)
            for K in (64, 1280):
                a = torch.randn(M, K, device=dev).to(dt)
                w = (torch.randn(N, K, device=dev) * 0.05).to(dt)
                o16 = torch.empty(M, N, device=dev, dtype=dt)
                o32 = torch.empty(M, N, device=dev)
                t16 = timeit(lambda: ops.gemm(a, w, out=o16))
                t32 = timeit(lambda: ops.gemm(a, w, out=o32, out_dtype=torch.float32))
                tb = timeit(lambda: ops.gemm(a, w, bias=bias, out=o16))
                tr = timeit(lambda: ops.gemm(a, w, bias=bias, residual=res, out=o32, out_dtype=torch.float32))
                print("M%6d N%6d K%5d cfg%d | bf16 out %7.1fus (%.2f TB/s C) | f32 out %7.1fus (%.2f TB/s) | +bias %7.1fus | "
                      "+bias+res f32 %7.1fus" % (M, N, K, cfg, t16 * 1e6, M * N * 2 / t16 / 1e12, t32 * 1e6,
                                                 M * N * 4 / t32 / 1e12, tb * 1e6, tr * 1e6), flush=True)
    lib.sx_gemm_force_tile(-1)
    x = torch.randn(16384 * 10240, device=dev)
    y = torch.empty_like(x)
    t = timeit(lambda: y.copy_(x))
    print("torch copy 671 MB: %.1f us → %.2f TB/s read + %.2f TB/s write" % (t * 1e6, x.numel() * 4 / t / 1e12, x.numel() * 4 / t / 1e12))
    t = timeit(lambda: y.fill_(1.0))
    print("torch fill 671 MB: %.1f us → %.2f TB/s write" % (t * 1e6, x.numel() * 4 / t / 1e12))


if __name__ == "__main__":
    main()
