"""The five GEMM call sites that carry a C=1280 SDXL transformer block at CFG batch 32 (M = 32768), with their real
epilogues: qkv (16-bit out), out-proj (+bias, +fp32 residual, fp32 out), q2, GEGLU (bias, exact-erf GELU gate, GLU product)
and ff2 (+bias +residual). Prints us / TFLOP/s / algorithmic GB/s per call.   SX_ROOT=<tree> python tools/bench_unet_gemms.py"""
import os
import sys
import time

ROOT = os.environ.get("SX_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from seedx_amd import ops


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


def main():
    dev, dt = torch.device("cuda:0"), torch.bfloat16
    from seedx_amd.llama import glu_pack_rows
    rows = []
    for M, C in ((32768, 1280), (131072, 640)):
        x = torch.randn(M, C, device=dev).to(dt)
        res = torch.randn(M, C, device=dev)
        wqkv = (torch.randn(3 * C, C, device=dev) * 0.03).to(dt)
        wo = (torch.randn(C, C, device=dev) * 0.03).to(dt)
        bo = torch.randn(C, device=dev)
        w1 = (torch.randn(8 * C, C, device=dev) * 0.03).to(dt)
        wff1 = glu_pack_rows(w1[:4 * C].contiguous(), w1[4 * C:].contiguous())
        bff1 = torch.randn(8 * C, device=dev)
        wff2 = (torch.randn(C, 4 * C, device=dev) * 0.03).to(dt)
        g = torch.randn(M, 4 * C, device=dev).to(dt)
        cases = [("qkv", lambda: ops.gemm(x, wqkv), 2 * M * 3 * C * C),
                 ("out+res", lambda: ops.gemm(x, wo, bias=bo, residual=res, out_dtype=torch.float32), 2 * M * C * C),
                 ("q2", lambda: ops.gemm(x, wo), 2 * M * C * C),
                 ("geglu", lambda: ops.gemm(x, wff1, bias=bff1, act="gelu", glu=True), 2 * M * 8 * C * C),
                 ("ff2+res", lambda: ops.gemm(g, wff2, bias=bo, residual=res, out_dtype=torch.float32), 2 * M * 4 * C * C)]
        for name, fn, fl in cases:
            t = timeit(fn)
            rows.append("%-8s M%6d C%4d %8.1f us %5.0f TF" % (name, M, C, t * 1e6, fl / t / 1e12))
    print(ROOT)
    print("\n".join(rows))


if __name__ == "__main__":
    main()
