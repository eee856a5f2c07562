"""Micro-benchmarks of the HIP kernels on the shapes of the SEED-X hot path (run on the GPU box)."""
import json
import math
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from seedx_amd import ops


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


def main():
    dev = torch.device("cuda:0")
    res = []
    dt = torch.bfloat16
    shapes = [(4096, 4096, 4096), (8192, 8192, 8192), (2048, 4992, 1664), (2048, 8192, 1664), (2048, 1664, 8192),
              (2048, 10240, 1280), (2048, 1280, 5120), (2048, 3840, 1280), (2048, 1280, 1280), (8192, 5120, 640),
              (8192, 640, 2560), (165, 15360, 5120), (64, 27648, 5120), (64, 5120, 13824)]
    for M, N, K in shapes:
        a = torch.randn(M, K, device=dev).to(dt)
        w = (torch.randn(N, K, device=dev) * 0.05).to(dt)
        t = timeit(lambda: ops.gemm(a, w))
        t_ref = timeit(lambda: torch.matmul(a, w.t()))
        res.append({"op": "gemm", "M": M, "N": N, "K": K, "ms": t * 1e3, "tflops": 2 * M * N * K / t / 1e12,
                    "torch_tflops": 2 * M * N * K / t_ref / 1e12})
        print(res[-1], flush=True)
    for B, H, W, Cin, Cout in [(2, 32, 32, 1280, 1280), (2, 64, 64, 640, 640), (2, 128, 128, 320, 320), (2, 32, 32, 2560, 1280)]:
        x = torch.randn(B, H, W, Cin, device=dev).to(dt)
        w = (torch.randn(Cout, 9 * Cin, device=dev) * 0.02).to(dt)
        t = timeit(lambda: ops.conv3x3(x, w))
        fl = 2 * B * H * W * Cout * 9 * Cin
        res.append({"op": "conv3x3", "B": B, "H": H, "Cin": Cin, "Cout": Cout, "ms": t * 1e3, "tflops": fl / t / 1e12})
        print(res[-1], flush=True)
    for B, H, S, D, causal in [(2, 16, 1024, 104, False), (2, 20, 1024, 64, False), (2, 10, 4096, 64, False),
                               (1, 40, 2048, 128, True), (16, 16, 2048, 128, False)]:
        q, k, v = (torch.randn(B, S, H, D, device=dev).to(dt) for _ in range(3))
        t = timeit(lambda: ops.attention(q, k, v, D ** -0.5, causal))
        fl = 4 * B * H * S * S * D * (0.5 if causal else 1.0)
        res.append({"op": "attn", "B": B, "H": H, "S": S, "D": D, "causal": causal, "ms": t * 1e3, "tflops": fl / t / 1e12})
        print(res[-1], flush=True)
    for N, K in [(15360, 5120), (5120, 5120), (27648, 5120), (5120, 13824), (32384, 5120)]:
        x = torch.randn(1, K, device=dev).to(dt)
        w = (torch.randn(N, K, device=dev) * 0.05).to(dt)
        t = timeit(lambda: ops.gemv(x, w), iters=50)
        res.append({"op": "gemv", "N": N, "K": K, "us": t * 1e6, "GBps": N * K * 2 / t / 1e9})
        print(res[-1], flush=True)
    for rows, cols in [(2048, 1664), (8192, 640), (2048, 1280)]:
        x = torch.randn(rows, cols, device=dev)
        g = torch.ones(cols, device=dev); b = torch.zeros(cols, device=dev)
        t = timeit(lambda: ops.layernorm(x, g, b, 1e-6, dt), iters=50)
        res.append({"op": "layernorm", "rows": rows, "cols": cols, "us": t * 1e6, "GBps": rows * cols * 6 / t / 1e9})
        print(res[-1], flush=True)
    for B, HW, C in [(2, 16384, 320), (2, 4096, 640), (2, 1024, 1280)]:
        x = torch.randn(B, HW, C, device=dev)
        g = torch.ones(C, device=dev); b = torch.zeros(C, device=dev)
        t = timeit(lambda: ops.groupnorm(x, g, b, 32, 1e-5, True, dt), iters=50)
        res.append({"op": "groupnorm", "B": B, "HW": HW, "C": C, "us": t * 1e6, "GBps": B * HW * C * 10 / t / 1e9})
        print(res[-1], flush=True)
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(res, open("gpurun_out/bench_kernels.json", "w"), indent=1)


if __name__ == "__main__":
    main()
