"""Sweep every GEMM tile config over UNet / ViT / LLM shapes at batch 1..8 and dump JSON lines (fits pick_tile's cost model).

Run on the GPU box: python tools/bench_tile_model.py > gpurun_out/tile_model.jsonl
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from seedx_amd import _lib, ops
from tools.bench_gemm_tiles import timeit

NCFG = 7


def main():
    dev = torch.device("cuda:0")
    lib = _lib.load()
    dt = torch.bfloat16
    shapes = [(8192, 8192, 2048)]
    for b in (1, 2, 4, 8, 16):  # UNet CFG batch 2b: 32x32 tokens @1280, 64x64 @640
        m32, m64 = 2 * b * 1024, 2 * b * 4096
        shapes += [(m32, 1280, 1280), (m32, 3840, 1280), (m32, 1280, 5120), (m64, 640, 640), (m64, 1920, 640),
                   (m64, 640, 2560)]
    shapes += [(2048, 1664, 1664), (2048, 4992, 1664), (2048, 8192, 1664), (2048, 1664, 8192), (165, 15360, 5120),
               (165, 5120, 13824), (520, 5120, 5120), (1320, 15360, 5120), (1040, 15360, 5120), (1040, 5120, 13824),
               (32768, 1664, 1664), (32768, 8192, 1664), (32768, 1664, 8192)]
    for M, N, K in shapes:
        a = torch.randn(M, K, device=dev).to(dt)
        w = (torch.randn(N, K, device=dev) * 0.05).to(dt)
        rec = {"kind": "linear", "M": M, "N": N, "K": K, "us": []}
        for c in range(NCFG):
            lib.sx_gemm_force_tile(c)
            rec["us"].append(round(timeit(lambda: ops.gemm(a, w), iters=20, warm=3) * 1e6, 2))
        lib.sx_gemm_force_tile(-1)
        rec["auto_us"] = round(timeit(lambda: ops.gemm(a, w), iters=20, warm=3) * 1e6, 2)
        print(json.dumps(rec), flush=True)
    for b in (1, 2, 4, 8, 16):
        for H, Cin, Cout in [(32, 1280, 1280), (32, 2560, 1280), (64, 640, 640), (64, 1280, 640), (128, 320, 320),
                             (128, 640, 320)]:
            B = 2 * b
            x = torch.randn(B, H, H, Cin, device=dev).to(dt)
            w = (torch.randn(Cout, 9 * Cin, device=dev) * 0.02).to(dt)
            rec = {"kind": "conv", "M": B * H * H, "N": Cout, "K": 9 * Cin, "us": []}
            for c in range(NCFG):
                lib.sx_gemm_force_tile(c)
                rec["us"].append(round(timeit(lambda: ops.conv3x3(x, w), iters=10, warm=2) * 1e6, 2))
            lib.sx_gemm_force_tile(-1)
            rec["auto_us"] = round(timeit(lambda: ops.conv3x3(x, w), iters=10, warm=2) * 1e6, 2)
            print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
