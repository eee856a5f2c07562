"""Decode GEMV: VALU path vs MFMA skinny GEMM at 1..16 rows on the Llama-13B shapes (GPU box only)."""
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
    for N, K, glu in [(15360, 5120, False), (5120, 5120, False), (27648, 5120, True), (5120, 13824, False), (32384, 5120, False)]:
        w = (torch.randn(N, K, device=dev) * 0.02).to(dt)
        for M in (1, 2, 4, 8, 16):
            x = torch.randn(M, K, device=dev).to(dt)
            row = []
            for valu in (1, 2):
                if (valu == 1 and M > 8) or (valu == 2 and M < 2):
                    continue
                lib.sx_gemv_force_valu(valu)
                t = timeit(lambda: ops.gemv(x, w, act="silu" if glu else None, glu=glu), iters=50)
                row.append("%s %6.1fus %4.2f TB/s" % ("valu" if valu == 1 else "mfma", t * 1e6, N * K * 2 / t / 1e12))
            print("N%6d K%6d glu%d M%2d | %s" % (N, K, glu, M, " | ".join(row)), flush=True)
    lib.sx_gemv_force_valu(0)


if __name__ == "__main__":
    main()
