"""Which tile is best for the fp32-residual epilogue GEMMs (out-proj / ff2 / proj_out of the UNet transformer blocks)?
The fitted cost model only saw plain 16-bit epilogues. Forces every tile config on those call sites."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from seedx_amd import _lib, ops

names = ["128x128", "128x80", "64x128", "64x64", "256x256", "256x320", "256x160"]


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


dev, dt = torch.device("cuda:0"), torch.bfloat16
lib = _lib.load()
for M, N, K in ((32768, 1280, 1280), (32768, 1280, 5120), (131072, 640, 640), (131072, 640, 2560), (16384, 1280, 1280)):
    x = torch.randn(M, K, device=dev).to(dt)
    w = (torch.randn(N, K, device=dev) * 0.03).to(dt)
    b = torch.randn(N, device=dev)
    res = torch.randn(M, N, device=dev)
    for label, kw in (("16b", {}), ("res-f32", dict(bias=b, residual=res, out_dtype=torch.float32))):
        row = []
        for c in (0, 4, 5, 6):
            lib.sx_gemm_force_tile(c)
            t = timeit(lambda: ops.gemm(x, w, **kw))
            row.append("%s %6.1f" % (names[c], t * 1e6))
        lib.sx_gemm_force_tile(-1)
        t = timeit(lambda: ops.gemm(x, w, **kw))
        print("M%6d N%5d K%5d %-8s | %s | auto(%s) %6.1f us" % (M, N, K, label, " | ".join(row),
                                                               names[lib.sx_gemm_pick_tile(M, N, K, 0, 0)], t * 1e6), flush=True)
