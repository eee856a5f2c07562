"""Conv3x3 tile sweep at CFG batch 16 (batch 8 generations): which 8-wave tile wins per UNet resolution. GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from seedx_amd import _lib, ops
from tools.bench_gemm_tiles import timeit
dev = torch.device("cuda:0"); lib = _lib.load(); dt = torch.bfloat16
names = ["128x128/2", "128x80/3", "64x128/3", "64x64/3", "256x256-8w", "256x128-8w"]
for B in (2, 8):
    for H, Cin, Cout in [(32, 1280, 1280), (32, 2560, 1280), (64, 640, 640), (64, 1280, 640), (128, 320, 320), (128, 640, 320)]:
        x = torch.randn(B, H, H, Cin, device=dev).to(dt)
        w = (torch.randn(Cout, 9 * Cin, device=dev) * 0.02).to(dt)
        row = []
        for c in range(6):
            lib.sx_gemm_force_tile(c)
            t = timeit(lambda: ops.conv3x3(x, w, out_dtype=torch.float32), iters=8)
            row.append("%s %4.0f" % (names[c], 2 * B * H * H * Cout * 9 * Cin / t / 1e12))
        lib.sx_gemm_force_tile(-1)
        t = timeit(lambda: ops.conv3x3(x, w, out_dtype=torch.float32), iters=8)
        print("conv B%d H%3d Cin%5d Cout%5d | %s | auto %4.0f" % (B, H, Cin, Cout, " | ".join(row), 2 * B * H * H * Cout * 9 * Cin / t / 1e12), flush=True)
