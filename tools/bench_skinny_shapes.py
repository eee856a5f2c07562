"""Per-shape timing of the decode step's skinny GEMMs (13B dims): plain flow (one 16-bit x block per 16 rows) vs precise flow (two planes),
16 and 32 lock-step sequences. Weight bytes / time = the HBM rate each launch reaches; the x operand is re-read from L2 by every workgroup
(64 B x K per operand block), which is what separates the four columns. `python tools/bench_skinny_shapes.py` on the GPU box."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from seedx_amd import ops  # noqa: E402
from seedx_amd.llama import glu_pack_rows  # noqa: E402

dev, dt = torch.device("cuda:0"), torch.float16
H, I = 5120, 13824
shapes = [("qkv", 3 * H, H, False, "t"), ("o (+res, 20-row tiles)", H, H, False, "t20"), ("gate|up (GLU)", 2 * I, H, True, "t"),
          ("down (+res, 20-row tiles)", H, I, False, "t20"), ("lm_head", 32384, H, False, "t")]
g = torch.Generator().manual_seed(0)
ws = torch.zeros(16384 + 8 * 64 * H * 4, dtype=torch.uint8, device=dev)
print(f"{'shape':28s} {'N':>6s} {'K':>6s}   " + "   ".join(f"{c:>22s}" for c in ("16 rows, 1 plane", "16 rows, 2 planes", "32 rows, 1 plane", "32 rows, 2 planes")))
from seedx_amd import _lib  # noqa: E402
R4 = int(os.environ.get("SKINNY_R4", "1"))      # sx_gemv_tune(3, v): 0 = 32-row workgroups only, 1 = 64-row where >= 400 remain (shipped), 2 = wherever legal
_lib.load().sx_gemv_tune(3, R4)
print(f"sx_gemv_tune(3, {R4})")
with torch.no_grad():
    for name, N, K, glu, layout in shapes:
        w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev, dt)
        wk = glu_pack_rows(w[: N // 2], w[N // 2:]) if glu else w
        wt = ops.pack_decode_tiles(wk)
        wt20 = ops.pack_decode_tiles20(wk) if layout == "t20" else None
        cells = []
        for M, planes in ((16, 1), (16, 2), (32, 1), (32, 2)):
            x = torch.randn(M, K, generator=g).to(dev)
            if planes == 2:
                xt = ops.split16(x, dt, tiled=True)
            else:
                xt = ops.Tiled16(M, K, dt, dev)
                xt.t.copy_(x.to(dt).view(M // 16, 16, K // 32, 32).permute(0, 2, 1, 3))
            res = torch.randn(M, N, generator=g).to(dev) if layout == "t20" else None
            kw = dict(w_tiles=wt, w_tiles20=wt20, workspace=ws, out_dtype=torch.float32, residual=res)
            if glu:
                kw.update(act="silu", glu=True)
            for _ in range(5):
                ops.gemv(xt, wk, **kw)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                ops.gemv(xt, wk, **kw)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / 50
            cells.append(f"{us:7.1f} us {N * K * 2 / us / 1e6:5.2f} TB/s")
        print(f"{name:28s} {N:6d} {K:6d}   " + "   ".join(f"{c:>22s}" for c in cells), flush=True)
